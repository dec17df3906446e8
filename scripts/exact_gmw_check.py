"""The exact path's forms on ONE matrix with active theta clamps (tiny diagonal entries under O(1) off-diagonals), through srukf_gmw_host with force_slow: left-looking k_gmw_col
(exact_rl 0), right-looking one pivot per launch (exact_rl 2), right-looking 8 pivots per launch (default) — clamp counts, D, P = S^T S against each other (against the oracle:
tests/test_gpu_parity_r4.py::test_exact_path_forms_under_active_theta_clamps).   python scripts/exact_gmw_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); srukf = pkg.srukf
rng = np.random.default_rng(5)
for n in (40, 300, 1204):
    A = rng.standard_normal((n + 10, n)); G = A.T @ A
    k = n // 3
    for kk in (k, k + 7, 2 * k): G[kk, kk] = 1e-7                     # tiny diagonal under O(1) off-diagonals: the theta clamp fires
    G[:, n - 5:] *= 1e-4; G[n - 5:, :] *= 1e-4
    out = {}
    for name, v in (("ll", 0), ("rl1", 2), ("rlb", 1)):
        srukf.debug_set_global("exact_rl", v)
        S, D, hit = srukf.gmw(G, force_slow=True)
        out[name] = (S, D, hit)
    srukf.debug_set_global("exact_rl", 1)
    P = {k_: v[0].T @ v[0] for k_, v in out.items()}
    print(n, "hits", {k_: v[2] for k_, v in out.items()}, "rl1==rlb", np.array_equal(out["rl1"][0], out["rlb"][0]),
          "max|D ll-rlb|/D %.2e" % (np.abs(out["ll"][1] - out["rlb"][1]) / np.abs(out["ll"][1])).max(),
          "max|P ll-rlb| %.2e (scale %.2e)" % (np.abs(P["ll"] - P["rlb"]).max(), np.abs(P["ll"]).max()))
