"""Which pivot rows get a frame flagged?  (a) the shipped a1..a4 = 8 at N = 200 (theta clamp), (b) the mixed-precision downdate in the rank-aware form.
One staged frame per call through the asynchronous replay; a flagged frame is reported (first flagged state row, number of flagged rows, aborted launches) and then
repeated through the synchronous call.   python scripts/flag_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf


def probe(name, N, F, p, storage=None, sets=(), quiet=False):
    sc = synth.make_scene(N, F, seed=1 if storage is None else 0, p=p)
    f = srukf.Filter(N, p)
    if storage is not None:
        f.debug_allow_mixed(True); f.set_storage(storage)
    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    for k, v in sets:
        f.debug_set(k, v)
    n = f.n
    print(f"== {name}: N = {N}, n = {n}, kept {f.debug_get('plan_kept')}, plan red_perm {f.debug_get('plan_red_perm')} fuse {f.debug_get('plan_fuse')}")
    for t in range(F):
        X0, S0 = f.get_state()
        try:
            f.run_frames_async(t, 1); f.synchronize()
            flagged = False
        except srukf.SrukfError as e:
            flagged = True
            fr, row = f.clamp_info()
            print(f"  frame {t}: FLAGGED first row {row} of {n} (landmark {row // 6} entry {row % 6}; robot rows start at {n - 4}), clamp_rows {f.debug_get('clamp_rows')}, aborts {f.debug_get('gmw_aborts')}: {str(e)[:100]}")
            f.set_state(X0, S0)
            f.run_frames(t, 1)
        if quiet and not flagged and t % 50:
            continue
        D = f.debug_copy("D", n)
        r = int(f.debug_get("plan_kept")) or n
        X, S = f.get_state()
        print(f"  frame {t}: {'exact' if flagged else 'clean'}  pose {X[-4:-2]} truth {sc['odo'][t + 1][:2]}  robot pivots {D[r - 4:r]}  min pivot {D[:r].min():.3e} at permuted {int(np.argmin(D[:r]))} / {r}, pivots < 1e-11: {int((D[:r] < 1e-11).sum())}, finite {np.isfinite(S).all()}")
        if not np.isfinite(S).all():
            break
    f.close()


p = synth.scene_params()
if len(sys.argv) > 1 and sys.argv[1] == "mixed":
    probe("mixed, rank-aware, robot / anchor tiles in FP64, N = 500", 500, int(sys.argv[2]) if len(sys.argv) > 2 else 300, p, srukf.STORAGE_F32_MIXED, quiet=True)
    sys.exit(0)
probe("theta clamp, shipped constants", 200, 12, synth.default_params())
probe("mixed, rank-aware", 200, 12, p, srukf.STORAGE_F32_MIXED)
probe("mixed, rank-aware, N = 500", 500, 8, p, srukf.STORAGE_F32_MIXED)
