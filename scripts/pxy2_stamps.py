"""Where inside the step-wise fast path's first launch (k_pxy2) do its three kinds of workgroups end?  Diagnostic build only:
  bash scripts/build_variants.sh srukf_factor.hip SRUKF_PXY2_DBG 1;  python scripts/pxy2_stamps.py [N] [lib]
Prints, for the last frames, microseconds from the start of the motion workgroup: the motion reduction's end, a tile's end, and the stations of landmark group 0's statistics."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "build", "variants", "libsrukf_hip_SRUKF_PXY2_DBG_1.so")
pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
srukf.load_library(lib)
p = synth.scene_params()
F = 24
sc = synth.make_scene(N, F + 2, seed=0, p=p)
for early in (2, 0):
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.debug_set("step_early", early)
    rows = []
    for t in range(F):
        f.predict_motion_next(sc["odo"][t + 1], sc["odo"][t + 2])
        f.predict_motion(sc["odo"][t], sc["odo"][t + 1])
        h, Si, vis = f.predict_measurement()
        f.update(sc["z"][t], sc["matched"][t] * vis)
        rows.append([f.debug_get("meas_flag_ticks")] + [f.debug_get(f"pxy2_stamp{k}") for k in (2, 6, 4, 5, 3, 7)])
    r = np.array(rows[6:], dtype=np.float64) * 0.01
    m = np.median(r, axis=0)
    print(f"N={N} step_early={early}: us from the motion workgroup's start (median over {len(r)} frames): motion end {m[1]:.1f}, first tile's end {m[6]:.1f} | statistics of group 0: "
          f"first job starts {m[2]:.1f}, last partial job done {m[3]:.1f}, last arrival knows it is last {m[4]:.1f}, final pass done {m[5]:.1f}; flag to the host (all groups) {m[0]:.1f}", flush=True)
    f.close()
