"""The hybrid mixed downdate (fp32-formed S^T S - U U^T over the kept rows, robot / shared-anchor tiles in FP64) beside the fp64 filter and fp32 storage, frame by frame:
pose difference, the robot block's pivots, the largest relative difference of the kept pivots.   python scripts/mixed_drift_probe.py [N] [frames] [every]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
if len(sys.argv) > 4:
    srukf.load_library(sys.argv[4])                 # an A/B build (scripts/build_variants.sh srukf_mixed.hip MX_KCHUNK 128)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
F = int(sys.argv[2]) if len(sys.argv) > 2 else 270
every = int(sys.argv[3]) if len(sys.argv) > 3 else 10
p = synth.scene_params()
sc = synth.make_scene(N, F, seed=0, p=p)


def mk(storage):
    f = srukf.Filter(N, p)
    if storage == srukf.STORAGE_F32_MIXED:
        f.debug_allow_mixed(True)
    if storage != srukf.STORAGE_F64:
        f.set_storage(storage)
    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    return f


fs = {"f64": mk(srukf.STORAGE_F64), "f32": mk(srukf.STORAGE_F32), "mix": mk(srukf.STORAGE_F32_MIXED)}
n = fs["f64"].n
r = int(fs["f64"].debug_get("plan_kept"))
names = ["xi", "yi", "zi", "theta", "phi", "rho"]
for t in range(F):
    out = {}
    for k, f in fs.items():
        e0 = f.debug_get("exact_frames")
        tr = f.run_frames(t, 1)
        out[k] = (tr[0], f.debug_copy("D", n)[:r].copy(), f.debug_get("exact_frames") - e0)
    if t % every == 0 or out["mix"][2] or t > F - 12:
        d64, dmx, d32 = out["f64"][1], out["mix"][1], out["f32"][1]
        rel = np.abs(dmx - d64) / d64
        rel32 = np.abs(d32 - d64) / d64
        a = int(np.argmax(rel[:r - 4]))
        print(f"frame {t:4d}: |pose mix - f64| {np.abs(out['mix'][0][:2] - out['f64'][0][:2]).max():.2e}  f32 - f64 {np.abs(out['f32'][0][:2] - out['f64'][0][:2]).max():.2e}  exact {out['mix'][2]}"
              f"  robot pivots f64 {d64[r - 4:]}  mix {dmx[r - 4:]}  max rel dD landmarks: mix {rel[:r - 4].max():.1e} (permuted {a}) f32 {rel32[:r - 4].max():.1e}")
