"""BASELINE configs[4], the tolerance study as a function of frames: N = 500 on the benchmark sequence, the same filter with the state STORED as fp64, as fp32
(SRUKF_STORAGE_F32: X and S live as float between frames, arithmetic fp64) and — with the clamp at 1e-8, where it is offered — with the mixed-precision downdate
(SRUKF_STORAGE_F32_MIXED: S^T S - U U^T on the fp32 matrix pipe).  Pose difference against the fp64 run and error against the truth at checkpoints.
  python scripts/f32_curve.py [frames] [tag]   -> gpurun_out/<tag>.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
F = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
tag = sys.argv[2] if len(sys.argv) > 2 else "r05_f32_curve_n500"
N = 500
marks = [m for m in (1, 10, 30, 100, 300, 1000, 1500, 2000, 2500, 3000) if m <= F]


def run(storage, eps=None):
    p = synth.scene_params()
    if eps is not None:
        p["epsilon"] = eps
    sc = synth.make_scene(N, F, seed=0, p=p)
    f = srukf.Filter(N, p)
    if storage != srukf.STORAGE_F64:
        f.set_storage(storage)
    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    tr = np.vstack([f.run_frames(a, min(250, F - a)) for a in range(0, F, 250)])
    info = {k: int(f.debug_get(k)) for k in ("split_form", "gmw_shared", "split_off")}
    f.close()
    return tr, sc["odo"][1:F + 1], info


t64, truth, i64 = run(srukf.STORAGE_F64)
t32, _, i32 = run(srukf.STORAGE_F32)
out = {"workload": f"N = {N}, benchmark sequence (seed 0), {F} frames through srukf_run_frames in blocks of 250, rank-aware default", "frames": marks,
       "pose_err_vs_truth_m": {"f64": [float(np.abs(t64[m - 1, :2] - truth[m - 1, :2]).max()) for m in marks], "f32_storage": [float(np.abs(t32[m - 1, :2] - truth[m - 1, :2]).max()) for m in marks]},
       "pose_diff_f32_storage_vs_f64_m": [float(np.abs(t32[:m, :2] - t64[:m, :2]).max()) for m in marks],
       "P_robot_rel_diff_f32_storage_vs_f64": [float((np.abs(t32[:m, 4:] - t64[:m, 4:]) / np.abs(t64[:m, 4:]).max()).max()) for m in marks],
       "plans": {"f64": i64, "f32_storage": i32}}
# the mixed downdate where it is offered (clamp 1e-8): against the fp64 run WITH THE SAME CLAMP
Fm = min(F, 300)
F_saved, F = F, Fm
t64e, truth_e, _ = run(srukf.STORAGE_F64, 1e-8)
tmx, _, _ = run(srukf.STORAGE_F32_MIXED, 1e-8)
mm = [m for m in marks if m <= Fm]
out["mixed_downdate_eps_1e-8"] = {"frames": mm, "pose_diff_vs_f64_same_clamp_m": [float(np.abs(tmx[:m, :2] - t64e[:m, :2]).max()) for m in mm],
                                  "pose_err_vs_truth_m": {"f64_same_clamp": [float(np.abs(t64e[m - 1, :2] - truth_e[m - 1, :2]).max()) for m in mm],
                                                          "f32_mixed": [float(np.abs(tmx[m - 1, :2] - truth_e[m - 1, :2]).max()) for m in mm]}}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", tag + ".json"), "w"), indent=1)
print(json.dumps(out, indent=1))
