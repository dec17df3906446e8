"""BASELINE configs[4] tolerance study: fp32 filter state (fp64 arithmetic) and the mixed-precision downdate (fp32 state,
S^T S - U U^T on the fp32 matrix pipe, FP64 pivots / trailing updates) against the fp64 run of the same sequence.
  python scripts/f32_study.py [N] [frames]      (GPU box; prints one JSON line)"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge

pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
F = int(sys.argv[2]) if len(sys.argv) > 2 else 60
p = synth.scene_params()
sc = synth.make_scene(N, F, seed=0, p=p)
out = {"landmarks": N, "state_dim": 6 * N + 4, "frames": F}
traj = {}
MIXED_EPS = 1e-8                                           # the mixed mode needs the clamp above the fp32 noise floor (include/srukf.h)
pm = dict(p); pm["epsilon"] = MIXED_EPS
out["mixed_epsilon"] = MIXED_EPS
for name, st in (("f64", srukf.STORAGE_F64), ("f32", srukf.STORAGE_F32), ("f32_mixed", srukf.STORAGE_F32_MIXED)):
    f = srukf.Filter(N, pm if name == "f32_mixed" else p); f.set_storage(st); f.set_state(sc["X0"], sc["S0"])
    f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    f.run_frames(0, 4)                                      # warm-up + graph capture
    f.set_state(sc["X0"], sc["S0"])
    t = time.perf_counter(); traj[name] = f.run_frames(0, F); dt = time.perf_counter() - t
    out[name + "_frames_per_s"] = F / dt
truth = sc["odo"][1:F + 1, :2]
for name in ("f32", "f32_mixed"):
    d = np.sqrt(np.sum((traj[name][:, :2] - traj["f64"][:, :2]) ** 2, axis=1))
    out[f"pose_diff_{name}_vs_f64_m"] = {"frame_1": float(d[0]), "frame_10": float(d[min(9, F - 1)]), "last": float(d[-1]), "max": float(d.max()),
                                         "rmse": float(np.sqrt(np.mean(d ** 2)))}
    out[f"robot_cov_rel_diff_last_{name}"] = float(np.abs(traj[name][-1, 4:] - traj["f64"][-1, 4:]).max() / np.abs(traj["f64"][-1, 4:]).max())
out["pose_rmse_vs_truth_m"] = {k: float(np.sqrt(np.mean(np.sum((traj[k][:, :2] - truth) ** 2, axis=1)))) for k in traj}
if len(sys.argv) > 3:                                       # per-kernel times of the mixed run
    f = srukf.Filter(N, pm); f.set_storage(srukf.STORAGE_F32_MIXED); f.set_state(sc["X0"], sc["S0"])
    f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); f.run_frames(0, 2)
    f.set_profiling(1); f.profile_reset(); f.run_frames(2, 4)
    out["mixed_kernels_us_per_frame"] = {k: round(v["ms"] / 4 * 1e3, 1) for k, v in f.profile().items() if v["launches"]}
    g = srukf.Filter(N, p); g.set_state(sc["X0"], sc["S0"])
    g.stage_sequence(sc["odo"], sc["z"], sc["matched"]); g.run_frames(0, 2)
    g.set_profiling(1); g.profile_reset(); g.run_frames(2, 4)
    out["f64_kernels_us_per_frame"] = {k: round(v["ms"] / 4 * 1e3, 1) for k, v in g.profile().items() if v["launches"]}
print(json.dumps(out))
