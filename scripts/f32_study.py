"""BASELINE configs[4] tolerance study: fp32 filter state (fp64 arithmetic) against the fp64 run of the same sequence.
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
for name, st in (("f64", srukf.STORAGE_F64), ("f32", srukf.STORAGE_F32)):
    f = srukf.Filter(N, p); f.set_storage(st); f.set_state(sc["X0"], sc["S0"])
    f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    f.run_frames(0, 4)                                      # warm-up + graph capture
    f.set_state(sc["X0"], sc["S0"])
    t = time.perf_counter(); traj[name] = f.run_frames(0, F); dt = time.perf_counter() - t
    out[name + "_frames_per_s"] = F / dt
d = np.sqrt(np.sum((traj["f32"][:, :2] - traj["f64"][:, :2]) ** 2, axis=1))
truth = sc["odo"][1:F + 1, :2]
out["pose_diff_f32_vs_f64_m"] = {"frame_1": float(d[0]), "frame_10": float(d[min(9, F - 1)]), "last": float(d[-1]), "max": float(d.max()),
                                 "rmse": float(np.sqrt(np.mean(d ** 2)))}
out["pose_rmse_vs_truth_m"] = {k: float(np.sqrt(np.mean(np.sum((traj[k][:, :2] - truth) ** 2, axis=1)))) for k in traj}
out["robot_cov_rel_diff_last"] = float(np.abs(traj["f32"][-1, 4:] - traj["f64"][-1, 4:]).max() / np.abs(traj["f64"][-1, 4:]).max())
print(json.dumps(out))
