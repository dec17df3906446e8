"""The drop-in path over the reference's capacity: F frames through srukf_predict_motion / srukf_predict_measurement / srukf_update (next odometry announced, host association =
the scene's z / matched) against the staged replay of the same sequence — state after every block compared bit for bit.   python scripts/step_soak.py [N] [F] [storage] [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
F = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
storage = sys.argv[3] if len(sys.argv) > 3 else "f64"
out = sys.argv[4] if len(sys.argv) > 4 else None
p = synth.scene_params()
sc = synth.make_scene(N, F + 2, seed=0, p=p)


def make():
    f = srukf.Filter(N, p)
    if storage == "f32":
        f.set_storage(srukf.STORAGE_F32)
    f.set_state(sc["X0"], sc["S0"])
    return f


# (both start from the state three replayed frames leave: the first frame of a jointly initialised state takes launch sequences of its own on either path)
W = 3
w = make(); w.stage_sequence(sc["odo"], sc["z"], sc["matched"]); w.run_frames(0, W); X3, S3 = w.get_state(); w.close()
a = make(); a.set_state(X3, S3); a.stage_sequence(sc["odo"], sc["z"], sc["matched"])
b = make(); b.set_state(X3, S3)
block, equal_blocks, t_step = 250, 0, 0.0
for f0 in range(W, F, block):
    a.run_frames(f0, min(block, F - f0))
    t0 = time.perf_counter()
    for t in range(f0, min(f0 + block, F)):
        b.predict_motion_next(sc["odo"][t + 1], sc["odo"][t + 2])
        b.predict_motion(sc["odo"][t], sc["odo"][t + 1])
        h, Si, vis = b.predict_measurement()
        b.update(sc["z"][t], sc["matched"][t])
    t_step += time.perf_counter() - t0
    Xa, Sa = a.get_state(); Xb, Sb = b.get_state()
    same = bool(np.array_equal(Xa, Xb) and np.array_equal(Sa, Sb))
    equal_blocks += same
    if not same:
        print("block", f0, "max |dX|", np.abs(Xa - Xb).max(), "max |dS|", np.abs(Sa - Sb).max(), flush=True)
res = {"workload": f"N = {N}, {F} frames, storage {storage}: step-wise API (look-ahead, Python host) against the staged replay, states compared after every {block} frames",
       "blocks": (F - W + block - 1) // block, "blocks_bit_identical": int(equal_blocks),
       "step_fast_frames": b.debug_get("step_fast"), "step_slow_frames": b.debug_get("step_slow"), "gmw_aborts": b.debug_get("gmw_aborts"), "clamp_rows": b.debug_get("clamp_rows"),
       "python_host_us_per_frame": round(t_step / (F - W) * 1e6, 1),
       "pose_err_vs_truth_m": float(np.abs(b.get_robot()[0][:2] - sc["odo"][F, :2]).max())}
a.close(); b.close()
print(json.dumps(res, indent=1))
if out:
    json.dump(res, open(out, "w"), indent=1)
