"""scene.bin + odo.txt for the C++ hosts (cslam_replay, cslam_step_bench, cslam_replay_multi):  python scripts/make_step_scene.py N F out_dir"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

synth = ge.load_package().synth
N, F, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.makedirs(out, exist_ok=True)
p = synth.scene_params()
sc = synth.make_scene(N, F, seed=0, p=p, obs_seed=1000)
with open(os.path.join(out, "scene.bin"), "wb") as fh:
    fh.write(struct.pack("ii", N, F))
    fh.write(np.array([p["a1"], p["a2"], p["a3"], p["a4"]], dtype=np.float64).tobytes())
    fh.write(np.ascontiguousarray(sc["X0"]).tobytes()); fh.write(np.ascontiguousarray(sc["S0"]).tobytes()); fh.write(np.ascontiguousarray(sc["z"]).tobytes())
with open(os.path.join(out, "odo.txt"), "w") as fh:                     # the reference's odometry text format (SLAM.cpp:475)
    for i, (x, y, th) in enumerate(sc["odo"]):
        fh.write(f"{i + 1} : {0.1 * i:.3f} {float(x)!r} {float(y)!r} {float(th)!r}\n")
print(out)
