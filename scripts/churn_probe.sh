#!/bin/bash
# where do the milliseconds of a map change go?  cslam_step_bench mode=facade churn=10 with set=timing:1 (srukf_debug_set(0, "timing", 1): phases of srukf_add_landmarks / srukf_delete_landmark on stderr)
#   bash scripts/churn_probe.sh [N] [frames]
N=${1:-200}; K=${2:-60}
python - <<PY
import struct, sys, os, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as ge
synth = ge.load_package().synth
N, F = $N, $K + 20
p = synth.scene_params(); sc = synth.make_scene(N, F, seed=0, p=p, obs_seed=1000)
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/churn_scene.bin", "wb") as fh:
    fh.write(struct.pack("ii", N, F)); fh.write(np.array([p["a1"], p["a2"], p["a3"], p["a4"]]).tobytes())
    fh.write(np.ascontiguousarray(sc["X0"]).tobytes()); fh.write(np.ascontiguousarray(sc["S0"]).tobytes()); fh.write(np.ascontiguousarray(sc["z"]).tobytes())
with open("gpurun_out/churn_odo.txt", "w") as fh:
    for i, (x, y, th) in enumerate(sc["odo"]):
        fh.write(f"{i + 1} : {0.1 * i:.3f} {float(x)!r} {float(y)!r} {float(th)!r}\n")
PY
cv-monoslam_amd/cslam_step_bench.bin gpurun_out/churn_scene.bin gpurun_out/churn_odo.txt mode=facade churn=10 frames=$K warmup=20 set=timing:1 2> gpurun_out/churn_timing.txt | cut -c1-1200
grep "map timing" gpurun_out/churn_timing.txt | tail -8
