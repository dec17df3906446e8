"""Kernel-trace fodder: 30 step-API frames (k_pxy without the statistics riding on it; k_meas_partial / k_meas_final as their own launches)
and 60 replay frames at N = 200.  Run under rocprofv3 --kernel-trace --stats."""
import sys
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = synth.scene_params(); F = 100
sc = synth.make_scene(N, F, seed=0, p=p)
f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
for t in range(30):
    f.predict_motion(sc["odo"][t], sc["odo"][t + 1]); f.predict_measurement(); f.update(sc["z"][t], sc["matched"][t])
f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
f.run_frames(30, 60)
print("done")
