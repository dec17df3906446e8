"""Timing of the two halves of the k_pxy2 launch alone (results are garbage while a half is skipped): run under rocprofv3 --kernel-trace."""
import sys
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N = 200
p = synth.scene_params(); F = 100
sc = synth.make_scene(N, F, seed=0, p=p)
for skip in (0, 1, 2):
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    f.debug_set("use_graph", 0)
    srukf.debug_set_global("pxy2_skip", skip)
    try:
        f.run_frames_async(0, 12); f.synchronize()
    except Exception as e:
        print("skip", skip, "flagged (expected with garbage):", str(e)[:80])
    f.close()
srukf.debug_set_global("pxy2_skip", 0)
print("done")
