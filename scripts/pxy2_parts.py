"""k_pxy2's two populations alone (srukf_debug_set "pxy2_skip": 1 = no statistics / motion jobs, 2 = no tiles; timing only — the frames' results are garbage):
HIP-event time of the launch over 6 eager frames at N = 200 / 500."""
import sys
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
for N in [int(a) for a in sys.argv[1:]] or [200]:
    p = synth.scene_params(); sc = synth.make_scene(N, 12, seed=0, p=p)
    for skip in (0, 1, 2, 0):
        f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        f.debug_set("pxy2_skip", skip); f.set_profiling(1)
        try:
            f.run_frames_async(0, 6); f.synchronize()
        except Exception as e:
            pass
        pr = f.profile()
        print(f"N={N} pxy2_skip={skip}: k_pxy2 {pr['k_pxy2']['ms'] / max(1, pr['k_pxy2']['launches']) * 1e3:.1f} us over {pr['k_pxy2']['launches']} launches", flush=True)
        f.debug_set("pxy2_skip", 0)
