"""Diagnostic: B filters in SRUKF_GPU_SHARED at N = 200 with the fused motion step on / off: frames/s, flagged filters, clamp info."""
import sys, os, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N, K, W = 200, 64, 16
for fused in (1, 0):
    for B in (1, 3):
        fs = []
        for b in range(B):
            sc = synth.make_scene(N, W + K, seed=0, p=synth.scene_params(), obs_seed=5000 + b)
            f = srukf.Filter(N, sc["params"], device=0); f.set_exclusive(srukf.GPU_SHARED); f.debug_set("fused_motion", fused)
            f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
        for f in fs: f.run_frames_async(0, W)
        msgs = []
        for f in fs:
            try: f.synchronize()
            except Exception as e: msgs.append(("warm", str(e)[:160], f.clamp_info()))
        t0 = time.perf_counter()
        for k0 in range(0, K, 16):
            for f in fs: f.run_frames_async(W + k0, 16)
        for f in fs:
            try: f.synchronize()
            except Exception as e: msgs.append(("timed", str(e)[:160], f.clamp_info()))
        dt = time.perf_counter() - t0
        print(f"fused_motion={fused} B={B}: {B * K / dt:.0f} frames/s aggregate; flagged: {msgs}", flush=True)
        # against a solo exclusive run of filter 0's sequence
        sc = synth.make_scene(N, W + K, seed=0, p=synth.scene_params(), obs_seed=5000)
        g = srukf.Filter(N, sc["params"], device=0); g.debug_set("fused_motion", 0); g.set_state(sc["X0"], sc["S0"]); g.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        g.run_frames(0, W + K)
        Xg, Sg = g.get_state(); Xf, Sf = fs[0].get_state()
        print("   filter 0 vs solo classic: max |dX|", np.abs(Xg - Xf).max(), " max |dP|", np.abs(Sg.T @ Sg - Sf.T @ Sf).max(), flush=True)
        for f in fs: f.close()
        g.close()
