"""A/B of the XCD-aware tile assignment of the persistent launch (srukf_debug_set "tile_xcd") at N landmarks: HIP-event times per launch class
(eager, 24 frames after 8) and frames/s of a 200-frame graph replay."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = synth.scene_params(); F = 260; sc = synth.make_scene(N, F, seed=0, p=p)
ref = None
for xcd in (0, 1, 0, 1):
    srukf.debug_set_global("tile_xcd", xcd)
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    f.set_profiling(1)
    f.run_frames_async(0, 8); f.synchronize(); f.profile_reset()
    f.run_frames_async(8, 24); f.synchronize()
    pr = f.profile(); f.set_profiling(0)
    f.set_state(sc["X0"], sc["S0"]); f.prepare_frames(200)
    f.run_frames_async(0, 40); f.synchronize()
    best = 1e9
    for rep in range(3):
        f.set_state(sc["X0"], sc["S0"]); f.run_frames_async(0, 40); f.synchronize()
        t0 = time.perf_counter(); f.run_frames_async(40, 200); f.synchronize(); best = min(best, time.perf_counter() - t0)
    X, S = f.get_state()
    if ref is None: ref = (X, S)
    same = np.array_equal(X, ref[0]) and np.array_equal(S, ref[1])
    print(f"tile_xcd={xcd}: {200 / best:8.1f} frames/s | " + "  ".join(f"{k} {v['ms'] / 24 * 1e3:.1f}" for k, v in pr.items() if v["launches"]) + f" | state identical to the first run: {same}")
