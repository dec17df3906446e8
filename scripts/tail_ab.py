"""A/B timing of the tail fold at N = 200 (graph replay of 200 frames, 3 repetitions): tail_fold 0 (table mode), 1 (tail mode),
2 (write-through rows + counters, no projection jobs: timing only)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N = 200; p = synth.scene_params(); F = 260; sc = synth.make_scene(N, F, seed=0, p=p)
for fold in (0, 1, 2, 2 + 4, 2 + 8, 2 + 4 + 8, 0):
    f = srukf.Filter(N, p); f.debug_set("tail_fold", fold)
    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    f.prepare_frames(200)
    f.run_frames_async(0, 40); f.synchronize() if hasattr(f, "synchronize") else None
    best = 1e9
    for rep in range(3):
        f.set_state(sc["X0"], sc["S0"])
        f.run_frames_async(0, 40); f.synchronize()
        t0 = time.perf_counter(); f.run_frames_async(40, 200); f.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"tail_fold={fold}: {200 / best:8.1f} frames/s  ({best / 200 * 1e6:6.1f} us per frame)  aborts {f.debug_get('gmw_aborts')} clamp {f.debug_get('clamp_rows')}")
