"""Rank-aware replay, frames/s at several sizes: owners fold S^T S - U U^T (default where a worker owns at most two tiles) against
k_syrk over the kept rows in permuted order (SRUKF_RANK_FOLD=0; the only form with memory tiles) against the full k_syrk +
permutation pass (SRUKF_RANK_FUSED=0)."""
import sys, time, os
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
p = synth.scene_params()
for N in [int(x) for x in sys.argv[1].split(",")]:
    F = 200 if N <= 300 else 100
    sc = synth.make_scene(N, F + 20, seed=0, p=p)
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    f.run_frames_async(0, 20); f.synchronize()
    best = 0
    for rep in range(2):
        f.set_state(sc["X0"], sc["S0"]); f.run_frames_async(0, 20); f.synchronize()
        t0 = time.perf_counter(); f.run_frames_async(20, F); f.synchronize(); best = max(best, F / (time.perf_counter() - t0))
    print(f"N={N} FOLD={os.environ.get('SRUKF_RANK_FOLD')} FUSED={os.environ.get('SRUKF_RANK_FUSED')}: {best:.0f} frames/s", flush=True)
    f.close()
