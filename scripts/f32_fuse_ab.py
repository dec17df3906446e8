"""fp32 storage in "fused tail" mode (srukf_debug_set "f32_fuse") against the launch sequence with k_quantize / k_rank_round / k_traj behind the tail:
same rounding points -> agreement at fp64 rounding level; the stored state is float-representable; frames/s."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
for N in [int(a) for a in sys.argv[1:]] or [200, 500]:
    p = synth.scene_params(); F = 70; K = 40; sc = synth.make_scene(N, F, seed=0, p=p)
    out = {}
    for ff in (0, 1):
        f = srukf.Filter(N, p); f.set_storage(srukf.STORAGE_F32); f.debug_set("f32_fuse", ff)
        f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        traj = np.vstack([f.run_frames(0, 3), f.run_frames(3, 5)])
        X, S = f.get_state(); X32, S32 = f.get_state_f32()
        rep = np.array_equal(X, X32.astype(np.float64)) and np.array_equal(np.triu(S), np.triu(S32).astype(np.float64)) and np.all(np.tril(S, -1) == 0.0)
        f.set_state(sc["X0"], sc["S0"]); f.prepare_frames(K)
        f.run_frames_async(0, 20); f.synchronize()
        best = 1e9
        for r_ in range(3):
            f.set_state(sc["X0"], sc["S0"]); f.run_frames_async(0, 20); f.synchronize()
            t0 = time.perf_counter(); f.run_frames_async(20, K); f.synchronize(); best = min(best, time.perf_counter() - t0)
        out[ff] = (traj, X, S.T @ S)
        print(f"N={N} f32_fuse={ff}: {K / best:8.1f} frames/s, state float-representable: {rep}, aborts {f.debug_get('gmw_aborts')} clamp {f.debug_get('clamp_rows')}", flush=True)
    a, b = out[0], out[1]
    print(f"   max |d traj pose| {np.abs(a[0][:, :4] - b[0][:, :4]).max():.2e}  |dX| {np.abs(a[1] - b[1]).max():.2e}  |dP| {np.abs(a[2] - b[2]).max():.2e}")
