""""table" / "fused tail" mode on the path where the owners do not fold (srukf_debug_set "table_perm"): agreement with the previous launch sequence
(k_project_motion, k_pxy, ...) and frames/s, at the sizes given (fp64)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
for N in [int(a) for a in sys.argv[1:]] or [100, 300, 500]:
    p = synth.scene_params(); F = 70; K = 40; sc = synth.make_scene(N, F, seed=0, p=p)
    out = {}
    for tp in (0, 1):
        f = srukf.Filter(N, p); f.debug_set("table_perm", tp)
        f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        traj = np.vstack([f.run_frames(0, 3), f.run_frames(3, 5)])
        X, S = f.get_state()
        f.set_state(sc["X0"], sc["S0"]); f.prepare_frames(K)
        f.run_frames_async(0, 20); f.synchronize()
        best = 1e9
        for rep in range(3):
            f.set_state(sc["X0"], sc["S0"]); f.run_frames_async(0, 20); f.synchronize()
            t0 = time.perf_counter(); f.run_frames_async(20, K); f.synchronize(); best = min(best, time.perf_counter() - t0)
        f.set_profiling(1); f.set_state(sc["X0"], sc["S0"]); f.run_frames_async(0, 4); f.synchronize(); f.profile_reset(); f.run_frames_async(4, 12); f.synchronize()
        pr = f.profile()
        out[tp] = (traj, X, S.T @ S)
        print(f"N={N} table_perm={tp}: {K / best:8.1f} frames/s | " + "  ".join(f"{k} {v['ms'] / 12 * 1e3:.1f}" for k, v in pr.items() if v["launches"]), flush=True)
    a, b = out[0], out[1]
    print(f"   max |d traj pose| {np.abs(a[0][:, :4] - b[0][:, :4]).max():.2e}  |dX| {np.abs(a[1] - b[1]).max():.2e}  |dP| {np.abs(a[2] - b[2]).max():.2e}")
