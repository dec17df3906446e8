"""BASELINE configs[4] (N = 500): per-kernel times of the staged replay (eager launches, HIP events) and frames/s of the graph replay,
fp64 and fp32 storage."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
p = synth.scene_params(); F = 140
sc = synth.make_scene(N, F, seed=0, p=p)
for storage in ("f64", "f32"):
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
    if storage == "f32": f.set_storage(srukf.STORAGE_F32)
    f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    f.run_frames(0, 10)
    f.set_profiling(1); f.profile_reset()
    f.run_frames(10, 10)
    pr = f.profile(); f.set_profiling(0)
    print(N, storage, {k: round(v["ms"] / 10 * 1e3, 1) for k, v in pr.items() if v["launches"]}, "sum", round(sum(v["ms"] for v in pr.values()) / 10 * 1e3, 1), flush=True)
    f.prepare_frames(100)
    f.run_frames_async(20, 10); f.synchronize()
    t0 = time.perf_counter(); f.run_frames_async(30, 100); f.synchronize(); dt = time.perf_counter() - t0
    print(N, storage, f"{100 / dt:.0f} frames/s; null directions {f.null_directions()}; pose err vs truth {np.abs(f.get_robot()[0][:2] - sc['odo'][130, :2]).max():.2e}", flush=True)
    f.close()
