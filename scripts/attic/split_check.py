"""Split form of the persistent factorisation (k_gmw_pivslab_persist + k_gmw_tiles_persist) against the memory-tile instance of k_gmw_persist, N >= 340 (GPU box):
state after F frames bit for bit, per-kernel times (eager, HIP events) and frames/s of the graph replay.  python scripts/split_check.py [N] [storage,..] [mem_split modes, e.g. 1,0 or 2,0] [rank_aware]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
storages = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f32", "f64"]
p = synth.scene_params(); F = 140
sc = synth.make_scene(N, F, seed=0, p=p)
modes = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 0]
rank_aware = int(sys.argv[4]) if len(sys.argv) > 4 else 1
for storage in storages:
    res = {}
    for split in modes:
        srukf.debug_set_global("mem_split", split)
        f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
        if not rank_aware: f.set_rank_aware(0)
        if storage == "f32": f.set_storage(srukf.STORAGE_F32)
        f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        t0 = time.perf_counter(); f.run_frames(0, 10); t1 = time.perf_counter() - t0
        X, S = f.get_state()
        f.set_profiling(1); f.profile_reset()
        f.run_frames(10, 10)
        pr = f.profile(); f.set_profiling(0)
        print(N, storage, "split" if split else "memtile", f"first 10 frames {t1 * 1e3:.0f} ms, aborts {f.debug_get('gmw_aborts')} mode {f.debug_get('gmw_shared')}", {k: round(v["ms"] / 10 * 1e3, 1) for k, v in pr.items() if v["launches"]}, flush=True)
        f.prepare_frames(100)
        f.run_frames_async(20, 10); f.synchronize()
        t0 = time.perf_counter(); f.run_frames_async(30, 100); f.synchronize(); dt = time.perf_counter() - t0
        X2, S2 = f.get_state()
        print(N, storage, "split" if split else "memtile", f"{100 / dt:.0f} frames/s; aborts {f.debug_get('gmw_aborts')} mode {f.debug_get('gmw_shared')}; pose err vs truth {np.abs(f.get_robot()[0][:2] - sc['odo'][130, :2]).max():.2e}", flush=True)
        res[split] = (X, S, X2, S2)
        f.close()
    a, b = res[modes[0]], res[modes[-1]]
    print(N, storage, "bit-identical after 10 frames:", bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])),
          " after 130:", bool(np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])),
          f" max |dS| {np.abs(a[3] - b[3]).max():.2e}", flush=True)
srukf.debug_set_global("mem_split", 1)
