"""fp32 storage with the rank-aware refactorisation: frames/s and agreement with the fp64 run (configs[4])."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
p = synth.scene_params()
for N, F in ((200, 200), (500, 100)):
    sc = synth.make_scene(N, F, seed=0, p=p)
    res = {}
    for name, st, ra in (("f64 rank-aware", srukf.STORAGE_F64, True), ("f32 storage rank-aware", srukf.STORAGE_F32, True), ("f32 storage full-rank", srukf.STORAGE_F32, False)):
        f = srukf.Filter(N, p); f.set_rank_aware(ra); f.set_storage(st); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        f.run_frames(0, F); f.set_state(sc["X0"], sc["S0"])
        t0 = time.perf_counter(); tr = f.run_frames(0, F); dt = time.perf_counter() - t0
        res[name] = tr
        print(f"N={N} {name}: {F / dt:.0f} frames/s, null directions {f.null_directions()}, max |dpose| vs f64 {np.abs(tr[:, :4] - res['f64 rank-aware'][:, :4]).max():.2e}, rel |dP_robot| {np.abs(tr[:, 4:] - res['f64 rank-aware'][:, 4:]).max() / np.abs(res['f64 rank-aware'][:, 4:]).max():.2e}", flush=True)
        f.close()
