"""The staged replay at N = 200 with and without the gain fold (srukf_debug_set "gain_fold"): frames/s over 5 x 200 frames of captured graphs, best and all.
python scripts/fold_bench.py [library.so]   (default: the shipped library)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
N, F = 200, 200
p = synth.scene_params()
sc = synth.make_scene(N, F, seed=0, p=p)
lib = sys.argv[1] if len(sys.argv) > 1 else None
if lib:
    srukf.load_library(os.path.join(ROOT, lib))
if True:
    for fold in (0, 1):
        f = srukf.Filter(N, p)
        f.debug_set("gain_fold", fold)
        rates = []
        for rep in range(6):
            f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
            f.synchronize()
            t0 = time.perf_counter(); f.run_frames(0, F); dt = time.perf_counter() - t0
            rates.append(F / dt)
        print(f"{lib or 'shipped'} gain_fold {fold}: best {max(rates[1:]):.0f} frames/s  all {[round(r) for r in rates[1:]]}  fold_seqs {f.debug_get('fold_seqs')}", flush=True)
        f.close()
