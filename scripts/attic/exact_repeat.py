import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); srukf = pkg.srukf; synth = pkg.synth
p = synth.default_params()
N = 200
sc = synth.make_scene(N, 20, seed=1, p=p)
for rep in range(4):
    f = srukf.Filter(N, p)
    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    ts = []
    e0 = 0
    for t in range(12):
        t0 = time.perf_counter(); f.run_frames(t, 1); dt = time.perf_counter() - t0
        e1 = f.debug_get("exact_frames")
        if e1 > e0: ts.append(round(dt * 1e3, 2))
        e0 = e1
    print("rep", rep, "flagged frame ms:", ts, flush=True)
    f.close()
