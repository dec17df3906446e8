"""Timing of map changes at N = 200 (GPU box): repeated add / delete cycles after the pools are warm."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
p = synth.scene_params(); N = 200
sc = synth.make_scene(N, 3, seed=0, p=p)
f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
f.predict_motion(sc["odo"][0], sc["odo"][1]); f.predict_measurement(); f.update(sc["z"][0], sc["matched"][0])
rng = np.random.default_rng(1)
for it in range(4):
    uv = np.column_stack([rng.uniform(60, 580, 5), rng.uniform(60, 420, 5)])
    t = time.perf_counter(); f.add_landmarks(uv); ta = time.perf_counter() - t
    td = []
    for k in range(5):
        t = time.perf_counter(); f.delete_landmark(f.N - 1); td.append(time.perf_counter() - t)
    print(f"cycle {it}: add 5 landmarks {ta*1e3:.2f} ms; delete x5 {[round(x*1e3,2) for x in td]} ms; N={f.N}", flush=True)
