"""Per-kernel times of the staged replay (eager launches with HIP events) at N landmarks."""
import sys
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = synth.scene_params(); F = 40
sc = synth.make_scene(N, F, seed=0, p=p)
f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
f.run_frames(0, 10)
f.set_profiling(1); f.profile_reset()
f.run_frames(10, 20)
pr = f.profile()
print(N, {k: round(v["ms"] / 20 * 1e3, 1) for k, v in pr.items() if v["launches"]}, "sum", round(sum(v["ms"] for v in pr.values()) / 20 * 1e3, 1))
