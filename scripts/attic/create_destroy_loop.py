"""Create / run / destroy cycles of a filter that takes the split form (N = 400): stream pairs are probed at every creation, rejected candidates destroyed — no leak, no abandoned launch.
python scripts/create_destroy_loop.py [cycles]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N, F = 400, 3
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
p = synth.scene_params(); sc = synth.make_scene(N, F, seed=0, p=p)
keep = []
t_create = []
for i in range(cycles):
    t0 = time.perf_counter(); f = srukf.Filter(N, p); t_create.append(time.perf_counter() - t0)
    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    f.run_frames(0, F)
    assert f.debug_get("split_form") == 1 and f.debug_get("gmw_shared") == 0, (i, f.debug_get("split_form"), f.debug_get("gmw_shared"))
    if i % 5 == 0: keep.append(f)           # some stay alive: the stream population changes
    else: f.close()
print(f"{cycles} cycles, {len(keep)} filters kept alive: all on the split form, nothing abandoned; create {np.median(t_create) * 1e3:.2f} ms median, {max(t_create) * 1e3:.2f} ms max")
for f in keep: f.close()
