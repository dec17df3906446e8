"""Several filters of a size that takes the split form (N = 400) in ONE process, run one after the other: every one must keep the split form (mode 0, nothing abandoned).
Before the stream-pair probe of split_ensure the second filter stream pair shared a hardware queue and it fell back to per-panel launches.
python scripts/multi_context_split.py [filters] [use_graph]"""
import sys
sys.path.insert(0, ".")
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N, F, B = 400, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = synth.scene_params()
scs = [synth.make_scene(N, F, seed=0, p=p, obs_seed=7100 + b) for b in range(B)]
fs = []
for sc in scs:
    f = srukf.Filter(N, p)
    if len(sys.argv) > 2: f.debug_set("use_graph", int(sys.argv[2]))
    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
for k in range(3):
    for b, f in enumerate(fs):
        f.run_frames(2 * k, 2)
        print("filter", b, "frames", 2 * k, "aborts", f.debug_get("gmw_aborts"), "mode", f.debug_get("gmw_shared"), "split", f.debug_get("split_form"), flush=True)
