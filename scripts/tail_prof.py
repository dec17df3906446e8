"""Per-launch-class HIP-event times (eager launches, 24 frames after 8) at N = 200 for the tail-fold variants of srukf_debug_set "tail_fold"."""
import sys
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N = 200; p = synth.scene_params(); F = 60; sc = synth.make_scene(N, F, seed=0, p=p)
folds = [a for a in sys.argv[1:]] or ["0", "1"]
for fold in folds:
    cap = 0
    if ":" in fold: fold, cap = fold.split(":")
    fold, cap = int(fold), int(cap)
    f = srukf.Filter(N, p); f.debug_set("tail_fold", fold); f.debug_set("tail_cap", cap)
    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    f.set_profiling(1)
    try:
        f.run_frames_async(0, 8); f.synchronize(); f.profile_reset()
        f.run_frames_async(8, 24); f.synchronize()
    except Exception as e:
        print(f"tail_fold={fold}: FAILED {e}"); continue
    pr = f.profile()
    print(f"tail_fold={fold} cap={cap}: " + "  ".join(f"{k} {v['ms'] / 24 * 1e3:.1f}" for k, v in pr.items() if v["launches"]))
