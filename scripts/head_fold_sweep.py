"""Where does the head fold (head tiles of S^T S - U U^T as helper workgroups of the persistent launch) pay?  frames/s of the default staged replay with the switch on / off.
  python scripts/head_fold_sweep.py N [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
srukf.debug_set_global("head_fold_free", 32)                   # (the sweep decides; 32 = one free CU per shader engine, below that the launch does not complete)
for N in [int(a) for a in sys.argv[1:]]:
    p = synth.scene_params()
    W, K = 10, 60
    sc = synth.make_scene(N, W + 3 * K, seed=0, p=p)
    row = []
    for hf in (1, 0):
        f = srukf.Filter(N, p)
        f.debug_set("head_fold", hf)
        f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        f.prepare_frames(K)
        f.run_frames_async(0, W); f.synchronize()
        best = 0.0
        for r in range(3):
            t0 = time.perf_counter()
            f.run_frames_async(W + r * K, K); f.synchronize()
            best = max(best, K / (time.perf_counter() - t0))
        plan = {k: f.debug_get("plan_" + k) for k in ("workers", "head_fold", "fold", "T", "Tp")}
        assert f.debug_get("gmw_aborts") == 0 and f.debug_get("gmw_shared") == 0
        row.append((hf, round(best, 1), plan))
        f.close()
    print(N, row, flush=True)
