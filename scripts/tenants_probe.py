"""Aggregate frames/s of B filters in SRUKF_GPU_SHARED at N = 200 against the number of persistent launches that share the GPU ("shared_tenants") and
the tiles per worker up to which the owners form their tiles of S^T S - U U^T themselves ("fold_tiles_pct": 106 = one tile per worker, 200 = two).
  python scripts/tenants_probe.py [tenants,tenants,...] [B,B,...] [pct,pct,...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N, K, W = 200, 96, 16
tenants_l = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 3, 4]
B_l = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [3, 4, 6, 8]
pct_l = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [106, 200]
scs = [synth.make_scene(N, W + K, seed=0, p=synth.scene_params(), obs_seed=5000 + b) for b in range(max(B_l))]
for pct in pct_l:
    srukf.debug_set_global("fold_tiles_pct", pct)
    for tenants in tenants_l:
        srukf.debug_set_global("shared_tenants", tenants)
        for B in B_l:
            fs = []
            for b in range(B):
                sc = scs[b]
                f = srukf.Filter(N, sc["params"], device=0); f.set_exclusive(srukf.GPU_SHARED)
                f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
            srukf.run_frames_batch(fs, 0, W)
            rates = []
            for rep in range(3):
                t0 = time.perf_counter(); srukf.run_frames_batch(fs, W, K); rates.append(B * K / (time.perf_counter() - t0))
            ab = sum(f.debug_get("gmw_aborts") for f in fs)
            print(f"fold_pct={pct} tenants={tenants} B={B}: best {max(rates):.0f} frames/s aggregate  reps {[round(r) for r in rates]}  aborts {ab}", flush=True)
            for f in fs: f.close()
