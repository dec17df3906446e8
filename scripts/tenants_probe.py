"""Aggregate frames/s of B filters through srukf_run_frames_batch at N = 200 against the cap of persistent launches that share the GPU ("batch_tenants"), the
tiles per worker up to which the owners form their tiles of S^T S - U U^T themselves ("fold_tiles_pct": 106 = one tile per worker, 200 = two) and the
assignment of two tiles to a worker ("pair_adjacent").  Every repetition replays its own block of frames.
  python scripts/tenants_probe.py [tenants,...] [B,...] [pct,...] [pair,...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N, K, W, R = 200, 96, 16, 3
arg = lambda i, d: [int(x) for x in sys.argv[i].split(",")] if len(sys.argv) > i else d
tenants_l, B_l, pct_l, pair_l = arg(1, [2, 3, 4]), arg(2, [3, 4, 6]), arg(3, [106, 200]), arg(4, [0, 1])
scs = [synth.make_scene(N, W + R * K, seed=0, p=synth.scene_params(), obs_seed=5000 + b) for b in range(max(B_l))]
for pct in pct_l:
    srukf.debug_set_global("fold_tiles_pct", pct)
    for pair in pair_l:
        srukf.debug_set_global("pair_adjacent", pair)
        for tenants in tenants_l:
            srukf.debug_set_global("batch_tenants", tenants)
            for B in B_l:
                fs = []
                for b in range(B):
                    sc = scs[b]
                    f = srukf.Filter(N, sc["params"], device=0)
                    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
                srukf.run_frames_batch(fs, 0, W)
                rates = []
                for rep in range(R):
                    t0 = time.perf_counter(); srukf.run_frames_batch(fs, W + rep * K, K); rates.append(B * K / (time.perf_counter() - t0))
                ab = sum(f.debug_get("gmw_aborts") + f.debug_get("clamp_rows") for f in fs)
                print(f"fold_pct={pct} pair={pair} tenants<={tenants} B={B}: median {np.median(rates):.0f} frames/s aggregate  reps {[round(r) for r in rates]}  flagged {ab}", flush=True)
                for f in fs: f.close()
