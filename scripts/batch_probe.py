"""Aggregate frames/s of B filters through srukf_run_frames_batch at N = 200: the batched launches (one launch per stage for all filters, one stream) against
round 3's form (batch_wide 0: one stream per filter, one tenant per filter up to four).  Every repetition replays its own block of frames.
  python scripts/batch_probe.py [B,B,...] [wide,wide,...] [N] [groups,...] [eager]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
if os.environ.get("LIB"):                                        # an A/B build (scripts/build_variants.sh)
    srukf.load_library(os.path.join(os.getcwd(), os.environ["LIB"]))
arg = lambda i, d: [int(x) for x in sys.argv[i].split(",")] if len(sys.argv) > i else d
B_l, wide_l = arg(1, [2, 4, 8, 12, 16]), arg(2, [1, 0])
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
G_l = arg(4, [0])
if len(sys.argv) > 5 and sys.argv[5] == "eager":                # eager launches instead of graphs (rocprofv3 --pmc)
    srukf.debug_set_global("graphs", 0)
if os.environ.get("BATCH_XCD") is not None:                      # 0: k_syrk_b in the solo launch's tile order (A/B of round 6's per-XCD order)
    srukf.debug_set_global("batch_xcd", int(os.environ["BATCH_XCD"]))
if os.environ.get("BATCH_K128") is not None:                     # 0: one pass over G per panel (K = 64) instead of one per pair
    srukf.debug_set_global("batch_k128", int(os.environ["BATCH_K128"]))
K, W, R = 96, 16, 3
scs = [synth.make_scene(N, W + R * K, seed=0, p=synth.scene_params(), obs_seed=5000 + b) for b in range(max(B_l))]
for wide, G in [(w, g) for w in wide_l for g in (G_l if w else [0])]:
    srukf.debug_set_global("batch_wide", wide); srukf.debug_set_global("batch_groups", G)
    for B in B_l:
        if not wide and B > 8: continue
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
        print(f"N={N} wide={wide} groups={G} B={B}: median {np.median(rates):.0f} frames/s aggregate  reps {[round(r) for r in rates]}  flagged {ab}", flush=True)
        for f in fs: f.close()
