"""Long run of the batched replay: B filters (one map, own measurement noise) x F frames (the reference's CAPACITY = 3000) at N = 200 through
srukf_run_frames_batch in blocks; per block the aggregate rate, flagged frames, pose error against the noise-free track; at the end filter 0 and filter B-1 against
the same sequences replayed ALONE (bit for bit).  Writes gpurun_out/<tag>.json.
  python scripts/batch_soak.py [B] [F] [block] [tag]"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
blk = int(sys.argv[3]) if len(sys.argv) > 3 else 500
tag = sys.argv[4] if len(sys.argv) > 4 else "r04_batch_soak"
N = 200
p = synth.scene_params()
scs = [synth.make_scene(N, F, seed=0, p=p, obs_seed=9000 + b) for b in range(B)]
fs = []
for sc in scs:
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
out = {"B": B, "frames": F, "block": blk, "N": N, "blocks": []}
trajs = []
for k0 in range(0, F, blk):
    n = min(blk, F - k0)
    t0 = time.perf_counter(); tr = srukf.run_frames_batch(fs, k0, n); dt = time.perf_counter() - t0
    trajs.append(tr)
    err = float(max(np.abs(tr[b][:, :2] - scs[b]["odo"][k0 + 1:k0 + n + 1, :2]).max() for b in range(B)))
    out["blocks"].append({"first": k0, "frames": n, "aggregate_frames_per_s": round(B * n / dt, 1), "max_pose_err_vs_noise_free_track_m": err,
                          "flagged": int(sum(f.debug_get("clamp_rows") + f.debug_get("gmw_aborts") for f in fs))})
    print(out["blocks"][-1], flush=True)
full = np.concatenate(trajs, axis=1)
same = True
for b in (0, B - 1):
    g = srukf.Filter(N, p); g.set_state(scs[b]["X0"], scs[b]["S0"]); g.stage_sequence(scs[b]["odo"], scs[b]["z"], scs[b]["matched"])
    ts = np.vstack([g.run_frames(k0, min(blk, F - k0)) for k0 in range(0, F, blk)])
    Xs, Ss = g.get_state(); Xb, Sb = fs[b].get_state()
    same = same and bool(np.array_equal(ts, full[b]) and np.array_equal(Xs, Xb) and np.array_equal(Ss, Sb))
    g.close()
out["filters_0_and_last_bit_identical_to_solo_runs_over_the_whole_run"] = same
rates = [b["aggregate_frames_per_s"] for b in out["blocks"]]
out["median_aggregate_frames_per_s"] = float(np.median(rates)); out["min_aggregate_frames_per_s"] = float(min(rates))
json.dump(out, open(f"gpurun_out/{tag}.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "blocks"}))
