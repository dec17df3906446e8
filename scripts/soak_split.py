"""Soak of the split form of the persistent factorisation (GPU box): N = 500, fp32 storage (BASELINE configs[4]), F frames of the benchmark sequence in blocks of 250;
frames/s per block (an abandoned launch or a flagged frame would show as a collapse), launches abandoned, finiteness, pose error against the ground truth.
python scripts/soak_split.py [F] [tag] -> gpurun_out/<tag>.json"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
F = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
tag = sys.argv[2] if len(sys.argv) > 2 else "r04_split_soak"
N = 500
p = synth.scene_params(); sc = synth.make_scene(N, F, seed=0, p=p)
f = srukf.Filter(N, p); f.set_storage(srukf.STORAGE_F32); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
rates, errs, flagged = [], [], 0
for b in range(0, F, 250):
    cnt = min(250, F - b)
    t = time.perf_counter(); tr = f.run_frames(b, cnt); dt = time.perf_counter() - t
    rates.append(cnt / dt); errs.append(float(np.abs(tr[:, :2] - sc["odo"][b + 1:b + cnt + 1, :2]).max()))
    flagged += int(f.debug_get("gmw_aborts") > 0)
    print(f"frames {b}-{b + cnt - 1}: {rates[-1]:.0f} frames/s, pose err vs truth max {errs[-1]:.2e}, split form {f.debug_get('split_form')}, mode {f.debug_get('gmw_shared')}", flush=True)
X, S = f.get_state()
out = {"workload": f"N = {N}, fp32 storage, {F} frames of the benchmark sequence through srukf_run_frames in blocks of 250 (graph replay), split form of the factorisation",
       "frames_per_s_per_block": [round(r, 1) for r in rates], "frames_per_s_median": float(np.median(rates)), "max_pose_err_vs_truth_m_per_block": errs,
       "split_form_at_the_end": int(f.debug_get("split_form")), "fell_back_to_per_panel_launches": int(f.debug_get("gmw_shared") == 2), "blocks_with_abandoned_launches": flagged,
       "state_finite": bool(np.isfinite(X).all() and np.isfinite(S).all()), "null_directions": f.null_directions(),
       "split_fold_sequences_enqueued": int(f.debug_get("split_fold_seqs")), "exact_frames": int(f.debug_get("exact_frames"))}
json.dump(out, open(f"gpurun_out/{tag}.json", "w"), indent=1)
print(json.dumps(out))
