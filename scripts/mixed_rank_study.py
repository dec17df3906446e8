"""Row g (BASELINE configs[4]: "fp32 SRUKF with mixed-precision sqrt-S downdate, tolerance study"): the mixed downdate — S^T S - U U^T formed on the fp32 matrix pipe
(SRUKF_STORAGE_F32_MIXED) — in the RANK-AWARE form the product runs (round 6: only the kept pivots are factored, the structurally null ones, whose 1e-13 clamp an
fp32-formed G cannot resolve, are not touched), at the reference's EPSILON = 1e-13, next to the fp64 filter and fp32 storage with FP64 arithmetic; and, for comparison,
the full-rank form of the mode that rounds 2 and 5 studied ("mixed_rank" 0).  Pose difference against the fp64 run, error against the truth, frames that had to be
repeated on the exact path, and the smallest kept pivot at the checkpoints.
  python scripts/mixed_rank_study.py [N] [frames] [tag]   -> gpurun_out/<tag>.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
F = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
tag = sys.argv[3] if len(sys.argv) > 3 else f"r06_mixed_rank_n{N}"
marks = [m for m in (1, 10, 30, 100, 300, 1000, 1500, 2000, 2500, 3000) if m <= F]
BLK = 50


def run(storage, mixed_rank=1, f64_robot=1, slow_limit_s=20.0):
    p = synth.scene_params()
    sc = synth.make_scene(N, F, seed=0, p=p)
    f = srukf.Filter(N, p)
    if storage == srukf.STORAGE_F32_MIXED:
        f.debug_allow_mixed(True)
    if storage != srukf.STORAGE_F64:
        f.set_storage(storage)
    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    if storage == srukf.STORAGE_F32_MIXED:
        f.debug_set("mixed_rank", mixed_rank); f.debug_set("mixed_f64_robot", f64_robot)
    rows, minpiv, stopped, t_run = [], {}, None, 0.0
    r = int(f.debug_get("plan_kept"))
    for a in range(0, F, BLK):
        t0 = time.time()
        try:
            rows.append(f.run_frames(a, min(BLK, F - a)))
        except srukf.SrukfError as e:
            stopped = {"frame": a, "error": str(e)[:120]}
            break
        dt = time.time() - t0
        t_run += dt
        if not np.isfinite(rows[-1]).all():
            stopped = {"frame": a, "error": "non-finite trajectory"}
            break
        done = a + len(rows[-1])
        if done in marks or a == 0:
            D = f.debug_copy("D", f.n)                       # pivots of the last factorisation, permuted order: the first r are the kept ones
            minpiv[str(done)] = float(np.min(D[:r])) if r > 0 else float(np.min(D))
        if dt > slow_limit_s:
            stopped = {"frame": done, "error": f"a block of {BLK} frames took {dt:.1f} s: flagged frames are being repeated column by column"}
            break
    tr = np.vstack(rows) if rows else np.zeros((0, 8))
    info = {"kept_pivots": r, "exact_frames": int(f.debug_get("exact_frames")), "frames_run": int(len(tr)), "stopped": stopped,
            "frames_per_s": round(len(tr) / t_run, 1) if t_run > 0 else None, "min_kept_pivot": minpiv,
            "plan": {k: int(f.debug_get("plan_" + k)) for k in ("red_perm", "fold", "motion", "fuse", "persist")}, "split_form": int(f.debug_get("split_form"))}
    f.close()
    return tr, sc["odo"][1:F + 1], info


def curve(t, ref, truth):
    mm = [m for m in marks if m <= len(t)]
    return {"frames": mm,
            "pose_diff_vs_f64_m": [float(np.abs(t[:m, :2] - ref[:m, :2]).max()) for m in mm],
            "pose_err_vs_truth_m": [float(np.abs(t[m - 1, :2] - truth[m - 1, :2]).max()) for m in mm],
            "P_robot_rel_diff_vs_f64": [float((np.abs(t[:m, 4:] - ref[:m, 4:]) / np.abs(ref[:m, 4:]).max()).max()) for m in mm]}


t64, truth, i64 = run(srukf.STORAGE_F64)
out = {"workload": f"N = {N}, benchmark sequence (seed 0), EPSILON = 1e-13 (the reference's), {F} frames through srukf_run_frames in blocks of {BLK}", "frames": marks,
       "f64": {"pose_err_vs_truth_m": [float(np.abs(t64[m - 1, :2] - truth[m - 1, :2]).max()) for m in marks], "info": i64}}
for name, st, kw in (("f32_storage", srukf.STORAGE_F32, {}),
                     ("f32_mixed_rank_aware_robot_and_anchor_tiles_f64", srukf.STORAGE_F32_MIXED, {"mixed_rank": 1, "f64_robot": 1}),
                     ("f32_mixed_rank_aware_every_tile_f32", srukf.STORAGE_F32_MIXED, {"mixed_rank": 1, "f64_robot": 0}),
                     ("f32_mixed_full_rank", srukf.STORAGE_F32_MIXED, {"mixed_rank": 0})):
    t, _, info = run(st, **kw)
    out[name] = curve(t, t64, truth)
    out[name]["info"] = info
    print(name, json.dumps(out[name]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", tag + ".json"), "w"), indent=1)
print(json.dumps(out, indent=1))
