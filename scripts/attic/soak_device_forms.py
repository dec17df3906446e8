"""Long-horizon replay on the device, both refactorisation forms (rank-aware = default, full-rank = srukf_set_rank_aware(0), "the
reference's own formulation": every null pivot factored and clamped to EPSILON), same staged inputs as scripts/soak_cpu_matched.py
(bench scene, seed 0).  Records per form the first frame flagged for the exact path (theta clamp / null-direction check), the pose at
the marks, the null rows' residue, and the differences to the CPU port's trajectory (profiles/r03_soak_cpu_n{N}_traj.npy) where it exists.

  python scripts/soak_device_forms.py N F -> profiles/r03_soak_n{N}.json"""
import json, os, sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
F = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
BLOCK = 50
MARKS = (100, 500, 1000, 1500, 1649, 2000, 2500, 3000)
p = synth.scene_params()
sc = synth.make_scene(N, F, seed=0, p=p)
null = [6 * k + c for k in range(1, N) for c in range(3)]
cpu_path = f"profiles/r03_soak_cpu_n{N}_traj.npy"
cpu = np.load(cpu_path) if os.path.exists(cpu_path) else None
out = dict(N=N, frames=F, scene_seed=0, cpu_trajectory=cpu_path if cpu is not None else None, forms={})
trajs = {}
for name, on in (("rank_aware", True), ("full_rank", False)):
    f = srukf.Filter(N, p); f.set_rank_aware(on); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    traj = np.full((F, 8), np.nan); first_flag = None; flagged = 0; null_at = {}; exact_frames = 0
    t0 = time.time(); b = 0
    while b < F:
        cnt = min(BLOCK, F - b)
        try:
            f.run_frames_async(b, cnt); f.synchronize()
            traj[b:b + cnt] = np.nan                           # (the async form keeps no host trajectory: take it from the sync form below)
            ok = True
        except Exception as e:                                  # SRUKF_ERR_CLAMP_PENDING: a frame of the block needs the exact path
            ok = False
            fr, row = f.clamp_info()
            if first_flag is None:
                first_flag = dict(frame=int(fr), row=int(row), landmark=int(row) // 6, component=int(row) % 6, message=str(e)[:200])
            flagged += 1
        b += cnt
        if not ok:
            break                                              # the state behind a flagged frame is not valid in the async form: stop this form here
        for q in MARKS:
            if b - cnt < q <= b and q == b:
                X, S = f.get_state(); e_ = np.sum(S * S, axis=1); d = np.diag(S)
                null_at[str(q)] = dict(max_row_energy=float(e_[null].max()), max_offdiag=float(np.abs(S[null] - np.diag(d)[null]).max()), min_diag=float(d.min()),
                                       pose=X[-4:].tolist())
    dt = time.time() - t0
    # trajectory of the valid part through the synchronous form on a fresh filter (same frames, graph replay; it recovers flagged frames by itself)
    g = srukf.Filter(N, p); g.set_rank_aware(on); g.set_state(sc["X0"], sc["S0"]); g.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    upto = F if first_flag is None else min(F, first_flag["frame"] + 60)
    t1 = time.time()
    for b2 in range(0, upto, BLOCK):
        c2 = min(BLOCK, upto - b2)
        traj[b2:b2 + c2] = g.run_frames(b2, c2)
        if time.time() - t1 > 120:                              # every frame behind the clamp goes column by column: bounded
            upto = b2 + c2; break
    trajs[name] = traj
    os.makedirs("gpurun_out", exist_ok=True)
    np.save(f"gpurun_out/r03_soak_n{N}_{name}_traj.npy", traj[:upto])
    err = np.abs(traj[:upto, :2] - sc["odo"][1:upto + 1, :2]).max(axis=1)
    marks = [q for q in MARKS if q <= upto]
    rec = dict(frames_async_valid=b if first_flag is None else first_flag["frame"], first_flagged_frame=first_flag, frames_per_s_async=(b / dt),
               frames_with_trajectory=upto, null_rows_at=null_at,
               pose_at={str(q): traj[q - 1, :4].tolist() for q in marks}, pose_err_vs_truth_at={str(q): float(err[q - 1]) for q in marks},
               pose_err_vs_truth_max=float(np.nanmax(err)))
    if cpu is not None:
        m = min(len(cpu), upto)
        d = np.abs(traj[:m, :4] - cpu[:m, :4]).max(axis=1)
        rec["pose_diff_vs_cpu_port_at"] = {str(q): float(d[q - 1]) for q in MARKS if q <= m}
        rec["pose_diff_vs_cpu_port_max"] = float(d.max()); rec["P_robot_diff_vs_cpu_port_max"] = float(np.abs(traj[:m, 4:] - cpu[:m, 4:]).max())
    out["forms"][name] = rec
    f.close(); g.close()
m = min(np.isfinite(trajs["rank_aware"][:, 0]).sum(), np.isfinite(trajs["full_rank"][:, 0]).sum())
d = np.abs(trajs["rank_aware"][:m, :4] - trajs["full_rank"][:m, :4]).max(axis=1)
out["pose_diff_rank_aware_vs_full_rank_at"] = {str(q): float(d[q - 1]) for q in MARKS if q <= m}
out["pose_diff_rank_aware_vs_full_rank_max"] = float(d.max()) if m else None
os.makedirs("profiles", exist_ok=True)
json.dump(out, open(f"profiles/r03_soak_n{N}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
