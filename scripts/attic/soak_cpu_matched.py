"""Long-horizon run of the algorithm-matched CPU port (oracle/srukf_matched.c: full-rank blocked GMW with the theta clamp verified
after the fact, exact orc_gmw fallback) on the bench scene: does the REFERENCE ALGORITHM itself meet the theta clamp
(modifiedCholeskyDecomposition, SLAM.cpp:2279-2285) in a long replay, and where?  Settles whether what the device's full-rank
form shows at N = 500 (flagged from frame ~1 649 on) is the algorithm or a device artefact.  CPU only; test infrastructure.

  python scripts/soak_cpu_matched.py N F [threads] -> profiles/r03_soak_cpu_n{N}.json (+ .npy trajectory beside it)

Stops early once `max_fallbacks` frames in a row went through the exact fallback (each is n^3/3 unblocked flops)."""
import json, os, sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
from oracle import oracle as O
pkg = ge.load_package(); synth = pkg.synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
F = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 0
max_fallbacks = 8
p = synth.scene_params()
sc = synth.make_scene(N, F, seed=0, p=p)
m = O.Matched(N, p, threads); m.set_state(sc["X0"], sc["S0"])
traj = np.full((F, 8), np.nan)
first_fb, fb_prev, run, t0 = None, 0, 0, time.time()
stopped = None
MARKS = (100, 500, 1000, 1500, 1649, 2000, 2500, 3000)
null = [6 * k + c for k in range(1, N) for c in range(3)]
null_at = {}


def null_stats():
    X, S = m.get_state(); e = np.sum(S * S, axis=1); d = np.diag(S)
    return dict(max_row_energy=float(e[null].max()), max_offdiag=float(np.abs(S[null] - np.diag(d)[null]).max()), min_diag=float(d.min()))
for f in range(F):
    traj[f] = m.run_frames(sc["odo"][f:f + 2], sc["z"][f:f + 1], sc["matched"][f:f + 1])[0]
    fb = m.clamp_fallbacks()
    if fb > fb_prev:
        if first_fb is None:
            first_fb = f
            null_info = null_stats()
        run += 1
    else:
        run = 0
    fb_prev = fb
    if f + 1 in MARKS:
        null_at[str(f + 1)] = null_stats()
    if run >= max_fallbacks:
        stopped = f; break
    if f % 100 == 99:
        print(f"frame {f + 1}: {(f + 1) / (time.time() - t0):.1f} frames/s, fallbacks {fb}, pose err vs truth {np.abs(traj[f, :2] - sc['odo'][f + 1, :2]).max():.2e}", flush=True)
done = (stopped + 1) if stopped is not None else F
err = np.abs(traj[:done, :2] - sc["odo"][1:done + 1, :2]).max(axis=1)
marks = [q for q in MARKS if q <= done]
out = dict(N=N, frames_requested=F, frames_run=done, threads=m.threads, isa=m.isa, seconds=time.time() - t0,
           first_theta_clamp_frame=first_fb, clamp_fallbacks=fb_prev, stopped_after_consecutive_fallbacks=stopped is not None,
           null_rows_at_first_clamp=null_info if first_fb is not None else None, null_rows_at=null_at,
           pose_at={str(q): traj[q - 1, :4].tolist() for q in marks}, P_robot_at={str(q): traj[q - 1, 4:].tolist() for q in marks},
           pose_err_vs_truth_at={str(q): float(err[q - 1]) for q in marks}, pose_err_vs_truth_max=float(err.max()),
           what="oracle/srukf_matched.c, full-rank form (every pivot factored, null pivots end at EPSILON), scene seed 0 = bench.py's")
os.makedirs("profiles", exist_ok=True)
json.dump(out, open(f"profiles/r03_soak_cpu_n{N}.json", "w"), indent=1)
np.save(f"profiles/r03_soak_cpu_n{N}_traj.npy", traj[:done])
print(json.dumps(out, indent=1))
