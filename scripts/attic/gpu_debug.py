"""First-contact diagnostics on the GPU box: stage-by-stage comparison against the oracle."""
import sys, os, time
if "torchfirst" in sys.argv:
    import torch; print("torch cuda available:", torch.cuda.is_available(), torch.version.hip)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import oracle as O
synth, srukf = pkg.synth, pkg.srukf
np.set_printoptions(linewidth=200, precision=4)

def gmw_checks():
    rng = np.random.default_rng(0)
    for n in (8, 40, 100, 200):
        A = rng.normal(size=(n, n)); G = A @ A.T + 0.1 * np.eye(n)
        S, D, hit = srukf.gmw(G)
        So, Do, Lo, ce, ct = O.gmw(G)
        print(f"gmw spd n={n}: |S^T S - G| {np.abs(S.T@S-G).max():.2e}  |S-So| {np.abs(S-So).max():.2e} |D-Do|rel {np.abs(D/Do-1).max():.2e} hit {hit}")
        S2, D2, hit2 = srukf.gmw(G, force_slow=True)
        print(f"   slow: |S-So| {np.abs(S2-So).max():.2e} hit {hit2}")
    # rank deficient PSD
    B = rng.normal(size=(30, 12)); G = B @ B.T
    S, D, hit = srukf.gmw(G); So, Do, Lo, ce, ct = O.gmw(G)
    print(f"gmw psd rank-def: |P-Po| {np.abs(S.T@S-So.T@So).max():.2e} hit {hit} oracle clamps eps={ce} theta={ct}")
    # indefinite -> theta clamp path
    G = rng.normal(size=(20, 20)); G = G + G.T
    S, D, hit = srukf.gmw(G); So, Do, Lo, ce, ct = O.gmw(G)
    print(f"gmw indefinite: |S-So| {np.abs(S-So).max():.2e} |D-Do| {np.abs(D-Do).max():.2e} hit {hit} oracle clamps eps={ce} theta={ct}")

def proj_checks():
    p = synth.scene_params(); rng = np.random.default_rng(1)
    sc = synth.make_scene(64, 1, seed=2, p=p)
    feat = sc['truth'] + rng.normal(0, 1e-3, sc['truth'].shape); feat[:, 5] = np.abs(feat[:, 5]) + 0.05
    pos = rng.normal(0, 0.05, (64, 3)); psi = rng.normal(0, 0.3, 64); err = rng.normal(0, 2, (64, 2))
    a = srukf.project(p, feat, pos, psi, err); b = O.project(p, feat, pos, psi, err)
    print("project: max |gpu-oracle|", np.abs(a - b).max())

def frame_checks(N, F, mode):
    p = synth.scene_params()
    sc = synth.make_scene(N, F, seed=3, p=p)
    f = srukf.Filter(N, p); f.set_state(sc['X0'], sc['S0'])
    o = O.Oracle(N, p); o.set_state(sc['X0'], sc['S0'])
    n = f.n
    for t in range(F):
        f.predict_motion(sc['odo'][t], sc['odo'][t+1]); o.predict_motion(sc['odo'][t], sc['odo'][t+1])
        X, S = f.get_state(); Xo, So = o.get_state()
        print(f"N={N} t={t} motion: |dX| {np.abs(X-Xo).max():.2e} |dP| {np.abs(S.T@S-So.T@So).max():.2e} (|P|max {np.abs(So.T@So).max():.2e})")
        h, Si, vis = f.predict_measurement(); ho, Sio, viso = o.predict_measurement()
        print(f"     meas: |dh| {np.abs(h-ho).max():.2e} |dSi| {np.abs(Si-Sio).max():.2e} |d|Si|| {np.abs(np.abs(Si)-np.abs(Sio)).max():.2e} vis eq {np.array_equal(vis, viso)}")
        f.update(sc['z'][t], sc['matched'][t], mode=mode); o.update(sc['z'][t], sc['matched'][t], 1, 0, mode)
        X, S = f.get_state(); Xo, So = o.get_state()
        print(f"     update(mode {mode}): |dX| {np.abs(X-Xo).max():.2e} |dP| {np.abs(S.T@S-So.T@So).max():.2e} pose err vs truth {np.abs(X[-4:-2]-sc['odo'][t+1,:2]).max():.2e}")
    pose, P4 = f.get_robot()
    Po = So.T @ So
    print("     get_robot: |dpose|", np.abs(pose - Xo[-4:]).max(), "|dP4|", np.abs(P4 - Po[-4:, -4:]).max())
    x6, P6 = f.get_landmark_block(1)
    print("     landmark block: ", np.abs(x6 - Xo[6:12]).max(), np.abs(P6 - Po[6:12, 6:12]).max())

def async_check(N, F):
    import torch
    p = synth.scene_params()
    sc = synth.make_scene(N, F, seed=4, p=p)
    f = srukf.Filter(N, p); f.set_state(sc['X0'], sc['S0'])
    f.stage_sequence(sc['odo'], sc['z'], sc['matched'])
    traj = torch.zeros(F, 8, dtype=torch.float64, device='cuda')
    t0 = time.time(); f.run_frames_async(0, F, srukf.UPDATE_BATCHED, traj.data_ptr()); f.synchronize(); dt = time.time() - t0
    o = O.Oracle(N, p); o.set_state(sc['X0'], sc['S0'])
    to = o.run_frames(sc['odo'], sc['z'], sc['matched'], 1)
    tg = traj.cpu().numpy()
    print(f"async N={N} F={F}: traj |dpose| {np.abs(tg[:, :4]-to[:, :4]).max():.2e} |dP| {np.abs(tg[:, 4:]-to[:, 4:]).max():.2e}  {F/dt:.1f} fps (eager)")

if __name__ == "__main__":
    what = sys.argv[1:] or ["gmw", "proj", "frames", "async"]
    if "gmw" in what: gmw_checks()
    if "proj" in what: proj_checks()
    if "frames" in what:
        frame_checks(8, 3, 1); frame_checks(8, 2, 0); frame_checks(20, 2, 1)
    if "async" in what: async_check(20, 10)
    if "perf" in what:
        p = synth.scene_params(); N = 200; F = 20
        sc = synth.make_scene(N, F, seed=0, p=p)
        f = srukf.Filter(N, p); f.set_state(sc['X0'], sc['S0']); f.stage_sequence(sc['odo'], sc['z'], sc['matched'])
        f.run_frames_async(0, 5); f.synchronize()
        t0 = time.time(); f.run_frames_async(5, 10); f.synchronize(); dt = time.time() - t0
        print(f"N=200 eager: {10/dt:.1f} fps")
        f.set_profiling(1); f.profile_reset(); f.run_frames_async(15, 5); f.synchronize()
        for k, v in f.profile().items():
            if v['launches']: print(f"  {k:16s} {v['ms']/5*1000:9.1f} us/frame  {v['launches']//5:4d} launches/frame  {v['alg_flops']/max(v['ms'],1e-9)/1e9:8.2f} TFLOP/s")
