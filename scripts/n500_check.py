import sys, time; sys.path.insert(0, '/root/repo')
import torch, numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
for N, F in ((50, 40), (500, 6)):
    p = synth.scene_params()
    t = time.time(); sc = synth.make_scene(N, F, seed=0, p=p); print('scene', N, time.time() - t)
    f = srukf.Filter(N, p); f.set_state(sc['X0'], sc['S0']); f.stage_sequence(sc['odo'], sc['z'], sc['matched'])
    f.run_frames(0, 2)
    t = time.time(); traj = f.run_frames(2, F - 2); dt = time.time() - t
    print(f'N={N}: {(F-2)/dt:.1f} fps; pose err vs truth {np.abs(traj[:, :2] - sc["odo"][3:F+1, :2]).max():.2e}')
    X, S = f.get_state(); print('  finite', np.isfinite(X).all() and np.isfinite(S).all(), 'min diag', np.diag(S).min())
