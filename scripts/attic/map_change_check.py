"""Map changes at the benchmark size (GPU box): augmentation of an N = 200 map by K = 40 landmarks against the oracle's
joint initialisation, the NEED_REORDER frame after it (rank check), deletion against the numpy marginal; timings."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
from oracle import oracle as O

pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
p = synth.scene_params()
N, K = 200, 40
sc = synth.make_scene(N, 3, seed=0, p=p)
f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
f.predict_motion(sc["odo"][0], sc["odo"][1]); f.predict_measurement(); f.update(sc["z"][0], sc["matched"][0])
X, S = f.get_state()
rng = np.random.default_rng(1)
uv = np.column_stack([rng.uniform(60, 580, K), rng.uniform(60, 420, K)])
t = time.perf_counter(); f.add_landmarks(uv); t_add = time.perf_counter() - t
Xa, Sa = f.get_state()
t = time.perf_counter(); Xo, So = O.joint_init(p, X, S, uv); t_or = time.perf_counter() - t
print(f"add_landmarks N={N}+{K}: device {t_add*1e3:.1f} ms (incl. context rebuild), oracle {t_or:.1f} s; |dX| {np.abs(Xa-Xo).max():.2e} |dP| {np.abs(Sa.T@Sa-So.T@So).max():.2e}")
f.predict_motion(sc["odo"][1], sc["odo"][2]); h, Si, vis = f.predict_measurement()
z = h + rng.normal(0, 0.5, h.shape)
t = time.perf_counter(); f.update(z, np.asarray(vis, dtype=np.int32), reorder=srukf.NEED_REORDER); t_ro = time.perf_counter() - t
Xr, Sr = f.get_state(); Pr = Sr.T @ Sr
ev = np.linalg.eigvalsh(Pr)
print(f"NEED_REORDER update n={f.n}: {t_ro*1e3:.1f} ms; rank(P) at 1e-9: {(ev > 1e-9).sum()} (n - 3K = {f.n - 3*K}); min eig {ev.min():.2e}; finite {np.isfinite(Xr).all()}")
t = time.perf_counter(); f.delete_landmark(17); t_del = time.perf_counter() - t
Xd, Sd = f.get_state()
keep = np.r_[0:6*17, 6*18:len(Xr)]
print(f"delete_landmark: {t_del*1e3:.1f} ms; |dP vs marginal| {np.abs(Sd.T@Sd - Pr[np.ix_(keep, keep)]).max():.2e}; X exact {np.array_equal(Xd, Xr[keep])}")
print("nan in device S:", np.isnan(Sa).sum(), " nan in oracle S:", np.isnan(So).sum(), " |dP| on finite:", np.nanmax(np.abs(Sa.T@Sa-So.T@So)))
