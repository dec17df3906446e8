"""Round-2 parity probe (GPU box): prints device-vs-oracle differences for the cases the round-1 verdict listed."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
from oracle import oracle as O


def run(N, F, p, mode, seed=1, tag=""):
    sc = synth.make_scene(N, F, seed=seed, p=p)
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
    o = O.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
    for t in range(F):
        f.predict_motion(sc["odo"][t], sc["odo"][t + 1]); o.predict_motion(sc["odo"][t], sc["odo"][t + 1])
        h, Si, vis = f.predict_measurement(); ho, Sio, viso = o.predict_measurement()
        f.update(sc["z"][t], sc["matched"][t], mode=mode); o.update(sc["z"][t], sc["matched"][t], 1, 0, mode)
        X, S = f.get_state(); Xo, So = o.get_state()
        P, Po = S.T @ S, So.T @ So
        print(f"{tag} N={N} mode={mode} t={t}: |dh|={np.abs(h-ho).max():.2e} |dSi|={np.abs(np.abs(Si)-np.abs(Sio)).max():.2e} vis_eq={np.array_equal(vis,viso)} "
              f"|dX|={np.abs(X-Xo).max():.2e} |dP|={np.abs(P-Po).max():.2e} |P|={np.abs(Po).max():.2e} clamp={o.clamp_stats()}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["wt", "def"]
    if "wt" in which:
        for wt in (1, 2):
            for mode in (0, 1):
                p = synth.scene_params(); p["weight_type"] = wt
                run(8, 4, p, mode, tag=f"wt{wt}")
            p = synth.scene_params(); p["weight_type"] = wt
            run(20, 3, p, 1, tag=f"wt{wt}")
    if "def" in which:
        p = synth.default_params()
        run(8, 3, p, 0, tag="defaults")
        run(8, 3, p, 1, tag="defaults")
