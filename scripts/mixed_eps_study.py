"""Mixed-precision downdate (SRUKF_STORAGE_F32_MIXED) against the fp64 run, as a function of the GMW clamp EPSILON.
  python scripts/mixed_eps_study.py [N] [frames]"""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
F = int(sys.argv[2]) if len(sys.argv) > 2 else 60
p = synth.scene_params(); sc = synth.make_scene(N, F, seed=0, p=p)
f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); ref = f.run_frames(0, F)
out = {"landmarks": N, "frames": F, "by_epsilon": {}}
for eps in (1e-13, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6):
    for name, st in (("f64", srukf.STORAGE_F64), ("f32_mixed", srukf.STORAGE_F32_MIXED)):
        q = dict(p); q["epsilon"] = eps
        g = srukf.Filter(N, q); g.debug_allow_mixed(True); g.set_storage(st); g.set_state(sc["X0"], sc["S0"]); g.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        try:
            t = g.run_frames(0, F)
            d = np.sqrt(np.sum((t[:, :2] - ref[:, :2]) ** 2, axis=1))
            out["by_epsilon"].setdefault(f"{eps:g}", {})[name] = {"pose_diff_vs_f64_eps1e-13_max_m": float(np.nanmax(d)) if np.isfinite(d).any() else None,
                                                                  "finite": bool(np.isfinite(t).all()), "last": float(d[-1])}
        except Exception as e:
            out["by_epsilon"].setdefault(f"{eps:g}", {})[name] = {"error": str(e)[:80]}
print(json.dumps(out))
