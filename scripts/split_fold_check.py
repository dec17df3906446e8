"""Split fold (k_gmw_tiles_fold: the split form's tile launch forms the tiles of S^T S - U U^T) against the k_syrk launch in front of the pair: same state bit for bit, frames/s.
python scripts/split_fold_check.py [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
if os.environ.get('LIB'):
    srukf.load_library(os.path.join(ROOT, os.environ['LIB']))      # an A/B build (scripts/build_variants.sh)
for k_, v_ in (("fold_head", os.environ.get("FOLD_HEAD")), ("fold_force", os.environ.get("FOLD_FORCE"))):
    if v_:
        srukf.debug_set_global(k_, int(v_))      # measurements: head rows of the split fold / the fold where it does not fit
for arg in (sys.argv[1:] or ["400", "500:f32"]):
    N = int(arg.split(":")[0]); storage = arg.split(":")[1] if ":" in arg else "f64"
    F = int(os.environ.get('FRAMES', 24))
    p = synth.scene_params()
    sc = synth.make_scene(N, F, seed=0, p=p)
    res = {}
    for fold in ([int(v) for v in os.environ['SPLIT_FOLD'].split(',')] if 'SPLIT_FOLD' in os.environ else [0, 1]):
        f = srukf.Filter(N, p)
        if storage == "f32":
            f.set_storage(srukf.STORAGE_F32)
        f.debug_set("split_fold", fold)
        rates = []
        for rep in range(3):
            f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
            f.synchronize()
            t0 = time.perf_counter(); traj = f.run_frames(0, F); dt = time.perf_counter() - t0
            rates.append(F / dt)
        X, S = f.get_state()
        res[fold] = (X, S, traj)
        print(f"N {N} {storage} split_fold {fold}: split_form {f.debug_get('split_form')}  frames/s {[round(r) for r in rates]}  aborts {f.debug_get('gmw_aborts')} clamp_rows {f.debug_get('clamp_rows')} exact {f.debug_get('exact_frames')}", flush=True)
        f.close()
    if len(res) < 2:
        continue
    same = all(np.array_equal(a, b) for a, b in zip(res[0], res[1]))
    print(f"N {N} {storage}: bit-identical {same}; max |dS| {np.abs(res[0][1] - res[1][1]).max():.3e}", flush=True)
