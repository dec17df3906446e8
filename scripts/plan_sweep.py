"""Which launch plan does the default staged replay take at which landmark count?  (GPU box; writes gpurun_out/plan_sweep.json)

For every N of the sweep: a filter with the jointly initialised state of the synthetic scene, two staged frames through the default path, then the
plan_* keys of srukf_debug_get.  Prints the N at which any key changes: the sizes tests/test_gpu_parity_r5.py holds to the oracle on both sides."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
KEYS = ["T", "Tp", "tiles", "workers", "persist", "register_form", "tiles_per_worker", "fold", "head_fold", "red_perm", "motion", "fuse", "kept"]
SHOW = ["persist", "register_form", "tiles_per_worker", "fold", "head_fold", "red_perm", "motion", "fuse", "split_form"]


def plan_of(N, rank_aware=1, storage="f64"):
    p = synth.scene_params()
    sc = synth.make_scene(N, 2, seed=3, p=p)
    f = srukf.Filter(N, p)
    if not rank_aware:
        f.set_rank_aware(0)
    if storage == "f32":
        f.set_storage(srukf.STORAGE_F32)
    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    f.run_frames(0, 2)
    d = {k: f.debug_get("plan_" + k) for k in KEYS}
    d["split_form"] = f.debug_get("split_form"); d["gmw_aborts"] = f.debug_get("gmw_aborts"); d["gmw_shared"] = f.debug_get("gmw_shared")
    f.close()
    return d


if __name__ == "__main__":
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 520)
    out = {}
    for label, ra, st in (("rank_aware_f64", 1, "f64"), ("full_rank_f64", 0, "f64"), ("rank_aware_f32", 1, "f32")):
        rows, prev = {}, None
        for N in range(lo, hi + 1):
            d = plan_of(N, ra, st)
            rows[N] = d
            sig = tuple(d[k] for k in SHOW)
            if sig != prev:
                print(label, "N =", N, {k: d[k] for k in SHOW}, "T", d["T"], "Tp", d["Tp"], "tiles", d["tiles"], "workers", d["workers"], flush=True)
                prev = sig
        out[label] = rows
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "plan_sweep.json"), "w"))
