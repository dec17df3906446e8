"""Why do staged frames at some sizes abandon their persistent launch?  (round 5: the plan sweep found gmw_shared = 2 at N = 267 .. 275)
  python scripts/abort_probe.py N [head_fold] [frames] [head_fold_free: CUs the head fold asks for beside pivot + workers]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
if len(sys.argv) > 5:
    srukf.load_library(sys.argv[5])                            # an A/B build (scripts/build_variants.sh), e.g. with a longer wait bound
import time

N = int(sys.argv[1]); head_fold = int(sys.argv[2]) if len(sys.argv) > 2 else 1; F = int(sys.argv[3]) if len(sys.argv) > 3 else 5
p = synth.scene_params()
sc = synth.make_scene(N, F, seed=3, p=p)
free = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if free:
    srukf.debug_set_global("head_fold_free", free)
f = srukf.Filter(N, p)
f.debug_set("head_fold", head_fold); f.debug_set("use_graph", 0)
f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
keys = ["plan_T", "plan_Tp", "plan_tiles", "plan_workers", "plan_persist", "plan_fold", "plan_head_fold", "plan_red_perm", "plan_fuse", "split_form"]
print("N", N, "head_fold", head_fold, {k: f.debug_get(k) for k in keys})
dbg = len(sys.argv) > 5 and "DBG" in sys.argv[5]
if dbg:
    f.debug_gmw_stamps()                                      # arms the time stamps of the diagnostic build (10 ns ticks of s_memrealtime)
for t in range(F):
    t0 = time.perf_counter()
    try:
        f.run_frames_async(t, 1); f.synchronize()
        st = "ok %.1f ms" % ((time.perf_counter() - t0) * 1e3)
    except srukf.SrukfError as e:
        st = f"{e}"
    code = f.debug_get("abort_code")
    print(" frame", t, st, {k: f.debug_get(k) for k in ("gmw_aborts", "clamp_rows", "gmw_shared", "plan_persist", "plan_fold", "plan_head_fold")}, "clamp_info", f.clamp_info(),
          "abort site", code >> 32, "workgroup", (code & 0xffffffff) - 1, "| diagnostic build: helpers started / finished", f.debug_get("pad1"), f.debug_get("pad2"),
          "head_done / head_crit at the last exit", f.debug_get("pad3"), f.debug_get("pad4"), "grid", f.debug_get("pad5"), "helpers", f.debug_get("pad6"))
    if dbg:
        st_ = f.debug_gmw_stamps().astype(np.int64)
        def row(p):
            r = st_[2048 + 8 * p:2048 + 8 * p + 8]
            return r
        base = row(128)[0]
        def rel(r):
            return [round((int(v) - int(base)) / 100.0, 1) if v else None for v in r]
        print("   stamps (us after the pivot's start): pivot [start, head tiles seen, end]", rel(row(128))[:3], "| worker 1 [start, own tiles formed, head tiles seen, end]", rel(row(129))[:4],
              "| last worker", rel(row(130))[:4], "| helpers [first start, first end, last end]", rel(row(131))[:3], "| helper job 0 [start, S part, U part, split-K sum, stores, barrier]", rel(row(132))[:7])
    if not st.startswith("ok"):
        print("   (%.1f ms)" % ((time.perf_counter() - t0) * 1e3))
        break
f.close()
