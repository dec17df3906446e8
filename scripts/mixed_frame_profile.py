"""One block of staged frames at N landmarks with the chosen storage mode, for rocprofv3 --kernel-trace --stats (and frames/s of a graph-replayed block without the profiler).
  python scripts/mixed_frame_profile.py [N] [f64|f32|mixed] [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
mode = sys.argv[2] if len(sys.argv) > 2 else "mixed"
F = int(sys.argv[3]) if len(sys.argv) > 3 else 60
p = synth.scene_params()
sc = synth.make_scene(N, F + 12, seed=0, p=p)
f = srukf.Filter(N, p)
if mode == "mixed":
    f.debug_allow_mixed(True); f.set_storage(srukf.STORAGE_F32_MIXED)
elif mode == "f32":
    f.set_storage(srukf.STORAGE_F32)
f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
f.run_frames(0, 6)
f.prepare_frames(F)
f.run_frames_async(6, 6); f.synchronize()
t0 = time.perf_counter()
f.run_frames_async(12, F); f.synchronize()
dt = time.perf_counter() - t0
print(f"N = {N} {mode}: {F / dt:.1f} frames/s, {dt / F * 1e6:.1f} us per frame; plan fuse {f.debug_get('plan_fuse')} red_perm {f.debug_get('plan_red_perm')} split {f.debug_get('split_form')}")
f.close()
