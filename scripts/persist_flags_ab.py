"""Compiler-flag A/B for srukf_gmw_persist.hip (built on the GPU box): each variant is linked with the in-tree objects of the other files into its own
library, loaded in a child process, and timed on a 200-frame graph replay at N = 200 (and, with a second argument, at N = 500 with fp32 storage)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "cv-monoslam_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "flagobj"); os.makedirs(out, exist_ok=True)
base = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 -w".split() + os.environ.get("AB_BASE", "").split()
SRC = os.environ.get("AB_SRC", "srukf_gmw_persist")
variants = {
    "default": [],
    "Os": ["-Os"],
    "Os+max-ilp": ["-Os", "-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "max-ilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
}
others = [f"{csrc}/{s}.o" for s in ("srukf_api", "srukf_predict", "srukf_factor", "srukf_gmw_persist", "srukf_augment", "srukf_assoc", "srukf_mixed", "srukf_rank") if s != SRC]
if os.environ.get("AB_DEFS"):
    variants = {"default": []}
    for dv in os.environ["AB_DEFS"].split(";"): variants[dv] = ["-D" + x for x in dv.split(",")]
child = r'''
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
srukf.LIB_PATH = sys.argv[1]
for N, st in ((200, 0), (500, 1)):
    if N == 500 and len(sys.argv) < 3: break
    p = synth.scene_params(); F = 260 if N == 200 else 70; K = 200 if N == 200 else 40
    sc = synth.make_scene(N, F, seed=0, p=p)
    f = srukf.Filter(N, p)
    if st: f.set_storage(srukf.STORAGE_F32)
    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); f.prepare_frames(K)
    f.run_frames_async(0, 20); f.synchronize()
    best = 1e9
    for rep in range(3):
        f.set_state(sc["X0"], sc["S0"]); f.run_frames_async(0, 20); f.synchronize()
        t0 = time.perf_counter(); f.run_frames_async(20, K); f.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"   N={N}: {K / best:8.1f} frames/s ({best / K * 1e6:.1f} us per frame)", flush=True)
'''
for name, fl in variants.items():
    obj = f"{out}/persist_{name}.o"; lib = f"{out}/libsrukf_{name}.so"
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + base + fl + ["-c", f"{csrc}/{SRC}.hip", "-o", obj], capture_output=True, text=True)
    if r.returncode: print(name, "does not compile:", r.stderr.strip().splitlines()[-1] if r.stderr else ""); continue
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others)
    print(name, flush=True)
    subprocess.run([sys.executable, "-c", child, lib] + sys.argv[1:2], cwd=ROOT)
