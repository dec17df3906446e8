"""Per-panel timeline of k_gmw_persist inside the REAL replay at N landmarks (diagnostic build of the library: the script builds
build/variants/libsrukf_hip_dbg.so with -DSRUKF_GMW_DBG (here, or on the GPU box), loads it instead of the product library, replays frames and
prints the pivot workgroup's time stamps of the last frame).  Ticks are 10 ns (s_memrealtime)."""
import os, subprocess, sys
sys.path.insert(0, ".")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "cv-monoslam_amd", "csrc")
dbgdir = os.path.join(ROOT, "gpurun_out", "dbgobj"); os.makedirs(dbgdir, exist_ok=True)
srcs = ["srukf_api", "srukf_step", "srukf_replay", "srukf_split", "srukf_batch", "srukf_map", "srukf_debug", "srukf_predict", "srukf_factor", "srukf_gmw_persist", "srukf_augment", "srukf_assoc", "srukf_mixed", "srukf_rank"]      # (csrc/Makefile: SRCS)
flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 -DSRUKF_GMW_DBG -w".split()
extra = {"srukf_gmw_persist": ["-Os", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]}      # (the Makefile's flags for that file: a representative timeline)
lib = os.path.join(ROOT, "build", "variants", "libsrukf_hip_dbg.so")          # built beforehand (hipcc cross-compiles without a GPU: the same flags as below) ...
if not os.path.exists(lib):                                                    # ... or here
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc"] + flags + extra.get(s, []) + ["-c", f"{csrc}/{s}.hip", "-o", f"{dbgdir}/{s}.o"]) for s in srcs]
    assert all(p.wait() == 0 for p in procs)
    lib = os.path.join(dbgdir, "libsrukf_hip_dbg.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [f"{dbgdir}/{s}.o" for s in srcs])
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
srukf.LIB_PATH = lib                                  # the diagnostic build instead of the product library (nothing has been loaded yet)
assert srukf._lib is None
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = synth.scene_params(); F = 60
sc = synth.make_scene(N, F, seed=0, p=p)
f = srukf.Filter(N, p)
for kv in sys.argv[3:]:                                # key=value pairs for srukf_debug_set (e.g. split_fold=0)
    f.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
f.run_frames(0, 20)
f.debug_gmw_stamps()                                   # arm
f.run_frames(20, 20)
st = f.debug_gmw_stamps()
Tp = int(sys.argv[2]) if len(sys.argv) > 2 else 10
print("p: afterA afterB afterF1 afterC1 | poll_begin poll_end | pivot_done w1_done w3_done | iter_end | since prev start")
prev = None
for pnl in range(Tp):
    t = st[2048 + 8 * pnl: 2048 + 8 * pnl + 8].astype(np.int64); u = st[2048 + 8 * (pnl + 64): 2048 + 8 * (pnl + 64) + 8].astype(np.int64)
    d = lambda x: int(x - t[0]) if x else -1
    print(f"p={pnl:02d}: {d(t[1]):6d} {d(t[2]):6d} {d(t[3]):6d} {d(t[4]):6d} | {d(t[5]):6d} {d(t[6]):6d} | {d(u[0]):6d} {d(u[1]):6d} {d(u[2]):6d} | {d(t[7]):6d} | {int(t[0] - prev) if prev is not None else 0}")
    prev = t[0]

print("factor 1, since afterB: pivot wave done, T wave done, wave 1 done, wave 3 done | factor 2, since afterC1: pivot wave done, T wave done, waves 1 / 3 done")
for pnl in range(Tp):
    t = st[2048 + 8 * pnl: 2048 + 8 * pnl + 8].astype(np.int64); u = st[2048 + 8 * (pnl + 64): 2048 + 8 * (pnl + 64) + 8].astype(np.int64); w = st[2048 + 8 * (pnl + 192): 2048 + 8 * (pnl + 192) + 8].astype(np.int64)
    a = lambda x, ref: int(x - ref) if x else -1
    print(f"p={pnl:02d}: {a(u[4], t[2]):6d} {a(u[5], t[2]):6d} {a(u[6], t[2]):6d} {a(u[7], t[2]):6d} | {a(u[0], t[4]):6d} {a(w[0], t[4]):6d} {a(u[1], t[4]):6d} {a(u[2], t[4]):6d}")
g = lambda p_, s_: int(st[2048 + 8 * p_ + s_])
t0 = g(128, 0)
print("pivot: entry 0, head ready %d, loop end %d   (ticks of 10 ns since the pivot's entry)" % (g(128, 1) - t0, g(128, 2) - t0))
for nm, q in (("first worker", 129), ("last worker", 130)):
    print(f"{nm}: entry {g(q, 0) - t0}, own tiles formed {g(q, 1) - t0}, head seen {g(q, 2) - t0}, done {g(q, 3) - t0}")
print("first helper: entry %d, out %d; last helper (by index) out %d" % (g(131, 0) - t0, g(131, 1) - t0, g(131, 2) - t0))
print("head job 0 (critical tile): entry %d, wave 0 after S^T S part %d, after its U U^T part %d, wave 3 after its part %d, after the split-K sum %d, stores issued %d, landed %d" % tuple(g(132, q) - t0 for q in (0, 1, 2, 6, 3, 4, 5)))
crit = [g(140 + q, 0) - t0 for q in range(32) if g(140 + q, 0)]
print("critical head tiles landed (ticks since the pivot's entry):", crit)

