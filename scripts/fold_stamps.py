"""Time stamps inside k_pxy2 with the gain fold (diagnostic build: bash scripts/build_variants.sh srukf_factor.hip SRUKF_FOLD_DBG 1): when do the motion reduction, the
statistics groups and the tile pairs end, when do the gain jobs start and end?   python scripts/fold_stamps.py [N]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
srukf.load_library(os.path.join(ROOT, "build", "variants", "libsrukf_hip_SRUKF_FOLD_DBG_1.so"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = synth.scene_params()
sc = synth.make_scene(N, 40, seed=0, p=p)
f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
f.run_frames(0, 30)
st = f.debug_copy("fold_dbg", 8192).view(np.uint64).astype(np.int64)
nmt, nbt = (2 * N + 63) // 64, (6 * N + 4 + 63) // 64
t0 = st[0]
us = lambda v: (v - t0) * 0.01
print(f"motion: reduce done {us(st[1]):.2f}, flag {us(st[2]):.2f}, robot tiles seen {us(st[3]):.2f}, commit done {us(st[4]):.2f}")
for g in range(nmt):
    print(f"stats group {g}: a slice-0 job starts {us(st[16 + 4 * g]):.2f}, flag raised {us(st[17 + 4 * g]):.2f}")
pair, waits, end = [], [], []
for mt in range(nmt):
    for bt in range(nbt):
        b = 512 + 4 * (mt * nbt + bt)
        if st[b + 1]:
            pair.append(us(st[b + 1])); waits.append(us(st[b + 2])); end.append(us(st[b + 3]))
pair, waits, end = np.array(pair), np.array(waits), np.array(end)
print(f"{len(pair)} tile pairs: complete at {pair.min():.2f} .. {pair.max():.2f} (median {np.median(pair):.2f}); waits over at {waits.min():.2f} .. {waits.max():.2f} (median {np.median(waits):.2f}); "
      f"gain jobs end at {end.min():.2f} .. {end.max():.2f} (median {np.median(end):.2f}); job length median {np.median(end - waits):.2f}, max {(end - waits).max():.2f}")
f.close()
