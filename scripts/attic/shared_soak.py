"""B filters in SRUKF_GPU_SHARED mode through a long block of frames (srukf_run_frames_batch): no abandoned launch, no flagged
frame, every trajectory equal to the filter's own run with the GPU to itself."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N, B, F = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 3, int(sys.argv[2]) if len(sys.argv) > 2 else 1500
p = synth.scene_params()
scs = [synth.make_scene(N, F, seed=0, p=p, obs_seed=9000 + b) for b in range(B)]
ref = []
for sc in scs[:2]:
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); ref.append(f.run_frames(0, F)); f.close()
fs = []
for sc in scs:
    f = srukf.Filter(N, p); f.set_exclusive(srukf.GPU_SHARED); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
srukf.run_frames_batch(fs, 0, 20)
t0 = time.perf_counter(); tr = srukf.run_frames_batch(fs, 20, F - 20); dt = time.perf_counter() - t0
print(f"B={B}: {B * (F - 20) / dt:.0f} frames/s aggregate over {F - 20} frames each; clamp info {[f.clamp_info() for f in fs]}")
for b in range(min(B, 2)):
    print(" filter", b, "equal to its exclusive run:", np.array_equal(tr[b], ref[b][20:]), " max |d|", np.abs(tr[b] - ref[b][20:]).max())
