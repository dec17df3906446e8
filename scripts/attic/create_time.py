import sys, time
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
p = synth.scene_params()
for i in range(6):
    t0 = time.perf_counter(); f = srukf.Filter(200, p); f.synchronize(); dt = time.perf_counter() - t0
    print(f"create {i}: {dt * 1e3:.1f} ms", flush=True)
    if i % 2: f.close()
