"""CPU-only sweep of the algorithm-matched baseline (oracle/srukf_matched.c) on the GPU box's host: thread counts x
binding policies.  Each configuration runs in its own process (libgomp reads OMP_* at load time)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
import __graft_entry__ as ge
synth = ge.load_package().synth
from oracle import oracle as O
N, th, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p = synth.scene_params(); sc = synth.make_scene(N, F + 2, seed=0, p=p)
m = O.Matched(N, p, threads=th); m.set_state(sc["X0"], sc["S0"])
m.run_frames(sc["odo"][:3], sc["z"][:2], sc["matched"][:2])
t0 = time.perf_counter(); m.run_frames(sc["odo"][2:], sc["z"][2:], sc["matched"][2:]); dt = time.perf_counter() - t0
print(json.dumps({"fps": F / dt, "phase_ms": {k: round(v / (F + 2) * 1e3, 3) for k, v in m.phase_times().items()}}))
''' % ROOT
if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a")
    print("affinity:", len(os.sched_getaffinity(0)), "cpus")
    for bind in ({}, {"OMP_PROC_BIND": "close", "OMP_PLACES": "cores"}, {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"},
                 {"OMP_WAIT_POLICY": "passive"}):
        for th in (8, 16, 32, 64, 128):
            env = dict(os.environ, **bind)
            r = subprocess.run([sys.executable, "-c", CHILD, str(N), str(th), "16"], env=env, capture_output=True, text=True, timeout=600)
            print(bind, th, r.stdout.strip()[-300:] or r.stderr[-300:], flush=True)
