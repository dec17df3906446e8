"""N > 1 path on CPU: two processes, gloo backend (the GPU run uses the same code with nccl = RCCL).
Checks the only collectives the path has: broadcast of the shared initial map from rank 0 at the
start, all-gather of the trajectories and max-over-ranks of the wall time at the end."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo(tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    assert all(o["same_map"] for o in out)                       # broadcast delivered rank 0's map everywhere
    assert out[0]["wall_max"] == out[1]["wall_max"] == 0.2       # max over ranks
    t0, t1 = np.array(out[0]["trajs"]), np.array(out[1]["trajs"])
    assert np.array_equal(t0, t1) and t0.shape == (2, 5, 8)      # all-gather: same view on both ranks
    assert not np.array_equal(t0[0], t0[1])                      # independent measurement noise per rank
    assert out[0]["z0"] != out[1]["z0"]
    assert np.abs(t0[0][:, :2] - t0[1][:, :2]).max() < 1e-3      # ...but the same underlying trajectory


def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 2` (no launcher, as the driver calls it) must start two rank processes itself.
    --launch-selftest runs the launcher + collectives on gloo with no filter; n_gpus is what the process group saw."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-selftest"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                      # exactly one JSON line, relayed from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["collectives_ok"] and out["wall_max"] == 0.2
    # what a scaling run is checked against: the world size the process group saw, one rate per rank, the rank that set the max, the map broadcast
    assert out["rccl_world_size"] == 2 and out["per_rank_wall_s"] == [0.1, 0.2] and len(out["per_rank_frames_per_s"]) == 2 and out["slowest_rank"] == 1
    assert out["map_broadcast"]["bytes"] == 8 * (16 + 16 * 16) and out["map_broadcast"]["ms"] > 0


def test_bench_gpus_flag_fails_in_the_children_without_gpus():
    """Without GPUs the real bench must fail inside the two spawned ranks (device acquisition), not run one rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())


def test_bench_force_dist_one_rank_goes_through_the_collectives():
    """`--gpus 1 --force-dist`: the launcher starts ONE rank and that rank still initialises the process group and runs the
    broadcast / barrier / all-reduce / all-gather (gloo here; the -m gpu twin of this test runs the same with nccl = RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--launch-selftest"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["collectives_ok"] and d["collectives"] == "gloo"
    assert d["rccl_world_size"] == 1 and len(d["per_rank_frames_per_s"]) == 1 and d["map_broadcast"]["bytes"] > 0
