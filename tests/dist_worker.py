"""Worker of tests/test_distributed_cpu.py: the multi-GPU driver logic of bench.py on the gloo
backend with CPU tensors (one process per "GPU"; the oracle stands in for the device filter)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, world, local = bench.dist_env()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    synth = ge.load_package().synth
    N, F = 6, 5
    # rank r builds its inputs with a DIFFERENT map seed on purpose: only the broadcast can make the maps equal
    sc = bench.build_inputs(synth, N, F, rank, map_seed=10 + rank)
    n = 6 * N + 4
    X0, S0 = bench.broadcast_map(torch, dist, sc, n, rank, world, torch.device("cpu"))
    ref = bench.build_inputs(synth, N, F, 0, map_seed=10)          # what rank 0 holds
    same_map = bool(np.array_equal(X0.numpy(), ref["X0"]) and np.array_equal(S0.numpy(), ref["S0"]))
    # every rank replays rank 0's odometry/map with its own measurement noise (Monte-Carlo run)
    mine = bench.build_inputs(synth, N, F, rank, map_seed=10)
    o = O.Oracle(N, mine["params"])
    o.set_state(X0.numpy(), S0.numpy())
    traj = torch.from_numpy(o.run_frames(mine["odo"], mine["z"], mine["matched"], O.Oracle.BATCHED))
    allt = [torch.empty_like(traj) for _ in range(world)]
    dist.all_gather(allt, traj)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    json.dump({"rank": rank, "world": world, "same_map": same_map, "wall_max": float(t.item()),
               "trajs": [a.numpy().tolist() for a in allt], "z0": mine["z"][0, :4].tolist()},
              open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
