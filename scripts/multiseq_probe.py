"""multi-sequence throughput probe: B contexts on one GPU (what bench.py's multi_sequence leg does)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, __graft_entry__ as ge
import torch
pkg = ge.load_package()
for B in (1, 2, 4, 8, 16):
    print(B, os.environ.get("GPU_MAX_HW_QUEUES"), round(bench.multi_sequence_throughput(torch, pkg.synth, pkg.srukf, 200, B, 64, 8, 0), 1), flush=True)
