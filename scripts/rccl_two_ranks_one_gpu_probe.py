"""Probe (not a test): can two RCCL ranks share GPU 0?  Run as two processes with RANK=0/1 WORLD_SIZE=2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spatialcore_amd import _lib, parallel
ctx = _lib.Context(0)
try:
    comm = parallel.connect(ctx, transport="rccl", timeout_s=30)
    print("rank", comm.rank, "gathered", comm.all_gather(np.full(3, float(comm.rank))).tolist(), flush=True)
    comm.close()
except Exception as e:
    print("rank", os.environ.get("RANK"), "FAILED:", type(e).__name__, str(e)[:300], flush=True)
