#!/usr/bin/env python3
"""World-size sweep of the xGMI collectives with every rank a process on cuda:0 (the 1-GPU stand-in for an
8-GPU node): python tools/xgmi_world8.py [world=8] [timeout_s=180].  Runs tests/test_gpu_xgmi.py's
`_collectives_worker` (every fusion of the all-reduce, the all-gather, hipGraph replay; bit-exact vs the oracle)
and prints the wall time.  GPU_MAX_HW_QUEUES=2 per process keeps 8 processes inside the GPU's hardware queues."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

if __name__ == "__main__":
    from tests import test_gpu_xgmi as t

    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 180
    t0 = time.time()
    t._spawn(t._collectives_worker, world, timeout=limit)
    print(f"world {world}: every rank bit-exact, {time.time() - t0:.1f} s")
