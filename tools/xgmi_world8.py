#!/usr/bin/env python3
"""World-size sweep of the xGMI collectives with every rank a process on cuda:0 (the 1-GPU stand-in for an
8-GPU node).
  python tools/xgmi_world8.py [world=8] [timeout_s=180]
      tests/test_gpu_xgmi.py's `_collectives_worker` (every fusion of the all-reduce, the all-gather, hipGraph replay;
      bit-exact vs the oracle) with kernels that really WAIT for each other: passes or times out with the GPU's time
      slicing of the processes (round 3: 4 processes passed once and timed out once, 8 timed out).
  python tools/xgmi_world8.py --split-phase [world=8] [repeats=3] [timeout_s=600]
      the same wiring (IPC handles, peer mapping, staged self-tests, one-shot / two-shot slicing) with every collective as
      contribute -> host barrier -> complete (CHITU_XGMI_SPLIT_PHASE=1, tests/test_gpu_xgmi.py::_split_phase_worker): no
      kernel waits for a peer, so the world size completes by construction; plus a 2-layer decode step at DeepSeek-R1's
      per-rank shapes, bit-identical logits on every rank.
GPU_MAX_HW_QUEUES=2 per process keeps 8 processes inside the GPU's hardware queues."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

if __name__ == "__main__":
    from tests import test_gpu_xgmi as t

    argv = sys.argv[1:]
    if argv and argv[0] == "--split-phase":
        world = int(argv[1]) if len(argv) > 1 else 8
        repeats = int(argv[2]) if len(argv) > 2 else 3
        limit = int(argv[3]) if len(argv) > 3 else 600
        for i in range(repeats):
            t0 = time.time()
            t._spawn(t._split_phase_worker, world, timeout=limit)
            print(f"split-phase world {world}, run {i + 1}/{repeats}: collectives bit-exact vs the oracle on every rank, R1-shaped "
                  f"2-layer step identical on every rank, {time.time() - t0:.1f} s", flush=True)
        sys.exit(0)
    world = int(argv[0]) if len(argv) > 0 else 8
    limit = int(argv[1]) if len(argv) > 1 else 180
    t0 = time.time()
    t._spawn(t._collectives_worker, world, timeout=limit)
    print(f"world {world}: every rank bit-exact, {time.time() - t0:.1f} s")
