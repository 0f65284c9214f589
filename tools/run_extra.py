#!/usr/bin/env python3
"""Run one of bench.py's extra workloads alone (for profiling): python tools/run_extra.py mixtral|llama|v2lite|ep8 [steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

torch.cuda.set_device(0)
from chitu_amd import _lib

_lib.apply_debug_options_from_env()
which, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16
fn = {"mixtral": bench.mixtral_extra, "llama": bench.llama3_8b_extra, "v2lite": bench.v2_lite_extra,
      "ep8": bench.ep8_rank_extra}[which]
print(json.dumps(fn(steps, 2, 1024)))
