#!/usr/bin/env python3
"""Derived per-kernel metrics from the counter passes of tools/pmc_passes.sh:
    CHITU_GIT_HEAD=<sha> python tools/pmc_report.py <pmc_dir> > profiles/r03_pmc_step.json
Per kernel (averages over its dispatches in an EAGER decode step, bs 16, ctx 1024):
  hbm_read_MB   = FETCH_SIZE [KB] * 1024 * 2   (gfx950: the counter tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM)
  hbm_write_MB  = WRITE_SIZE [KB] * 1024       (uncalibrated on gfx950: an indication only)
  mfma_util     = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): share of SIMD-cycles with the matrix
                  pipe busy while the kernel ran (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
  lds_conflict  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: share of LDS-array cycles lost to bank conflicts
  avg_us        = kernel duration in the FETCH_SIZE pass (counter collection serialises dispatches; durations under
                  counters run a few % above the plain kernel trace)
  read_TBs      = hbm_read_MB / avg_us
"""
import glob
import json
import os
import re
import sys

d = sys.argv[1]


def load(prefix):
    f = glob.glob(os.path.join(d, f"pmc_{prefix}*.json"))
    return json.load(open(f[0])) if f else {}


def durations(prefix):
    f = glob.glob(os.path.join(d, f"pmc_{prefix}*.durations.txt"))
    out = {}
    if not f:
        return out
    for line in open(f[0]):
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", line)
        if m:
            out[m.group(7).strip().split("(")[0][:80]] = float(m.group(3))
    return out


fetch, write, mfma, lds, ins = load("FETCH"), load("WRITE"), load("SQ_VALU_MFMA"), load("SQ_LDS"), load("SQ_INSTS")
dur = durations("FETCH")
res = {}
for k in sorted(fetch):
    if k.startswith("_"):
        continue
    g = lambda tbl, name: tbl.get(k, {}).get(name, {}).get("avg")
    r = {"dispatches": fetch[k]["FETCH_SIZE"]["dispatches"]}
    fs, ws = g(fetch, "FETCH_SIZE"), g(write, "WRITE_SIZE")
    if fs is not None:
        r["hbm_read_MB"] = round(fs * 1024 * 2 / 1e6, 3)
    if ws is not None:
        r["hbm_write_MB_uncalibrated"] = round(ws * 1024 / 1e6, 3)
    busy, gui = g(mfma, "SQ_VALU_MFMA_BUSY_CYCLES"), g(mfma, "GRBM_GUI_ACTIVE")
    if busy is not None and gui:
        r["mfma_busy_cycles"] = busy
        r["mfma_util"] = round(busy / (gui / 8 * 1024), 4)
    bc, act = g(lds, "SQ_LDS_BANK_CONFLICT"), g(lds, "SQ_LDS_IDX_ACTIVE")
    if bc is not None and act:
        r["lds_bank_conflict_frac"] = round(bc / act, 4)
    m8, m16 = g(ins, "SQ_INSTS_VALU_MFMA_MOPS_F8"), g(ins, "SQ_INSTS_VALU_MFMA_MOPS_BF16")
    if m8 is not None:
        r["mfma_mops_f8"], r["mfma_mops_bf16"] = m8, m16
    name = next((n for n in dur if n.startswith(k[:60])), None)
    if name:
        r["avg_us_under_counters"] = dur[name]
        if "hbm_read_MB" in r:
            r["read_TBs"] = round(r["hbm_read_MB"] / dur[name], 3)
    res[k] = r
# identity of the code the counters were taken on: the git head handed in by the caller (the GPU box has no .git:
# CHITU_GIT_HEAD=$(git rev-parse --short HEAD) in the gpurun command line) and digests of the kernel sources as found
# next to this script -- bench.py withholds a kernel's counters when its source has changed since
import hashlib

csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "chitu_amd", "csrc")
digests = {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest() for f in sorted(glob.glob(os.path.join(csrc, "*.hip")))}
print(json.dumps({"source": "rocprofv3 --kernel-trace --pmc <group>, one pass per group (tools/pmc_passes.sh), eager decode step "
                            "of 8 R1 TP=8-rank layers, bs 16, ctx 1024", "git_head": os.environ.get("CHITU_GIT_HEAD"),
                  "source_sha256": digests, "kernels": res}, indent=1))
