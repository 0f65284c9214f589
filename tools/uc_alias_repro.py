"""Root cause experiment 3 (round 4): memory that was an UNCACHED allocation (hipExtMallocWithFlags(hipDeviceMallocUncached),
the xGMI exchange buffers of chitu_hip_comm_create) and is handed out again as ordinary memory reads STALE L2 lines on some XCD.

What the whole-suite probe showed (profiles/r04_graph_mismatch_probe_suite_run2.txt): the rejected graph's private pool sat on
the address range of a freed uncached buffer; one eighth of a GEMM's workgroups (one XCD) read old activations; a 64 MB fill
(every L2 evicted) cured it for good; sleeping, allocating, re-capturing did not.

Here, without any graph: per trial
  1. ordinary memory at address A is written and read by workgroups on every XCD (all eight L2s hold its lines), freed;
  2. an uncached buffer is allocated (same size class), used, freed -- optionally several times;
  3. ordinary memory again: if it lands on A, a producer kernel writes new values and a consumer kernel whose
     workgroup -> data mapping differs (so data crosses XCDs) reads them back.
Reports how often step 3 reads anything but the new values, with and without an L2 sweep (a 256 MB fill) after step 2.
"""

import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

hip = ctypes.CDLL("libamdhip64.so")
UNCACHED = 0x3  # hipDeviceMallocUncached


def uc_alloc(nbytes):
    p = ctypes.c_void_p()
    e = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(nbytes), ctypes.c_uint(UNCACHED))
    assert e == 0, e
    return p


def sweep(mb=256):
    j = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    j.fill_(1)
    torch.cuda.synchronize()
    del j


def trial(nbytes, uc_rounds, do_sweep, use_comm):
    n = nbytes // 4
    # 1. ordinary life: every XCD's L2 gets lines of the block
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    addr = a.data_ptr()
    a.fill_(1.0)
    for _ in range(3):
        _ = (a + a.flip(0)).sum().item()  # reads through two different workgroup -> address maps
    del a, _
    torch.cuda.synchronize()
    torch.cuda.empty_cache()  # hipFree
    # 2. uncached life
    hit = False
    for _ in range(uc_rounds):
        if use_comm:
            from chitu_amd.xgmi import XgmiComm

            c = XgmiComm(0, 1, max_rows=64, max_dim=8192, timeout_ms=200)
            p = c.local_ptr()
            hit = hit or (p <= addr < p + (40 << 20)) or (addr <= p < addr + nbytes)
            c.close()
        else:
            p = uc_alloc(nbytes)
            hit = hit or p.value == addr
            hip.hipMemset(p, 0x5A, ctypes.c_size_t(nbytes))
            hip.hipDeviceSynchronize()
            hip.hipFree(p)
    if do_sweep:
        sweep()
    # 3. ordinary life again
    b = torch.empty(n, dtype=torch.float32, device="cuda")
    same_addr = b.data_ptr() == addr
    want = torch.arange(n, dtype=torch.float32, device="cuda")  # written elsewhere
    b.copy_(want * 3.0)  # producer: workgroup i writes chunk i
    got_flip = b.flip(0).clone()  # consumer: workgroup i reads chunk n-1-i
    got_strided = b.view(-1, 64).t().contiguous()  # consumer: another map
    torch.cuda.synchronize()
    bad1 = int((got_flip != (want * 3.0).flip(0)).sum())
    bad2 = int((got_strided != (want * 3.0).view(-1, 64).t()).sum())
    del b
    torch.cuda.empty_cache()
    return same_addr, hit, bad1, bad2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=6)
    a = ap.parse_args()
    for nbytes in (2 << 20, 20 << 20, 64 << 20):
        for use_comm in (False, True):
            for do_sweep in (False, True):
                res = [trial(nbytes, 2, do_sweep, use_comm) for _ in range(a.trials)]
                print(f"{nbytes >> 20:3d} MB, uncached via {'comm_create' if use_comm else 'hipExtMallocWithFlags'}, "
                      f"sweep={do_sweep}: same address {sum(r[0] for r in res)}/{len(res)}, uncached on it {sum(r[1] for r in res)}, "
                      f"stale elements (flip / transpose reads): {[r[2] for r in res]} {[r[3] for r in res]}", flush=True)


if __name__ == "__main__":
    main()
