"""Root cause experiment 4 (round 5): the stale-L2 reproducer WITH the IPC-shared leg that tools/uc_alias_repro.py lacked.

What round 4's whole-suite probe showed: a hipGraph's private pool sat on memory that had been an xGMI exchange buffer
(hipExtMallocWithFlags(hipDeviceMallocUncached), written by a PEER PROCESS through hipIpcOpenMemHandle with sc0 sc1 stores);
one XCD then read stale lines until every L2 was swept.  Experiment 3 recycled an uncached buffer inside ONE process and never
saw a stale element.  Here the peer leg is real: per trial

  1. the parent (which plays the pytest process: a live HIP context that survives the children) gives a block an ordinary
     life -- written and read through two workgroup -> address maps, so all eight L2s hold lines of it -- and returns it
     to the driver (empty_cache: hipFree);
  2. two child PROCESSES on the same GPU create the exchange buffers (uncached), map each other's over IPC handles, run the
     all-reduce cases of tests/test_gpu_xgmi.py (one- and two-shot), and leave -- either by the library's close() path
     (hipIpcCloseMemHandle, buffers parked, context torn down at exit) or ABRUPTLY (os._exit right after the last kernel:
     handles open, nothing closed: the driver reclaims everything);
  3. the parent allocates again (the block's size, then a large span that covers whatever the children returned), a producer
     kernel writes new values, consumers with two other workgroup -> address maps read them back; repeated three times with
     new values (a captured graph replays on the same memory, so a stale line would show on a later pass).
Reports the stale elements per trial, with and without an L2 sweep after step 2.

    python tools/uc_alias_repro_ipc.py [--trials 3] [--span-gb 4]
"""

import argparse
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _child(rank, world, port, q, abrupt):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist

        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from chitu_amd import tensor_parallel as tp

        tp.init_tp(world, 1)
        ok = tp.enable_xgmi(max_rows=32, max_dim=8192, gather_bytes=16 * 16160 * 2, timeout_ms=8000)
        if not ok:
            q.put((rank, "xGMI setup refused"))
            return
        comm = tp.xgmi_comm()
        g = torch.Generator().manual_seed(rank)
        for salt in range(4):
            comm.set_two_shot(0 if salt >= 2 else 256 << 10)
            for rows, dim in ((1, 7168), (16, 7168), (32, 7168), (3, 8192)):
                part = (torch.randn(rows, dim, generator=g) * 0.7).to(torch.bfloat16).cuda()
                x = torch.randn(rows, dim, generator=g).to(torch.bfloat16).cuda()
                w = torch.ones(dim, dtype=torch.bfloat16).cuda()
                comm.allreduce_rmsnorm(part, x, w, 1e-6, out_bf16=True, quant="act")
        torch.cuda.synchronize()
        st = comm.status()
        dist.barrier()
        if abrupt:
            q.put((rank, f"ok status {st} (abrupt exit)"))
            time.sleep(0.2)  # let the queue's feeder thread flush
            os._exit(0)
        tp.disable_xgmi()
        dist.destroy_process_group()
        q.put((rank, f"ok status {st} (closed)"))
    except BaseException as e:  # noqa: BLE001 -- reported to the parent
        q.put((rank, "fail: " + repr(e)))


def run_children(abrupt):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_child, args=(r, 2, port, q, abrupt)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    deadline = time.time() + 240
    while len(res) < 2 and time.time() < deadline:
        try:
            res.append(q.get(timeout=2.0))
        except Exception:  # noqa: BLE001 -- queue.Empty
            if all(p.exitcode is not None for p in procs):
                break
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()  # our own child, by handle
    return res


def sweep(mb=256):
    j = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    j.fill_(1)
    torch.cuda.synchronize()
    del j


def ordinary_life(n):
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    a.fill_(1.0)
    for _ in range(3):
        _ = (a + a.flip(0)).sum().item()
        _ = a.view(-1, 64).t().contiguous().sum().item()
    addr = a.data_ptr()
    del a
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return addr


def probe(n):
    """producer / consumer passes over a fresh block of n floats: stale elements seen by each consumer map, per pass"""
    b = torch.empty(n, dtype=torch.float32, device="cuda")
    want = torch.arange(n, dtype=torch.float32, device="cuda")
    bad = []
    for k in range(3):
        b.copy_(want * float(3 + k))
        f = b.flip(0).clone()
        t = b.view(-1, 64).t().contiguous()
        torch.cuda.synchronize()
        ref = want * float(3 + k)
        bad.append((int((f != ref.flip(0)).sum()), int((t != ref.view(-1, 64).t()).sum())))
    addr = b.data_ptr()
    del b, want
    torch.cuda.empty_cache()
    return addr, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=3)
    ap.add_argument("--span-gb", type=float, default=4.0)
    a = ap.parse_args()
    torch.cuda.init()
    n_small = (40 << 20) // 4  # the size class of an exchange buffer
    n_span = int(a.span_gb * (1 << 30)) // 4 // 64 * 64
    for abrupt in (False, True):
        for do_sweep in (False, True):
            for t in range(a.trials):
                addr0 = ordinary_life(n_small)
                ordinary_life(n_span)
                res = run_children(abrupt)
                if do_sweep:
                    sweep()
                addr1, bad_small = probe(n_small)
                _, bad_span = probe(n_span)
                print(f"children exit {'abruptly' if abrupt else 'by close()'}, sweep={do_sweep}, trial {t}: children {sorted(res)}; "
                      f"block re-issued at the same address: {addr0 == addr1}; stale elements (flip, transpose) per pass: "
                      f"40 MB block {bad_small}, {a.span_gb:g} GB span {bad_span}", flush=True)


if __name__ == "__main__":
    main()
