#!/usr/bin/env python3
"""MLA paged decode outputs of the loaded build over a fixed set of cases (ragged lengths, empty sequences, 1 / 3 / default
splits, 64- and 128-token pages), one sha256 per case: run it once per build (CHITU_HIP_LIB=<other build>) and diff the two
listings -- a kernel rewrite that claims unchanged arithmetic must produce the same listing.
    python tools/mla_decode_outputs.py > a.txt; CHITU_HIP_LIB=build_probe/lib_x.so python tools/mla_decode_outputs.py > b.txt; diff a.txt b.txt"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chitu_amd.attn_backend import HipAttnBackend  # noqa: E402


@torch.inference_mode()
def main():
    H, C, R = 16, 512, 64
    for heads in (16, 5):
        for page_size in (64, 128):
            for lens_l in ([1], [63, 64, 65], [1000, 17, 0, 300], [1024] * 16, [1041] * 16, [5000, 4097, 64, 1], [20000, 3]):
                torch.manual_seed(len(lens_l) * 131 + page_size + heads)
                bs = len(lens_l)
                per = max(lens_l) // page_size + 2
                cache = (torch.randn(bs * per, page_size, C + R, device="cuda") * 0.5).to(torch.bfloat16)
                table = torch.randperm(bs * per, device="cuda").to(torch.int32).view(bs, per)
                lens = torch.tensor(lens_l, dtype=torch.int32, device="cuda")
                qn = torch.randn(bs, heads, C, device="cuda").to(torch.bfloat16)
                qp = torch.randn(bs, heads, R, device="cuda").to(torch.bfloat16)
                for splits in (None, 1, 3):
                    be = HipAttnBackend(local_n_heads=heads, max_seq_len=max(lens_l) + 64)
                    o = be.mla_decode(qn, qp, cache, lens, table, 0.1352, num_splits=splits)
                    torch.cuda.synchronize()
                    h = hashlib.sha256(o.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
                    print(f"heads {heads} page {page_size} lens {lens_l[:4]}{'...' if bs > 4 else ''} splits {splits}: {h}")


if __name__ == "__main__":
    main()
