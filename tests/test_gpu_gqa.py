"""HIP paged GQA decode (attn_with_kvcache with block_table) vs the oracle."""

import pytest
import torch

from oracle import gqa as ogqa
from tests.util import max_rel_to_peak

pytestmark = pytest.mark.gpu


def make(bs, Hq, Hkv, lens, page=256, seed=0):
    g = torch.Generator().manual_seed(seed)
    per = [(l + 1 + page - 1) // page for l in lens]
    pages = sum(per) + 2
    kc = torch.randn(pages, page, Hkv, 128, generator=g).to(torch.bfloat16)
    vc = torch.randn(pages, page, Hkv, 128, generator=g).to(torch.bfloat16)
    perm = torch.randperm(pages, generator=g)
    table = torch.zeros(bs, max(per) + 1, dtype=torch.int32)
    o = 0
    for b in range(bs):
        table[b, : per[b]] = perm[o : o + per[b]].to(torch.int32)
        o += per[b]
    q = (torch.randn(bs, 1, Hq, 128, generator=g) * 0.5).to(torch.bfloat16)
    k = torch.randn(bs, 1, Hkv, 128, generator=g).to(torch.bfloat16)
    v = torch.randn(bs, 1, Hkv, 128, generator=g).to(torch.bfloat16)
    return q, kc, vc, k, v, torch.tensor(lens, dtype=torch.int32), table


@pytest.mark.parametrize(
    "bs,Hq,Hkv,lens,page",
    [
        (1, 32, 8, [0], 256),                 # first decode token: attends only to itself
        (1, 32, 8, [255], 256),               # append fills the page
        (1, 32, 8, [256], 256),               # append opens a new page
        (4, 32, 8, [5, 300, 1000, 2047], 256),  # Llama-3-8B heads, ragged
        (2, 32, 32, [100, 17], 256),          # MHA (Llama-2-7B): group of 1
        (3, 16, 1, [40, 41, 700], 64),        # MQA, 16 heads per kv head, small pages
        (16, 32, 8, [1024] * 16, 256),
    ],
)
@pytest.mark.parametrize("splits", [None, 1, 5])
def test_vs_oracle(bs, Hq, Hkv, lens, page, splits):
    from chitu_amd.attn_backend import HipAttnBackend

    q, kc, vc, k, v, sl, table = make(bs, Hq, Hkv, lens, page, seed=bs + Hq + len(lens))
    ref, kc_ref, vc_ref = ogqa.attn_with_kvcache(q, kc, vc, k, v, sl, table)
    kd, vd = kc.cuda(), vc.cuda()
    out = HipAttnBackend(local_n_heads=Hq).attn_with_kvcache(
        q.cuda(), kd, vd, k.cuda(), v.cuda(), cache_seqlens=sl.cuda(), block_table=table.cuda(), num_splits=splits
    )
    assert tuple(out.shape) == (bs, 1, Hq, 128)
    assert torch.equal(kd.cpu(), kc_ref) and torch.equal(vd.cpu(), vc_ref)  # in-place append, exact
    assert max_rel_to_peak(out, ref) < 1e-2


def test_no_append_and_poisoned_tail():
    from chitu_amd.attn_backend import HipAttnBackend

    q, kc, vc, k, v, sl, table = make(2, 32, 8, [70, 300], 256, seed=3)
    ref, _, _ = ogqa.attn_with_kvcache(q, kc, vc, None, None, sl, table)
    kp, vp = kc.clone(), vc.clone()
    kp[table[0, 0].item(), 70:] = float("nan")
    vp[table[0, 0].item(), 70:] = float("inf")
    out = HipAttnBackend(local_n_heads=32).attn_with_kvcache(q.cuda(), kp.cuda(), vp.cuda(), cache_seqlens=sl.cuda(),
                                                             block_table=table.cuda())
    assert torch.isfinite(out.float()).all() and max_rel_to_peak(out, ref) < 1e-2
