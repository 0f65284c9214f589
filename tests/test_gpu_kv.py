"""HIP paged-KV append and RoPE vs the oracle / reference fixtures."""

import numpy as np
import pytest
import torch

from oracle import kv as okv
from tests.util import bf16, bits16, golden, pattern_cache

pytestmark = pytest.mark.gpu


def test_append_fixture_exact():
    from chitu_amd import ops

    g = golden("append_rope")
    pages, page, dim = g["cache_shape"].tolist()
    cache = pattern_cache(pages, page, dim).cuda()
    ops.append_to_paged_kv_cache(
        cache, torch.from_numpy(g["table"]).cuda(), bf16(g["kv"]).cuda(), torch.from_numpy(g["lens"]).cuda()
    )
    expect = pattern_cache(pages, page, dim)
    ch = torch.from_numpy(g["changed"])
    expect[ch[:, 0], ch[:, 1]] = bf16(g["changed_rows"])
    assert torch.equal(cache.cpu(), expect)


@pytest.mark.parametrize("page,shape", [(64, (576,)), (256, (8, 128)), (16, (3, 5))])
def test_append_vs_oracle(page, shape):
    from chitu_amd import ops

    g = torch.Generator().manual_seed(page)
    bs, pages, per = 5, 40, 6
    cache = torch.randn(pages, page, *shape, generator=g).to(torch.bfloat16)
    table = torch.stack([torch.randperm(pages, generator=g)[:per] for _ in range(bs)]).to(torch.int32)
    lens = torch.tensor([0, page - 1, page, 3 * page + 7, per * page - 1], dtype=torch.int32)
    kv = torch.randn(bs, 1, *shape, generator=g).to(torch.bfloat16)
    ref = okv.append_to_paged_kv_cache(cache, table, kv, lens)
    c = cache.cuda()
    ops.append_to_paged_kv_cache(c, table.cuda(), kv.cuda(), lens.cuda())
    assert torch.equal(c.cpu(), ref)


def test_rope_fixture_and_reference_test_recipe():
    from chitu_amd import ops

    g = golden("append_rope")
    oq, ok = ops.apply_rotary_pos_emb(
        bf16(g["q"]).cuda(), bf16(g["k"]).cuda(), torch.from_numpy(g["cos"]).cuda(), torch.from_numpy(g["sin"]).cuda(), "llama"
    )
    assert np.array_equal(bits16(oq), g["oq_torch"]) and np.array_equal(bits16(ok), g["ok_torch"])
    # the reference's own test: test/pytest/test_rotary_triton.py:16-32 (fp32, tol 1e-5)
    gen = torch.Generator().manual_seed(0)
    q = torch.randn(16, 64, 256, generator=gen)
    k = torch.randn(16, 256, generator=gen)
    cos = torch.randn(16, 128, generator=gen) * 2
    sin = torch.randn(16, 128, generator=gen)
    rq, rk = okv.apply_rotary_pos_emb(q, k, cos, sin, "llama")
    oq, ok = ops.apply_rotary_pos_emb(q.cuda(), k.cuda(), cos.cuda(), sin.cuda(), "llama")
    assert torch.allclose(oq.cpu(), rq, rtol=1e-5, atol=1e-5) and torch.allclose(ok.cpu(), rk, rtol=1e-5, atol=1e-5)
    assert torch.equal(oq.cpu(), rq) and torch.equal(ok.cpu(), rk)


@pytest.mark.parametrize("rtype", ["llama", "hf-llama"])
def test_rope_strided_views(rtype):
    """MLA calls RoPE on q[..., 128:] / kv[..., 512:] views (model_deepseek_v3.py:493-500)."""
    from chitu_amd import ops

    gen = torch.Generator().manual_seed(2)
    q_full = torch.randn(7, 16, 192, generator=gen).to(torch.bfloat16)
    kv_full = torch.randn(7, 576, generator=gen).to(torch.bfloat16)
    cos = torch.randn(7, 32, generator=gen)
    sin = torch.randn(7, 32, generator=gen)
    q_pe, k_pe = q_full[..., 128:], kv_full[..., 512:]
    rq, rk = okv.apply_rotary_pos_emb(q_pe, k_pe, cos, sin, rtype)
    oq, ok = ops.apply_rotary_pos_emb(q_full.cuda()[..., 128:], kv_full.cuda()[..., 512:], cos.cuda(), sin.cuda(), rtype)
    assert torch.equal(oq.cpu(), rq) and torch.equal(ok.cpu(), rk)


@pytest.mark.parametrize("bs,dim,vocab_start,rows", [(1, 7168, 0, 16160), (16, 7168, 32320, 16160), (5, 512, 100, 64)])
def test_embed_rope_gather_is_the_masked_lookup_and_the_table_rows(bs, dim, vocab_start, rows):
    """chitu_hip_embed_rope_gather == VocabParallelEmbedding's mask / lookup / zero-fill (tensor_parallel.py:199-208)
    + the rotary-row gather of prepare_freqs_cis_decode (model.py:429-448), bit for bit."""
    from chitu_amd import ops

    g = torch.Generator().manual_seed(bs + dim)
    table = torch.randn(rows, dim, generator=g).to(torch.bfloat16)
    toks = torch.randint(vocab_start - 5, vocab_start + rows + 5, (bs,), generator=g)
    toks[0] = vocab_start  # first and last row of the slice, and ids on both sides of it
    if bs > 2:
        toks[1], toks[2] = vocab_start + rows - 1, vocab_start + rows
    cos_t, sin_t = torch.randn(300, 32, generator=g), torch.randn(300, 32, generator=g)
    pos = torch.randint(0, 300, (bs + 3,), generator=g).to(torch.int32)  # the cache's buffer is longer than the batch
    h, c, s = ops.embed_rope_gather(toks.cuda(), table.cuda(), vocab_start, pos.cuda(), cos_t.cuda(), sin_t.cuda())
    local = toks - vocab_start
    foreign = (local < 0) | (local >= rows)
    want = torch.nn.functional.embedding(local.masked_fill(foreign, 0), table).masked_fill(foreign.unsqueeze(-1), 0)
    assert torch.equal(h.cpu(), want) and not foreign.all() and (bs < 3 or foreign.any())
    assert torch.equal(c.cpu(), cos_t[pos[:bs].long()]) and torch.equal(s.cpu(), sin_t[pos[:bs].long()])
    h2, c2, s2 = ops.embed_rope_gather(toks.cuda(), table.cuda(), vocab_start)
    assert torch.equal(h2, h) and c2 is None and s2 is None
