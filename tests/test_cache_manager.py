"""chitu_amd.cache_manager.PagedKVCacheManager against the reference's own class
(tests/golden/gen_cache_manager.py ran chitu/cache_manager.py:12-225 through the same scripted life and recorded what
does not depend on which physical pages were handed out).  Host-only."""

import json
import os

import pytest
import torch

from tests.util import cache_manager_scenario

HERE = os.path.dirname(os.path.abspath(__file__))


def _mine():
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.deepseek_v3 import VarLens

    def make_manager(layers, page, width, max_reqs, max_seq_len):
        return PagedKVCacheManager(0, layers, num_hot_req=max_reqs, block_size=page, max_seq_len=max_seq_len, device="cpu",
                                   kv_shape_per_sample=(width,), dtype=torch.bfloat16)

    return cache_manager_scenario(make_manager, lambda toks: VarLens(toks, "cpu"))


def test_scripted_life_matches_the_reference_manager():
    with open(os.path.join(HERE, "golden", "cache_manager.json")) as f:
        want = json.load(f)
    got = json.loads(json.dumps(_mine()))  # tuples -> lists, like the fixture
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w, (g["tag"], {k: (g[k], w[k]) for k in w if g.get(k) != w[k]})
    assert all(o["pages_distinct"] for o in got) and all(o.get("device_table_matches", True) for o in got)


def test_running_out_of_pages_raises_like_the_reference():
    from chitu_amd.cache_manager import PagedKVCacheManager

    mgr = PagedKVCacheManager(0, 1, num_hot_req=1, block_size=4, max_seq_len=8, device="cpu", kv_shape_per_sample=(8,),
                              dtype=torch.bfloat16)
    for _ in range(mgr.num_blocks):
        mgr.get_free_block()
    with pytest.raises(Exception, match="No more free blocks"):  # chitu/cache_manager.py:163-164
        mgr.get_free_block()
