"""Golden fixture for SURVEY 8 row a11: the REFERENCE'S PagedKVCacheManager (chitu/cache_manager.py:12-225) driven
through tests/util.py::cache_manager_scenario on CPU.  Only page-assignment-independent observations are recorded (the
reference hands out `list(set)[0]`, chitu_amd a deque's head: different physical pages, same logical cache).

Run in the build container only:   python tests/golden/gen_cache_manager.py   -> tests/golden/cache_manager.json
"""

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_shims  # noqa: E402

ref_shims.install()
import torch  # noqa: E402


class AD(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def main():
    import chitu.global_vars as gv

    gv.set_global_variables(AD(models=AD(), infer=AD(tp_size=1, pp_size=1, max_reqs=3, cache_type="paged")))
    from chitu.cache_manager import PagedKVCacheManager
    from chitu.utils import VarLens

    from tests.util import cache_manager_scenario

    torch.set_default_dtype(torch.bfloat16)  # the reference allocates its pages in the default dtype

    def make_manager(layers, page, width, max_reqs, max_seq_len):
        return PagedKVCacheManager(0, layers, num_hot_req=max_reqs, block_size=page, max_seq_len=max_seq_len, device="cpu",
                                   kv_shape_per_sample=(width,))

    obs = cache_manager_scenario(make_manager, lambda toks: VarLens(toks, "cpu"))
    with open(os.path.join(HERE, "cache_manager.json"), "w") as f:
        json.dump(obs, f, indent=1)
    print(len(obs), "observations;", obs[-1]["seq_lens"], "free", obs[-1]["free"])


if __name__ == "__main__":
    main()
