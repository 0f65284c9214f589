"""Golden fixture for the Mixtral router (BASELINE config 4's routing): the REFERENCE'S SparseMoeBlockHFMixtral.forward
(chitu/models/model_hf_mixtral.py:53-94) run on CPU in bf16 with every expert replaced by a probe that returns the
one-hot row of its own index -- the block's output column e is then exactly the renormalised, dtype-cast routing
weight the reference gives expert e for that token (0 where it is not selected), i.e. the two tensors the fused MoE
consumes, observed through the reference's own forward.

Run in the build container only:   python tests/golden/gen_mixtral_router.py   -> tests/golden/mixtral_router.npz
"""

import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
import numpy as np  # noqa: E402
import torch  # noqa: E402

DIM, HIDDEN, EXPERTS, TOPK, TOKENS, SEED = 256, 512, 8, 2, 96, 99


class AD(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class Probe(torch.nn.Module):
    def __init__(self, index):
        super().__init__()
        self.index = index

    def forward(self, x):
        out = torch.zeros_like(x)
        out[:, self.index] = 1.0
        return out


def main():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist

    dist.init_process_group("gloo")
    import chitu.global_vars as gv

    gv.set_global_variables(AD(models=AD(), infer=AD(tp_size=1, pp_size=1, max_reqs=1, cache_type="paged", op_impl="torch", soft_fp8=False, use_cuda_graph=False, attn_type="ref", pp_layer_partition=None, max_seq_len=64)))
    from chitu import tensor_parallel as rtp

    rtp.init_tp(1, 1)
    torch.set_default_dtype(torch.bfloat16)
    from chitu.models.model_hf_mixtral import SparseMoeBlockHFMixtral

    block = SparseMoeBlockHFMixtral(DIM, HIDDEN, EXPERTS, TOPK, op_impl="torch")
    g = torch.Generator().manual_seed(SEED)
    gate_w = (torch.randn(EXPERTS, DIM, generator=g, dtype=torch.float32) * DIM ** -0.5).to(torch.bfloat16)
    block.gate.weight.data.copy_(gate_w)
    block.experts = torch.nn.ModuleList(Probe(e) for e in range(EXPERTS))
    x = torch.randn(TOKENS, DIM, generator=g, dtype=torch.float32).to(torch.bfloat16)
    x[-8:] = x[-8:].abs() * 0.01  # nearly uniform router logits: top-2 decided by small differences
    with torch.inference_mode():
        out = block(x.clone())
    weights = out[:, :EXPERTS].contiguous()
    assert (out[:, EXPERTS:] == 0).all() and ((weights != 0).sum(-1) == TOPK).all()
    np.savez_compressed(os.path.join(HERE, "mixtral_router.npz"), x=x.view(torch.int16).numpy(),
                        gate_w=gate_w.view(torch.int16).numpy(), weights=weights.view(torch.int16).numpy(),
                        topk=np.array([TOPK]))
    print("routing weights of token 0:", weights[0].float().tolist())


if __name__ == "__main__":
    main()
