"""Tensor-parallel groups and linears with the reference's `chitu/tensor_parallel.py` surface.

Reference (read-only): chitu/tensor_parallel.py:1-208.  One process per GPU; the process group
backend string "nccl" IS RCCL on ROCm (collectives run over xGMI and are hipGraph-capturable),
"gloo" is used by the CPU tests.  `linear_op` stays the GEMM plug point (tensor_parallel.py:50,
:125): the DeepSeek modules pass the HIP fp8 linear through it.

All collectives go through `all_reduce` / `all_gather_last_dim` below so the decode step has ONE
place where communication is issued (and where a custom xGMI all-reduce can later replace RCCL).
"""

__all__ = [
    "init_tp",
    "get_tp_group",
    "get_tp_size",
    "get_tp_rank",
    "ColumnParallelLinear",
    "RowParallelLinear",
    "VocabParallelEmbedding",
]

import torch
import torch.distributed as dist

tp_comm_group = None


def generate_tp_rank_list(tp_size: int, pp_size: int):
    return torch.arange(tp_size * pp_size).reshape(pp_size, tp_size).tolist()


def init_tp(tp_size: int, pp_size: int = 1):
    """Create the TP groups (ranks laid out [pp, tp], tensor_parallel.py:16-27)."""
    global tp_comm_group
    rank_list = generate_tp_rank_list(tp_size, pp_size)
    global_rank = dist.get_rank()
    for ranks in rank_list:
        group = dist.new_group(ranks)
        if global_rank in ranks:
            tp_comm_group = group


def reset_tp():
    global tp_comm_group
    tp_comm_group = None


def get_tp_group():
    return tp_comm_group


def get_tp_size():
    return tp_comm_group.size() if tp_comm_group is not None else 1


def get_tp_rank():
    return dist.get_rank(group=get_tp_group()) if tp_comm_group is not None else 0


# Piecewise hipGraph capture (deepseek_v3.DeepSeekV3Decoder.decode, mode "piecewise"): while a step is
# being captured this is a callable `cut(run)`; every collective below then ENDS the graph piece being
# recorded, hands its own launch (`run`, or None when the group has one rank) to the capture to be issued
# eagerly between the pieces at replay time, and a new piece begins.  No collective is ever recorded into
# a graph that way, so the N > 1 step does not depend on RCCL being capturable.
_graph_break = None


def all_reduce(t: torch.Tensor) -> torch.Tensor:
    """Sum over the TP group, in place (tensor_parallel.py:166, model_deepseek_v3.py:1011)."""
    run = (lambda: dist.all_reduce(t, group=get_tp_group())) if get_tp_size() > 1 else None
    if _graph_break is not None:
        _graph_break(run)
    elif run is not None:
        run()
    return t


def all_gather_last_dim(y: torch.Tensor) -> torch.Tensor:
    """Concatenate the last dimension over TP ranks (tensor_parallel.py:94-102: the reference
    gathers on a permuted [N/tp, ...] layout so the result is rank-major along the last dim)."""
    tp = get_tp_size()
    if tp == 1:
        return y
    y_t = y.permute(-1, *range(y.dim() - 1)).contiguous()
    shape = list(y_t.shape)
    shape[0] *= tp
    gathered = y.new_empty(shape)
    run = lambda: dist.all_gather_into_tensor(gathered, y_t, group=get_tp_group())
    if _graph_break is not None:
        _graph_break(run)
    else:
        run()
    return gathered.permute(*range(1, y.dim()), 0)


class _ShardedLinear(torch.nn.Module):
    """Shared part of the two Megatron linears: group bookkeeping and the `[out, in]` weight (and bias) of THIS
    rank, `split` naming the dimension the full matrix is cut along ("out": rows / column-parallel, "in":
    columns / row-parallel).  Parameters are uninitialised, like the reference's (the loader fills them)."""

    def __init__(self, in_features, out_features, split, has_bias, dtype, bias_dtype, linear_op):
        super().__init__()
        self.tp_group, self.tp_size, self.rank = get_tp_group(), get_tp_size(), get_tp_rank()
        self.in_features, self.out_features, self.linear_op = in_features, out_features, linear_op
        cut = out_features if split == "out" else in_features
        assert cut % self.tp_size == 0, f"{split}_features must be divisible by tp_size"
        rows = out_features // self.tp_size if split == "out" else out_features
        cols = in_features if split == "out" else in_features // self.tp_size
        self.weight = torch.nn.Parameter(torch.empty(rows, cols, dtype=dtype), requires_grad=False)
        self.bias = torch.nn.Parameter(torch.empty(rows, dtype=bias_dtype or dtype), requires_grad=False) if has_bias else None


class ColumnParallelLinear(_ShardedLinear):
    """y = x W_r^T (+ b_r) with the OUTPUT features split over the ranks; gather_output concatenates the ranks'
    slices along the last dimension (tensor_parallel.py:42-103)."""

    def __init__(self, in_features, out_features, has_bias=True, gather_output=True, dtype=None, bias_dtype=None,
                 linear_op=torch.nn.functional.linear):
        super().__init__(in_features, out_features, "out", has_bias, dtype, bias_dtype, linear_op)
        self.gather_output = gather_output

    def forward(self, x):
        y = self.linear_op(x, self.weight, self.bias)
        return all_gather_last_dim(y) if self.gather_output and self.tp_size > 1 else y


class RowParallelLinear(_ShardedLinear):
    """y = sum over ranks of x_r W_r^T with the INPUT features split; the bias is added once (by rank 0, before the
    all-reduce); an input that is not already this rank's slice is sliced here (tensor_parallel.py:106-169)."""

    def __init__(self, in_features, out_features, has_bias=True, input_is_parallel=False, dtype=None, bias_dtype=None,
                 linear_op=torch.nn.functional.linear):
        super().__init__(in_features, out_features, "in", has_bias, dtype, bias_dtype, linear_op)
        self.input_is_parallel = input_is_parallel

    def forward(self, x):
        if self.tp_size == 1:
            return self.linear_op(x, self.weight, self.bias)
        if not self.input_is_parallel:
            x = x.unflatten(-1, (self.tp_size, -1)).select(-2, self.rank)
        return all_reduce(self.linear_op(x, self.weight, self.bias if self.rank == 0 else None))


class VocabParallelEmbedding(torch.nn.Module):
    """Embedding table split by vocabulary rows: ids outside this rank's range look up row 0 and are zeroed, the
    all-reduce assembles the batch (tensor_parallel.py:172-208).  The caller's id tensor is not modified."""

    def __init__(self, num_embeddings, embedding_dim, dtype=None):
        super().__init__()
        self.tp_group, self.tp_size, self.rank = get_tp_group(), get_tp_size(), get_tp_rank()
        assert num_embeddings % self.tp_size == 0, "num_embeddings must be divisible by tp_size"
        per_rank = num_embeddings // self.tp_size
        self.vocab_start_idx, self.vocab_end_idx = self.rank * per_rank, (self.rank + 1) * per_rank
        self.weight = torch.nn.Parameter(torch.empty(per_rank, embedding_dim, dtype=dtype), requires_grad=False)

    def forward(self, x):
        if self.tp_size == 1:
            return torch.nn.functional.embedding(x, self.weight)
        local = x - self.vocab_start_idx
        foreign = (local < 0) | (local >= self.weight.shape[0])
        y = torch.nn.functional.embedding(local.masked_fill(foreign, 0), self.weight)
        return all_reduce(y.masked_fill(foreign.unsqueeze(-1), 0))
