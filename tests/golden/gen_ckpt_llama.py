"""Golden for the Llama-family half of chitu_amd/checkpoint.py from the REFERENCE'S OWN loader methods, run on CPU here:

    HF-named tiny checkpoint (tests/util.py::tiny_hf_llama_checkpoint: "llama", "merged", "mixtral")
      -> the "model." prefix stripped                                   (backend.py:374-380)
      -> Mixtral only: key renames                                      (model_hf_mixtral.py:171-178)
      -> world > 1: merged qkv / gate_up split again                    (model_hf_llama.py:428-504, 595-600)
      -> _chunk_checkpoint_for_tensor_parallel(rank, world)             (models/model.py:332-370)
      -> _process_state_dict_for_merging_qkv / _gate_up                 (model_hf_llama.py:506-566, 616-618)
    in the order TransformerHFLlama / TransformerHFMixtral.load_state_dict_parallel + load_state_dict apply them,
    for world = 1 and world = 2.

Writes tests/golden/ckpt_preprocess_llama.json: per kind and world, per rank, the ordered (name, shape, dtype, sha1).
Run:  python tests/golden/gen_ckpt_llama.py   (~5 s)
"""

import json
import os
import sys
from types import SimpleNamespace

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_shims  # noqa: E402

ref_shims.install()

from tests.util import HF_LLAMA_TINY, tensor_digest, tiny_hf_llama_checkpoint  # noqa: E402


def reference_pipeline(kind, rank, world):
    import chitu.models.model as rmodel
    from chitu.models.model_hf_llama import TransformerHFLlama
    from chitu.models.model_hf_mixtral import TransformerHFMixtral

    cls = TransformerHFMixtral if kind == "mixtral" else TransformerHFLlama
    m = object.__new__(cls)  # the loader methods only read .params (and the class's name lists): no module is built
    c = HF_LLAMA_TINY
    object.__setattr__(m, "params", SimpleNamespace(n_heads=c["n_heads"], n_kv_heads=c["n_kv_heads"], dim=c["dim"], name="tiny",
                                                    type="hf-mixtral" if kind == "mixtral" else "hf-llama"))
    rmodel.get_tp_rank = lambda: rank  # row-parallel biases stay on TP rank 0 (model.py:361-363); no process group here
    st = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in tiny_hf_llama_checkpoint(kind).items()}
    if kind == "mixtral":
        def map_mixtral_key(k):  # model_hf_mixtral.py:171-176 (a closure there: restated, the only step not called)
            for a, b in ((".block_sparse_moe.", ".mlp."), (".w1.", ".gate_proj."), (".w3.", ".up_proj."), (".w2.", ".down_proj.")):
                k = k.replace(a, b)
            return k

        st = {map_mixtral_key(k): v for k, v in st.items()}
    if world > 1:
        st = m._process_state_dict_for_splitting_qkv(st)
        st = m._process_state_dict_for_splitting_gate_up(st)
        st = m._chunk_checkpoint_for_tensor_parallel(st, rank, world)
    st = m._process_state_dict_for_merging_qkv(st)
    st = m._process_state_dict_for_merging_gate_up(st)
    return st


def main():
    out = {}
    for kind in ("llama", "merged", "mixtral"):
        out[kind] = {}
        for world in (1, 2):
            out[kind][str(world)] = [[[k] + tensor_digest(v) for k, v in reference_pipeline(kind, rank, world).items()]
                                     for rank in range(world)]
    with open(os.path.join(HERE, "ckpt_preprocess_llama.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote ckpt_preprocess_llama.json", {k: {w: [len(r) for r in v] for w, v in d.items()} for k, d in out.items()})


if __name__ == "__main__":
    main()
