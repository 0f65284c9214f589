"""Golden for chitu_amd/checkpoint.py from the REFERENCE'S OWN loader functions, run on CPU in the build container:

    HF-named tiny checkpoint (tests/util.py::tiny_hf_checkpoint, written as a safetensors file)
      -> chitu.backend.load_state_dict_deepseek_v3            (names, MTP layer dropped; backend.py:431-481)
      -> model._chunk_checkpoint_for_tensor_parallel(rank, 2)  (models/model.py:332-370)
      -> _process_state_dict_for_merging_qkv / _gate_up / _experts  (model_deepseek_v3.py:1167-1271)
    in the order load_state_dict_parallel + TransformerDeepSeekV3.load_state_dict apply them.

Writes tests/golden/ckpt_preprocess.json: per TP rank the ordered list of (name, shape, dtype, sha1 of the bytes).
Run:  python tests/golden/gen_ckpt.py   (~20 s)
"""

import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_shims  # noqa: E402

ref_shims.install()
from gen_ref_model import build_reference_model  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

from tests.util import CKPT_TINY, tensor_digest, tiny_hf_checkpoint  # noqa: E402


def main():
    model, _, _ = build_reference_model(CKPT_TINY, max_seq_len=64)
    import chitu.backend as rbackend

    with tempfile.TemporaryDirectory() as d:
        save_file({k: v.contiguous() for k, v in tiny_hf_checkpoint().items()}, os.path.join(d, "model-00001-of-00001.safetensors"))
        named = rbackend.load_state_dict_deepseek_v3(d)
    out = {"tp": 2, "names_after_rename": list(named.keys()), "ranks": []}
    for rank in range(2):
        st = model._chunk_checkpoint_for_tensor_parallel(named, rank, 2)
        st = model._process_state_dict_for_merging_qkv(st)
        st = model._process_state_dict_for_merging_gate_up(st)
        st = model._process_state_dict_for_merging_experts(st)
        out["ranks"].append([[k] + tensor_digest(v) for k, v in st.items()])
        if rank == 0:  # the reference model accepts its own preprocessed dict: names and shapes are what it expects
            # (tp 1 model here, so only check the key set)
            # (the reference only creates the router bias at dim 7168, model_deepseek_v3.py:804-808)
            diff = set(st.keys()) ^ set(dict(model.named_parameters()).keys())
            assert diff <= {"layers.1.ffn.gate.bias"}, sorted(diff)
    with open(os.path.join(HERE, "ckpt_preprocess.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote ckpt_preprocess.json", [len(r) for r in out["ranks"]])


if __name__ == "__main__":
    main()
