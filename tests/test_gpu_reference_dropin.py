"""SURVEY 8b end to end: the reference's UNMODIFIED model classes (TransformerDeepSeekV3) on chitu_amd's operator
surface, on the GPU.  Needs a copy of the reference tree on the GPU box: set CHITU_REFERENCE_DIR (the build container
has it at /root/reference; the GPU box does not, so the driver's run skips this file -- the log of a run made with a
staged copy is profiles/r03_reference_dropin.txt).  The work happens in tests/dropin_worker.py, one fresh process per
mode (it rewires sys.modules)."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REF = os.environ.get("CHITU_REFERENCE_DIR", "")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(soft: int):
    # DROPIN_TRACE: the worker's per-op watchdog (enter / leave lines + a Python stack every 30 s on stderr), so a run that
    # does not finish names the op it stopped in (round 5 had one such run and no record of where: profiles/r06_reference_dropin.txt)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", REF_MASTER_PORT=str(29560 + soft), DROPIN_TRACE="1", GLOO_SOCKET_IFNAME="lo")
    env.pop("TRITON_INTERPRET", None)
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_worker.py"), REF, str(soft)], capture_output=True,
                           text=True, timeout=200, env=env, cwd=ROOT)  # (on expiry subprocess.run kills the worker: no orphan on the GPU)
    except subprocess.TimeoutExpired as e:
        tail = (e.stderr or b"")
        tail = tail.decode(errors="replace") if isinstance(tail, bytes) else tail
        pytest.fail("drop-in worker did not finish in 200 s; last trace lines:\n" + tail[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("DROPIN ")]
    assert p.returncode == 0 and lines, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.loads(lines[-1][len("DROPIN "):])
    print(json.dumps(res))
    return res


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "chitu")), reason="CHITU_REFERENCE_DIR is not a reference checkout")
def test_reference_model_runs_unmodified_on_the_hip_operator_surface():
    r = _run(0)
    assert r["fused_experts_calls"] == [[True, False, "torch.float8_e4m3fn"]]  # model_deepseek_v3.py:958-966
    # same weights, same kernels, the reference's wiring vs chitu_amd's fused wiring
    assert max(r["vs_chitu_amd_decoder"]) < 2e-2 and all(r["greedy_equal_chitu_amd_decoder"]) and r["kv_pages_equal_chitu_amd_decoder"]
    assert max(r["vs_same_run_with_oracle_moe"]) < 2e-2
    # the reference's own CPU run of this model went through the Triton interpreter's defective fp8 / bf16 casts
    # (tests/test_oracle_golden.py), which this tiny random model amplifies layer by layer: same function, loose bar
    assert max(r["vs_reference_cpu_run"]) < 0.3


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "chitu")), reason="CHITU_REFERENCE_DIR is not a reference checkout")
def test_reference_model_soft_fp8_branches_on_the_hip_operator_surface():
    """infer.soft_fp8=True (the README's launch line) on a non-NVIDIA device: weight_dequant_soft_fp8 + F.linear for the
    linears (model_deepseek_v3.py:85-98), dequantised experts through fused_experts(use_fp8_w8a8=False) (:975-993)."""
    r = _run(1)
    assert r["fused_experts_calls"] == [[False, False, "torch.bfloat16"]]
    assert max(r["vs_same_run_with_oracle_moe"]) < 2e-2
    assert max(r["vs_reference_cpu_run"]) < 0.3  # unquantised activations vs the fixture's W8A8 run: same model, looser still
