"""INT8 W8A8 linear: the reference's own test recipe (test/pytest/test_w8a8.py) + oracle parity."""

import numpy as np
import pytest
import torch

from oracle import w8a8 as ow

pytestmark = pytest.mark.gpu


def test_reference_recipe_gemm_and_gemv():
    from chitu_amd.quantize.w8a8 import w8a8_linear

    torch.manual_seed(0)
    # test_w8a8.py:13-29  (m=1024 there; a decode-sized slice of it here plus the full gemv case)
    for m, n, k in [(64, 2048, 4096), (2, 4096, 11008)]:
        a = (torch.randn([m, k]) * 4).to(torch.int8)
        b = (torch.randn([n, k]) * 4).to(torch.int8)
        c = w8a8_linear(a.cuda(), torch.ones(m).cuda(), b.cuda(), torch.ones(n).cuda(), None, torch.float16)
        c1 = torch.mm(a.float(), b.float().T).to(torch.float16)
        assert torch.allclose(c.cpu(), c1, rtol=5e-3, atol=5e-3)
        assert torch.equal(c.cpu(), ow.w8a8_linear(a, torch.ones(m), b, torch.ones(n)))  # exact dot, same scaling


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,K", [(1, 4096), (5, 11008), (33, 256)])
def test_quant_act_bit_exact(rows, K, dtype):
    from chitu_amd.quantize.w8a8 import quant_act

    g = torch.Generator().manual_seed(rows + K)
    x = (torch.randn(rows, K, generator=g) * 3).to(dtype)
    x[0, :7] = 0
    q_ref, s_ref = ow.quant_act(x)
    q, s = quant_act(x.cuda())
    assert torch.equal(s.cpu(), s_ref) and torch.equal(q.cpu(), q_ref)
    z, sz = quant_act(torch.zeros(2, 256, dtype=dtype).cuda())  # clamp(…, 1e-5) path
    assert (z == 0).all() and torch.allclose(sz.cpu(), torch.full((2,), 1e-5 / 127))


@pytest.mark.parametrize("M", [1, 4, 16, 40])
@pytest.mark.parametrize("N,K", [(4096, 4096), (14336, 4096), (4096, 14336), (1000, 384)])
def test_module_vs_oracle(M, N, K):
    """W8A8Linear.from_float(...)(x) vs the oracle pipeline, Mixtral / Llama shapes."""
    from chitu_amd.quantize.w8a8 import W8A8Linear

    if M > 16 and N * K > 3e7:
        pytest.skip("large-M only on small shapes")
    g = torch.Generator().manual_seed(M + N)
    lin = torch.nn.Linear(K, N, bias=True)
    lin.weight.data = torch.randn(N, K, generator=g) * 0.05
    lin.bias.data = torch.randn(N, generator=g) * 0.1
    x = torch.randn(M, 1, K, generator=g).to(torch.float16)
    qw, sw = ow.quant_weight(lin.weight.data.cpu())
    mod_gpu_quant = W8A8Linear.from_float(lin.cuda())  # load-time quantisation with torch ops on the device
    assert (mod_gpu_quant.weight.cpu().int() - qw.int()).abs().max() <= 1  # CPU/GPU torch.div rounding only
    mod = W8A8Linear(K, N, bias=True).cuda()
    mod.weight, mod.scale_channel, mod.bias = qw.cuda(), sw.cuda(), lin.bias.data.half().cuda()
    y = mod(x.cuda())
    assert y.shape == (M, 1, N) and y.dtype == torch.float16
    qx, sx = ow.quant_act(x.view(M, K))
    ref = ow.w8a8_linear(qx, sx, qw, sw, lin.bias.data.cpu().half())
    err = (y.view(M, N).cpu().float() - ref.float()).abs().max() / ref.float().abs().max()
    assert err < 2e-3, err  # identical integers; fp16 rounding of the scaled sum only
    full = torch.nn.functional.linear(x.view(M, K).float(), lin.weight.data.cpu(), lin.bias.data.cpu())
    assert ((y.view(M, N).cpu().float() - full).abs().max() / full.abs().max()) < 5e-2  # int8 quantisation noise


def test_quant_act_matches_reference_fixture():
    """chitu_hip_quant_act_int8 vs the reference's own quant_act output (tests/golden/w8a8_quant.npz):
    incl. an all-zero row (scale clamp) and a 60000 outlier."""
    import numpy as np

    from chitu_amd.quantize import w8a8
    from tests.util import golden

    g = golden("w8a8_quant")
    x = torch.from_numpy(g["x"].view(np.int16)).view(torch.float16)
    q, s = w8a8.quant_act(x.cuda())
    assert np.array_equal(q.cpu().numpy(), g["qx"]) and np.array_equal(s.cpu().numpy(), g["sx"])


@pytest.mark.parametrize("rows,dim", [(1, 4096), (2, 4096), (16, 4096), (5, 7168), (3, 1000)])
def test_rms_norm_with_the_int8_quantiser_in_its_launch_is_bit_identical(rows, dim):
    """ops.rms_norm(add=..., quant="int8") (round 6: the residual add, the norm and quant_act of its rounded output in one
    launch -- Mixtral's ffn_norm in front of the int8 experts) == ops.rms_norm(add=...) followed by quantize.w8a8.quant_act:
    same x_new, same y, same int8 codes, same scales; a row of zeros takes quant_act's 1e-5 floor."""
    from chitu_amd import ops
    from chitu_amd.quantize.w8a8 import quant_act

    g = torch.Generator().manual_seed(rows * 31 + dim)
    x = torch.randn(rows, dim, generator=g).to(torch.bfloat16).cuda()
    a = (torch.randn(rows, dim, generator=g) * 0.3).to(torch.bfloat16).cuda()
    w = (torch.rand(dim, generator=g) + 0.5).to(torch.bfloat16).cuda()
    if rows > 2:
        x[1].zero_()
        a[1].zero_()
    x_new0, y0 = ops.rms_norm(x, w, 1e-5, add=a)
    q0, s0 = quant_act(y0)
    x_new, y, q, s = ops.rms_norm(x, w, 1e-5, add=a, quant="int8")
    assert torch.equal(x_new, x_new0) and torch.equal(y, y0)
    assert q.dtype == torch.int8 and tuple(q.shape) == (rows, dim) and tuple(s.shape) == (rows,)
    assert torch.equal(q, q0) and torch.equal(s, s0)
