"""Where does the fused MLA tail differ from the two launches?  (debug aid for tests/test_gpu_mla.py's bit-identity test)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chitu_amd import ops, workspace
from chitu_amd.attn_backend import HipAttnBackend
from tests.test_gpu_mla import make_case, _uv_weights

bs, H, lens = 16, 16, [1024] * 16
pages = sum((l + 63) // 64 for l in lens) + 2
q_nope, q_pe, cache, table, sl = make_case(bs, H, lens, pages, seed=bs * 7 + H)
be = HipAttnBackend(local_n_heads=H)
dev = [t.cuda() for t in (q_nope, q_pe, cache, sl, table)]
w_uv, sc = _uv_weights(H)
part = be.mla_decode(*dev, 0.1352, return_partials=True)
S = part[1]
nb = bs * H * S * (512 * 2 + 4)
saved = part[0][:nb].clone()
want = ops.mla_merge_absorb_uv_quant_fp8(part[0], S, bs, w_uv, sc, 4, 8, 1)
wq, wsc = want[0].clone(), want[1].clone()
for poison in (False, True, True):
    if poison:
        part[0][:nb].fill_(0x7F)  # bf16 0x7f7f = 3.4e38, lse 0x7f7f7f7f = huge
        torch.cuda.synchronize()
    got = be.mla_decode_merge_uv_quant(*dev, 0.1352, w_uv, sc, 4, 8, 1, num_splits=S)
    torch.cuda.synchronize()
    ws_eq = torch.equal(part[0][:nb], saved)
    dq = (got[0].view(torch.uint8) != wq.view(torch.uint8)).view(bs, H, 128)
    ds = (got[1].view(torch.int32) != wsc.view(torch.int32))
    bad = dq.any(-1) | ds
    print(f"poison={poison} workspace identical after fused launch: {ws_eq}; rows (b,h) that differ: {int(bad.sum())} of {bs*H}")
    if bad.any():
        idx = bad.nonzero()[:12].tolist()
        print("  first differing (b,h):", idx)
        b0, h0 = idx[0]
        print("  scales got/want:", got[1][b0, h0].item(), wsc[b0, h0].item(), " codes differing in row:", int(dq[b0, h0].sum()))
        print("  per-column-tile diff count:", dq[b0, h0].view(8, 16).sum(-1).tolist())
    if not ws_eq:
        d = (part[0][:nb] != saved).nonzero().flatten()
        print("  workspace bytes differing:", d.numel(), "first at", d[:8].tolist())
