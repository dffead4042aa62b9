"""ipoke_conv_wgrad_batched (the flow engine's batched weight gradients of the coupling nets, macow_utils.py:270-281) against torch's
conv weight gradient on the same bf16 operands: the 128 x 128 tiles of wide outputs (conv1 / conv2) and the 64 x 256 tiles of narrow
ones (conv3: <= 64 output rows), several problems per launch, ragged output widths."""
import struct
from ctypes import byref

import pytest
import torch
import torch.nn.functional as F

from ipoke_amd import _lib, ops
from ipoke_amd._lib import WgradDesc, check

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("B,Cin,Cout,k,nb", [(3, 256, 64, 3, 2), (5, 512, 60, 3, 3), (4, 256, 24, 3, 1), (2, 128, 64, 3, 2),
                                               (3, 64, 192, 3, 2), (3, 256, 256, 1, 2)])
def test_batched_weight_gradient_matches_torch(B, Cin, Cout, k, nb):
    lib = _lib.lib()
    M = B * 64
    gen = torch.Generator().manual_seed(B * 100 + Cout + k)
    pad = k // 2
    ldy = -(-Cout // 8) * 8
    xs = [(torch.randn(B, Cin, 8, 8, generator=gen) * 0.5).to(torch.bfloat16) for _ in range(nb)]
    dys = [(torch.randn(B, Cout, 8, 8, generator=gen) * 0.5).to(torch.bfloat16) for _ in range(nb)]
    # reference: d/dW of sum(conv(x, W) * dy) on the bf16-rounded operands, fp32 accumulation
    refs = []
    for x, dy in zip(xs, dys):
        w = torch.zeros(Cout, Cin, k, k, requires_grad=True)
        y = F.conv2d(x.float(), w, padding=pad)
        (y * dy.float()).sum().backward()
        refs.append(w.grad.clone())
    # device operands: channels-last activations [M][Cin], gradients [M][ldy] (zero padded columns), one flat fp32 buffer of dW
    a_all = torch.stack([x.permute(0, 2, 3, 1).reshape(M, Cin) for x in xs]).contiguous().to(DEV)
    y_all = torch.zeros(nb, M, ldy, dtype=torch.bfloat16)
    for i, dy in enumerate(dys):
        y_all[i, :, :Cout] = dy.permute(0, 2, 3, 1).reshape(M, Cout)
    y_all = y_all.to(DEV)
    dw = torch.full((nb, Cout, Cin, k * k), float("nan"), device=DEV)
    es = lib.ipoke_wgrad_batch_entry_size()
    assert es == struct.calcsize("qqqiiii")
    raw = b"".join(struct.pack("qqqiiii", i * M * Cin * 2, i * M * ldy * 2, i * Cout * Cin * k * k, k, k, pad, pad) for i in range(nb))
    entries = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(DEV)
    d = WgradDesc()
    d.NB, d.Di, d.Hi, d.Wi, d.Do, d.Ho, d.Wo = B, 1, 8, 8, 1, 8, 8
    d.kd, d.kh, d.kw, d.sd, d.sh, d.sw, d.pd, d.ph, d.pw = 1, k, k, 1, 1, 1, 0, pad, pad
    d.a_f32 = 0; d.a_sn = 64 * Cin; d.a_sh = 8 * Cin; d.a_sw = Cin; d.a_sc = 1; d.Kc_real = Cin; d.Kc = Cin
    d.ldy = ldy; d.Nout = Cout
    d.w_sn = Cin * k * k; d.w_sc = k * k; d.w_st = 1
    check(lib.ipoke_conv_wgrad_batched(byref(d), entries.data_ptr(), nb, a_all.data_ptr(), y_all.data_ptr(), dw.data_ptr(), _lib.BF16,
                                       ops._s()))
    torch.cuda.synchronize()
    got = dw.cpu().view(nb, Cout, Cin, k, k)
    assert not torch.isnan(got).any(), "unwritten weight-gradient elements"
    for i in range(nb):
        err = (got[i] - refs[i]).abs().max().item()
        assert err <= 2e-3 * max(1.0, refs[i].abs().max().item()), (i, err)
