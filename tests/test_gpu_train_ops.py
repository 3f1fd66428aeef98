"""GPU: the operators of the training step (csrc/train_kernels.hip, attention.hip) one by one against torch
autograd in float64 on the same seeded inputs: convolution input / weight gradients, GroupNorm32 / FiLM / SiLU backward, attention and
linear backward, the device weight packer, resampling backward, the one-channel convolutions' gradients, the loss. The step itself
(src/models/DDPM_2D.py:114-135): tests/test_gpu_training.py."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(engine_factory):
    return engine_factory(timesteps=50, max_batch=2, max_h=32, max_w=32)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().float().cuda()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous().cpu()


@pytest.mark.parametrize("Cin,Cout,k,H,W", [(128, 128, 3, 8, 32), (256, 128, 3, 12, 20), (384, 256, 3, 8, 40), (512, 256, 1, 8, 16),
                                             (128, 32, 3, 8, 32), (256, 768, 1, 8, 8)])
def test_conv_dgrad_vs_autograd(eng, Cin, Cout, k, H, W):
    """dL/dx of y = conv2d(x, w, padding = k // 2) for an upstream gradient dy (OpenAI_Unet.py: every Conv2d of a ResBlock,
    skip_connection, qkv / proj_out as 1x1)"""
    torch.manual_seed(Cin + Cout + k)
    B = 2
    x = torch.randn(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, dtype=torch.float64) / (Cin * k * k) ** 0.5
    dy = torch.randn(B, Cout, H, W, dtype=torch.float64)
    F.conv2d(x, w, None, padding=k // 2).backward(dy)
    got = nchw(eng.op_conv_dgrad(nhwc(dy), w.float())).double()
    # same bound as the forward kernel's tests: 2^-20 * sum |dy| |w| per output (+ the fp32 rounding of the float64 weights)
    lim = 2.0 ** -19 * F.conv2d(dy.abs(), w.abs().transpose(0, 1).flip(2, 3), None, padding=k // 2) + 1e-30
    err = (got - x.grad).abs()
    assert float((err / lim).max()) <= 1.0, (float(err.max()), float(x.grad.abs().max()))


@pytest.mark.parametrize("C,film,silu", [(128, False, True), (256, True, True), (384, True, True), (256, False, False), (512, True, True)])
def test_gn_film_silu_backward_vs_autograd(eng, C, film, silu):
    """a = act(GroupNorm32(x) * (1 + scale) + shift) (OpenAI_Unet.py:284-286 in_layers, :325-330 out_layers with FiLM; the
    attention block's norm has no activation): dx, dgamma, dbeta, dscale, dshift for an upstream gradient da"""
    torch.manual_seed(C + int(film) + 2 * int(silu))
    B, H, W = 2, 12, 20
    x = (torch.randn(B, C, H, W, dtype=torch.float64) * 1.7 + 0.6).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(C, dtype=torch.float64)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, dtype=torch.float64)).requires_grad_(True)
    fl = (0.3 * torch.randn(B, 2 * C, dtype=torch.float64)).requires_grad_(True) if film else None
    da = torch.randn(B, C, H, W, dtype=torch.float64)
    u = F.group_norm(x, 32, gamma, beta, eps=1e-5)
    if film:
        u = u * (1 + fl[:, :C, None, None]) + fl[:, C:, None, None]
    a = F.silu(u) if silu else u
    a.backward(da)
    dx, dg, db, dfl = eng.op_gn_silu_backward(nhwc(x.detach()), nhwc(da), gamma.detach().float(), beta.detach().float(),
                                              fl.detach().float().cuda() if film else None, silu=silu)
    def rel(g, r):
        return float((g.double().cpu() - r).abs().max() / (r.abs().max() + 1e-12))
    r = {"dx": rel(nchw(dx), x.grad), "dgamma": rel(dg, gamma.grad), "dbeta": rel(db, beta.grad)}
    if film:
        r["dfilm"] = rel(dfl, fl.grad)
    print(C, film, silu, {k: f"{v:.2e}" for k, v in r.items()})
    assert max(r.values()) < 2e-5, r


@pytest.mark.parametrize("C0,C1,Cout,k,H,W,act", [(128, 0, 128, 3, 8, 32, True), (256, 0, 128, 3, 12, 20, True), (128, 0, 256, 3, 16, 40, False),
                                                  (256, 128, 128, 3, 8, 64, True), (64, 0, 64, 3, 4, 8, True), (256, 256, 256, 3, 8, 8, True),
                                                  (256, 0, 768, 1, 8, 8, True), (256, 256, 256, 1, 8, 16, False), (128, 0, 256, 1, 12, 36, False)])
def test_conv_wgrad_vs_autograd(eng, C0, C1, Cout, k, H, W, act):
    """dL/dW and dL/db of y = conv2d(act(cat[x0, x1]), w, b, padding = k // 2), act = [SiLU]((x - mean) * a + d) with per-(sample,
    channel) coefficients (what GroupNorm / FiLM fold into; OpenAI_Unet.py:284-338; the output path concatenates a skip tensor,
    :948; 1x1: skip_connection on the raw input, qkv behind a GroupNorm without activation, proj_out) for an upstream gradient dy"""
    torch.manual_seed(C0 + C1 + Cout + H + k)
    B, Cin = 2, C0 + C1
    x = torch.randn(B, Cin, H, W, dtype=torch.float64)
    coef = torch.stack([torch.randn(B, Cin) * 0.2, 1 + 0.2 * torch.randn(B, Cin), torch.randn(B, Cin) * 0.2]).double()
    a = x
    if act:
        a = (x - coef[0][:, :, None, None]) * coef[1][:, :, None, None] + coef[2][:, :, None, None]
        if k == 3:
            a = F.silu(a)
    w = (torch.randn(Cout, Cin, k, k, dtype=torch.float64) / (Cin * k * k) ** 0.5).requires_grad_(True)
    bias = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(B, Cout, H, W, dtype=torch.float64)
    F.conv2d(a, w, bias, padding=k // 2).backward(dy)
    dw, db = eng.op_conv_wgrad(nhwc(x[:, :C0]), nhwc(x[:, C0:]) if C1 else None, coef.float().cuda() if act else None, act and k == 3,
                               nhwc(dy), ksize=k)
    ew = float((dw.double().cpu() - w.grad).abs().max() / w.grad.abs().max())
    eb = float((db.double().cpu() - bias.grad).abs().max() / bias.grad.abs().max())
    print(C0, C1, Cout, k, H, W, act, f"dW rel {ew:.2e}  db rel {eb:.2e}")
    assert ew < 1e-5 and eb < 1e-5


@pytest.mark.parametrize("N", [16, 64, 576, 1024])
def test_attention_backward_vs_autograd(eng, N):
    """QKVAttention (OpenAI_Unet.py:457-476, new order): dL/dq, dL/dk, dL/dv; N = 576 is the 96x96 experiment's middle block
    (not a multiple of the 64-wide GEMM tiles), 1024 the 128x128 one"""
    torch.manual_seed(N)
    B, C, ch = 2, 256, 64
    heads = C // ch
    qkv = torch.randn(B, 3 * C, N, dtype=torch.float64, requires_grad=True)
    q, k, v = qkv.chunk(3, dim=1)
    s = 1 / (ch ** 0.25)
    w = torch.softmax(torch.einsum("bct,bcs->bts", (q * s).reshape(B * heads, ch, N), (k * s).reshape(B * heads, ch, N)), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v.reshape(B * heads, ch, N)).reshape(B, C, N)
    da = torch.randn(B, C, N, dtype=torch.float64)
    a.backward(da)
    got = eng.op_attention_backward(qkv.detach().permute(0, 2, 1).contiguous().float().cuda(),
                                    da.permute(0, 2, 1).contiguous().float().cuda()).cpu().double().permute(0, 2, 1)
    rel = float((got - qkv.grad).abs().max() / qkv.grad.abs().max())
    print(N, f"dqkv rel {rel:.2e}")
    assert rel < 1e-5


@pytest.mark.parametrize("M,N,K,silu", [(2, 256, 1024, True), (16, 11776, 1024, True), (7, 512, 128, False), (1000, 512, 512, True)])
def test_linear_backward_vs_autograd(eng, M, N, K, silu):
    """emb_layers (SiLU -> Linear(1024, 2 Cout), all 27 ResBlocks as one [11776, 1024] matrix), time_embed / label_emb MLPs"""
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(N, K, dtype=torch.float64) / K ** 0.5).requires_grad_(True)
    b = torch.zeros(N, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(M, N, dtype=torch.float64)
    F.linear(F.silu(x) if silu else x, w, b).backward(dy)
    dw, db, dx = eng.op_linear_backward(x.detach().float().cuda(), w.detach().float().cuda(), dy.float().cuda(), silu_in=silu)
    r = [float((g.cpu().double() - t).abs().max() / t.abs().max()) for g, t in ((dw, w.grad), (db, b.grad), (dx, x.grad))]
    print(M, N, K, silu, [f"{v:.2e}" for v in r])
    assert max(r) < 1e-5


@pytest.mark.parametrize("Cout,Cin,k", [(128, 128, 3), (256, 384, 3), (768, 256, 1), (128, 384, 1)])
def test_device_weight_packer_is_the_host_packer(eng, Cout, Cin, k):
    """cddpm_op_pack_conv (the training step re-packs the updated weights on the device every step) writes, bit for bit, the image of the
    host packer cddpm_pack_conv_weights for the same exponent: forward image, and the transposed / flipped image of the input gradient"""
    import ctypes as C
    torch.manual_seed(Cout + Cin + k)
    w = (torch.randn(Cout, Cin, k, k) * 0.05).contiguous()
    lib, taps = eng.lib, k * k
    for mode in (0, 1):
        O, I = (Cout, Cin) if mode == 0 else (Cin, Cout)
        src = w if mode == 0 else w.transpose(0, 1).flip(2, 3).contiguous()
        nb = lib.cddpm_packed_conv_bytes(O, I, taps)
        host = np.zeros(nb, dtype=np.uint8)
        e = C.c_int(-1)
        fmt = lib.cddpm_pack_conv_weights(src.numpy().ctypes.data_as(C.POINTER(C.c_float)), O, I, taps, host.ctypes.data, C.byref(e))
        assert fmt == 2 and 0 <= e.value <= 24
        dev = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        wd = w.cuda()
        rc = lib.cddpm_op_pack_conv(eng._h, wd.data_ptr(), Cout, Cin, k, mode, e.value, dev.data_ptr(), None)
        assert rc == 0, lib.cddpm_last_error(eng._h)
        torch.cuda.synchronize()
        assert np.array_equal(dev.cpu().numpy(), host), (mode, int((dev.cpu().numpy() != host).sum()))


def test_batched_device_packer_equals_the_single_image_packer(eng):
    """cddpm_op_pack_conv_batch (one launch over a device table of jobs: what the training step runs after every update) writes the same
    bytes as one cddpm_op_pack_conv per image -- forward, input-gradient and the four classes of a folded-upsample image, mixed sizes"""
    lib = eng.lib
    torch.manual_seed(7)
    specs = [(128, 128, 3, 0, 5), (256, 384, 3, 1, 7), (768, 256, 1, 0, 9), (256, 256, 3, 2, 6), (128, 384, 1, 1, 11)]     # Cout, Cin, k, mode, exponent
    job = np.dtype([("w", "<u8"), ("dst", "<u8"), ("O", "<i4"), ("I", "<i4"), ("taps", "<i4"), ("mode", "<i4"), ("wexp", "<i4"), ("cls", "<i4")])
    rows, keep, want = [], [], []
    for Cout, Cin, k, mode, e in specs:
        w = (torch.randn(Cout, Cin, k, k) * 0.05).cuda().contiguous()
        O, I = (Cin, Cout) if mode == 1 else (Cout, Cin)
        taps = 4 if mode == 2 else k * k
        nb = (4 if mode == 2 else 1) * lib.cddpm_packed_conv_bytes(O, I, taps)
        single, batched = torch.zeros(nb, dtype=torch.uint8, device="cuda"), torch.zeros(nb, dtype=torch.uint8, device="cuda")
        assert lib.cddpm_op_pack_conv(eng._h, w.data_ptr(), Cout, Cin, k, mode, e, single.data_ptr(), None) == 0, lib.cddpm_last_error(eng._h)
        rows += [(w.data_ptr(), batched.data_ptr(), O, I, taps, mode, e, cls) for cls in range(4 if mode == 2 else 1)]
        keep.append(w); want.append((single, batched))
    tab = torch.from_numpy(np.array(rows, dtype=job).view(np.uint8).copy()).cuda()
    units = max((r[2] // 128) * (r[3] // 32) * r[4] * 512 for r in rows)
    assert lib.cddpm_op_pack_conv_batch(eng._h, tab.data_ptr(), len(rows), units, None) == 0, lib.cddpm_last_error(eng._h)
    torch.cuda.synchronize()
    for i, (single, batched) in enumerate(want):
        assert bool(single.any()) and torch.equal(single, batched), specs[i]


@pytest.mark.parametrize("B,C0,Cout,k,H,W,up", [(11, 128, 128, 3, 6, 20, False), (16, 128, 64, 1, 8, 8, False), (9, 128, 128, 3, 8, 16, True),
                                                (8, 256, 128, 3, 5, 9, False)])
def test_conv_wgrad_batch_groups_and_ragged_tiles(eng, B, C0, Cout, k, H, W, up):
    """the 16-bit-pipe weight-gradient kernel contracts over (pixel, sample mod 8): batches that do not fill their last group of 8,
    images that do not fill their last 2 x 8 pixel tile, and the upsampled input of an 'up' ResBlock's first convolution"""
    torch.manual_seed(B + C0 + Cout + H)
    hs, ws = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(B, C0, hs, ws, dtype=torch.float64)
    coef = torch.stack([torch.randn(B, C0) * 0.2, 1 + 0.2 * torch.randn(B, C0), torch.randn(B, C0) * 0.2]).double()
    a = F.silu((x - coef[0][:, :, None, None]) * coef[1][:, :, None, None] + coef[2][:, :, None, None])
    if up:
        a = F.interpolate(a, scale_factor=2, mode="nearest")
    w = (torch.randn(Cout, C0, k, k, dtype=torch.float64) / (C0 * k * k) ** 0.5).requires_grad_(True)
    bias = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(B, Cout, H, W, dtype=torch.float64)
    F.conv2d(a, w, bias, padding=k // 2).backward(dy)
    dw, db = eng.op_conv_wgrad(nhwc(x), None, coef.float().cuda(), True, nhwc(dy), ksize=k, upsample=up)
    ew = float((dw.double().cpu() - w.grad).abs().max() / w.grad.abs().max())
    eb = float((db.double().cpu() - bias.grad).abs().max() / bias.grad.abs().max())
    print(B, C0, Cout, k, H, W, up, f"dW rel {ew:.2e}  db rel {eb:.2e}")
    assert ew < 1e-5 and eb < 1e-5


def _ptr(t):
    return None if t is None else t.data_ptr()


def test_resampling_backward_ops(eng):
    """AvgPool2d(2) backward (cddpm_op_unpool2, scale 1/4) and nearest x2 upsample backward (cddpm_op_sumpool2), plain and accumulating"""
    torch.manual_seed(1)
    B, Cc, H, W = 2, 64, 6, 10
    x = torch.randn(B, Cc, H, W, dtype=torch.float64, requires_grad=True)
    dyp = torch.randn(B, Cc, H // 2, W // 2, dtype=torch.float64)
    F.avg_pool2d(x, 2).backward(dyp)
    out = torch.empty(B, H, W, Cc, device="cuda")
    assert eng.lib.cddpm_op_unpool2(eng._h, nhwc(dyp).data_ptr(), out.data_ptr(), B, H, W, Cc, C.c_float(0.25), 0, None) == 0
    base = torch.randn(B, H, W, Cc, device="cuda")
    acc = base.clone()
    assert eng.lib.cddpm_op_unpool2(eng._h, nhwc(dyp).data_ptr(), acc.data_ptr(), B, H, W, Cc, C.c_float(0.25), 1, None) == 0
    torch.cuda.synchronize()
    assert float((nchw(out).double() - x.grad).abs().max()) < 1e-6
    assert float((acc - base - out).abs().max()) < 1e-6
    xs = torch.randn(B, Cc, H // 2, W // 2, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(B, Cc, H, W, dtype=torch.float64)
    F.interpolate(xs, scale_factor=2, mode="nearest").backward(dy)
    outp = torch.empty(B, H // 2, W // 2, Cc, device="cuda")
    assert eng.lib.cddpm_op_sumpool2(eng._h, nhwc(dy).data_ptr(), outp.data_ptr(), B, H, W, Cc, 0, None) == 0
    torch.cuda.synchronize()
    assert float((nchw(outp).double() - xs.grad).abs().max()) < 2e-6


def test_one_channel_convolution_gradients_and_loss(eng):
    """the UNet's two one-channel convolutions: head Conv2d(C -> 1) behind GroupNorm + SiLU (input gradient cddpm_op_head_dgrad, weight
    gradient cddpm_op_chan_image_corr with sign -1) and the input Conv2d(1 -> C) (weight gradient, sign +1; bias cddpm_op_bias_grad);
    the loss operator (L1 / L2, p2 weights, loss scale)"""
    torch.manual_seed(2)
    B, Cc, H, W = 2, 128, 8, 12
    lib, h = eng.lib, eng._h
    x = torch.randn(B, Cc, H, W, dtype=torch.float64)
    coef = torch.stack([torch.randn(B, Cc) * 0.2, 1 + 0.2 * torch.randn(B, Cc), torch.randn(B, Cc) * 0.2]).double()
    a = F.silu((x - coef[0][:, :, None, None]) * coef[1][:, :, None, None] + coef[2][:, :, None, None]).requires_grad_(True)
    w = (torch.randn(1, Cc, 3, 3, dtype=torch.float64) / (9 * Cc) ** 0.5).requires_grad_(True)
    out = F.conv2d(a, w, None, padding=1)
    dout = torch.randn_like(out)
    out.backward(dout)
    w9 = w.detach().float().reshape(Cc, 9).t().contiguous().cuda()
    doutg = dout.float().cuda().contiguous()
    dact = torch.empty(B, H, W, Cc, device="cuda")
    assert lib.cddpm_op_head_dgrad(h, doutg.data_ptr(), w9.data_ptr(), dact.data_ptr(), B, H, W, Cc, None) == 0
    dw = torch.empty(Cc, 9, device="cuda")
    assert lib.cddpm_op_chan_image_corr(h, nhwc(x).data_ptr(), coef.float().cuda().contiguous().data_ptr(), 1, doutg.data_ptr(), -1, dw.data_ptr(),
                                        B, H, W, Cc, None) == 0
    torch.cuda.synchronize()
    assert float((nchw(dact).double() - a.grad).abs().max() / a.grad.abs().max()) < 2e-6
    assert float((dw.cpu().double().reshape(1, Cc, 3, 3) - w.grad).abs().max() / w.grad.abs().max()) < 5e-6
    # input convolution 1 -> C
    img = torch.rand(B, 1, H, W, dtype=torch.float64)
    w_in = (torch.randn(Cc, 1, 3, 3, dtype=torch.float64) / 3).requires_grad_(True)
    b_in = torch.zeros(Cc, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(img, w_in, b_in, padding=1)
    dy = torch.randn_like(y)
    y.backward(dy)
    dwi, dbi = torch.empty(Cc, 9, device="cuda"), torch.empty(Cc, device="cuda")
    assert lib.cddpm_op_chan_image_corr(h, nhwc(dy).data_ptr(), None, 0, img.float().cuda().contiguous().data_ptr(), 1, dwi.data_ptr(), B, H, W, Cc, None) == 0
    assert lib.cddpm_op_bias_grad(h, nhwc(dy).data_ptr(), B * H * W, Cc, dbi.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert float((dwi.cpu().double().reshape(Cc, 1, 3, 3) - w_in.grad).abs().max() / w_in.grad.abs().max()) < 5e-6
    assert float((dbi.cpu().double() - b_in.grad).abs().max() / b_in.grad.abs().max()) < 5e-6
    # loss
    for l2 in (0, 1):
        o = torch.randn(B, 1, H, W, dtype=torch.float64, requires_grad=True)
        tgt, p2w, S = torch.randn(B, 1, H, W, dtype=torch.float64), torch.tensor([0.7, 1.3], dtype=torch.float64), 256.0
        d = o - tgt
        per = ((d ** 2) if l2 else d.abs()).reshape(B, -1).mean(dim=1) * p2w
        per.mean().backward()
        dog, lb = torch.empty(B, 1, H, W, device="cuda"), torch.empty(B, device="cuda")
        assert lib.cddpm_op_loss(h, o.detach().float().cuda().data_ptr(), tgt.float().cuda().data_ptr(), p2w.float().cuda().data_ptr(), l2, B, H * W,
                                 C.c_float(S), dog.data_ptr(), lb.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert float((lb.cpu().double() - per.detach()).abs().max()) < 1e-6
        assert float((dog.cpu().double() / S - o.grad).abs().max() / o.grad.abs().max()) < 2e-6
