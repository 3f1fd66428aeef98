"""GPU parity tests of the individual HIP kernels, called through the C ABI (cddpm_op_*), against plain
torch fp32 CPU ops on the same seeded inputs. Tolerances: fp32 sums of K <= 4608 products in a different
order than torch -> |err| <= 2e-5 * (1 + |ref|) (observed ~1e-6)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(engine_factory):
    return engine_factory(timesteps=50, max_batch=2, max_h=32, max_w=32)


def nhwc(x):  # NCHW cpu -> NHWC cuda
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous().cpu()


def close(got, ref, tol=2e-5):
    err = (got - ref).abs()
    lim = tol * (1 + ref.abs())
    assert bool((err <= lim).all()), f"max err {float(err.max()):.3e} (ref max {float(ref.abs().max()):.3e})"
    return float(err.max())


def test_noise_fill_matches_numpy_philox(eng, synth):
    from conftest import load_pkg
    B, H, W = 3, 16, 24
    for (stream, t, fn) in ((synth.STREAM_XT, 0, lambda: synth.noise_xT(7, 5, B, H, W)),
                            (synth.STREAM_Z, 9, lambda: synth.noise_z(7, 9, 5, B, H, W))):
        got = eng.noise_fill(B, H, W, seed=7, stream_id=stream, t=t, slice0=5).cpu().numpy()
        ref = fn()
        assert np.abs(got - ref).max() < 2e-5
    big = eng.noise_fill(2, 128, 128, seed=3, stream_id=synth.STREAM_Z, t=1).cpu().numpy()
    assert abs(big.mean()) < 0.02 and abs(big.std() - 1) < 0.02


@pytest.mark.parametrize("C0,C1,film", [(128, 0, False), (256, 128, False), (256, 0, True), (256, 256, True)])
def test_gn_coef(eng, C0, C1, film):
    torch.manual_seed(C0 + C1)
    B, H, W = 2, 8, 12
    C = C0 + C1
    x = torch.randn(B, C, H, W) * 1.7 + 0.9
    gamma, beta = 1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C)
    fl = torch.randn(B, 2 * C) * 0.3 if film else None
    x0 = nhwc(x[:, :C0])
    x1 = nhwc(x[:, C0:]) if C1 else None
    coef = eng.op_gn_coef(x0, x1, gamma, beta, fl.cuda() if film else None).cpu()
    xg = x.reshape(B, 32, -1).double()
    mean = xg.mean(-1)
    rstd = 1.0 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-5)
    cpg = C // 32
    mean_c = mean.repeat_interleave(cpg, 1).float()
    a = (rstd.repeat_interleave(cpg, 1) * gamma.double()).float()
    d = beta.expand(B, C).clone()
    if film:
        a = a * (1 + fl[:, :C])
        d = d * (1 + fl[:, :C]) + fl[:, C:]
    close(coef[0], mean_c, 1e-6)
    close(coef[1], a, 1e-5)
    close(coef[2], d, 1e-6)
    # and the normalised tensor it implies equals torch's group_norm (+FiLM)
    ref = F.group_norm(x, 32, gamma, beta, eps=1e-5)
    if film:
        ref = ref * (1 + fl[:, :C, None, None]) + fl[:, C:, None, None]
    got = (x - coef[0][:, :, None, None]) * coef[1][:, :, None, None] + coef[2][:, :, None, None]
    close(got, ref, 1e-5)


CONV_CASES = [
    # name, C0, C1, Cout, k, H, W, coef, silu, upsample, residual ('none'|'same'|'up')
    ("3x3_plain", 128, 0, 128, 3, 8, 32, False, False, False, "none"),
    ("3x3_act_res", 128, 0, 128, 3, 12, 20, True, True, False, "same"),
    ("3x3_concat", 256, 128, 256, 3, 8, 40, True, True, False, "none"),
    ("3x3_up", 256, 0, 256, 3, 16, 24, True, True, True, "up"),
    ("3x3_up_folded", 256, 0, 256, 3, 16, 24, True, True, 2, "up"),
    ("3x3_up_folded_wide", 128, 0, 128, 3, 24, 80, True, True, 2, "none"),
    ("1x1_qkv", 256, 0, 768, 1, 8, 8, True, False, False, "none"),
    ("1x1_proj_res", 256, 0, 256, 1, 12, 36, False, False, False, "same"),
    ("3x3_wide", 128, 0, 256, 3, 4, 96, False, False, False, "none"),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv(eng, case):
    name, C0, C1, Cout, k, H, W, use_coef, silu, up, resmode = case
    torch.manual_seed(len(name) * 7 + C0)
    B = 2
    Cin = C0 + C1
    h, w = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(B, Cin, h, w)
    wt = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout) * 0.1
    coef = None
    v = x
    if use_coef:
        coef = torch.stack([torch.randn(B, Cin) * 0.2, 1 + 0.2 * torch.randn(B, Cin), torch.randn(B, Cin) * 0.2])
        v = (x - coef[0][:, :, None, None]) * coef[1][:, :, None, None] + coef[2][:, :, None, None]
    if silu:
        v = F.silu(v)
    if up:
        v = F.interpolate(v, scale_factor=2, mode="nearest")
    ref = F.conv2d(v, wt, bias, padding=k // 2)
    res = None
    if resmode == "same":
        res = torch.randn(B, Cout, H, W)
        ref = ref + res
    elif resmode == "up":
        res = torch.randn(B, Cout, H // 2, W // 2)
        ref = ref + F.interpolate(res, scale_factor=2, mode="nearest")
    got = eng.op_conv(nhwc(x[:, :C0]), nhwc(x[:, C0:]) if C1 else None, coef.cuda() if use_coef else None, silu, int(up),
                      wt, bias, nhwc(res) if res is not None else None, resmode == "up", k)
    close(nchw(got), ref)


@pytest.mark.parametrize("N", [16, 64, 576, 1024])
def test_attention(eng, N):
    torch.manual_seed(N)
    B, C = 2, 256
    qkv = torch.randn(B, 3 * C, N)
    # reference math: QKVAttention (new order), fp32
    q, k, v = qkv.chunk(3, dim=1)
    heads, ch = C // 64, 64
    s = 1 / (ch ** 0.25)
    wgt = torch.einsum("bct,bcs->bts", (q * s).reshape(B * heads, ch, N), (k * s).reshape(B * heads, ch, N))
    wgt = torch.softmax(wgt.float(), dim=-1)
    ref = torch.einsum("bts,bcs->bct", wgt, v.reshape(B * heads, ch, N)).reshape(B, C, N)
    got = eng.op_attention(qkv.permute(0, 2, 1).contiguous().cuda()).cpu().permute(0, 2, 1)
    close(got, ref, 1e-5)
    # a spiky case: one key dominates each row (exercises the running-max rescale)
    qkv2 = qkv.clone()
    qkv2[:, C:2 * C, N // 2] *= 25.0
    q, k, v = qkv2.chunk(3, dim=1)
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", (q * s).reshape(B * heads, ch, N), (k * s).reshape(B * heads, ch, N)), dim=-1)
    ref2 = torch.einsum("bts,bcs->bct", wgt, v.reshape(B * heads, ch, N)).reshape(B, C, N)
    got2 = eng.op_attention(qkv2.permute(0, 2, 1).contiguous().cuda()).cpu().permute(0, 2, 1)
    close(got2, ref2, 1e-5)
