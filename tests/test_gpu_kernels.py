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


# ------------------------------------------------------------------------------------------------------------------
# Edges of the arithmetic domain. The default convolution family carries fp32 products on the
# fp16 matrix pipe: every operand is hi + mid (two fp16 terms), a*b = hi*hi + hi*mid + mid*hi. Error budget per product:
#   operand representation   <= 2^-23 |x| each while mid is a normal fp16 (|x| >= 2^-2); ABSOLUTE <= 2^-25 below that
#                               (activations as they are; weights are pre-scaled so that max|w| lands in [2^13, 2^14),
#                               i.e. their absolute term is 2^-25 * 2^-13 max|w| -- negligible)
#   dropped mid*mid           <= 2^-22 |ab|
#   fp32 accumulation         ~ sqrt(K) 2^-24 of the partial sums
# => |err| <= 2^-20 sum|a||w| + 2^-24 sum|w|  (documented bound, a factor ~4 above the worst case of the model), where
# the second term is the absolute error of activations below 2^-2. The exact families (CDDPM_CONV=x6 / f32) have no
# absolute term. Inputs with |activation| >= 65504 are OUTSIDE the domain of the default family: fp16 overflows and the
# result is non-finite (loud), the exact families stay finite. All references below are float64.
# ------------------------------------------------------------------------------------------------------------------
import os

FAMILY = {"f32": "f32", "x6": "x6"}.get(os.environ.get("CDDPM_CONV", ""), "h3")


def conv_bound(v64, w64, pad):
    s1 = F.conv2d(v64.abs(), w64.abs(), None, padding=pad)
    s2 = F.conv2d(torch.ones_like(v64), w64.abs(), None, padding=pad)
    # the f32 family is a plain k-ordered fmaf chain folded per 288 products (bitwise what an fp32 VALU loop gives, and what the
    # reference's CPU convolution does): its rounding error follows the magnitude of the running partial sum, so one huge term
    # costs every later add an ulp of IT -- four times the relative allowance of the split families' three-level accumulation
    rel, ab = (2.0 ** -18 if FAMILY == "f32" else 2.0 ** -20), (2.0 ** -24 if FAMILY == "h3" else 0.0)
    return rel * s1 + ab * s2 + 1e-30


def check_conv64(got_nhwc, v, wt, bias, pad, extra=None, extra_bound=None):
    v64, w64 = v.double(), wt.double()
    ref = F.conv2d(v64, w64, bias.double(), padding=pad)
    lim = conv_bound(v64, w64, pad)
    if extra is not None:
        ref = ref + extra
        lim = lim + extra_bound
    err = (nchw(got_nhwc).double() - ref).abs()
    ratio = float((err / lim).max())
    assert ratio <= 1.0, f"max err/bound {ratio:.3f} (max err {float(err.max()):.3e}, max |ref| {float(ref.abs().max()):.3e})"
    return ratio


SCALES = [("act_1e3", 1e3, 1.0), ("act_1e4", 1e4, 1.0), ("act_1e-4", 1e-4, 1.0), ("act_1e-6", 1e-6, 1.0),
          ("w_1e3", 1.0, 1e3), ("w_1e-5", 1.0, 1e-5)]


@pytest.mark.parametrize("name,sa,sw", SCALES, ids=[s[0] for s in SCALES])
def test_conv_domain_scaled_operands(eng, name, sa, sw):
    """activations at 10^3..10^4 and at 10^-4..10^-6 (far below the 2^-2 threshold of the relative bound), weights at
    10^3 / 10^-5 (the pre-scale exponent absorbs them)"""
    torch.manual_seed(hash(name) % 1000)
    B, Cin, Cout, H, W = 2, 128, 128, 8, 32
    x = torch.randn(B, Cin, H, W) * sa
    wt = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5 * sw
    bias = torch.randn(Cout) * 0.1 * sa * sw
    got = eng.op_conv(nhwc(x), None, None, False, 0, wt, bias, None, False, 3)
    r = check_conv64(got, x, wt, bias, 1)
    print(name, FAMILY, f"err/bound {r:.3f}")


def test_conv_domain_mixed_magnitudes_and_weight_outliers(eng):
    """activations log-uniform over 10^-6 .. 10^4 with random signs; a weight tensor with a few 100x outliers, which push the
    per-convolution pre-scale exponent down so that ordinary weights sit 2^7 lower in the fp16 range"""
    torch.manual_seed(5)
    B, Cin, Cout, H, W = 2, 256, 128, 8, 32
    x = torch.sign(torch.randn(B, Cin, H, W)) * 10.0 ** (torch.rand(B, Cin, H, W) * 10 - 6)
    wt = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    idx = torch.randint(0, wt.numel(), (7,))
    wt.view(-1)[idx] *= 100.0
    bias = torch.randn(Cout)
    got = eng.op_conv(nhwc(x), None, None, False, 0, wt, bias, None, False, 3)
    r = check_conv64(got, x, wt, bias, 1)
    print("mixed magnitudes + outliers", FAMILY, f"err/bound {r:.3f}")
    # through the fused transform too: GroupNorm coefficients that blow the activation up to ~10^3 before SiLU
    coef = torch.stack([torch.randn(B, Cin) * 0.2, 300.0 * (1 + 0.2 * torch.randn(B, Cin)), torch.randn(B, Cin)])
    x2 = torch.randn(B, Cin, H, W)
    v = F.silu(((x2.double() - coef[0].double()[:, :, None, None]) * coef[1].double()[:, :, None, None] + coef[2].double()[:, :, None, None]))
    got = eng.op_conv(nhwc(x2), None, coef.cuda(), True, 0, wt, bias, None, False, 3)
    # the transform itself runs in fp32 on the device (affine + SiLU: a few ulp of |v|): widen the bound by 2^-21 |v| per operand
    v64, w64 = v, wt.double()
    ref = F.conv2d(v64, w64, bias.double(), padding=1)
    lim = conv_bound(v64, w64, 1) + 2.0 ** -20 * F.conv2d(v64.abs(), w64.abs(), None, padding=1)
    err = (nchw(got).double() - ref).abs()
    assert float((err / lim).max()) <= 1.0, float((err / lim).max())


def test_conv_skip_segment_with_large_raw_residual_stream(eng):
    """the fused 1x1 skip_connection reads the RAW residual stream (no GroupNorm in front of it, OpenAI_Unet.py:338): feed it
    magnitudes of 10^3..10^4 next to an O(1) main segment; both weight tensors share one pre-scale exponent"""
    torch.manual_seed(11)
    B, Cin, S0, Cout, H, W = 2, 128, 256, 128, 8, 32
    x = torch.randn(B, Cin, H, W)
    sk = torch.randn(B, S0, H, W) * 10.0 ** (3 + torch.rand(B, S0, H, W))
    wt = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    ws = torch.randn(Cout, S0, 1, 1) / S0 ** 0.5
    bias = torch.randn(Cout)
    got = eng.op_conv_skip(nhwc(x), None, False, wt, bias, nhwc(sk), ws)
    extra = F.conv2d(sk.double(), ws.double())
    r = check_conv64(got, x, wt, bias, 1, extra=extra, extra_bound=conv_bound(sk.double(), ws.double(), 0))
    print("skip segment, raw stream 1e3..1e4", FAMILY, f"err/bound {r:.3f}")


def test_conv_beyond_the_fp16_range_is_loud(eng):
    """|activation| >= 65504 is outside the default family's domain: the result must be NON-FINITE (never a finite wrong
    number); the exact families compute it. 65000 (inside) must still meet the bound."""
    torch.manual_seed(3)
    B, Cin, Cout, H, W = 1, 128, 128, 8, 32
    wt = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    bias = torch.zeros(Cout)
    x = torch.randn(B, Cin, H, W)
    x[0, 5, 3, 7] = 65000.0
    check_conv64(eng.op_conv(nhwc(x), None, None, False, 0, wt, bias, None, False, 3), x, wt, bias, 1)
    x[0, 5, 3, 7] = 1.0e5
    got = nchw(eng.op_conv(nhwc(x), None, None, False, 0, wt, bias, None, False, 3))
    if FAMILY == "h3":
        bad = ~torch.isfinite(got)
        assert bool(bad[0, :, 2:5, 6:9].all()), "every output that reads the out-of-range activation must be non-finite"
        assert bool(torch.isfinite(got[0, :, 6:, 20:]).all())
    else:
        check_conv64(nhwc(got), x, wt, bias, 1)


@pytest.mark.parametrize("ratio", [0.0, 10.0, 100.0])
def test_groupnorm_statistics_from_the_conv_epilogue_with_large_mean(eng, ratio):
    """GroupNorm statistics come from the producing convolution's epilogue as fp32 per-tile records, folded in fp64
    (conv_x6.hip epilogue, norm_kernels.hip). A channel group with |mean| >> sigma is the hard case: here the conv bias
    puts the output mean at ratio * sigma. Reference: float64 statistics of the conv OUTPUT AS STORED (isolates the
    statistics path from the convolution's own rounding). Bound on the normalised value: 1e-5 * (1 + ratio)."""
    torch.manual_seed(int(ratio) + 1)
    B, Cin, Cout, H, W = 2, 128, 256, 16, 32
    x = torch.randn(B, Cin, H, W)
    wt = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5          # output sigma ~ 1
    bias = ratio * (1 + 0.05 * torch.randn(Cout))
    gamma, beta = 1 + 0.1 * torch.randn(Cout), 0.1 * torch.randn(Cout)
    out, coef = eng.op_conv_gn(nhwc(x), wt, bias, gamma, beta)
    o64 = nchw(out).double()
    coef = coef.cpu().double()
    ref = F.group_norm(o64, 32, gamma.double(), beta.double(), eps=1e-5)
    got = (o64 - coef[0][:, :, None, None]) * coef[1][:, :, None, None] + coef[2][:, :, None, None]
    err = float((got - ref).abs().max())
    print(f"|mean|/sigma = {ratio:g}: normalised-value max err {err:.3e}")
    assert err <= 1e-5 * (1 + ratio), err
