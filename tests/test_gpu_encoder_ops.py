"""GPU: the training-mode operators of the context encoder one by one (csrc/encoder_train.hip behind cddpm_op_enc_*) against torch
autograd in float64 -- shapes beyond the ResNet-50's own (odd sizes, every kernel / stride combination) so that a failure of the
end-to-end gradient test (tests/test_gpu_encoder_training.py) can be pinned to an operator. Reference semantics: torch.nn.Conv2d(bias=False,
padding=k//2), BatchNorm2d in training mode, MaxPool2d(3, 2, 1), as timm's ResNet uses them (reference src/models/modules/DDPM_encoder.py:21-23)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    tr = load_pkg("training")
    o = tr.UNetTrainer({"w": torch.zeros(64)}, device=torch.device("cuda", 0))
    o._fit(4, 64, 64)
    return o


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().float().cuda()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous().cpu().double()


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("Cin,Cout,K,stride,H,W", [(64, 64, 3, 1, 7, 9), (64, 128, 3, 2, 9, 7), (128, 64, 1, 1, 5, 6), (64, 128, 1, 2, 8, 5), (256, 64, 3, 2, 4, 4)])
def test_enc_conv_forward_input_and_weight_gradients(ops, Cin, Cout, K, stride, H, W):
    torch.manual_seed(Cin + Cout + K + stride)
    B = 3
    x = torch.randn(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, K, K, dtype=torch.float64) / (Cin * K * K) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=stride, padding=K // 2)
    dy = torch.randn_like(y)
    y.backward(dy)
    lib, h = ops.lib, ops.h
    wd = w.detach().float().cuda().contiguous()
    wf, wdt = torch.empty(wd.numel(), device="cuda"), torch.empty(wd.numel(), device="cuda")
    assert lib.cddpm_op_enc_pack_w(h, wd.data_ptr(), Cout, Cin, K, wf.data_ptr(), wdt.data_ptr(), None) == 0
    xg, dyg = nhwc(x.detach()), nhwc(dy)
    Ho, Wo = y.shape[2], y.shape[3]
    yg = torch.empty(B, Ho, Wo, Cout, device="cuda")
    assert lib.cddpm_op_enc_conv(h, xg.data_ptr(), wf.data_ptr(), yg.data_ptr(), B, H, W, Cin, Cout, K, stride, 0, None) == 0, lib.cddpm_last_error(h)
    dxg = torch.empty(B, H, W, Cin, device="cuda")
    assert lib.cddpm_op_enc_conv(h, dyg.data_ptr(), wdt.data_ptr(), dxg.data_ptr(), B, H, W, Cin, Cout, K, stride, 1, None) == 0, lib.cddpm_last_error(h)
    dwg = torch.empty(Cout, Cin, K, K, device="cuda")
    assert lib.cddpm_op_enc_conv_wgrad(h, xg.data_ptr(), dyg.data_ptr(), dwg.data_ptr(), B, H, W, Cin, Cout, K, stride, None) == 0, lib.cddpm_last_error(h)
    torch.cuda.synchronize()
    e = (rel(nchw(yg), y.detach()), rel(nchw(dxg), x.grad), rel(dwg.cpu().double(), w.grad))
    print(Cin, Cout, K, stride, H, W, "y %.1e dx %.1e dw %.1e" % e)
    assert max(e) < 5e-6


@pytest.mark.parametrize("C_,relu,res,drop", [(64, True, False, False), (128, True, True, True), (256, False, False, False), (64, True, True, False)])
def test_enc_batchnorm_training_forward_backward(ops, C_, relu, res, drop):
    torch.manual_seed(C_ + relu + 2 * res)
    B, H, W = 4, 6, 5
    z = (torch.randn(B, C_, H, W, dtype=torch.float64) * 1.5 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C_, dtype=torch.float64)).requires_grad_(True)
    beta = (0.2 * torch.randn(C_, dtype=torch.float64)).requires_grad_(True)
    r = torch.randn(B, C_, H, W, dtype=torch.float64).requires_grad_(True) if res else None
    ss = torch.tensor([1 / 0.8, 0.0, 1 / 0.8, 1 / 0.8], dtype=torch.float64) if drop else None
    rm, rv = torch.zeros(C_, dtype=torch.float64), torch.ones(C_, dtype=torch.float64)
    lib, h = ops.lib, ops.h
    zg = nhwc(z.detach())
    rmg, rvg = rm.float().cuda(), rv.float().cuda()
    mr, yg = torch.empty(2, C_, device="cuda"), torch.empty_like(zg)
    rg = nhwc(r.detach()) if res else None
    ssg = ss.float().cuda() if drop else None
    p = lambda t: None if t is None else t.data_ptr()
    assert lib.cddpm_op_enc_bn_forward(h, zg.data_ptr(), gamma.detach().float().cuda().data_ptr(), beta.detach().float().cuda().data_ptr(), p(ssg), p(rg),
                                       int(relu), C.c_float(1e-5), C.c_float(0.1), rmg.data_ptr(), rvg.data_ptr(), mr.data_ptr(), yg.data_ptr(),
                                       B * H * W, H * W, C_, None) == 0, lib.cddpm_last_error(h)
    # float64 reference on the implementation's ReLU mask (see oracle/encoder_oracle.py)
    u = F.batch_norm(z, rm, rv, gamma, beta, True, 0.1, 1e-5)
    if drop:
        u = u * ss.reshape(-1, 1, 1, 1)
    if res:
        u = u + r
    mask = nchw(yg) > 0
    y = u * mask if relu else u
    dy = torch.randn_like(y)
    y.backward(dy)
    gam_g = gamma.detach().float().cuda()
    dz, dres = torch.empty_like(zg), (torch.empty_like(zg) if res else None)
    dg, db = torch.empty(C_, device="cuda"), torch.empty(C_, device="cuda")
    assert lib.cddpm_op_enc_bn_backward(h, zg.data_ptr(), yg.data_ptr(), nhwc(dy).data_ptr(), mr.data_ptr(), gam_g.data_ptr(), p(ssg), int(relu),
                                        dz.data_ptr(), p(dres), dg.data_ptr(), db.data_ptr(), B * H * W, H * W, C_, None) == 0, lib.cddpm_last_error(h)
    torch.cuda.synchronize()
    e = {"y": rel(nchw(yg), y.detach()), "dz": rel(nchw(dz), z.grad), "dgamma": rel(dg.cpu().double(), gamma.grad), "dbeta": rel(db.cpu().double(), beta.grad),
         "run_mean": rel(rmg.cpu().double(), rm), "run_var": rel(rvg.cpu().double(), rv)}
    if res:
        e["dres"] = rel(nchw(dres), r.grad)
    print(C_, relu, res, drop, {k: f"{v:.1e}" for k, v in e.items()})
    assert max(e.values()) < 2e-5


def test_enc_maxpool_avgpool_and_stem(ops):
    torch.manual_seed(5)
    lib, h = ops.lib, ops.h
    B, Cc, H, W = 3, 64, 9, 11
    x = torch.randn(B, Cc, H, W, dtype=torch.float64, requires_grad=True)
    y = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn_like(y)
    y.backward(dy)
    xg, dyg = nhwc(x.detach()), nhwc(dy)
    yg, dxg = torch.empty(B, y.shape[2], y.shape[3], Cc, device="cuda"), torch.empty_like(xg)
    assert lib.cddpm_op_enc_maxpool(h, xg.data_ptr(), yg.data_ptr(), B, H, W, Cc, 0, None, None, None) == 0
    assert lib.cddpm_op_enc_maxpool(h, xg.data_ptr(), None, B, H, W, Cc, 1, dyg.data_ptr(), dxg.data_ptr(), None) == 0
    torch.cuda.synchronize()
    # fp32 rounding of the float64 inputs can only matter at exact ties, which random inputs do not have
    assert rel(nchw(yg), y.detach()) < 1e-6 and float((nchw(dxg) - x.grad).abs().max()) < 1e-6
    # global average pool and its backward
    g = torch.empty(B, Cc, device="cuda")
    assert lib.cddpm_op_enc_avgpool(h, xg.data_ptr(), g.data_ptr(), B, H * W, Cc, 0, None) == 0
    dg = torch.randn(B, Cc, device="cuda")
    dxa = torch.empty_like(xg)
    assert lib.cddpm_op_enc_avgpool(h, dg.data_ptr(), dxa.data_ptr(), B, H * W, Cc, 1, None) == 0
    torch.cuda.synchronize()
    assert rel(g.cpu().double(), x.detach().mean(dim=(2, 3))) < 1e-6
    assert rel(nchw(dxa), (dg.cpu().double() / (H * W))[:, :, None, None].expand(B, Cc, H, W)) < 1e-6
    # 7x7 / 2 single-channel stem: forward and weight gradient
    Hs, Ws = 21, 18
    xs = torch.rand(B, 1, Hs, Ws, dtype=torch.float64)
    w = (torch.randn(64, 1, 7, 7, dtype=torch.float64) / 7).requires_grad_(True)
    z = F.conv2d(xs, w, None, stride=2, padding=3)
    dz = torch.randn_like(z)
    z.backward(dz)
    xsg, wg = xs.float().cuda().contiguous(), w.detach().float().cuda().contiguous()
    zg = torch.empty(B, z.shape[2], z.shape[3], 64, device="cuda")
    assert lib.cddpm_op_enc_stem(h, xsg.data_ptr(), wg.data_ptr(), zg.data_ptr(), B, Hs, Ws, None) == 0
    dwg = torch.empty(64, 1, 7, 7, device="cuda")
    assert lib.cddpm_op_enc_stem_wgrad(h, xsg.data_ptr(), nhwc(dz).data_ptr(), dwg.data_ptr(), B, Hs, Ws, None) == 0
    torch.cuda.synchronize()
    assert rel(nchw(zg), z.detach()) < 5e-6 and rel(dwg.cpu().double(), w.grad) < 5e-6
