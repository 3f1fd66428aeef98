#!/usr/bin/env python3
"""md5 of the fused convolution's output bytes on fixed random inputs, per library build (CDDPM_LIB): two builds that claim the same
arithmetic in the same order must print the same digests.  usage: CDDPM_LIB=<so> python tools/conv_bits.py"""
import hashlib
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
eng_mod = importlib.import_module("conditioned-diffusion-models-uad_amd.engine")
e = eng_mod.CddpmEngine(timesteps=10, max_batch=4, max_h=64, max_w=64)
torch.manual_seed(0)
for (C0, C1, Cout, H, W, coef, res) in [(128, 0, 128, 64, 64, True, False), (256, 128, 256, 32, 40, True, True), (256, 0, 128, 24, 24, False, False)]:
    B = 3
    x0 = torch.randn(B, H, W, C0, device="cuda")
    x1 = torch.randn(B, H, W, C1, device="cuda") if C1 else None
    cf = torch.stack([torch.randn(B, C0 + C1) * 0.2, 1 + 0.2 * torch.randn(B, C0 + C1), torch.randn(B, C0 + C1) * 0.2]).cuda().contiguous() if coef else None
    w = torch.randn(Cout, C0 + C1, 3, 3) / ((C0 + C1) * 9) ** 0.5
    bias = torch.randn(Cout) * 0.1
    r = torch.randn(B, H, W, Cout, device="cuda") if res else None
    out = e.op_conv(x0, x1, cf, coef, 0, w, bias, r, False, 3)
    torch.cuda.synchronize()
    print((C0, C1, Cout, H, W), hashlib.md5(out.cpu().numpy().tobytes()).hexdigest(), float(out.abs().mean()))
