"""ORACLE -- test infrastructure, not product code.  PARITY UNPINNED (see below).

CPU restatement (torch functional ops, dtype-generic) of the context encoder the reference builds with
`timm.create_model('resnet50', pretrained=False, in_chans=1, num_classes=cond_dim)` (src/models/modules/DDPM_encoder.py:21-23;
called once per slice batch by DDPM_2D.forward, src/models/DDPM_2D.py:98-104), in eval mode.

timm (pinned 0.6.7, requirements.txt.backup:34) is NOT installed in the build image and the reference tree does not
vendor it, so this file restates timm's published ResNet-50: 7x7/2 stem (no bias) + BatchNorm + ReLU, 3x3/2 max-pool
(pad 1), four stages of [3, 4, 6, 3] bottlenecks (1x1 -> 3x3 (stride here, "v1.5") -> 1x1 x4, BatchNorm after each conv, ReLU
after the first two and after the residual add, 1x1/stride conv + BatchNorm shortcut in the first block of a stage),
global average pool, Linear(2048, num_classes); BatchNorm eps 1e-5, running statistics. State-dict names are timm's.
Neither the reference nor any fixture pins these numbers: tests compare the HIP encoder with THIS restatement only."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

STAGES = ((64, 3), (128, 4), (256, 6), (512, 3))


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def resnet50_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """x [B,1,H,W] -> [B,num_classes]"""
    h = F.relu(_bn(F.conv2d(x, sd["conv1.weight"], None, stride=2, padding=3), sd, "bn1"))
    h = F.max_pool2d(h, kernel_size=3, stride=2, padding=1)
    for s, (planes, nblocks) in enumerate(STAGES):
        for i in range(nblocks):
            p = f"layer{s + 1}.{i}"
            stride = 2 if (i == 0 and s > 0) else 1
            o = F.relu(_bn(F.conv2d(h, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
            o = F.relu(_bn(F.conv2d(o, sd[p + ".conv2.weight"], None, stride=stride, padding=1), sd, p + ".bn2"))
            o = _bn(F.conv2d(o, sd[p + ".conv3.weight"]), sd, p + ".bn3")
            sc = h
            if i == 0:
                sc = _bn(F.conv2d(h, sd[p + ".downsample.0.weight"], None, stride=stride), sd, p + ".downsample.1")
            h = F.relu(o + sc)
    g = h.mean(dim=(2, 3))
    return F.linear(g, sd["fc.weight"], sd["fc.bias"])
