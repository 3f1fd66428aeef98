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


def _bn(x, sd, p, training=False, stats=None):
    if not training:
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)
    # training mode (what DDPM_2D.training_step runs the encoder in): batch statistics; the running buffers are updated on COPIES that
    # `stats` collects (momentum 0.1, unbiased variance -- torch.nn.BatchNorm2d), so that the caller's dict stays untouched
    rm, rv = sd[p + ".running_mean"].detach().clone(), sd[p + ".running_var"].detach().clone()
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, 1e-5)
    if stats is not None:
        stats[p + ".running_mean"], stats[p + ".running_var"] = rm, rv
    return y


def resnet50_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], training: bool = False, drop_scales=None, stats=None,
                     relu_masks=None) -> torch.Tensor:
    """x [B,1,H,W] -> [B,num_classes]. training: BatchNorm on batch statistics; drop_scales: {block name: [B] tensor} = timm's drop_path
    realised as a per-sample scale (0 or 1 / keep_prob) of the residual branch, applied where timm applies it (after bn3, before the add).
    relu_masks: {"bn1", "<block>.bn1", "<block>.bn2", "<block>.out": bool [B,C,H,W]} -- ReLU evaluated as a multiplication by a GIVEN mask:
    gradient tests pass the masks of the implementation under test, so that the float64 yardstick differentiates the same smooth branch of
    the network (an activation within rounding of zero flips its mask between fp32 and float64 and, with 16 samples per channel in the last
    stage, moves that channel's BatchNorm gradients by several percent: an ambiguity of the yardstick, not an error)."""
    bn = lambda t, p: _bn(t, sd, p, training, stats)

    def relu(t, key):
        return F.relu(t) if relu_masks is None else t * relu_masks[key].to(t.dtype)

    h = relu(bn(F.conv2d(x, sd["conv1.weight"], None, stride=2, padding=3), "bn1"), "bn1")
    h = F.max_pool2d(h, kernel_size=3, stride=2, padding=1)
    for s, (planes, nblocks) in enumerate(STAGES):
        for i in range(nblocks):
            p = f"layer{s + 1}.{i}"
            stride = 2 if (i == 0 and s > 0) else 1
            o = relu(bn(F.conv2d(h, sd[p + ".conv1.weight"]), p + ".bn1"), p + ".bn1")
            o = relu(bn(F.conv2d(o, sd[p + ".conv2.weight"], None, stride=stride, padding=1), p + ".bn2"), p + ".bn2")
            o = bn(F.conv2d(o, sd[p + ".conv3.weight"]), p + ".bn3")
            if drop_scales is not None and p in drop_scales:
                o = o * drop_scales[p].to(o.dtype).reshape(-1, 1, 1, 1)
            sc = h
            if i == 0:
                sc = bn(F.conv2d(h, sd[p + ".downsample.0.weight"], None, stride=stride), p + ".downsample.1")
            h = relu(o + sc, p + ".out")
    g = h.mean(dim=(2, 3))
    return F.linear(g, sd["fc.weight"], sd["fc.bias"])
