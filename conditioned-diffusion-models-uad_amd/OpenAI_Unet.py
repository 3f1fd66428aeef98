"""Host-side mirror of the reference denoiser class, `UNetModel`
(reference src/models/modules/OpenAI_Unet.py:483-1006).

Same constructor arguments, same `state_dict` keys and shapes (so a Lightning checkpoint of the reference
loads with strict=True), same `forward(x, timesteps, cond=None, context=None)` /
`forward_with_cond_scale(*args, cond_scale=..., **kwargs)` / `convert_to_fp16()` surface -- but the nn.Module
tree below only HOLDS parameters. `forward` hands them (packed once) to the HIP engine; nothing is computed
by torch modules and there is no CPU path.

Supported configuration = the one DDPM_2D builds (reference src/models/DDPM_2D.py:37-59): dims=2,
use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True, num_head_channels=64,
use_spatial_transformer=False, in/out channels 1, model_channels a multiple of 128.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .backend import HipBackend


def _zeroed(module: nn.Module) -> nn.Module:
    for p in module.parameters():
        p.detach().zero_()
    return module


class GroupNorm32(nn.GroupNorm):
    """parameter holder for GroupNorm(32, C) (reference util.py:214-216)"""


class _ResBlockParams(nn.Module):
    """Parameters of one ResBlock, registered under the reference's names (OpenAI_Unet.py:223-268):
    in_layers.{0,2}, emb_layers.1, out_layers.{0,3}, skip_connection."""

    def __init__(self, channels, emb_channels, dropout, out_channels, up=False, down=False):
        super().__init__()
        self.channels, self.out_channels, self.up, self.down = channels, out_channels, up, down
        self.in_layers = nn.Sequential(GroupNorm32(32, channels), nn.SiLU(), nn.Conv2d(channels, out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, 2 * out_channels))
        self.out_layers = nn.Sequential(GroupNorm32(32, out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        _zeroed(nn.Conv2d(out_channels, out_channels, 3, padding=1)))
        self.skip_connection = nn.Identity() if out_channels == channels else nn.Conv2d(channels, out_channels, 1)


class _AttentionParams(nn.Module):
    """Parameters of one AttentionBlock (OpenAI_Unet.py:349-382): norm, qkv (Conv1d), proj_out (Conv1d)."""

    def __init__(self, channels):
        super().__init__()
        self.norm = GroupNorm32(32, channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = _zeroed(nn.Conv1d(channels, channels, 1))


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=True, num_heads=1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, legacy=True, num_mem_kv=0):
        super().__init__()
        unsupported = []
        if dims != 2: unsupported.append(f"dims={dims}")
        if not use_scale_shift_norm: unsupported.append("use_scale_shift_norm=False")
        if not resblock_updown: unsupported.append("resblock_updown=False")
        if not use_new_attention_order: unsupported.append("use_new_attention_order=False")
        if use_spatial_transformer or context_dim is not None: unsupported.append("use_spatial_transformer")
        if num_head_channels != 64: unsupported.append(f"num_head_channels={num_head_channels}")
        if in_channels != 1 or out_channels != 1: unsupported.append("in/out channels != 1")
        if model_channels % 128: unsupported.append(f"model_channels={model_channels} (need a multiple of 128)")
        if unsupported:
            raise NotImplementedError("the HIP path implements the cDDPM configuration only (DDPM_2D.py:37-59); "
                                      "unsupported: " + ", ".join(unsupported))
        self.image_size = image_size          # stored, never read in forward (as in the reference)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.model_channels = model_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = tuple(attention_resolutions)
        self.dropout = dropout
        self.channel_mult = tuple(int(m) for m in channel_mult)
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float32            # convert_to_fp16 is a no-op in the reference (:23-28): fp32 it is
        self.num_heads, self.num_head_channels, self.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample
        self.features_info = {}               # the reference's write-only debug collector; kept empty

        C = model_channels
        if num_classes is not None:
            half = 4 * C
            emb_dim = 2 * half
            self.label_emb = nn.Sequential(nn.Linear(num_classes, half), nn.SiLU(), nn.Linear(half, half))
        else:
            half = emb_dim = 4 * C
        self.time_embed = nn.Sequential(nn.Linear(C, half), nn.SiLU(), nn.Linear(half, half))

        self.input_blocks = nn.ModuleList([nn.Sequential(nn.Conv2d(in_channels, C, 3, padding=1))])
        chans, ch, ds = [C], C, 1
        for level, mult in enumerate(self.channel_mult):
            for _ in range(num_res_blocks):
                layers = [_ResBlockParams(ch, emb_dim, dropout, mult * C)]
                ch = mult * C
                if ds in self.attention_resolutions:
                    layers.append(_AttentionParams(ch))
                self.input_blocks.append(nn.Sequential(*layers))
                chans.append(ch)
            if level != len(self.channel_mult) - 1:
                self.input_blocks.append(nn.Sequential(_ResBlockParams(ch, emb_dim, dropout, ch, down=True)))
                chans.append(ch)
                ds *= 2
        self.middle_block = nn.Sequential(_ResBlockParams(ch, emb_dim, dropout, ch), _AttentionParams(ch),
                                          _ResBlockParams(ch, emb_dim, dropout, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(self.channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [_ResBlockParams(ch + chans.pop(), emb_dim, dropout, mult * C)]
                ch = mult * C
                if ds in self.attention_resolutions:
                    layers.append(_AttentionParams(ch))
                if level and i == num_res_blocks:
                    layers.append(_ResBlockParams(ch, emb_dim, dropout, ch, up=True))
                    ds //= 2
                self.output_blocks.append(nn.Sequential(*layers))
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(), _zeroed(nn.Conv2d(ch, out_channels, 3, padding=1)))
        self._hip = HipBackend(self)

    # ---- reference surface -------------------------------------------------------------------------
    def convert_to_fp16(self):
        """no-op, as in the reference (OpenAI_Unet.py:23-28, :799-805): weights stay fp32"""

    def convert_to_fp32(self):
        """no-op"""

    def forward_with_cond_scale(self, *args, cond_scale=2., **kwargs):
        # the reference ignores cond_scale (OpenAI_Unet.py:814-821)
        return self.forward(*args, **kwargs)

    @torch.no_grad()
    def forward(self, x, timesteps, cond=None, context=None):
        """x [N,1,H,W] fp32 on the GPU, timesteps [N] (integer valued), cond [N, num_classes] -> [N,1,H,W]."""
        if context is not None:
            raise NotImplementedError("cross-attention context (SpatialTransformer) is not part of the cDDPM path")
        if self.num_classes is None:
            cond = None
        elif cond is None:
            raise ValueError("this UNet is conditional (num_classes set): cond is required")
        if torch.is_autocast_enabled():
            x = x.float()
        B, _c, H, W = x.shape
        eng = self._hip.get(self, B, H, W, x.device)
        t = timesteps
        if isinstance(t, torch.Tensor):
            if t.is_floating_point() and bool((t != t.round()).any()):
                raise NotImplementedError("fractional timesteps are not supported by the table-driven embedding")
            t = t.to(torch.int32)
            if t.numel() == 1:
                t = t.expand(B).contiguous()
        return eng.unet_forward(x.float(), t, cond.float() if cond is not None else None)

    def hip_engine(self, B, H, W, device):
        """the packed engine for this module (used by GaussianDiffusion to run the whole loop natively)"""
        return self._hip.get(self, B, H, W, torch.device(device))
