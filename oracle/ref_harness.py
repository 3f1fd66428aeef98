"""ORACLE tooling -- build-container only. Imports the REFERENCE implementation from /root/reference
by path so that oracle/make_golden.py can record its outputs as golden vectors. Never runs on the
GPU box (the reference does not travel) and is never imported by the shipped package.

How the reference is made importable here: `src.models.modules.cond_DDPM` has three
top-level imports that are unused on this path and absent from the image -- torchvision
(cond_DDPM.py:15), ema_pytorch (:21) and numba via src.utils.generate_noise (:24). Empty stand-in
modules are registered for those names only; no reference source is copied or modified.
One attribute, `use_spatial_transformer`, is read by model_predictions (:401) but never assigned
anywhere in the reference; the harness sets it to False (the experiment's value,
configs/experiment/cDDPM/DDPM_cond_spark_2D.yaml:31).
"""
from __future__ import annotations

import sys
import types
from contextlib import contextmanager

import torch

REF_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns (UNetModel, GaussianDiffusion) classes of the reference."""
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.transforms = _stub("torchvision.transforms")
        tv.utils = _stub("torchvision.utils")
    if "ema_pytorch" not in sys.modules:
        _stub("ema_pytorch", EMA=object)
    if "numba" not in sys.modules:
        def njit(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f
        _stub("numba", njit=njit, prange=range)
    from src.models.modules.OpenAI_Unet import UNetModel  # type: ignore
    from src.models.modules.cond_DDPM import GaussianDiffusion  # type: ignore
    return UNetModel, GaussianDiffusion


def build_reference(sd_torch, image_size=(128, 128), timesteps=1000, objective="pred_x0",
                    model_channels=128, channel_mult=(1, 2, 2), num_classes=128):
    """Reference UNet + GaussianDiffusion with the ctor arguments of src/models/DDPM_2D.py:37-77."""
    UNetModel, GaussianDiffusion = import_reference()
    model = UNetModel(
        image_size=image_size, in_channels=1, model_channels=model_channels, out_channels=1,
        num_res_blocks=3, attention_resolutions=(3, 6, 12), dropout=0, channel_mult=list(channel_mult),
        conv_resample=True, dims=2, num_classes=num_classes, use_checkpoint=False, use_fp16=True,
        num_heads=1, num_head_channels=64, num_heads_upsample=-1, use_scale_shift_norm=True,
        resblock_updown=True, use_new_attention_order=True, use_spatial_transformer=False,
        transformer_depth=1)
    model.convert_to_fp16()  # a no-op in the reference (OpenAI_Unet.py:23-28)
    missing, unexpected = model.load_state_dict(sd_torch, strict=True)
    assert not missing and not unexpected
    diff = GaussianDiffusion(model, image_size=image_size, timesteps=timesteps, sampling_timesteps=timesteps,
                             objective=objective, channels=1, loss_type="l1", p2_loss_weight_gamma=0, cfg=None)
    diff.use_spatial_transformer = False
    diff.eval()
    return model, diff


@contextmanager
def injected_randn(draws):
    """Replace torch.randn / torch.randn_like by an iterator of prepared tensors, so the reference's
    Gaussian branch (cond_DDPM.py:454, :440) consumes OUR counter-RNG draws in its own call order."""
    it = iter(draws)
    orig_randn, orig_like = torch.randn, torch.randn_like

    def randn(*shape, **kw):
        v = next(it)
        shp = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        assert tuple(v.shape) == shp, (v.shape, shp)
        return v.clone()

    def randn_like(x, **kw):
        v = next(it)
        assert v.shape == x.shape
        return v.clone()

    torch.randn, torch.randn_like = randn, randn_like
    try:
        yield
    finally:
        torch.randn, torch.randn_like = orig_randn, orig_like
