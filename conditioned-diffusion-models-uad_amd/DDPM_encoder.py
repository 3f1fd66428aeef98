"""Host-side mirror of the reference's context encoder factory (reference src/models/modules/DDPM_encoder.py:6-29):
`get_encoder(cfg) -> (encoder, out_features)`, where `encoder` is what `timm.create_model('resnet50', pretrained=False,
in_chans=1, num_classes=cond_dim)` would be -- same constructor semantics, same state_dict names (conv1, bn1,
layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}, fc; `num_batches_tracked` buffers included), eval-mode
forward `x [B,1,H,W] -> [B, cond_dim]` -- but running on the MI355X through libcddpm_hip.so (csrc/encoder.hip).

PARITY UNPINNED: timm is not importable in the build image; the network follows timm's published ResNet-50 v1.5 and is
checked against the torch restatement in oracle/encoder_oracle.py (tests/test_gpu_encoder.py), not against timm.
No CPU path: without a GPU, or on CPU tensors, forward raises. Training mode (BatchNorm batch statistics, gradients)
is not implemented: the cDDPM evaluation path calls the encoder under torch.no_grad with `.eval()` semantics.
Only `resnet50` is built natively; other backbones / the SparK sparse encoder raise NotImplementedError."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .synth import encoder_param_shapes


class ResNet50Encoder(nn.Module):
    """parameter / buffer tree with timm's names; forward on the device via the C ABI (cddpm_encoder_*)"""

    def __init__(self, num_classes: int = 128, in_chans: int = 1):
        super().__init__()
        if in_chans != 1:
            raise NotImplementedError("the cDDPM context encoder takes single-channel slices (in_chans=1)")
        self.num_classes = int(num_classes)
        self._shapes = encoder_param_shapes(self.num_classes)
        self._flat = {}
        for name, shape in self._shapes.items():
            leaf = name.rsplit(".", 1)[1]
            if leaf in ("running_mean", "running_var"):
                t = torch.zeros(shape) if leaf == "running_mean" else torch.ones(shape)
                self._register(name, t, buffer=True)
                if leaf == "running_var":
                    self._register(name.rsplit(".", 1)[0] + ".num_batches_tracked", torch.zeros((), dtype=torch.long), buffer=True)
            else:
                t = torch.empty(shape)
                if len(shape) > 1:
                    nn.init.kaiming_normal_(t, mode="fan_out", nonlinearity="relu") if len(shape) == 4 else nn.init.normal_(t, std=0.01)
                else:
                    t = torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
                self._register(name, t, buffer=False)
        self._h = None
        self._key = None

    def _register(self, dotted: str, tensor: torch.Tensor, buffer: bool):
        """register `tensor` under a dotted name by creating the intermediate empty Modules (state_dict keys = timm's)"""
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        if buffer:
            mod.register_buffer(parts[-1], tensor)
        else:
            mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))

    # ---- engine plumbing ---------------------------------------------------------------------------
    def _engine(self, B: int, H: int, W: int, device: torch.device):
        lib = _lib.load_library()
        sd = {k: v for k, v in self.state_dict().items() if not k.endswith("num_batches_tracked")}
        key = (device.index or 0, max(B, 64), max(H, 128), max(W, 128), tuple((v.data_ptr(), v._version) for v in sd.values()))
        k0 = self._key
        if self._h is not None and k0 is not None and k0[0] == key[0] and B <= k0[1] and H <= k0[2] and W <= k0[3] and k0[4] == key[4]:
            return lib, self._h
        self.close()
        h = C.c_void_p()
        if lib.cddpm_encoder_create(C.byref(h), self.num_classes, key[1], key[2], key[3], key[0]) != 0:
            raise RuntimeError("cddpm_encoder_create: " + (lib.cddpm_encoder_last_error(None) or b"").decode())
        arrs = {k: v.detach().cpu().float().contiguous().numpy() for k, v in sd.items()}
        names = (C.c_char_p * len(arrs))(*[k.encode() for k in arrs])
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs.values()])
        nums = (C.c_int64 * len(arrs))(*[a.size for a in arrs.values()])
        if lib.cddpm_encoder_load_weights(h, names, ptrs, nums, len(arrs)) != 0:
            msg = (lib.cddpm_encoder_last_error(h) or b"").decode()
            lib.cddpm_encoder_destroy(h)
            raise RuntimeError("cddpm_encoder_load_weights: " + msg)
        self._h, self._key = h, key
        return lib, h

    def close(self):
        if self._h is not None:
            _lib.load_library().cddpm_encoder_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("ResNet50Encoder runs on the MI355X only: pass a CUDA/HIP tensor (no CPU fallback)")
        if x.dim() != 4 or x.shape[1] != 1:
            raise RuntimeError(f"expected [B,1,H,W], got {tuple(x.shape)}")
        x = x.float().contiguous()
        B, _c, H, W = x.shape
        lib, h = self._engine(B, H, W, x.device)
        out = torch.empty((B, self.num_classes), device=x.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if lib.cddpm_encoder_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), B, H, W, C.c_void_p(stream)) != 0:
            raise RuntimeError("cddpm_encoder_forward: " + (lib.cddpm_encoder_last_error(h) or b"").decode())
        return out


def _cfg_get(cfg, key, default=None):
    try:
        v = cfg.get(key, default)
    except AttributeError:
        v = getattr(cfg, key, default)
    return default if v is None else v


class SparK_2D_encoder(nn.Module):
    """reference src/models/modules/spark/Spark_2D.py:268-290: a thin wrapper whose `.encoder` is
    build_encoder(cfg.version, cond_dim, ...) = timm create_model(version, in_chans=1, num_classes=cond_dim,
    drop_path_rate=0.05) (spark/models.py:89-109; stochastic depth is the identity in eval mode). What
    `experiment=cDDPM/DDPM_cond_spark_2D` builds (backbone: Spark_Encoder_2D, version: resnet50): checkpoints carry the
    weights under `encoder.encoder.*`."""

    def __init__(self, cfg):
        super().__init__()
        version = str(_cfg_get(cfg, "version", "resnet50"))
        if version != "resnet50":
            raise NotImplementedError(f"only version=resnet50 of the SparK encoder is built natively (got {version!r})")
        self.cfg = cfg
        self.encoder = ResNet50Encoder(num_classes=int(_cfg_get(cfg, "cond_dim", 128)), in_chans=1)

    def forward(self, x):
        return self.encoder(x)


def get_encoder(cfg):
    """reference get_encoder (DDPM_encoder.py:6-29): (encoder, out_features) with out_features = cfg.cond_dim (default 256);
    'spark' in the backbone name selects SparK_2D_encoder, anything else is a timm model name."""
    backbone = str(_cfg_get(cfg, "backbone", "resnet50"))
    dim = int(_cfg_get(cfg, "cond_dim", 256))
    if "spark" in backbone.lower():
        return SparK_2D_encoder(cfg), dim
    if backbone != "resnet50":
        raise NotImplementedError(f"only the resnet50 context encoder is built natively (got backbone={backbone!r})")
    if _cfg_get(cfg, "pretrained_backbone", False):
        raise NotImplementedError("pretrained_backbone=True would download ImageNet weights; load a checkpoint instead")
    return ResNet50Encoder(num_classes=dim, in_chans=1), dim
