"""Counter-based synthetic data for the cDDPM reverse path (host side, numpy only).

Everything random that the path consumes -- synthetic UNet weights, context vectors
``cond``, the start image ``x_T`` and the per-step Gaussian draws ``z_t`` -- comes from
Philox4x32-10 keyed by ``(seed, stream)`` and counted by ``(element quad, t, slice)``.
A slice's values depend only on its GLOBAL slice index, never on the batch it sits in
or the rank that owns it, so any shard can regenerate its part independently and a
1-GPU run equals an N-GPU run bit for bit.

The device generator in ``csrc/step_kernels.hip`` (``cddpm_noise_fill`` and the fused
posterior step) uses the same key/counter convention; the integer stream is identical,
the Box-Muller floats agree to a few ulp (device ``logf``/``sincosf`` vs numpy).

Reference behaviour this replaces: ``torch.randn(shape)`` for x_T and one
``torch.randn_like(x)`` per step t = T-1 .. 1 (reference src/models/modules/cond_DDPM.py:454, :440).
"""
from __future__ import annotations

import math
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)

# stream ids (counter word 3): what a draw is for
STREAM_XT = 0x1001      # start image x_T
STREAM_Z = 0x1002       # per-step posterior noise z_t
STREAM_COND = 0x1003    # synthetic context vectors
STREAM_INPUT = 0x1004   # synthetic input slices in [0,1]
STREAM_WEIGHT = 0x2000  # + tensor ordinal: synthetic weights


def philox4x32(c0, c1, c2, c3, k0: int, k1: int):
    """Philox4x32-10. c* are uint32 arrays (broadcastable), k* python ints. Returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64)
    c3 = np.asarray(c3, dtype=np.uint64)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 &= 0xFFFFFFFF
    k1 &= 0xFFFFFFFF
    for _ in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0 = p0 >> np.uint64(32)
        lo0 = p0 & _MASK32
        hi1 = p1 >> np.uint64(32)
        lo1 = p1 & _MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n1 = lo1
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def _u01(x):
    """uint32 -> float32 uniform strictly inside (0,1): ((x >> 9) + 0.5) * 2^-23 (exact in fp32)."""
    return ((x >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)


def _box_muller(ua, ub):
    r = np.sqrt(np.float32(-2.0) * np.log(ua, dtype=np.float32), dtype=np.float32)
    th = np.float32(2.0 * math.pi) * ub
    return r * np.cos(th, dtype=np.float32), r * np.sin(th, dtype=np.float32)


def normal_quads(nquads: int, c1: int, c2: int, stream: int, seed: int) -> np.ndarray:
    """4*nquads float32 N(0,1) values: quad q -> elements 4q..4q+3."""
    q = np.arange(nquads, dtype=np.uint32)
    x0, x1, x2, x3 = philox4x32(q, np.uint32(c1 & 0xFFFFFFFF), np.uint32(c2 & 0xFFFFFFFF),
                                np.uint32(stream & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    n0, n1 = _box_muller(_u01(x0), _u01(x1))
    n2, n3 = _box_muller(_u01(x2), _u01(x3))
    return np.stack([n0, n1, n2, n3], axis=-1).reshape(-1)


def uniform_quads(nquads: int, c1: int, c2: int, stream: int, seed: int) -> np.ndarray:
    q = np.arange(nquads, dtype=np.uint32)
    xs = philox4x32(q, np.uint32(c1 & 0xFFFFFFFF), np.uint32(c2 & 0xFFFFFFFF),
                    np.uint32(stream & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return np.stack([_u01(x) for x in xs], axis=-1).reshape(-1)


def _per_slice(fn, n_elems: int, t: int, slice0: int, nslices: int, stream: int, seed: int) -> np.ndarray:
    nq = (n_elems + 3) // 4
    out = np.empty((nslices, n_elems), dtype=np.float32)
    for i in range(nslices):
        out[i] = fn(nq, t, slice0 + i, stream, seed)[:n_elems]
    return out


def noise_xT(seed: int, slice0: int, nslices: int, H: int, W: int) -> np.ndarray:
    """x_T for global slices [slice0, slice0+nslices): float32 [n,1,H,W]."""
    return _per_slice(normal_quads, H * W, 0, slice0, nslices, STREAM_XT, seed).reshape(nslices, 1, H, W)


def noise_z(seed: int, t: int, slice0: int, nslices: int, H: int, W: int) -> np.ndarray:
    """z_t (drawn while stepping t -> t-1, t >= 1): float32 [n,1,H,W]."""
    return _per_slice(normal_quads, H * W, t, slice0, nslices, STREAM_Z, seed).reshape(nslices, 1, H, W)


def synth_cond(seed: int, slice0: int, nslices: int, dim: int = 128) -> np.ndarray:
    """context vectors c ~ N(0,1): float32 [n, dim] (stands in for the Spark encoder output)."""
    return _per_slice(normal_quads, dim, 0, slice0, nslices, STREAM_COND, seed)


def synth_slices(seed: int, slice0: int, nslices: int, H: int, W: int) -> np.ndarray:
    """input slices in (0,1): float32 [n,1,H,W] (only needed for residual maps)."""
    return _per_slice(uniform_quads, H * W, 0, slice0, nslices, STREAM_INPUT, seed).reshape(nslices, 1, H, W)


# ----------------------------------------------------------------------------------------------
# synthetic UNet weights with the reference's state_dict names and shapes
# ----------------------------------------------------------------------------------------------

def unet_param_shapes(model_channels=128, channel_mult=(1, 2, 2), num_res_blocks=3, num_classes=128,
                      in_channels=1, out_channels=1, attention_resolutions=(3, 6, 12)):
    """Ordered {name: shape} of the UNet state_dict, in nn.Module registration order of the
    reference constructor (src/models/modules/OpenAI_Unet.py:513-797; names listed in SURVEY 8a)."""
    shapes = {}
    C = model_channels
    half = 4 * C  # time_embed_dim // fac when num_classes is set; == time_embed_dim otherwise
    E = 2 * half if num_classes is not None else half

    def lin(prefix, i, o):
        shapes[prefix + ".weight"] = (o, i)
        shapes[prefix + ".bias"] = (o,)

    def conv(prefix, i, o, k):
        shapes[prefix + ".weight"] = (o, i, k, k)
        shapes[prefix + ".bias"] = (o,)

    def gn(prefix, c):
        shapes[prefix + ".weight"] = (c,)
        shapes[prefix + ".bias"] = (c,)

    def resblock(prefix, cin, cout):
        gn(prefix + ".in_layers.0", cin)
        conv(prefix + ".in_layers.2", cin, cout, 3)
        lin(prefix + ".emb_layers.1", E, 2 * cout)
        gn(prefix + ".out_layers.0", cout)
        conv(prefix + ".out_layers.3", cout, cout, 3)
        if cin != cout:
            conv(prefix + ".skip_connection", cin, cout, 1)

    def attn(prefix, c):
        gn(prefix + ".norm", c)
        shapes[prefix + ".qkv.weight"] = (3 * c, c, 1)
        shapes[prefix + ".qkv.bias"] = (3 * c,)
        shapes[prefix + ".proj_out.weight"] = (c, c, 1)
        shapes[prefix + ".proj_out.bias"] = (c,)

    if num_classes is not None:
        lin("label_emb.0", num_classes, half)
        lin("label_emb.2", half, half)
    lin("time_embed.0", C, half)
    lin("time_embed.2", half, half)
    conv("input_blocks.0.0", in_channels, C, 3)
    chans = [C]
    ch = C
    ds = 1
    idx = 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            resblock(f"input_blocks.{idx}.0", ch, mult * C)
            ch = mult * C
            if ds in attention_resolutions:
                attn(f"input_blocks.{idx}.1", ch)
            chans.append(ch)
            idx += 1
        if level != len(channel_mult) - 1:
            resblock(f"input_blocks.{idx}.0", ch, ch)
            chans.append(ch)
            idx += 1
            ds *= 2
    resblock("middle_block.0", ch, ch)
    attn("middle_block.1", ch)
    resblock("middle_block.2", ch, ch)
    idx = 0
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            ich = chans.pop()
            resblock(f"output_blocks.{idx}.0", ch + ich, mult * C)
            ch = mult * C
            sub = 1
            if ds in attention_resolutions:
                attn(f"output_blocks.{idx}.{sub}", ch)
                sub += 1
            if level and i == num_res_blocks:
                resblock(f"output_blocks.{idx}.{sub}", ch, ch)
                ds //= 2
            idx += 1
    gn("out.0", ch)
    conv("out.2", C, out_channels, 3)
    return shapes


def synth_state_dict(seed: int = 0, **unet_kwargs):
    """Synthetic, everywhere non-zero UNet weights (numpy float32), reference names.

    conv/linear weight and bias ~ U(-1/sqrt(fan_in), +1/sqrt(fan_in)); GroupNorm gamma = 1 + 0.1 n,
    beta = 0.1 n, n ~ N(0,1). The reference zero-initialises out.2, every out_layers.3 and proj_out
    (OpenAI_Unet.py:241-245, :380, :793-797) which makes a fresh model output exactly 0; parity on
    such weights would be vacuous, so nothing here is zero.
    """
    shapes = unet_param_shapes(**unet_kwargs)
    sd = {}
    for ordinal, (name, shape) in enumerate(shapes.items()):
        n = int(np.prod(shape))
        nq = (n + 3) // 4
        stream = STREAM_WEIGHT + ordinal
        is_norm = (".in_layers.0." in name or ".out_layers.0." in name or ".norm." in name
                   or name.startswith("out.0."))
        if is_norm:
            z = normal_quads(nq, 0, 0, stream, seed)[:n]
            v = (np.float32(1.0) + np.float32(0.1) * z) if name.endswith("weight") else np.float32(0.1) * z
        else:
            wshape = shapes[name[: name.rfind(".")] + ".weight"]
            fan_in = int(np.prod(wshape[1:]))
            bound = np.float32(1.0 / math.sqrt(fan_in))
            u = uniform_quads(nq, 0, 0, stream, seed)[:n]
            v = (np.float32(2.0) * u - np.float32(1.0)) * bound
        sd[name] = np.ascontiguousarray(v.astype(np.float32).reshape(shape))
    return sd


# ----------------------------------------------------------------------------------------------
# synthetic context-encoder weights (timm resnet50, in_chans=1) with timm's state_dict names and shapes
# ----------------------------------------------------------------------------------------------
STREAM_ENC_WEIGHT = 0x3000   # + tensor ordinal
ENCODER_STAGES = ((64, 3), (128, 4), (256, 6), (512, 3))


def encoder_param_shapes(num_classes: int = 128):
    """name -> shape of timm's ResNet-50 (v1.5) with a 1-channel stem, in state_dict order (num_batches_tracked omitted)"""
    shapes = {"conv1.weight": (64, 1, 7, 7)}

    def bn(p, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"{p}.{k}"] = (c,)
    bn("bn1", 64)
    inplanes = 64
    for s, (planes, nblocks) in enumerate(ENCODER_STAGES):
        for i in range(nblocks):
            p = f"layer{s + 1}.{i}"
            shapes[p + ".conv1.weight"] = (planes, inplanes, 1, 1); bn(p + ".bn1", planes)
            shapes[p + ".conv2.weight"] = (planes, planes, 3, 3); bn(p + ".bn2", planes)
            shapes[p + ".conv3.weight"] = (planes * 4, planes, 1, 1); bn(p + ".bn3", planes * 4)
            if i == 0:
                shapes[p + ".downsample.0.weight"] = (planes * 4, inplanes, 1, 1); bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    shapes["fc.weight"] = (num_classes, 2048)
    shapes["fc.bias"] = (num_classes,)
    return shapes


def synth_encoder_state_dict(seed: int = 0, num_classes: int = 128):
    """Synthetic encoder weights (numpy float32): conv / fc ~ U(+-sqrt(3 / fan_in)) (unit gain, keeps activations O(1)
    through 50 layers), BatchNorm gamma = 1 + 0.1 n, beta = 0.1 n, running_mean = 0.1 n, running_var = 1 + 0.2 u."""
    shapes = encoder_param_shapes(num_classes)
    sd = {}
    for ordinal, (name, shape) in enumerate(shapes.items()):
        n = int(np.prod(shape))
        nq = (n + 3) // 4
        stream = STREAM_ENC_WEIGHT + ordinal
        leaf = name.rsplit(".", 1)[1]
        if len(shape) == 1 and not name.startswith("fc."):
            if leaf == "running_var":
                v = np.float32(1.0) + np.float32(0.2) * uniform_quads(nq, 0, 0, stream, seed)[:n]
            else:
                z = normal_quads(nq, 0, 0, stream, seed)[:n]
                v = (np.float32(1.0) + np.float32(0.1) * z) if leaf == "weight" else np.float32(0.1) * z
        else:
            wshape = shapes[name[: name.rfind(".")] + ".weight"]
            fan_in = int(np.prod(wshape[1:]))
            bound = np.float32(math.sqrt(3.0 / fan_in))
            v = (np.float32(2.0) * uniform_quads(nq, 0, 0, stream, seed)[:n] - np.float32(1.0)) * bound
        sd[name] = np.ascontiguousarray(v.astype(np.float32).reshape(shape))
    return sd
