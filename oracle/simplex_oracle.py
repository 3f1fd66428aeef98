"""ORACLE -- test infrastructure, not product code.

CPU restatement (numpy, float64, vectorised over pixels) of the reference's simplex noise generator
`gen_noise` (src/utils/generate_noise.py:8-52): OpenSimplex 2-D, 6 octaves, persistence 0.8, start frequency 64,
the same field repeated over the batch, converted float64 -> float16 the way torch's `.half()` does (:12).

Pinned against the reference's own functions (`_init`, `_noise2`, `Simplex_CLASS.rand_2d_octaves`) executed in
pure Python in the build container -- numba is not installed, `@njit` is the identity there (oracle/ref_harness.py)
-- with the outputs stored in tests/golden/simplex.npz by oracle/make_golden_simplex.py.
All citations relative to /root/reference/src/utils/generate_noise.py.
"""
from __future__ import annotations

import numpy as np

STRETCH_CONSTANT2 = -0.211324865405187   # :194
SQUISH_CONSTANT2 = 0.366025403784439     # :195
NORM_CONSTANT2 = 47                      # :201
GRADIENTS2 = np.array([5, 2, 2, 5, -5, 2, -2, 5, 5, -2, 2, -5, -5, -2, -2, -5], dtype=np.int64)   # :143-150


def _wrap64(x: int) -> int:
    """two's-complement int64 overflow, as `overflow()` does through c_int64 (:206-211)"""
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x


def init_perm(seed: int) -> np.ndarray:
    """`_init` (:214-232): 256-entry permutation from three warm-up LCG steps plus one per entry."""
    perm = np.zeros(256, dtype=np.int64)
    source = list(range(256))
    for _ in range(3):
        seed = _wrap64(seed * 6364136223846793005 + 1442695040888963407)
    for i in range(255, -1, -1):
        seed = _wrap64(seed * 6364136223846793005 + 1442695040888963407)
        r = int((seed + 31) % (i + 1))      # Python modulo: result already in [0, i]
        if r < 0:
            r += i + 1
        perm[i] = source[r]
        source[r] = source[i]
    return perm


def _extrapolate2(perm, xsb, ysb, dx, dy):
    """`_extrapolate2` (:235-239), vectorised"""
    index = perm[(perm[xsb & 0xFF] + ysb) & 0xFF] & 0x0E
    return GRADIENTS2[index] * dx + GRADIENTS2[index + 1] * dy


def noise2(x: np.ndarray, y: np.ndarray, perm: np.ndarray) -> np.ndarray:
    """`_noise2` (:252-352) on arrays; every arithmetic step in the reference's order (float64, no fusing)."""
    stretch_offset = (x + y) * STRETCH_CONSTANT2
    xs = x + stretch_offset
    ys = y + stretch_offset
    xsb = np.floor(xs).astype(np.int64)
    ysb = np.floor(ys).astype(np.int64)
    squish_offset = (xsb + ysb) * SQUISH_CONSTANT2
    xb = xsb + squish_offset
    yb = ysb + squish_offset
    xins = xs - xsb
    yins = ys - ysb
    in_sum = xins + yins
    dx0 = x - xb
    dy0 = y - yb
    value = np.zeros_like(x, dtype=np.float64)

    def contrib(val, dx, dy, xv, yv):
        attn = 2 - dx * dx - dy * dy
        a2 = attn * attn
        c = a2 * a2 * _extrapolate2(perm, xv, yv, dx, dy)
        return val + np.where(attn > 0, c, 0.0)

    dx1 = dx0 - 1 - SQUISH_CONSTANT2
    dy1 = dy0 - 0 - SQUISH_CONSTANT2
    value = contrib(value, dx1, dy1, xsb + 1, ysb + 0)
    dx2 = dx0 - 0 - SQUISH_CONSTANT2
    dy2 = dy0 - 1 - SQUISH_CONSTANT2
    value = contrib(value, dx2, dy2, xsb + 0, ysb + 1)

    inside0 = in_sum <= 1
    # --- in_sum <= 1: triangle at (0,0)
    zins_a = 1 - in_sum
    near_a = (zins_a > xins) | (zins_a > yins)
    gt = xins > yins
    xe_a = np.where(near_a, np.where(gt, xsb + 1, xsb - 1), xsb + 1)
    ye_a = np.where(near_a, np.where(gt, ysb - 1, ysb + 1), ysb + 1)
    dxe_a = np.where(near_a, np.where(gt, dx0 - 1, dx0 + 1), dx0 - 1 - 2 * SQUISH_CONSTANT2)
    dye_a = np.where(near_a, np.where(gt, dy0 + 1, dy0 - 1), dy0 - 1 - 2 * SQUISH_CONSTANT2)
    # --- else: triangle at (1,1)
    zins_b = 2 - in_sum
    near_b = (zins_b < xins) | (zins_b < yins)
    xe_b = np.where(near_b, np.where(gt, xsb + 2, xsb + 0), xsb)
    ye_b = np.where(near_b, np.where(gt, ysb + 0, ysb + 2), ysb)
    dxe_b = np.where(near_b, np.where(gt, dx0 - 2 - 2 * SQUISH_CONSTANT2, dx0 + 0 - 2 * SQUISH_CONSTANT2), dx0)
    dye_b = np.where(near_b, np.where(gt, dy0 + 0 - 2 * SQUISH_CONSTANT2, dy0 - 2 - 2 * SQUISH_CONSTANT2), dy0)
    xsv_ext = np.where(inside0, xe_a, xe_b)
    ysv_ext = np.where(inside0, ye_a, ye_b)
    dx_ext = np.where(inside0, dxe_a, dxe_b)
    dy_ext = np.where(inside0, dye_a, dye_b)
    xsb2 = np.where(inside0, xsb, xsb + 1)
    ysb2 = np.where(inside0, ysb, ysb + 1)
    dx0b = np.where(inside0, dx0, dx0 - 1 - 2 * SQUISH_CONSTANT2)
    dy0b = np.where(inside0, dy0, dy0 - 1 - 2 * SQUISH_CONSTANT2)

    value = contrib(value, dx0b, dy0b, xsb2, ysb2)
    value = contrib(value, dx_ext, dy_ext, xsv_ext, ysv_ext)
    return value / NORM_CONSTANT2


def rand_2d_octaves(perm, H: int, W: int, octaves: int = 6, persistence: float = 0.8, frequency: float = 64) -> np.ndarray:
    """`Simplex_CLASS.rand_2d_octaves` (:97-114) over `_noise2a` (:355-361): noise[i][j] = noise2(x[j], y[i]).
    The reference's flat index `i * y.size + j` assumes a square field; so does this restatement."""
    if H != W:
        raise ValueError("the reference's _noise2a indexing (generate_noise.py:359) is only valid for square fields")
    y = np.arange(0, H)
    x = np.arange(0, W)
    noise = np.zeros((H, W))
    amplitude = 1
    for _ in range(octaves):
        xx, yy = np.meshgrid(x / frequency, y / frequency)     # xx[i][j] = x[j] / f, yy[i][j] = y[i] / f
        noise += amplitude * noise2(xx, yy, perm)
        frequency /= 2
        amplitude *= persistence
    return noise


def gen_noise(seed: int, shape) -> np.ndarray:
    """`gen_noise` (:8-15) + `generate_simplex_noise` (:19-52) for noisetype 'simplex', random_param=False:
    one field per call (seed = what `newSeed` drew), repeated over batch and channel, float64 -> float16."""
    import torch
    B, C, H, W = shape
    field = rand_2d_octaves(init_perm(seed), H, W)
    # torch's `.half()` on a float64 tensor rounds twice (float64 -> float32 -> float16); numpy's astype(float16)
    # rounds once. The reference uses torch (:12), so the restatement does too.
    f16 = torch.from_numpy(field).half().numpy()
    return np.broadcast_to(f16, (B, C, H, W)).copy()
