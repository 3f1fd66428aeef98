"""Host-side mirror of the reference's `gen_noise` (reference src/utils/generate_noise.py:8-15), on the device.

    gen_noise(cfg, shape) -> float16 tensor [B, C, H, W]

Same call, same result type, same field for every batch item; the 6-octave OpenSimplex field is computed by
csrc/simplex.hip in float64 and is bit-exact with the reference's numba/CPU code for the seed `Simplex_CLASS.newSeed`
would draw. Like the reference, the seed comes from numpy's global RNG (`np.random.randint(-1e10, 1e10)`, :62)
unless one is passed explicitly. The reference builds the field on the CPU and copies it to the GPU on every reverse
step of its simplex branch (src/models/modules/cond_DDPM.py:442); here it never leaves the GPU.
"""
from __future__ import annotations

import numpy as np
import torch

OCTAVES, PERSISTENCE, FREQUENCY = 6, 0.8, 64     # generate_simplex_noise defaults (:19-21)


def draw_seed() -> int:
    """what Simplex_CLASS.newSeed() draws (:60-63)"""
    return int(np.random.randint(-10000000000, 10000000000))


def gen_noise(cfg, shape, *, engine=None, device=None, seed=None):
    """cfg.noisetype must be 'simplex' (the only type the reference implements, :10-14). `engine`: a CddpmEngine
    (any handle on the target device works: the generator needs no model state)."""
    noisetype = cfg.get("noisetype", None) if hasattr(cfg, "get") else getattr(cfg, "noisetype", None)
    if noisetype != "simplex":
        raise ValueError("Noise type not recognized")
    B, C, H, W = (int(v) for v in shape)
    if engine is None:
        raise RuntimeError("gen_noise needs the HIP engine of the model (engine=...); there is no CPU fallback")
    if seed is None:
        # the reference constructs Simplex_CLASS() (one draw, :59) and generate_simplex_noise re-seeds (:26): two draws,
        # the second one is used
        draw_seed()
        seed = draw_seed()
    field = engine.simplex_noise(B, H, W, seed=seed, octaves=OCTAVES, persistence=PERSISTENCE, frequency=FREQUENCY)
    return field if C == 1 else field.expand(B, C, H, W).contiguous()
