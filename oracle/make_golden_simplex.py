"""ORACLE tooling -- build-container only: run the REFERENCE's simplex generator (pure Python: numba is absent and
`@njit` is stubbed to the identity by oracle/ref_harness.py) and store its outputs under tests/golden/.

    python oracle/make_golden_simplex.py        # ~1 min
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_harness as R  # noqa: E402
import simplex_oracle as S  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

if __name__ == "__main__":
    R.import_reference()
    from src.utils import generate_noise as G  # type: ignore

    out = {}
    worst = 0.0
    for seed, n in ((3, 32), (-9876543210, 32), (1234567, 96), (9999999999, 128), (0, 16)):
        sx = G.Simplex_CLASS()
        sx.newSeed(seed if seed else None) if seed else None
        if seed == 0:      # `if not seed` in newSeed draws a random one: pin numpy's RNG instead and record what it drew
            np.random.seed(17)
            drawn = int(np.random.randint(-10000000000, 10000000000))
            np.random.seed(17)
            sx.newSeed()
            seed_used = drawn
        else:
            seed_used = seed
        perm_ref = np.array(sx._perm)
        field = sx.rand_2d_octaves((n, n), 6, 0.8, 64)
        half = torch.from_numpy(field).half().numpy()
        key = f"seed{seed_used}_n{n}"
        out[key + "_perm"] = perm_ref.astype(np.int16)
        out[key + "_f16"] = half.view(np.uint16)
        out[key + "_f64"] = field
        mine = S.rand_2d_octaves(S.init_perm(seed_used), n, n)
        worst = max(worst, float(np.abs(mine - field).max()))
        assert np.array_equal(S.init_perm(seed_used), perm_ref), key
        assert np.array_equal(S.gen_noise(seed_used, (1, 1, n, n))[0, 0].view(np.uint16), half.view(np.uint16)), key
        print(key, "oracle == reference (perm, f64 maxdiff", float(np.abs(mine - field).max()), ", f16 bits)")
    # gen_noise end to end (shape handling, batch repeat), numpy RNG pinned
    np.random.seed(5)

    class Cfg:
        noisetype = "simplex"
    ns = G.gen_noise(Cfg(), (3, 1, 32, 32))
    np.random.seed(5)
    drawn = [int(np.random.randint(-10000000000, 10000000000)) for _ in range(2)]   # __init__ + newSeed in generate_simplex_noise
    out["gen_noise_seed"] = np.int64(drawn[1])
    out["gen_noise_f16"] = ns.numpy().view(np.uint16)
    assert ns.dtype == torch.float16 and tuple(ns.shape) == (3, 1, 32, 32)
    assert np.array_equal(S.gen_noise(drawn[1], (3, 1, 32, 32)).view(np.uint16), ns.numpy().view(np.uint16))
    out["oracle_vs_reference_f64_maxabs"] = np.float64(worst)
    np.savez_compressed(os.path.join(GOLD, "simplex.npz"), **out)
    print("saved; worst f64 diff", worst)
