"""CPU: what the scipy calls behind the reference's residual post-processing mean (utils_eval.py:447-464), pinned by
brute-force restatements, so that the HIP kernels can be checked against either."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import eval_oracle as EO  # noqa: E402


def test_median_filter_is_rank_half_with_reflect_boundary():
    rng = np.random.default_rng(0)
    for shape, k in (((6, 7, 5), 5), ((4, 3, 9), 3), ((2, 5, 1), 5), ((3, 3, 3), 5)):
        v = rng.random(shape, dtype=np.float32)
        assert np.array_equal(EO.apply_3d_median_filter(v, k), EO.median3d_bruteforce(v, k)), (shape, k)


def test_iterated_cross_erosion_is_a_diamond_with_background_outside():
    rng = np.random.default_rng(1)
    import scipy.ndimage
    strel = scipy.ndimage.generate_binary_structure(2, 1)
    for n in (1, 2, 5):
        m = rng.random((40, 52)) > 0.03
        m[8:30, 10:45] = True
        got = scipy.ndimage.binary_erosion(m, structure=strel, iterations=n)
        assert np.array_equal(got, EO.diamond_erosion_bruteforce(m, n)), n
    # iterations = 0 (fewer than 25 columns in the reference's formula): scipy erodes until nothing changes
    assert not scipy.ndimage.binary_erosion(np.ones((12, 12), bool), structure=strel, iterations=0).any()


def test_brainmask_volume_layout():
    rng = np.random.default_rng(2)
    vol = rng.random((1, 50, 75, 3), dtype=np.float32)
    mask = np.zeros((50, 75, 3), np.float32)
    mask[5:45, 6:70, :] = 2.0
    out = EO.apply_brainmask_volume(vol, mask)
    assert out.shape == vol.shape
    n = 75 // 25
    inner = np.zeros((50, 75), bool)
    inner[5 + n:45 - n, 6 + n:70 - n] = True
    # a rectangle eroded by a diamond keeps its interior rectangle
    assert np.array_equal(out[0, :, :, 1] != 0, inner & (vol[0, :, :, 1] != 0))
