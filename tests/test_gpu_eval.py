"""GPU: residual-map post-processing (eval_post.hip through cddpm_residual_postprocess) against the scipy calls the
reference makes (oracle/eval_oracle.py; utils_eval.py:29-33, :447-464). Selections and 0/1 products: bit-exact."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_pkg

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import eval_oracle as EO  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(engine_factory):
    return engine_factory(timesteps=10, max_batch=1, max_h=32, max_w=32)


@pytest.fixture(scope="module")
def UE():
    return load_pkg("utils_eval")


def volume(seed, H, W, S):
    rng = np.random.default_rng(seed)
    orig = rng.random((H, W, S), dtype=np.float32)
    recon = np.clip(orig + 0.1 * rng.standard_normal((H, W, S)).astype(np.float32), 0, 1).astype(np.float32)
    recon[rng.random((H, W, S)) < 0.1] = orig[rng.random((H, W, S)) < 0.1].mean()      # exact ties and repeated values
    yy, xx = np.mgrid[0:H, 0:W]
    mask = np.zeros((H, W, S), np.float32)
    for s in range(S):
        r = 0.42 - 0.02 * (s % 3)
        mask[:, :, s] = 3.0 * ((((yy - H / 2) / (r * H)) ** 2 + ((xx - W / 2) / (r * W)) ** 2) < 1.0)
    mask[0, :, 0] = 1.0            # foreground touching the border erodes away
    holes = rng.random((H, W, S)) < 0.002
    mask[holes] = 0.0
    return orig, recon, mask


@pytest.mark.parametrize("H,W,S", [(50, 75, 7), (128, 128, 12), (30, 26, 2)])
def test_residual_postprocess_matches_scipy(eng, UE, H, W, S):
    orig, recon, mask = volume(H * 1000 + W, H, W, S)
    d_o, d_r, d_m = (torch.from_numpy(a).cuda() for a in (orig, recon, mask))
    diff = EO.residual(orig, recon)
    assert np.array_equal(UE.residual_volume(eng, d_o, d_r).cpu().numpy(), diff)
    masked = EO.apply_brainmask_volume(diff, mask)
    got = UE.apply_brainmask_volume(eng, torch.from_numpy(diff).cuda(), d_m).cpu().numpy()
    assert np.array_equal(got, masked)
    assert (masked != 0).any() and (masked == 0).any()
    for k in (3, 5):
        ref = EO.apply_3d_median_filter(masked, k)
        got = UE.apply_3d_median_filter(eng, torch.from_numpy(masked).cuda(), k).cpu().numpy()
        assert np.array_equal(got, ref), k
    # the fused call = the three steps of _test_step in a row
    fused = UE.postprocess_residual(eng, d_o, d_r, d_m).cpu().numpy()
    assert np.array_equal(fused, EO.apply_3d_median_filter(masked, 5))
    # signed data through the stand-alone median (selection on the total order of floats)
    sv = (orig - 0.5).astype(np.float32)
    assert np.array_equal(UE.apply_3d_median_filter(eng, torch.from_numpy(sv).cuda(), 3).cpu().numpy(),
                          EO.apply_3d_median_filter(sv, 3))


def test_reference_layout_with_singleton_axes_and_narrow_slices(eng, UE):
    orig, recon, mask = volume(5, 40, 20, 3)          # 20 columns: W // 25 == 0 -> scipy erodes until empty
    v = torch.from_numpy(EO.residual(orig, recon)).cuda().reshape(1, 40, 20, 3)
    out = UE.apply_brainmask_volume(eng, v, torch.from_numpy(mask).cuda())
    assert out.shape == v.shape and float(out.abs().max()) == 0.0
    assert np.array_equal(out.cpu().numpy().squeeze(), EO.apply_brainmask_volume(v.cpu().numpy(), mask).squeeze())


def test_single_slice_volume_through_the_engine(eng):
    """S = 1 (the reference's squeeze() would lose the axis): every window reflects onto the one slice."""
    orig, recon, mask = volume(9, 64, 96, 1)
    diff = EO.residual(orig, recon)
    shw = lambda a: torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).cuda()
    got = eng.residual_postprocess(shw(orig), shw(recon), shw(mask), erode_iterations=96 // 25, median_k=5).cpu().numpy()
    import scipy.ndimage
    er = scipy.ndimage.binary_erosion(mask[:, :, 0] > 0, structure=scipy.ndimage.generate_binary_structure(2, 1), iterations=96 // 25)
    ref = EO.apply_3d_median_filter((er * diff[:, :, 0])[:, :, None].astype(np.float32), 5)
    assert np.array_equal(got[0], ref[:, :, 0])


def test_errors_are_loud(eng):
    a = torch.zeros((2, 8, 8), device="cuda")
    with pytest.raises(RuntimeError):
        eng.residual_postprocess(a, a, None, median_k=4)
    with pytest.raises(RuntimeError):
        eng.residual_postprocess(a.cpu(), a, None)
    with pytest.raises(RuntimeError):
        eng.residual_postprocess(a, torch.zeros((2, 8, 4), device="cuda"), None)
