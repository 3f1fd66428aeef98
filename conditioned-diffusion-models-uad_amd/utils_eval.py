"""Device-side mirrors of the residual-map helpers of the reference's evaluation step (src/utils/utils_eval.py):
same names, argument meaning and array layout ([H, W, S] volumes, squeezable singleton axes), on HIP tensors.

    residual_volume          utils_eval.py:29-33   |orig - recon| (the reference's `residualmode` test is always true: L1)
    apply_brainmask_volume   utils_eval.py:454-460 per slice: volume * binary_erosion(mask, cross, iterations = W // 25)
    apply_3d_median_filter   utils_eval.py:462-464 scipy.ndimage.median_filter(volume, (k, k, k))

The metrics behind them (AUROC / AUPRC / Dice / Hausdorff, utils_eval.py:80-194) stay with sklearn / monai: out of scope
(SURVEY.md section 8, row f4). No CPU fallback: the functions need an engine (a loaded libcddpm_hip.so) and HIP tensors.
"""
from typing import Optional

import torch


def _to_shw(vol: torch.Tensor) -> torch.Tensor:
    v = vol.squeeze()
    if v.dim() != 3:
        raise RuntimeError(f"expected a volume that squeezes to [H, W, S], got shape {tuple(vol.shape)}")
    return v.permute(2, 0, 1).contiguous().float()


def _from_shw(v: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    return v.permute(1, 2, 0).contiguous().reshape(like.shape)


def residual_volume(engine, data_orig: torch.Tensor, final_volume: torch.Tensor, *, squared: bool = False) -> torch.Tensor:
    """diff_volume of _test_step: torch.abs(data_orig - final_volume) (or the square), same shape as data_orig."""
    out = engine.residual_postprocess(_to_shw(data_orig), _to_shw(final_volume), None, squared=squared)
    return _from_shw(out, data_orig)


def apply_brainmask_volume(engine, vol: torch.Tensor, mask_vol: torch.Tensor, erode: bool = True, iterations: int = 10) -> torch.Tensor:
    """The reference ignores `erode` and `iterations`: it always erodes, vol.squeeze().shape[1] // 25 times (:458).
    With fewer than 25 columns that count is 0, which scipy reads as "erode until nothing changes": an empty mask."""
    v = _to_shw(vol)
    n = v.shape[2] // 25
    if n == 0:
        return torch.zeros_like(vol) * vol
    return _from_shw(engine.residual_postprocess(v, None, _to_shw(mask_vol), erode_iterations=n), vol)


def apply_3d_median_filter(engine, volume: torch.Tensor, kernelsize: int = 5) -> torch.Tensor:
    return _from_shw(engine.residual_postprocess(_to_shw(volume), None, None, median_k=kernelsize), volume)


def postprocess_residual(engine, data_orig: torch.Tensor, final_volume: torch.Tensor, data_mask: Optional[torch.Tensor], *,
                         erodeBrainmask: bool = True, medianFiltering: bool = True, kernelsize_median: int = 5) -> torch.Tensor:
    """The three steps of _test_step (:29-33, :64-71) in one call: one pass over the volume + one median pass."""
    o = _to_shw(data_orig)
    n = o.shape[2] // 25
    use_mask = erodeBrainmask and data_mask is not None
    if use_mask and n == 0:
        return torch.zeros_like(data_orig)
    out = engine.residual_postprocess(o, _to_shw(final_volume), _to_shw(data_mask) if use_mask else None,
                                      erode_iterations=n if use_mask else 0, median_k=kernelsize_median if medianFiltering else 0)
    return _from_shw(out, data_orig)
