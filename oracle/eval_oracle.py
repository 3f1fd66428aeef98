"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md section 1): CPU restatement of the residual-map helpers of
the reference's evaluation step, src/utils/utils_eval.py:

    residual                 :29-33    torch.abs(data_orig - final_volume)
    apply_brainmask          :447-452  np.multiply(binary_erosion(mask, generate_binary_structure(2, 1), iterations), x)
    apply_brainmask_volume   :454-460  per slice along axis 2, iterations = vol.shape[1] // 25 (the arguments are ignored)
    apply_3d_median_filter   :462-464  scipy.ndimage.median_filter(volume, (k, k, k))

The reference module itself cannot be imported here (it imports monai, skimage, wandb, which are not installed), so the
functions are restated: the same scipy calls on the same [H, W, S] arrays (scipy IS the reference's implementation of the
two filters), plus brute-force numpy versions of both filters that pin what those scipy calls mean (tests/test_eval_oracle.py).
Only tests/ may import this file; the product path (eval_post.hip) never does.
"""
import numpy as np
import scipy.ndimage


def residual(data_orig: np.ndarray, final_volume: np.ndarray, squared: bool = False) -> np.ndarray:
    d = data_orig.astype(np.float32) - final_volume.astype(np.float32)
    return (d * d if squared else np.abs(d)).astype(np.float32)


def apply_brainmask(x, brainmask, erode, iterations):
    strel = scipy.ndimage.generate_binary_structure(2, 1)
    brainmask = np.expand_dims(brainmask, 2)
    if erode:
        brainmask = scipy.ndimage.binary_erosion(np.squeeze(brainmask), structure=strel, iterations=iterations)
    return np.multiply(np.squeeze(brainmask), np.squeeze(x))


def apply_brainmask_volume(vol: np.ndarray, mask_vol: np.ndarray) -> np.ndarray:
    vol = np.array(vol, dtype=np.float32, copy=True)
    v, m = vol.squeeze(), np.asarray(mask_vol).squeeze()
    out = np.empty_like(v)
    for s in range(v.shape[2]):
        out[:, :, s] = apply_brainmask(v[:, :, s], m[:, :, s] > 0, erode=True, iterations=v.shape[1] // 25)
    return out.reshape(vol.shape)


def apply_3d_median_filter(volume: np.ndarray, kernelsize: int = 5) -> np.ndarray:
    return scipy.ndimage.median_filter(volume, (kernelsize, kernelsize, kernelsize))


# ---- brute-force statements of the two filters (small volumes only)
def _reflect(i: int, n: int) -> int:
    if n == 1:
        return 0
    p = 2 * n
    i %= p
    return i if i < n else p - 1 - i


def median3d_bruteforce(vol: np.ndarray, k: int) -> np.ndarray:
    A, B, C = vol.shape
    r = k // 2
    out = np.empty_like(vol)
    for a in range(A):
        for b in range(B):
            for c in range(C):
                vals = [vol[_reflect(a + da, A), _reflect(b + db, B), _reflect(c + dc, C)]
                        for da in range(-r, r + 1) for db in range(-r, r + 1) for dc in range(-r, r + 1)]
                out[a, b, c] = sorted(vals)[len(vals) // 2]
    return out


def diamond_erosion_bruteforce(mask2d: np.ndarray, n: int) -> np.ndarray:
    H, W = mask2d.shape
    out = np.zeros((H, W), bool)
    for y in range(H):
        for x in range(W):
            ok = True
            for dy in range(-n, n + 1):
                for dx in range(-(n - abs(dy)), n - abs(dy) + 1):
                    yy, xx = y + dy, x + dx
                    if not (0 <= yy < H and 0 <= xx < W and mask2d[yy, xx]):
                        ok = False
            out[y, x] = ok
    return out
