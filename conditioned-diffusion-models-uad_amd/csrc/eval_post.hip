// Residual-map post-processing of the reference's _test_step (src/utils/utils_eval.py:29-33, :64-73), on the device:
//   diff   = |orig - recon|  (or its square)                                                     utils_eval.py:30-33
//   diff  *= erode(mask > 0)   per slice, 2-D cross structuring element, `iterations` times      utils_eval.py:447-460
//   diff   = median_filter(diff, (k, k, k))   scipy.ndimage default boundary mode 'reflect'      utils_eval.py:462-464
// In the reference the last two run in scipy on the CPU after a device -> host copy of the volume. Both are exact here:
// n erosions by the 4-connected cross are one erosion by the diamond |dx| + |dy| <= n (pixels outside the slice count as
// background, scipy's border_value = 0), and the median is a selection, not arithmetic.
// Volumes are [S][H][W] (slice-major, the layout the reconstruction produces); the reference indexes [H][W][S] -- the
// filters are symmetric under that permutation (the Python mirror permutes).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace cddpm {

__global__ __launch_bounds__(256) void residual_mask_kernel(const float* __restrict__ orig, const float* __restrict__ recon,
                                                            const float* __restrict__ mask, float* __restrict__ out,
                                                            int S, int H, int W, int squared, int iters) {
    const unsigned total = (unsigned)S * H * W;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const unsigned row = i / (unsigned)W;
    const int x = (int)(i - row * W);
    const unsigned s = row / (unsigned)H;
    const int y = (int)(row - s * H);
    float d = orig[i];                       // recon == nullptr: `orig` already is the residual volume
    if (recon) {
        const float d0 = d - recon[i];
        d = squared ? d0 * d0 : fabsf(d0);
    }
    if (mask) {
        const float* m = mask + (size_t)s * H * W;
        // iters == 0: the mask as it is. Otherwise the whole diamond must be foreground, and a diamond cut by the slice
        // border contains background (its four tips inside <=> all of it inside).
        bool keep = (y - iters >= 0) && (y + iters < H) && (x - iters >= 0) && (x + iters < W);
        for (int dy = -iters; keep && dy <= iters; ++dy) {
            const int r = iters - (dy < 0 ? -dy : dy);
            const float* mr = m + (y + dy) * W + x;
            for (int dx = -r; dx <= r; ++dx) keep = keep && (mr[dx] > 0.f);
        }
        d = keep ? d : 0.f * d;      // bool * float, as numpy multiplies (NaN stays NaN)
    }
    out[i] = d;
}

// scipy 'reflect' (half-sample symmetric): ... c b a | a b c ... x y z | z y x ...
__device__ __forceinline__ int reflect_index(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

// order-preserving map float -> uint32 (total order of the bit patterns; -0 < +0)
__device__ __forceinline__ uint32_t sort_key(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// One thread per voxel. Its k^3 neighbours go into a private LDS column ([k^3][256] keys: bank = thread, conflict-free);
// the element of rank k^3 / 2 is found by radix selection, most significant bit first: 32 counting passes over the column.
__global__ __launch_bounds__(256) void median3d_kernel(const float* __restrict__ in, float* __restrict__ out, int S, int H,
                                                       int W, int k) {
    extern __shared__ uint32_t col[];
    const int tid = threadIdx.x;
    const unsigned total = (unsigned)S * H * W;
    const unsigned i = blockIdx.x * 256u + tid;
    const bool live = i < total;
    const unsigned ic = live ? i : total - 1;
    const unsigned row = ic / (unsigned)W;
    const int x = (int)(ic - row * W);
    const unsigned s = row / (unsigned)H;
    const int y = (int)(row - s * H);
    const int r = k >> 1, n = k * k * k;
    int e = 0;
    for (int ds = -r; ds <= r; ++ds) {
        const int ss = reflect_index((int)s + ds, S);
        for (int dy = -r; dy <= r; ++dy) {
            const int yy = reflect_index(y + dy, H);
            const float* line = in + ((size_t)ss * H + yy) * W;
            for (int dx = -r; dx <= r; ++dx) col[(e++) * 256 + tid] = sort_key(line[reflect_index(x + dx, W)]);
        }
    }
    const int rank = n >> 1;          // scipy.ndimage.median_filter: rank_filter with rank = size // 2
    uint32_t res = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = res | (1u << bit);
        int below = 0;
        for (int j = 0; j < n; ++j) below += (col[j * 256 + tid] < cand) ? 1 : 0;
        if (below <= rank) res = cand;      // the element of that rank is >= cand
    }
    if (live) out[i] = key_value(res);
}

void launch_residual_mask(const float* orig, const float* recon, const float* mask, float* out, int S, int H, int W,
                          int squared, int iters, hipStream_t stream) {
    const unsigned total = (unsigned)S * H * W;
    hipLaunchKernelGGL(residual_mask_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, orig, recon, mask, out, S, H,
                       W, squared, iters);
}

void launch_median3d(const float* in, float* out, int S, int H, int W, int k, hipStream_t stream) {
    const unsigned total = (unsigned)S * H * W;
    const size_t lds = (size_t)k * k * k * 256 * sizeof(uint32_t);       // 125 KB at k = 5
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(median3d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(median3d_kernel, dim3((total + 255) / 256), dim3(256), lds, stream, in, out, S, H, W, k);
}

}  // namespace cddpm
