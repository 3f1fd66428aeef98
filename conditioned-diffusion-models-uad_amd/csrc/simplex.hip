// 2-D OpenSimplex fractal noise on the device, bit-exact with the reference's CPU generator.
//
// Replaces gen_noise / generate_simplex_noise / Simplex_CLASS.rand_2d_octaves / _noise2 / _init
// (src/utils/generate_noise.py:8-52, :97-114, :214-232, :252-361): 6 octaves, persistence 0.8, start frequency 64,
// one field per call repeated over the batch, float64 arithmetic, result converted float64 -> float32 -> float16
// exactly as torch's `.half()` does (:12). The reference calls this numba CPU code and copies host -> device on
// EVERY reverse step of its simplex branch (src/models/modules/cond_DDPM.py:442).
// Every floating-point operation is an individually rounded IEEE double operation in the reference's order:
// contraction into FMAs is switched off for this file.
#include "kernels.h"

#pragma clang fp contract(off)

namespace cddpm {

struct SimplexPerm { unsigned char p[256]; };

__device__ __forceinline__ double simplex_extrapolate2(const SimplexPerm& pm, long long xsb, long long ysb, double dx, double dy) {
    // GRADIENTS2 (generate_noise.py:143-150), index = perm[(perm[xsb & 0xFF] + ysb) & 0xFF] & 0x0E
    const int g2[16] = {5, 2, 2, 5, -5, 2, -2, 5, 5, -2, 2, -5, -5, -2, -2, -5};
    const int index = pm.p[(pm.p[xsb & 0xFF] + ysb) & 0xFF] & 0x0E;
    return (double)g2[index] * dx + (double)g2[index + 1] * dy;
}

__device__ double simplex_noise2(double x, double y, const SimplexPerm& pm) {
    const double STRETCH = -0.211324865405187, SQUISH = 0.366025403784439;
    const double stretch_offset = (x + y) * STRETCH;
    const double xs = x + stretch_offset, ys = y + stretch_offset;
    long long xsb = (long long)floor(xs), ysb = (long long)floor(ys);
    const double squish_offset = (double)(xsb + ysb) * SQUISH;
    const double xb = (double)xsb + squish_offset, yb = (double)ysb + squish_offset;
    const double xins = xs - (double)xsb, yins = ys - (double)ysb;
    const double in_sum = xins + yins;
    double dx0 = x - xb, dy0 = y - yb;
    double value = 0.0;

    const double dx1 = dx0 - 1 - SQUISH, dy1 = dy0 - 0 - SQUISH;
    double attn1 = 2 - dx1 * dx1 - dy1 * dy1;
    if (attn1 > 0) { attn1 *= attn1; value += attn1 * attn1 * simplex_extrapolate2(pm, xsb + 1, ysb + 0, dx1, dy1); }
    const double dx2 = dx0 - 0 - SQUISH, dy2 = dy0 - 1 - SQUISH;
    double attn2 = 2 - dx2 * dx2 - dy2 * dy2;
    if (attn2 > 0) { attn2 *= attn2; value += attn2 * attn2 * simplex_extrapolate2(pm, xsb + 0, ysb + 1, dx2, dy2); }

    long long xsv_ext, ysv_ext;
    double dx_ext, dy_ext;
    if (in_sum <= 1) {
        const double zins = 1 - in_sum;
        if (zins > xins || zins > yins) {
            if (xins > yins) { xsv_ext = xsb + 1; ysv_ext = ysb - 1; dx_ext = dx0 - 1; dy_ext = dy0 + 1; }
            else             { xsv_ext = xsb - 1; ysv_ext = ysb + 1; dx_ext = dx0 + 1; dy_ext = dy0 - 1; }
        } else {
            xsv_ext = xsb + 1; ysv_ext = ysb + 1;
            dx_ext = dx0 - 1 - 2 * SQUISH; dy_ext = dy0 - 1 - 2 * SQUISH;
        }
    } else {
        const double zins = 2 - in_sum;
        if (zins < xins || zins < yins) {
            if (xins > yins) { xsv_ext = xsb + 2; ysv_ext = ysb + 0; dx_ext = dx0 - 2 - 2 * SQUISH; dy_ext = dy0 + 0 - 2 * SQUISH; }
            else             { xsv_ext = xsb + 0; ysv_ext = ysb + 2; dx_ext = dx0 + 0 - 2 * SQUISH; dy_ext = dy0 - 2 - 2 * SQUISH; }
        } else {
            dx_ext = dx0; dy_ext = dy0; xsv_ext = xsb; ysv_ext = ysb;
        }
        xsb += 1; ysb += 1;
        dx0 = dx0 - 1 - 2 * SQUISH; dy0 = dy0 - 1 - 2 * SQUISH;
    }
    double attn0 = 2 - dx0 * dx0 - dy0 * dy0;
    if (attn0 > 0) { attn0 *= attn0; value += attn0 * attn0 * simplex_extrapolate2(pm, xsb, ysb, dx0, dy0); }
    double attn_ext = 2 - dx_ext * dx_ext - dy_ext * dy_ext;
    if (attn_ext > 0) { attn_ext *= attn_ext; value += attn_ext * attn_ext * simplex_extrapolate2(pm, xsv_ext, ysv_ext, dx_ext, dy_ext); }
    return value / 47;
}

__global__ __launch_bounds__(256) void simplex_kernel(unsigned short* __restrict__ out, const SimplexPerm pm, int B, int H,
                                                      int W, int octaves, double persistence, double frequency) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * W) return;
    const int i = idx / W, j = idx - i * W;      // noise[i][j] = noise2(x[j] / f, y[i] / f)   (_noise2a, :355-361)
    double noise = 0.0, amplitude = 1.0, f = frequency;
    for (int o = 0; o < octaves; ++o) {
        noise += amplitude * simplex_noise2((double)j / f, (double)i / f, pm);
        f /= 2;
        amplitude *= persistence;
    }
    const _Float16 h = (_Float16)(float)noise;   // float64 -> float32 -> float16, both round-to-nearest-even (torch .half())
    unsigned short bits;
    __builtin_memcpy(&bits, &h, 2);
    for (int b = 0; b < B; ++b) out[(size_t)b * H * W + idx] = bits;
}

void launch_simplex(unsigned short* out, long long seed, int B, int H, int W, int octaves, double persistence,
                    double frequency, hipStream_t stream) {
    // `_init` (:214-232): LCG over int64 with wrap-around, Python's floor modulo
    SimplexPerm pm;
    unsigned char source[256];
    for (int i = 0; i < 256; ++i) source[i] = (unsigned char)i;
    unsigned long long s = (unsigned long long)seed;
    for (int k = 0; k < 3; ++k) s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    for (int i = 255; i >= 0; --i) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        const __int128 v = (__int128)(long long)s + 31;
        long long r = (long long)(v % (i + 1));
        if (r < 0) r += i + 1;
        pm.p[i] = source[r];
        source[r] = source[i];
    }
    const int n = H * W;
    hipLaunchKernelGGL(simplex_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, out, pm, B, H, W, octaves, persistence,
                       frequency);
}

}  // namespace cddpm
