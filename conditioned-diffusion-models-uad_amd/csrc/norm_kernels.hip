// GroupNorm32 statistics and FiLM coefficients for gfx950.
//
// Replaces GroupNorm32(32, C) (src/models/LDM/modules/diffusionmodules/util.py:199-216) and the
// scale-shift modulation `out_norm(h) * (1 + scale) + shift` (src/models/modules/OpenAI_Unet.py:325-330).
// Nothing here writes a normalised tensor: the normalisation is folded into three per-(sample, channel)
// numbers (mean, a, d) that conv_mfma.hip applies while staging its input, v -> (v - mean) * a + d, so a
// GroupNorm costs one statistics read of its input instead of a read + write + re-read.
//
//   gn_partial : coalesced NHWC sweep, per-channel sum / sum of squares in fp64, one record per
//                (sample, pixel-range split) -- no atomics, bitwise reproducible.
//   gn_finalize: folds the splits and the channels of each of the 32 groups (biased variance, eps 1e-5),
//                then a = rstd * gamma * (1 + scale), d = beta * (1 + scale) + shift, where
//                (scale | shift) = emb_layers(emb) = table[t_b] + cond_part[b]  (see cddpm_api.hip).
#include "kernels.h"

namespace cddpm {

int gn_nsplit(int B, int HW) {
    // The pixel-range split depends on the image size ONLY (never on the batch): a slice's statistics are
    // then summed in the same order whatever batch or rank it sits in -> sharded runs are bitwise identical.
    (void)B;
    const int ppb = (HW >= 4096) ? 256 : 64;
    return (HW + ppb - 1) / ppb;
}

__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ src, int C, int Ctot, int coff,
                                                         int HW, int nsplit, double* __restrict__ part) {
    __shared__ double red[256][8];
    const int tid = threadIdx.x;
    const int split = blockIdx.x, b = blockIdx.y;
    const int ncq = C >> 2;             // channel quads (<= 256)
    const int npl = 256 / ncq;          // pixel lanes
    const int cq = tid % ncq, pl = tid / ncq;
    const int ppb = (HW + nsplit - 1) / nsplit;
    const int p0 = split * ppb;
    const int p1 = min(HW, p0 + ppb);
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (pl < npl) {
        const float* base = src + ((size_t)b * HW) * C + 4 * cq;
        for (int p = p0 + pl; p < p1; p += npl) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)p * C);
            const double x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
            s[0] += x0; q[0] += x0 * x0;
            s[1] += x1; q[1] += x1 * x1;
            s[2] += x2; q[2] += x2 * x2;
            s[3] += x3; q[3] += x3 * x3;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[tid][i] = s[i]; red[tid][4 + i] = q[i]; }
    __syncthreads();
    if (tid < ncq) {
        double ts[4] = {0, 0, 0, 0}, tq[4] = {0, 0, 0, 0};
        for (int l = 0; l < npl; ++l) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { ts[i] += red[l * ncq + tid][i]; tq[i] += red[l * ncq + tid][4 + i]; }
        }
        double* o = part + (((size_t)b * nsplit + split) * Ctot + coff + 4 * tid) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] = ts[i]; o[2 * i + 1] = tq[i]; }
    }
}

void launch_gn_partial(const float* src, int C, int Ctot, int coff, int B, int HW, int nsplit, double* part,
                       hipStream_t stream) {
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nsplit, B), dim3(256), 0, stream, src, C, Ctot, coff, HW, nsplit, part);
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, int nsplit, int C, int B,
                                                          int HW, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ tab,
                                                          const float* __restrict__ cpart, int sumE, int eoff,
                                                          const int* __restrict__ t_dev,
                                                          const float* __restrict__ film_direct,
                                                          float* __restrict__ coef) {
    __shared__ double chS[1024], chQ[1024];
    __shared__ float gm[32], gr[32];
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int c = tid; c < C; c += 256) {
        double s = 0, q = 0;
        for (int sp = 0; sp < nsplit; ++sp) {
            const double* p = part + (((size_t)b * nsplit + sp) * C + c) * 2;
            s += p[0];
            q += p[1];
        }
        chS[c] = s;
        chQ[c] = q;
    }
    __syncthreads();
    const int cpg = C >> 5;
    if (tid < 32) {
        double s = 0, q = 0;
        for (int i = 0; i < cpg; ++i) { s += chS[tid * cpg + i]; q += chQ[tid * cpg + i]; }
        const double n = (double)cpg * (double)HW;
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0) var = 0;
        gm[tid] = (float)mean;
        gr[tid] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    const size_t plane = (size_t)B * C;
    for (int c = tid; c < C; c += 256) {
        const int g = c / cpg;
        float av = gr[g] * gamma[c];
        float dv = beta[c];
        float sc = 0.f, sh = 0.f;
        bool film = false;
        if (tab) {
            const int t = t_dev[b];
            const float* tr = tab + (size_t)t * sumE + eoff;
            const float* cr = cpart + (size_t)b * sumE + eoff;
            sc = tr[c] + cr[c];
            sh = tr[C + c] + cr[C + c];
            film = true;
        } else if (film_direct) {
            sc = film_direct[(size_t)b * 2 * C + c];
            sh = film_direct[(size_t)b * 2 * C + C + c];
            film = true;
        }
        if (film) {
            const float f = 1.0f + sc;
            av *= f;
            dv = dv * f + sh;
        }
        coef[(size_t)b * C + c] = gm[g];
        coef[plane + (size_t)b * C + c] = av;
        coef[2 * plane + (size_t)b * C + c] = dv;
    }
}

void launch_gn_finalize(const double* part, int nsplit, int C, int B, int HW, const float* gamma, const float* beta,
                        const float* tab, const float* cpart, int sumE, int eoff, const int* t_dev,
                        const float* film_direct, float* coef, hipStream_t stream) {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, stream, part, nsplit, C, B, HW, gamma, beta, tab,
                       cpart, sumE, eoff, t_dev, film_direct, coef);
}

}  // namespace cddpm
