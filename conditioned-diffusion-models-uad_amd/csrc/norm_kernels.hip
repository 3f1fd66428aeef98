// GroupNorm32 statistics and FiLM coefficients for gfx950.
//
// Replaces GroupNorm32(32, C) (src/models/LDM/modules/diffusionmodules/util.py:199-216) and the
// scale-shift modulation `out_norm(h) * (1 + scale) + shift` (src/models/modules/OpenAI_Unet.py:325-330).
// Nothing here writes a normalised tensor: the normalisation is folded into three per-(sample, channel)
// numbers (mean, a, d) that conv_mfma.hip applies while staging its input, v -> (v - mean) * a + d, so a
// GroupNorm costs one statistics read of its input instead of a read + write + re-read.
//
//   statistics : per-channel (sum, sum of squares) RECORDS over pixel subsets, [B][records][C][2] fp32. Normally the
//                producing convolution's epilogue writes them (conv_mfma.hip, one record per wave tile), so the
//                tensor is not re-read at all; gn_partial is the stand-alone sweep for tensors no fused conv
//                produced (input_blocks.0). No atomics: bitwise reproducible, independent of the batch.
//   gn_finalize: four workgroups per sample fold the records (fp64) and the channels of each of the 32 groups (biased variance, eps 1e-5),
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

__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ src, int C, int HW, int nsplit,
                                                         float* __restrict__ rec) {
    __shared__ double red[256][9];   // 9: odd stride, conflict-free column walks
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
        float* o = rec + (((size_t)b * nsplit + split) * C + 4 * tid) * 2;
        *reinterpret_cast<float4*>(o) = make_float4((float)ts[0], (float)tq[0], (float)ts[1], (float)tq[1]);
        *reinterpret_cast<float4*>(o + 4) = make_float4((float)ts[2], (float)tq[2], (float)ts[3], (float)tq[3]);
    }
}

void launch_gn_partial(const float* src, int C, int B, int HW, int nsplit, float* rec, hipStream_t stream) {
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nsplit, B), dim3(256), 0, stream, src, C, HW, nsplit, rec);
}

// Four workgroups per sample, eight groups each: 256 threads = (record lane, channel pair of the workgroup's 8 C / 32 channels).
// A thread folds the records of its channel pair, rec[r][c .. c + 1] for r = lane, lane + nrl, ... in fp64 (16-B loads; a record
// row of the workgroup is one contiguous run of 64 C / 32 bytes), the lanes are combined in lane order, then the channels of each
// group (biased variance, eps 1e-5), and the three coefficients of the workgroup's channels are written. The fold is latency-bound
// (a few hundred records per channel): four times the workgroups of a one-per-sample kernel cut the launch from 7-9 us to ~3 (56
// launches per UNet forward) while every load stays a full 16-B lane access. The summation order depends on the record count only
// -- never on the batch. The channels of one group may come from both concatenated sources (C0 is not always a multiple of C / 32;
// a channel PAIR never straddles them: C0 is a multiple of 32).
constexpr int GN_WG_PER_SAMPLE = 4;
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ rec0, int C0, int n0,
                                                          const float* __restrict__ rec1, int C1, int n1, int B, int HW,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ tab, const float* __restrict__ cpart,
                                                          int sumE, int eoff, const int* __restrict__ t_dev,
                                                          const float* __restrict__ film_direct, float* __restrict__ coef) {
    __shared__ double red[256][5];            // per thread (s0, q0, s1, q1); 5: odd stride
    __shared__ double chS[384], chQ[384];     // per channel of this workgroup (8 groups x <= 48 channels)
    __shared__ float gm[8], gr[8];
    const int tid = threadIdx.x, w = blockIdx.x, b = blockIdx.y;
    const int C = C0 + C1;
    const int cpg = C >> 5;                   // channels per group: 4 .. 48
    const int nch = 8 * cpg;                  // channels of this workgroup
    const int cbase = w * nch;
    const int P = nch >> 1;                   // channel pairs: 16 .. 192
    const int nrl = 256 / P;                  // record lanes (>= 1)
    const int jp = tid % P, rl = tid / P;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (rl < nrl) {
        const int c = cbase + 2 * jp;
        const bool second = c >= C0;
        const float* rec = second ? rec1 : rec0;
        const int Cs = second ? C1 : C0, ns = second ? n1 : n0, cl = second ? c - C0 : c;
        const float* base = rec + ((size_t)b * ns) * Cs * 2 + 2 * cl;
        int r = rl;
        for (; r + 7 * nrl < ns; r += 8 * nrl) {             // eight records in flight per thread; same order as the tail loop
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(base + (size_t)(r + k * nrl) * Cs * 2);
#pragma unroll
            for (int k = 0; k < 8; ++k) { a0 += v[k].x; a1 += v[k].y; a2 += v[k].z; a3 += v[k].w; }
        }
        for (; r < ns; r += nrl) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)r * Cs * 2);
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
    }
    red[tid][0] = a0; red[tid][1] = a1; red[tid][2] = a2; red[tid][3] = a3;
    __syncthreads();
    if (tid < P) {
        double s0 = 0, q0 = 0, s1 = 0, q1 = 0;
        for (int l = 0; l < nrl; ++l) {
            const double* p = red[l * P + tid];
            s0 += p[0]; q0 += p[1]; s1 += p[2]; q1 += p[3];
        }
        chS[2 * tid] = s0; chQ[2 * tid] = q0;
        chS[2 * tid + 1] = s1; chQ[2 * tid + 1] = q1;
    }
    __syncthreads();
    if (tid < 8) {
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
    for (int i = tid; i < nch; i += 256) {
        const int cc = cbase + i, g = i / cpg;
        float av = gr[g] * gamma[cc];
        float dv = beta[cc];
        float sc = 0.f, sh = 0.f;
        bool film = false;
        if (tab) {
            const int t = t_dev[b];
            const float* tr = tab + (size_t)t * sumE + eoff;
            const float* cr = cpart + (size_t)b * sumE + eoff;
            sc = tr[cc] + cr[cc];
            sh = tr[C + cc] + cr[C + cc];
            film = true;
        } else if (film_direct) {
            sc = film_direct[(size_t)b * 2 * C + cc];
            sh = film_direct[(size_t)b * 2 * C + C + cc];
            film = true;
        }
        if (film) {
            const float f = 1.0f + sc;
            av *= f;
            dv = dv * f + sh;
        }
        coef[(size_t)b * C + cc] = gm[g];
        coef[plane + (size_t)b * C + cc] = av;
        coef[2 * plane + (size_t)b * C + cc] = dv;
    }
}

void launch_gn_finalize(const float* rec0, int C0, int n0, const float* rec1, int C1, int n1, int B, int HW,
                        const float* gamma, const float* beta, const float* tab, const float* cpart, int sumE, int eoff,
                        const int* t_dev, const float* film_direct, float* coef, hipStream_t stream) {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(GN_WG_PER_SAMPLE, B), dim3(256), 0, stream, rec0, C0, n0, rec1, C1, n1, B, HW,
                       gamma, beta, tab, cpart, sumE, eoff, t_dev, film_direct, coef);
}

}  // namespace cddpm
