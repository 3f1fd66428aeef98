// First pieces of the TRAINING step (src/models/DDPM_2D.py:114-135 -> cond_DDPM.py:565-645 with
// gradients) for gfx950. Backward of the two operator groups every ResBlock is made of (src/models/modules/OpenAI_Unet.py:284-338):
//
//   a = SiLU( GroupNorm32(x) * (1 + scale) + shift )       (in_layers.0/1: scale = shift = 0;  out_layers.0-2 with FiLM)
//   y = Conv3x3(a)                                          (in_layers.2 / out_layers.3, skip_connection: Conv1x1)
//
// * conv dgrad (dL/da from dL/dy) needs no new kernel: it IS the fused forward convolution (conv_x6.hip / conv_mfma.hip) run on
//   dL/dy with the weight tensor transposed in (Cout, Cin) and flipped in (ky, kx), packed by the same host packer
//   (cddpm_op_conv_dgrad in cddpm_api.hip) -- every Cin of the UNet is a multiple of 128, every Cout of 32.
// * GroupNorm/FiLM/SiLU backward (here): with x^ = (x - mu_g) r_g, u = x^ g' + b' (g' = gamma (1 + scale), b' = beta (1 + scale) +
//   shift), a = SiLU(u) and du = da * SiLU'(u):
//       S1[b,c] = sum_p du,  S2[b,c] = sum_p du x^          -> d b' = S1, d g' = S2
//       dgamma[c] = sum_b S2 (1 + scale),  dbeta[c] = sum_b S1 (1 + scale),  dscale[b,c] = S2 gamma + S1 beta,  dshift[b,c] = S1
//       m1[b,g] = mean_{c in g, p}(du g') = sum_c g' S1 / n,   m2[b,g] = mean(du g' x^) = sum_c g' S2 / n
//       dx = r_g (du g' - m1 - x^ m2)
//   Two passes over (x, da): gn_bwd_partial_kernel (per-channel S1, S2 over pixel splits, fp64 inside) and gn_bwd_apply_kernel
//   (elementwise dx), with gn_bwd_finalize_kernel (tiny) in between. The statistics (mu, r) are an input: the forward's.
// Not here yet: conv wgrad, attention backward, the embedding MLPs, Adam, the gradient all-reduce.
#include "kernels.h"

namespace cddpm {

__device__ __forceinline__ float sigmoid_t(float v) {
    const float t = fminf(-v * 1.44269502162933349609375f, 126.0f);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
// du = da * SiLU'(u), SiLU'(u) = s (1 + u (1 - s)), s = sigmoid(u); identity when !silu
__device__ __forceinline__ float dact(float u, float da, int silu) {
    if (!silu) return da;
    const float s = sigmoid_t(u);
    return da * (s * (1.0f + u * (1.0f - s)));
}

// planes: [4][B][C] = mean_g, rstd_g, g', b' per (sample, channel)
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ da,
                                                             const float* __restrict__ planes, int B, int C, int HW, int nsplit,
                                                             int silu, double* __restrict__ part /*[B][nsplit][C][2]*/) {
    __shared__ double red[256][9];
    const int tid = threadIdx.x, split = blockIdx.x, b = blockIdx.y;
    const int ncq = C >> 2, npl = 256 / ncq;
    const int cq = tid % ncq, pl = tid / ncq;
    const int ppb = (HW + nsplit - 1) / nsplit;
    const int p0 = split * ppb, p1 = min(HW, p0 + ppb);
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (pl < npl) {
        const size_t pc = (size_t)B * C, bc = (size_t)b * C + 4 * cq;
        const float4 mu = *reinterpret_cast<const float4*>(planes + bc), rs = *reinterpret_cast<const float4*>(planes + pc + bc);
        const float4 gp = *reinterpret_cast<const float4*>(planes + 2 * pc + bc), bp = *reinterpret_cast<const float4*>(planes + 3 * pc + bc);
        const float m[4] = {mu.x, mu.y, mu.z, mu.w}, r[4] = {rs.x, rs.y, rs.z, rs.w};
        const float g[4] = {gp.x, gp.y, gp.z, gp.w}, bt[4] = {bp.x, bp.y, bp.z, bp.w};
        for (int p = p0 + pl; p < p1; p += npl) {
            const size_t off = ((size_t)b * HW + p) * C + 4 * cq;
            const float4 xv = *reinterpret_cast<const float4*>(x + off), dv = *reinterpret_cast<const float4*>(da + off);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (xs[i] - m[i]) * r[i];
                const float du = dact(xh * g[i] + bt[i], ds[i], silu);
                s1[i] += (double)du;
                s2[i] += (double)du * (double)xh;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[tid][i] = s1[i]; red[tid][4 + i] = s2[i]; }
    __syncthreads();
    if (tid < ncq) {
        double t1[4] = {0, 0, 0, 0}, t2[4] = {0, 0, 0, 0};
        for (int l = 0; l < npl; ++l)
#pragma unroll
            for (int i = 0; i < 4; ++i) { t1[i] += red[l * ncq + tid][i]; t2[i] += red[l * ncq + tid][4 + i]; }
        double* o = part + (((size_t)b * nsplit + split) * C + 4 * tid) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] = t1[i]; o[2 * i + 1] = t2[i]; }
    }
}

// one workgroup per sample: fold the splits (fixed order), group means, FiLM / affine gradients of this sample.
// out_bc: [4][B][C] = S1, S2 (= d b', d g'), m1_g, m2_g broadcast per channel;  dfilm: [B][2C] (dscale | dshift) or nullptr
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const double* __restrict__ part, const float* __restrict__ planes,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int B, int C, int HW, int nsplit, float* __restrict__ out_bc,
                                                              float* __restrict__ dfilm) {
    __shared__ double sS1[1024], sS2[1024];
    __shared__ double gm1[32], gm2[32];
    const int tid = threadIdx.x, b = blockIdx.x;
    const size_t pc = (size_t)B * C;
    for (int c = tid; c < C; c += 256) {
        double a = 0, q = 0;
        for (int s = 0; s < nsplit; ++s) {
            const double* p = part + (((size_t)b * nsplit + s) * C + c) * 2;
            a += p[0]; q += p[1];
        }
        sS1[c] = a; sS2[c] = q;
    }
    __syncthreads();
    const int cpg = C >> 5;
    if (tid < 32) {
        double a = 0, q = 0;
        for (int i = 0; i < cpg; ++i) {
            const int c = tid * cpg + i;
            const double gp = planes[2 * pc + (size_t)b * C + c];
            a += gp * sS1[c]; q += gp * sS2[c];
        }
        const double n = (double)cpg * (double)HW;
        gm1[tid] = a / n; gm2[tid] = q / n;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const size_t bc = (size_t)b * C + c;
        out_bc[bc] = (float)sS1[c];
        out_bc[pc + bc] = (float)sS2[c];
        out_bc[2 * pc + bc] = (float)gm1[c / cpg];
        out_bc[3 * pc + bc] = (float)gm2[c / cpg];
        if (dfilm) {
            dfilm[(size_t)b * 2 * C + c] = (float)(sS2[c] * (double)gamma[c] + sS1[c] * (double)beta[c]);   // dscale
            dfilm[(size_t)b * 2 * C + C + c] = (float)sS1[c];                                                 // dshift
        }
    }
}

// dgamma[c] = sum_b S2 (1 + scale), dbeta[c] = sum_b S1 (1 + scale)     (film: [B][2C] scale | shift, or nullptr)
__global__ void gn_bwd_param_kernel(const float* __restrict__ out_bc, const float* __restrict__ film, int B, int C,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const size_t pc = (size_t)B * C;
    double g = 0, bt = 0;
    for (int b = 0; b < B; ++b) {
        const double f = film ? 1.0 + (double)film[(size_t)b * 2 * C + c] : 1.0;
        g += (double)out_bc[pc + (size_t)b * C + c] * f;
        bt += (double)out_bc[(size_t)b * C + c] * f;
    }
    dgamma[c] = (float)g;
    dbeta[c] = (float)bt;
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ da,
                                                           const float* __restrict__ planes, const float* __restrict__ out_bc,
                                                           int B, int C, int HW, int silu, float* __restrict__ dx) {
    const int ncq = C >> 2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * HW * ncq) return;
    const int cq = (int)(e % ncq);
    const long long bp = e / ncq;
    const int b = (int)(bp / HW);
    const size_t pc = (size_t)B * C, bc = (size_t)b * C + 4 * cq, off = (size_t)bp * C + 4 * cq;
    const float4 mu = *reinterpret_cast<const float4*>(planes + bc), rs = *reinterpret_cast<const float4*>(planes + pc + bc);
    const float4 gp = *reinterpret_cast<const float4*>(planes + 2 * pc + bc), bt = *reinterpret_cast<const float4*>(planes + 3 * pc + bc);
    const float4 m1 = *reinterpret_cast<const float4*>(out_bc + 2 * pc + bc), m2 = *reinterpret_cast<const float4*>(out_bc + 3 * pc + bc);
    const float4 xv = *reinterpret_cast<const float4*>(x + off), dv = *reinterpret_cast<const float4*>(da + off);
    float4 o;
#define CDDPM_GNB(f) { const float xh = (xv.f - mu.f) * rs.f; const float du = dact(xh * gp.f + bt.f, dv.f, silu); \
                       o.f = rs.f * (du * gp.f - m1.f - xh * m2.f); }
    CDDPM_GNB(x) CDDPM_GNB(y) CDDPM_GNB(z) CDDPM_GNB(w)
#undef CDDPM_GNB
    *reinterpret_cast<float4*>(dx + off) = o;
}

// forward statistics for the standalone op: planes[0] = mean_g, [1] = rstd_g, [2] = g', [3] = b' from fp32 (sum, sum of squares) records
__global__ __launch_bounds__(256) void gn_bwd_planes_kernel(const float* __restrict__ rec, int nrec, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ film, int B, int C,
                                                            int HW, float* __restrict__ planes) {
    __shared__ double sS[1024], sQ[1024];
    __shared__ float gmu[32], grs[32];
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int c = tid; c < C; c += 256) {
        double s = 0, q = 0;
        for (int r = 0; r < nrec; ++r) {
            const float* p = rec + (((size_t)b * nrec + r) * C + c) * 2;
            s += p[0]; q += p[1];
        }
        sS[c] = s; sQ[c] = q;
    }
    __syncthreads();
    const int cpg = C >> 5;
    if (tid < 32) {
        double s = 0, q = 0;
        for (int i = 0; i < cpg; ++i) { s += sS[tid * cpg + i]; q += sQ[tid * cpg + i]; }
        const double n = (double)cpg * (double)HW, mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0) var = 0;
        gmu[tid] = (float)mean;
        grs[tid] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    const size_t pc = (size_t)B * C;
    for (int c = tid; c < C; c += 256) {
        const size_t bc = (size_t)b * C + c;
        const float sc = film ? film[(size_t)b * 2 * C + c] : 0.f, sh = film ? film[(size_t)b * 2 * C + C + c] : 0.f;
        planes[bc] = gmu[c / cpg];
        planes[pc + bc] = grs[c / cpg];
        planes[2 * pc + bc] = gamma[c] * (1.0f + sc);
        planes[3 * pc + bc] = beta[c] * (1.0f + sc) + sh;
    }
}

void launch_gn_bwd_planes(const float* rec, int nrec, const float* gamma, const float* beta, const float* film, int B, int C, int HW,
                          float* planes, hipStream_t stream) {
    hipLaunchKernelGGL(gn_bwd_planes_kernel, dim3(B), dim3(256), 0, stream, rec, nrec, gamma, beta, film, B, C, HW, planes);
}

void launch_gn_silu_backward(const float* x, const float* da, const float* planes, const float* gamma, const float* beta,
                             const float* film, int silu, int B, int C, int HW, int nsplit, double* part, float* out_bc, float* dx,
                             float* dgamma, float* dbeta, float* dfilm, hipStream_t stream) {
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nsplit, B), dim3(256), 0, stream, x, da, planes, B, C, HW, nsplit, silu, part);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B), dim3(256), 0, stream, part, planes, gamma, beta, B, C, HW, nsplit, out_bc, dfilm);
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, out_bc, film, B, C, dgamma, dbeta);
    const long long total = (long long)B * HW * (C / 4);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, da, planes, out_bc, B, C, HW,
                       silu, dx);
}

}  // namespace cddpm
