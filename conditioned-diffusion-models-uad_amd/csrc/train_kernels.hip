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
// * conv wgrad (below): a pixel-contraction GEMM on the fp32 MFMA with the activation recomputed while staging.
// Further down: the weight-gradient families, the batched fp32 GEMM and the linear backward, resampling backward, the one-channel
// convolutions' gradients, the loss, Adam. (Attention backward: attention.hip; the encoder: encoder_train.hip; sequencing: training.py.)
#include <atomic>
#include "kernels.h"
#include <cstdlib>
#include <cstring>

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
// x: the GroupNorm input, possibly the channel concatenation of two tensors (x [.., C0] | x1 [.., C - C0]; C0 = C when there is one)
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ x1, int C0,
                                                             const float* __restrict__ da,
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
        const bool first = 4 * cq < C0;
        const float* xsrc = first ? x + 4 * cq : x1 + (4 * cq - C0);
        const int Cs = first ? C0 : C - C0;
        for (int p = p0 + pl; p < p1; p += npl) {
            const size_t off = ((size_t)b * HW + p) * C + 4 * cq;
            const float4 xv = *reinterpret_cast<const float4*>(xsrc + ((size_t)b * HW + p) * Cs), dv = *reinterpret_cast<const float4*>(da + off);
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

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ x1, int C0,
                                                           float* __restrict__ dx1, const float* __restrict__ da,
                                                           const float* __restrict__ planes, const float* __restrict__ out_bc,
                                                           int B, int C, int HW, int silu, const float* __restrict__ add,
                                                           float* __restrict__ dx) {
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
    const bool first = 4 * cq < C0;
    const int Cs = first ? C0 : C - C0;
    const size_t soff = (size_t)bp * Cs + (first ? 4 * cq : 4 * cq - C0);      // offset inside the source (and destination) tensor
    const float4 xv = *reinterpret_cast<const float4*>((first ? x : x1) + soff), dv = *reinterpret_cast<const float4*>(da + off);
    float4 o;
#define CDDPM_GNB(f) { const float xh = (xv.f - mu.f) * rs.f; const float du = dact(xh * gp.f + bt.f, dv.f, silu); \
                       o.f = rs.f * (du * gp.f - m1.f - xh * m2.f); }
    CDDPM_GNB(x) CDDPM_GNB(y) CDDPM_GNB(z) CDDPM_GNB(w)
#undef CDDPM_GNB
    if (add) {      // the gradient arriving through the block's skip path (identity or 1x1 skip_connection): saves a separate pass
        const float4 av = *reinterpret_cast<const float4*>(add + off);
        o.x += av.x; o.y += av.y; o.z += av.z; o.w += av.w;
    }
    *reinterpret_cast<float4*>((first ? dx : dx1) + soff) = o;
}

// forward statistics for the standalone op: planes[0] = mean_g, [1] = rstd_g, [2] = g', [3] = b' from fp32 (sum, sum of squares) records.
// Four workgroups per sample (8 of the 32 groups each); a workgroup's 256 threads = (channel, record slice): the records of a 128 x 128
// convolution output are 64 per sample, which one thread per channel walked serially (46 us per launch when the records came from the
// convolution epilogues instead of the 2 ... 8 of a sweep).
__global__ __launch_bounds__(256) void gn_bwd_planes_kernel(const float* __restrict__ rec, int nrec, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ film, int B, int C,
                                                            int HW, float* __restrict__ planes) {
    __shared__ double sS[256], sQ[256];
    __shared__ float gmu[8], grs[8];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int Cq = C >> 2, c0 = blockIdx.y * Cq, cpg = C >> 5;          // this workgroup's channels: 8 groups
    const int nsl = 256 / Cq > 0 ? 256 / Cq : 1;                         // record slices (Cq <= 256)
    const int cl = tid % Cq, sl = tid / Cq;
    double s = 0, q = 0;
    if (sl < nsl)
        for (int r = sl; r < nrec; r += nsl) {
            const float* p = rec + (((size_t)b * nrec + r) * C + c0 + cl) * 2;
            s += p[0]; q += p[1];
        }
    // fold the slices in order through LDS, one slice per round
    for (int k = 0; k < nsl; ++k) {
        if (sl == k) { sS[cl] = (k ? sS[cl] : 0.0) + s; sQ[cl] = (k ? sQ[cl] : 0.0) + q; }
        __syncthreads();
    }
    if (tid < 8) {
        double gs = 0, gq = 0;
        for (int i = 0; i < cpg; ++i) { gs += sS[tid * cpg + i]; gq += sQ[tid * cpg + i]; }
        const double n = (double)cpg * (double)HW, mean = gs / n;
        double var = gq / n - mean * mean;
        if (var < 0) var = 0;
        gmu[tid] = (float)mean;
        grs[tid] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    const size_t pc = (size_t)B * C;
    for (int i = tid; i < Cq; i += 256) {
        const int c = c0 + i;
        const size_t bc = (size_t)b * C + c;
        const float sc = film ? film[(size_t)b * 2 * C + c] : 0.f, sh = film ? film[(size_t)b * 2 * C + C + c] : 0.f;
        planes[bc] = gmu[i / cpg];
        planes[pc + bc] = grs[i / cpg];
        planes[2 * pc + bc] = gamma[c] * (1.0f + sc);
        planes[3 * pc + bc] = beta[c] * (1.0f + sc) + sh;
    }
}

void launch_gn_bwd_planes(const float* rec, int nrec, const float* gamma, const float* beta, const float* film, int B, int C, int HW,
                          float* planes, hipStream_t stream) {
    hipLaunchKernelGGL(gn_bwd_planes_kernel, dim3(B, 4), dim3(256), 0, stream, rec, nrec, gamma, beta, film, B, C, HW, planes);
}

void launch_gn_silu_backward(const float* x, const float* x1, int C0, float* dx1, const float* da, const float* planes, const float* gamma, const float* beta,
                             const float* film, int silu, int B, int C, int HW, int nsplit, double* part, float* out_bc, float* dx,
                             float* dgamma, float* dbeta, float* dfilm, const float* add, hipStream_t stream) {
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nsplit, B), dim3(256), 0, stream, x, x1, C0, da, planes, B, C, HW, nsplit, silu, part);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B), dim3(256), 0, stream, part, planes, gamma, beta, B, C, HW, nsplit, out_bc, dfilm);
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, out_bc, film, B, C, dgamma, dbeta);
    const long long total = (long long)B * HW * (C / 4);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, x1, C0, dx1, da, planes, out_bc, B, C, HW,
                       silu, add, dx);
}

// ------------------------------------------------------------------------------------------------------------------
// conv wgrad: dW[co][ci][ky][kx] = sum_{b,y,x} dy[b,y,x,co] * a[b, y + ky - 1, x + kx - 1, ci],  a = act(GroupNorm(x)) recomputed
// from the conv's input x and the forward's coefficient planes while staging (the activated tensor was never stored).
//
// GEMM view: M = Cout, N = 32 input channels x 9 taps, K = pixels. The contraction runs over PIXELS, and both operands are NHWC
// ([pixel][channel]); the fp32 MFMA v_mfma_f32_32x32x2_f32 takes ONE float per lane for each operand -- lanes 0..31 the 32 rows /
// columns at k = 0, lanes 32..63 at k = 1 -- so a fragment is one ds_read_b32 of 32 consecutive channels of one pixel (conflict
// free), a tap is a row offset in the patch, and no transposed image of either tensor is needed; products are exact fp32 (the
// gradients then carry fp32 accuracy like autograd's; config 5's bf16 autocast can later use the 16-bit pipe the way the forward
// split kernels do).
// Workgroup = 4 waves = (32-channel block of the 64 output channels) x (taps 0..4 | 5..8); it walks a contiguous range of
// 4 x 32-pixel tiles: per tile the dy tile [128 px][64 co] and the activated patch [6 x 34 px][32 ci] go to LDS (59 KB: two
// workgroups per CU), 64 k-steps of 2 pixels, chains folded per tile into a running total (the three-level accumulation of the
// forward kernels). P workgroups per (co block, ci chunk) write P partial tiles; wgrad_reduce_kernel adds them in fixed order.
// ------------------------------------------------------------------------------------------------------------------
typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
    const float* x0; const float* x1;   // conv input before the activation, NHWC [B,H,W,C0] (+ [B,H,W,C1]: channels C0.. of the concatenation)
    int C0, C1;
    const float* coef;     // [3][B][Cin] (mean, a, d) over the concatenated channels: act input = (x - mean) * a + d, or nullptr
    int silu;
    int up;                // 1: the conv input is the nearest x2 upsampling of act(x): x is [B,H/2,W/2,C], read at (y >> 1, x >> 1)
    const float* dy;       // NHWC [B,H,W,Cout]
    int B, H, W, Cout;
    float* part;           // [P][Cout/64][Cin/CK][64 co][TAPS][CK ci]
    int P;
};

__device__ __forceinline__ float silu_t(float v) { return v * sigmoid_t(v); }

// TAPS = 9: 3x3, padding 1, 32 input channels per workgroup, wave = (co 32-block, taps 0..4 | 5..8)
// TAPS = 1: 1x1 (skip_connection, qkv, proj_out), 64 input channels per workgroup, wave = (co 32-block, ci 32-block)
template <int TAPS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int TR = 4;                                   // tile rows
    constexpr int PAD = (TAPS == 9) ? 1 : 0;
    constexpr int PW = 32 + 2 * PAD, PH = TR + 2 * PAD, NPIX = PW * PH;
    constexpr int CK = (TAPS == 9) ? 32 : 64;               // input channels per workgroup
    constexpr int NT = (TAPS == 9) ? 5 : 1;                 // accumulators per wave
    __shared__ float dys[TR * 32 * 64];                     // [128 px][64 co]
    __shared__ float xs[NPIX * CK];                         // [204 px][32 ci] | [128 px][64 ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, tg = wave >> 1;                // co 32-block; tap group | ci 32-block (wave-uniform)
    const int Cin = a.C0 + a.C1;
    const int nchunk = Cin / CK, ncb = a.Cout >> 6;
    int bid = blockIdx.x;
    const int chunk = bid % nchunk; bid /= nchunk;
    const int cb = bid % ncb;
    const int p = bid / ncb;
    const int tilesX = (a.W + 31) >> 5, tilesY = a.H / TR;
    const int ntile = a.B * tilesY * tilesX;
    const int t0 = (int)((long long)ntile * p / a.P), t1 = (int)((long long)ntile * (p + 1) / a.P);
    const int tapbase = (TAPS == 9) ? tg * 5 : 0, ntap = (TAPS == 9) ? (tg ? 4 : 5) : 1;
    // source tensor of this chunk (a chunk never straddles the two sources: C0 is a multiple of 64)
    const int ch0 = chunk * CK;
    const float* xsrc = (ch0 < a.C0) ? a.x0 : a.x1;
    const int Cs = (ch0 < a.C0) ? a.C0 : a.C1, cs0 = (ch0 < a.C0) ? ch0 : ch0 - a.C0;
    wg_f32x16 acc[NT], tot[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; tot[i][r] = 0.f; }
    const int li = lane & 31, lh = lane >> 5;
    for (int tile = t0; tile < t1; ++tile) {
        const int tx = tile % tilesX, ty = (tile / tilesX) % tilesY, b = tile / (tilesX * tilesY);
        const int y0 = ty * TR, x0 = tx * 32;
        __syncthreads();                          // the previous tile's fragments have been read
        // dy tile: 128 px x 64 co = 2048 float4
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i, q = e >> 4, c4 = e & 15;
            const int y = y0 + (q >> 5), x = x0 + (q & 31);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (x < a.W) v = *reinterpret_cast<const float4*>(a.dy + ((size_t)(b * a.H + y) * a.W + x) * a.Cout + cb * 64 + 4 * c4);
            *reinterpret_cast<float4*>(dys + q * 64 + 4 * c4) = v;
        }
        // activated patch: NPIX px x CK ci
        for (int e = tid; e < NPIX * (CK / 4); e += 256) {
            const int q = e / (CK / 4), c4 = e % (CK / 4);
            const int pr = q / PW, pc = q - pr * PW;
            const int y = y0 + pr - PAD, x = x0 + pc - PAD;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
                const size_t sp = a.up ? ((size_t)(b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1)) : ((size_t)(b * a.H + y) * a.W + x);
                v = *reinterpret_cast<const float4*>(xsrc + sp * Cs + cs0 + 4 * c4);
                if (a.coef) {
                    const size_t pl = (size_t)a.B * Cin, bc = (size_t)b * Cin + ch0 + 4 * c4;
                    const float4 m = *reinterpret_cast<const float4*>(a.coef + bc), g = *reinterpret_cast<const float4*>(a.coef + pl + bc);
                    const float4 d = *reinterpret_cast<const float4*>(a.coef + 2 * pl + bc);
                    v.x = (v.x - m.x) * g.x + d.x; v.y = (v.y - m.y) * g.y + d.y; v.z = (v.z - m.z) * g.z + d.z; v.w = (v.w - m.w) * g.w + d.w;
                }
                if (a.silu) { v.x = silu_t(v.x); v.y = silu_t(v.y); v.z = silu_t(v.z); v.w = silu_t(v.w); }
            }
            *reinterpret_cast<float4*>(xs + q * CK + 4 * c4) = v;
        }
        __syncthreads();
        // 64 k-steps of two pixels (x, x + 1 of one row); A = dy[px][co], B = act[px + tap][ci]
#pragma unroll 4
        for (int k = 0; k < 64; ++k) {
            const int py = k >> 4, pxx = (k & 15) * 2 + lh;
            const float av = dys[(py * 32 + pxx) * 64 + cw * 32 + li];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (i < ntap) {
                    const int t = tapbase + i, ky = (TAPS == 9) ? t / 3 : 0, kx = (TAPS == 9) ? t - 3 * ky : 0;
                    const float bv = xs[((py + ky) * PW + pxx + kx) * CK + ((TAPS == 9) ? 0 : tg * 32) + li];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            tot[i] += acc[i];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
    }
    // partial tile: [64 co][TAPS][CK ci]; D row = co = (r & 3) + 8 (r >> 2) + 4 lh, column = ci = li
    float* o = a.part + ((((size_t)p * ncb + cb) * nchunk + chunk) * 64) * TAPS * CK;
#pragma unroll
    for (int i = 0; i < NT; ++i)
        if (i < ntap) {
            const int t = tapbase + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                o[((size_t)co * TAPS + t) * CK + ((TAPS == 9) ? 0 : tg * 32) + li] = tot[i][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------
// conv wgrad on the 16-bit matrix pipe (the default): the same contraction, products formed like the forward kernel's -- both operands
// split into NS fp16 terms (NS = 2: hi . hi + hi . mid + mid . hi, fp32-grade; NS = 1: plain fp16 operands with fp32 accumulation,
// the arithmetic of the reference trainer's `precision: 16`), v_mfma_f32_16x16x32_f16.
//
// That MFMA wants, per lane, 8 consecutive k of one row. The contraction index is (sample, y, x), the operands are NHWC, and a tap
// shifts (y, x) by one -- an 8-vector along x or y would be misaligned for eight of the nine taps. So the 8-vector runs over the
// BATCH: k = (pixel, sample mod 8). A staging thread owns one (pixel, channel quad), loads the 8 samples of a batch group (8 coalesced
// float4 rows, one per sample), applies GroupNorm / FiLM / SiLU, splits, and writes for each of its 4 channels ONE 16-byte unit
// [8 samples] to LDS at [channel][pixel]: no scattered 2-byte stores, no shuffles, and a tap is a plain (aligned) pixel offset.
// Channel rows are padded by one unit so that the 16 channels a fragment read touches fall into distinct bank groups.
// Workgroup = 4 waves on a 64 co x 32 ci (x 9 taps) tile, pixel tile 2 x 8 (+ halo), one batch group at a time: 80 KB of LDS, two
// workgroups per CU -- one stages while the other multiplies. Wave = (co 32-block) x (taps 0..4 | 5..8) [3x3] or (ci 16-half) [1x1].
// ------------------------------------------------------------------------------------------------------------------
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
typedef float wg_v4f __attribute__((ext_vector_type(4)));

template <int TAPS, int NS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x_kernel(const WgradArgs a, int G) {
    constexpr int PAD = (TAPS == 9) ? 1 : 0;
    constexpr int TX = 8, TY = 2, NPX = TX * TY;
    constexpr int PW = TX + 2 * PAD, PH = TY + 2 * PAD, NHP = PW * PH;          // 10 x 4 = 40 halo pixels | 16
    constexpr int CK = 32;
    constexpr int APITCH = NHP + 1, DPITCH = NPX + 1;                           // 16-byte units per channel row
    constexpr int NT = (TAPS == 9) ? 5 : 1, NJ = (TAPS == 9) ? 2 : 1;          // taps and ci 16-blocks per wave
    __shared__ uint4 actL[NS][CK * APITCH];
    __shared__ uint4 dyL[NS][64 * DPITCH];
    __shared__ float coefL[3][8][CK];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, tg = wave >> 1;
    const int r = lane & 15, g = lane >> 4;
    const int Cin = a.C0 + a.C1, nchunk = Cin / CK, ncb = a.Cout >> 6;
    int bid = blockIdx.x;
    const int chunk = bid % nchunk; bid /= nchunk;
    const int cb = bid % ncb;
    const int p = bid / ncb;
    const int tilesX = (a.W + TX - 1) / TX, tilesY = (a.H + TY - 1) / TY, tpg = tilesX * tilesY;
    const int ntile = G * tpg;
    const int t0 = (int)((long long)ntile * p / a.P), t1 = (int)((long long)ntile * (p + 1) / a.P);
    const int tapbase = (TAPS == 9) ? tg * 5 : 0, ntap = (TAPS == 9) ? (tg ? 4 : 5) : 1;
    const int ch0 = chunk * CK;
    const float* xsrc = (ch0 < a.C0) ? a.x0 : a.x1;
    const int Cs = (ch0 < a.C0) ? a.C0 : a.C1, cs0 = (ch0 < a.C0) ? ch0 : ch0 - a.C0;
    const int sH = a.up ? a.H >> 1 : a.H, sW = a.up ? a.W >> 1 : a.W;
    wg_v4f acc[2][NJ][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[i][j][t] = wg_v4f{0.f, 0.f, 0.f, 0.f};
    int cur_grp = -1;
    for (int tile = t0; tile < t1; ++tile) {
        const int grp = tile / tpg, tt = tile - grp * tpg;
        const int y0 = (tt / tilesX) * TY, x0 = (tt % tilesX) * TX;
        __syncthreads();                          // the previous tile's fragments have been read
        if (grp != cur_grp) {                     // (mean, a, d) of the group's 8 samples x 32 channels
            for (int e = tid; e < 3 * 8 * CK; e += 256) {
                const int pl = e / (8 * CK), i = (e / CK) & 7, c = e & (CK - 1), b = grp * 8 + i;
                coefL[pl][i][c] = (a.coef && b < a.B) ? a.coef[((size_t)pl * a.B + b) * Cin + ch0 + c] : (pl == 1 ? 1.f : 0.f);
            }
            cur_grp = grp;
            __syncthreads();
        }
        // dy tile: 16 pixels x 16 channel quads, one task per thread
        {
            const int n = tid >> 4, q = tid & 15;
            const int y = y0 + (n >> 3), x = x0 + (n & 7);
            const bool in = y < a.H && x < a.W;
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int b = grp * 8 + i;
                v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (in && b < a.B) v[i] = *reinterpret_cast<const float4*>(a.dy + ((size_t)(b * a.H + y) * a.W + x) * a.Cout + cb * 64 + 4 * q);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                wg_f16x8 h, m;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float f = c == 0 ? v[i].x : c == 1 ? v[i].y : c == 2 ? v[i].z : v[i].w;
                    h[i] = (_Float16)f;
                    m[i] = (_Float16)(f - (float)h[i]);
                }
                dyL[0][(4 * q + c) * DPITCH + n] = __builtin_bit_cast(uint4, h);
                if (NS == 2) dyL[NS - 1][(4 * q + c) * DPITCH + n] = __builtin_bit_cast(uint4, m);
            }
        }
        // activated patch: NHP pixels x 8 channel quads
        for (int task = tid; task < NHP * 8; task += 256) {
            const int hp = task >> 3, q = task & 7;
            const int gy = y0 - PAD + hp / PW, gx = x0 - PAD + hp % PW;
            const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const int sy = a.up ? gy >> 1 : gy, sx = a.up ? gx >> 1 : gx;
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int b = grp * 8 + i;
                v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (in && b < a.B) v[i] = *reinterpret_cast<const float4*>(xsrc + ((size_t)(b * sH + sy) * sW + sx) * Cs + cs0 + 4 * q);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                wg_f16x8 h, m;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float f = c == 0 ? v[i].x : c == 1 ? v[i].y : c == 2 ? v[i].z : v[i].w;
                    f = (f - coefL[0][i][4 * q + c]) * coefL[1][i][4 * q + c] + coefL[2][i][4 * q + c];
                    if (a.silu) f = silu_t(f);
                    if (!(in && grp * 8 + i < a.B)) f = 0.f;          // padding pixels and samples past the batch contribute nothing
                    h[i] = (_Float16)f;
                    m[i] = (_Float16)(f - (float)h[i]);
                }
                actL[0][(4 * q + c) * APITCH + hp] = __builtin_bit_cast(uint4, h);
                if (NS == 2) actL[NS - 1][(4 * q + c) * APITCH + hp] = __builtin_bit_cast(uint4, m);
            }
        }
        __syncthreads();
        // 4 k-steps of 4 pixels x 8 samples; A = dy[co][k], B = act[k + tap][ci]
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ty = ks >> 1, tx = 4 * (ks & 1) + g;
            wg_f16x8 fa[NS][2];
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[s2][i] = __builtin_bit_cast(wg_f16x8, dyL[s2][(cw * 32 + i * 16 + r) * DPITCH + ty * TX + tx]);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < ntap) {
                    const int tap = tapbase + t, ky = (TAPS == 9) ? tap / 3 : 0, kx = (TAPS == 9) ? tap - 3 * ky : 0;
                    wg_f16x8 fb[NS][NJ];
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const int ci = (TAPS == 9) ? j * 16 + r : tg * 16 + r;
                            fb[s2][j] = __builtin_bit_cast(wg_f16x8, actL[s2][ci * APITCH + (ty + ky) * PW + tx + kx]);
                        }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            if (NS == 2) {
                                acc[i][j][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[NS - 1][i], fb[0][j], acc[i][j][t], 0, 0, 0);
                                acc[i][j][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[NS - 1][j], acc[i][j][t], 0, 0, 0);
                            }
                            acc[i][j][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[0][j], acc[i][j][t], 0, 0, 0);
                        }
                }
            }
        }
    }
    // partial tile [64 co][TAPS][32 ci]; D row = co = 4 g + e, column = ci = r
    float* o = a.part + ((((size_t)p * ncb + cb) * nchunk + chunk) * 64) * TAPS * CK;
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < ntap) {
            const int tap = tapbase + t;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int co = cw * 32 + i * 16 + 4 * g + e, ci = (TAPS == 9) ? j * 16 + r : tg * 16 + r;
                        o[((size_t)co * TAPS + tap) * CK + ci] = acc[i][j][t][e];
                    }
        }
}

// ------------------------------------------------------------------------------------------------------------------
// The 3x3 weight gradient in two passes (the default): (1) wgrad_image_kernel writes each operand ONCE as its "k-image" -- activated
// (GroupNorm / FiLM / SiLU, upsampled where the convolution reads an upsampled input), split into NS fp16 planes, laid out
// [plane][batch group][channel][y][x][8 samples] (16-byte units) -- so the transform and the SiLU are paid once per element instead of
// once per (output-channel block x halo overlap) as in conv_wgrad_x_kernel above; (2) conv_wgrad_img_kernel is then a pure fp16 GEMM
// over those images: staging is 16-byte copies (a 10-pixel halo row = 160 contiguous bytes), the next tile's units are fetched into
// registers while the current tile's MFMAs run, and two workgroups share a CU. Same tile, k order and partial layout as conv_wgrad_x_kernel (wave = co 32-block x ci 16-block, all nine taps); the images cost one extra write + read of each operand (= its fp32 size per pass), ~1 % of the kernel's time.
// ------------------------------------------------------------------------------------------------------------------
struct ImageArgs {
    const float* x0; const float* x1; int C0, C1;
    const float* coef; int silu, up;
    int B, H, W, G;
    uint4* img;            // [NS][G][C0 + C1][H * W]
    int NS;
    float* bsum;           // optional [gridDim.x * G][C0 + C1]: per-workgroup channel sums of the RAW source (the dy image pass: the bias
                           // gradient's partial sums come for free, the tensor is not read a second time), folded by bias_fold_f32_kernel
};

__global__ __launch_bounds__(256) void wgrad_image_kernel(const ImageArgs a) {
    // workgroup = 8 consecutive pixels x 32 channel quads; wave = 8 pixels x 8 quads: loads and stores are whole 128-byte lines
    const int tid = threadIdx.x, px = tid & 7, q = blockIdx.y * 32 + (tid >> 3), grp = blockIdx.z;
    const int C = a.C0 + a.C1, HW = a.H * a.W;
    const int n = blockIdx.x * 8 + px;
    if (4 * q >= C) return;                          // (whole groups of 8 lanes leave together)
    // a ragged last pixel group: its lanes beyond the image stay in the kernel (they take part in the 8-lane sums below with zeros:
    // the cross-lane steps need every lane of the group at the same instruction) and skip their loads and stores
    const bool live = n < HW;
    if (!live && !a.bsum) return;
    const int nn = live ? n : 0;
    const int y = nn / a.W, x = nn - y * a.W;
    const int c0 = 4 * q;
    const float* src = (c0 < a.C0) ? a.x0 : a.x1;
    const int Cs = (c0 < a.C0) ? a.C0 : a.C1, cs = (c0 < a.C0) ? c0 : c0 - a.C0;
    const int sH = a.up ? a.H >> 1 : a.H, sW = a.up ? a.W >> 1 : a.W, sy = a.up ? y >> 1 : y, sx = a.up ? x >> 1 : x;
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int b = grp * 8 + i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < a.B && live) {
            v[i] = *reinterpret_cast<const float4*>(src + ((size_t)(b * sH + sy) * sW + sx) * Cs + cs);
            if (a.coef) {
                const size_t pl = (size_t)a.B * C, bc = (size_t)b * C + c0;
                const float4 m = *reinterpret_cast<const float4*>(a.coef + bc), gg = *reinterpret_cast<const float4*>(a.coef + pl + bc);
                const float4 d = *reinterpret_cast<const float4*>(a.coef + 2 * pl + bc);
                v[i].x = (v[i].x - m.x) * gg.x + d.x; v[i].y = (v[i].y - m.y) * gg.y + d.y;
                v[i].z = (v[i].z - m.z) * gg.z + d.z; v[i].w = (v[i].w - m.w) * gg.w + d.w;
            }
            if (a.silu) { v[i].x = silu_t(v[i].x); v[i].y = silu_t(v[i].y); v[i].z = silu_t(v[i].z); v[i].w = silu_t(v[i].w); }
        }
    }
    if (a.bsum) {       // (dy pass: no coefficients, no SiLU -- v holds the raw gradients) sum over the 8 samples, then over the 8 pixels
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s0 += v[i].x; s1 += v[i].y; s2 += v[i].z; s3 += v[i].w; }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) { s0 += __shfl_xor(s0, m, 8); s1 += __shfl_xor(s1, m, 8); s2 += __shfl_xor(s2, m, 8); s3 += __shfl_xor(s3, m, 8); }
        if (px == 0)
            *reinterpret_cast<float4*>(a.bsum + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x)) * C + c0) = make_float4(s0, s1, s2, s3);
    }
    if (!live) return;
    const size_t plane = (size_t)a.G * C * HW;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        wg_f16x8 h, m;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float f = c == 0 ? v[i].x : c == 1 ? v[i].y : c == 2 ? v[i].z : v[i].w;
            h[i] = (_Float16)f;
            m[i] = (_Float16)(f - (float)h[i]);
        }
        const size_t u = ((size_t)grp * C + c0 + c) * HW + n;
        a.img[u] = __builtin_bit_cast(uint4, h);
        if (a.NS == 2) a.img[plane + u] = __builtin_bit_cast(uint4, m);
    }
}

struct WgradImgArgs {
    const uint4* act;      // [NS][G][Cin][H * W]
    const uint4* dy;       // [NS][G][Cout][H * W]
    int Cin, Cout, H, W, G;
    float* part;           // [P][Cout/64][Cin/32][64 co][9][32 ci]
    int P;
};

template <int TAPS, int NS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_img_kernel(const WgradImgArgs a) {
    // TAPS = 9: 32 input channels per workgroup, wave = (co 32-block) x (ci 16-block), all nine taps: 18 accumulator tiles of 16 x 16
    // TAPS = 1: 64 input channels per workgroup (no halo, no tap reuse: the wider tile halves the staging per MFMA), wave = (co 32-block) x
    //           (ci 32-block): 4 accumulator tiles
    constexpr int PAD = (TAPS == 9) ? 1 : 0;
    constexpr int TX = 8, TY = 2, NPX = 16, PW = TX + 2 * PAD, PH = TY + 2 * PAD, NHP = PW * PH, CK = (TAPS == 9) ? 32 : 64;
    constexpr int NJ = (TAPS == 9) ? 1 : 2;                                  // ci 16-blocks per wave
    // 16-byte units per channel row: pitch = 2 (mod 8). ds_read_b128 serves a wave in four groups of 16 lanes, each holding the 16 channel
    // rows of a fragment once, 8 of them (r) at k-group g and the other 8 (r + 8) at g + 1: rows r and r + 8 then share an even 16-byte bank
    // slot, and the +1 of the second half moves it to the odd one -- conflict-free (a pitch of 41 measured 41 % conflict cycles)
    constexpr int APITCH = NHP + 2, DPITCH = NPX + 2;
    constexpr int NA = NS * CK * NHP, ND = NS * 64 * NPX;                   // units per tile: 2560 + 2048 (3x3, NS = 2)
    constexpr int LA = NA / 256, LD = ND / 256;                             // per thread: 10 + 8
    __shared__ uint4 actL[NS][CK * APITCH];
    __shared__ uint4 dyL[NS][64 * DPITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, cj = wave >> 1;
    const int r = lane & 15, g = lane >> 4;
    const int nchunk = a.Cin / CK, ncb = a.Cout >> 6, HW = a.H * a.W;
    // workgroups go round-robin over the 8 XCDs (each with its own L2): the ncb x nchunk workgroups of one pixel range p share its dy tiles
    // (across ci chunks) and act tiles (across co blocks), so they are numbered to run on ONE XCD at the same time. P is a multiple of 8.
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3, per = ncb * nchunk;
    const int inner = jj % per, chunk = inner % nchunk, cb = inner / nchunk;
    const int p = (jj / per) * 8 + xcd;
    const int tilesX = (a.W + TX - 1) / TX, tilesY = (a.H + TY - 1) / TY, tpg = tilesX * tilesY;
    const int ntile = a.G * tpg;
    const int t0 = (int)((long long)ntile * p / a.P), t1 = (int)((long long)ntile * (p + 1) / a.P);
    const size_t aplane = (size_t)a.G * a.Cin * HW, dplane = (size_t)a.G * a.Cout * HW;
    wg_v4f acc[2][NJ][TAPS];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[i][j][t] = wg_v4f{0.f, 0.f, 0.f, 0.f};
    uint4 ra[LA], rd[LD];
    auto fetch = [&](int tile) {
        const int grp = tile / tpg, tt = tile - grp * tpg;
        const int y0 = (tt / tilesX) * TY, x0 = (tt % tilesX) * TX;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int e = tid + 256 * i, s2 = e / (CK * NHP), rem = e - s2 * (CK * NHP), ch = rem / NHP, hp = rem - ch * NHP;
            const int gy = y0 - PAD + hp / PW, gx = x0 - PAD + hp % PW;
            ra[i] = make_uint4(0u, 0u, 0u, 0u);
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                ra[i] = a.act[s2 * aplane + ((size_t)grp * a.Cin + chunk * CK + ch) * HW + gy * a.W + gx];
        }
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            const int e = tid + 256 * i, s2 = e / (64 * NPX), rem = e - s2 * (64 * NPX), co = rem >> 4, n = rem & 15;
            const int y = y0 + (n >> 3), x = x0 + (n & 7);
            rd[i] = make_uint4(0u, 0u, 0u, 0u);
            if (y < a.H && x < a.W) rd[i] = a.dy[s2 * dplane + ((size_t)grp * a.Cout + cb * 64 + co) * HW + y * a.W + x];
        }
    };
    if (t0 < t1) fetch(t0);
    for (int tile = t0; tile < t1; ++tile) {
        __syncthreads();                          // the previous tile's fragments have been read
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int e = tid + 256 * i, s2 = e / (CK * NHP), rem = e - s2 * (CK * NHP), ch = rem / NHP, hp = rem - ch * NHP;
            actL[s2][ch * APITCH + hp] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            const int e = tid + 256 * i, s2 = e / (64 * NPX), rem = e - s2 * (64 * NPX), co = rem >> 4, n = rem & 15;
            dyL[s2][co * DPITCH + n] = rd[i];
        }
        __syncthreads();
        if (tile + 1 < t1) fetch(tile + 1);       // in flight while this tile multiplies
#pragma unroll 1
        for (int ks = 0; ks < 4; ++ks) {      // not unrolled: the compiler otherwise hoists all 44 fragment reads and spills
            const int ty = ks >> 1, tx = 4 * (ks & 1) + g;
            wg_f16x8 fa[NS][2];
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[s2][i] = __builtin_bit_cast(wg_f16x8, dyL[s2][(cw * 32 + i * 16 + r) * DPITCH + ty * TX + tx]);
            // fragments of tap t + 1 are read while tap t multiplies
            wg_f16x8 fb[2][NS][NJ];
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    fb[0][s2][j] = __builtin_bit_cast(wg_f16x8, actL[s2][((cj * NJ + j) * 16 + r) * APITCH + ty * PW + tx]);
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                if (t + 1 < TAPS) {
                    const int ky = (t + 1) / 3, kx = (t + 1) - 3 * ky;
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            fb[(t + 1) & 1][s2][j] =
                                __builtin_bit_cast(wg_f16x8, actL[s2][((cj * NJ + j) * 16 + r) * APITCH + (ty + ky) * PW + tx + kx]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (NS == 2) {
                            acc[i][j][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[NS - 1][i], fb[t & 1][0][j], acc[i][j][t], 0, 0, 0);
                            acc[i][j][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[t & 1][NS - 1][j], acc[i][j][t], 0, 0, 0);
                        }
                        acc[i][j][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[t & 1][0][j], acc[i][j][t], 0, 0, 0);
                    }
                if (TAPS == 9) {
                    // keep the program order "reads of tap t + 1, then the MFMAs of tap t" (the scheduler otherwise clusters the reads of two
                    // taps right in front of their MFMAs and the LDS latency is exposed every 12 MFMAs)
                    if (t < 8) __builtin_amdgcn_sched_group_barrier(0x100, NS, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NS == 2 ? 6 : 2, 0);
                }
            }
        }
    }
    float* o = a.part + ((((size_t)p * ncb + cb) * nchunk + chunk) * 64) * TAPS * CK;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[((size_t)(cw * 32 + i * 16 + 4 * g + e) * TAPS + t) * CK + (cj * NJ + j) * 16 + r] = acc[i][j][t][e];
}

// convolution weight-gradient family: CDDPM_WGRAD = h3 (default: fp16 two-term split, fp32-grade), h1 (plain fp16 operands), f32 (the
// fp32-MFMA kernel above)
// training arithmetic of the process: 32 = fp32-grade (two-term fp16 splits), 16 = plain fp16 operands with fp32 accumulation (the reference
// trainer's `precision: 16`). Initial value from CDDPM_TRAIN_PRECISION, changed by cddpm_set_train_precision (the DDPM_2D mirror passes the
// Trainer's precision).
static std::atomic<int> g_train_precision{-1};
int train_precision() {
    int v = g_train_precision.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* tp = getenv("CDDPM_TRAIN_PRECISION");
        v = (tp && !strcmp(tp, "16")) ? 16 : 32;
        g_train_precision.store(v, std::memory_order_relaxed);
    }
    return v;
}
int set_train_precision(int bits) {
    const int prev = train_precision();
    g_train_precision.store(bits == 16 ? 16 : 32, std::memory_order_relaxed);
    return prev;
}
int wgrad_mode() {
    static const int forced = [] {
        const char* e = getenv("CDDPM_WGRAD");
        return !e ? -1 : !strcmp(e, "f32") ? 0 : !strcmp(e, "h1") ? 1 : 2;
    }();
    if (forced >= 0) return forced;
    return train_precision() == 16 ? 1 : 2;         // precision 16: plain fp16 operands everywhere in the training step
}

// dW[co][ci][t] (PyTorch layout) = sum over the P partial tiles in the order of p
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, int P, int Cout, int Cin, int taps, int CK,
                                                                float* __restrict__ dw) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)Cout * Cin * taps) return;
    const int t = (int)(e % taps);
    const int ci = (int)((e / taps) % Cin), co = (int)(e / ((long long)taps * Cin));
    const int ncb = Cout >> 6, nchunk = Cin / CK;
    const size_t tile = (size_t)64 * taps * CK;
    const size_t inner = (((size_t)(co >> 6) * nchunk + ci / CK) * 64 + (co & 63)) * taps * CK + (size_t)t * CK + (ci % CK);
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += part[(size_t)p * ncb * nchunk * tile + inner];
    dw[e] = s;
}

// dL/db = sum over pixels of dy: pixel chunks across workgroups (coalesced float4 rows, fp64 partial sums), then one fold
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const float* __restrict__ dy, long long npix, int C, int nchunk,
                                                                double* __restrict__ part) {
    __shared__ double red[1024];
    const int nq = C >> 2, rows = 256 / nq, tid = threadIdx.x, q = tid % nq, r = tid / nq;
    const long long per = (npix + nchunk - 1) / nchunk, p0 = per * blockIdx.x, p1 = p0 + per < npix ? p0 + per : npix;
    double s[4] = {0, 0, 0, 0};
    if (r < rows)
        for (long long p = p0 + r; p < p1; p += rows) {
            const float4 v = *reinterpret_cast<const float4*>(dy + (size_t)p * C + 4 * q);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        }
    for (int rr = 0; rr < rows; ++rr) {          // fold the rows one at a time through a C-wide buffer
        __syncthreads();
        if (r == rr) for (int i = 0; i < 4; ++i) red[4 * q + i] = (rr ? red[4 * q + i] : 0.0) + s[i];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) part[(size_t)blockIdx.x * C + c] = red[c];
}
__global__ __launch_bounds__(256) void bias_grad_fold_kernel(const double* __restrict__ part, int nchunk, int C, float* __restrict__ db) {
    // 16 channels per workgroup, the chunks in 16 interleaved slices (fixed order within a slice, slices folded in order)
    __shared__ double red[16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    double t = 0;
    if (c < C)
        for (int k = sl; k < nchunk; k += 16) t += part[(size_t)k * C + c];
    red[sl][cl] = t;
    __syncthreads();
    if (sl == 0 && c < C) {
        double u = 0;
        for (int k = 0; k < 16; ++k) u += red[k][cl];
        db[c] = (float)u;
    }
}
// the same fold over fp32 partial rows (the rows the dy image pass of the weight gradient leaves behind): db[c] = sum_k part[k][c], fp64
// blockIdx.y = row range [per * y, per * (y + 1)): out[y][c] (one range: the gradient itself; several: partial rows for a second call)
__global__ __launch_bounds__(256) void bias_fold_f32_kernel(const float* __restrict__ part, int nrow, int per, int C, float* __restrict__ out) {
    __shared__ double red[16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    const int r0 = per * blockIdx.y, r1 = min(nrow, r0 + per);
    double t = 0;
    if (c < C)
        for (int k = r0 + sl; k < r1; k += 16) t += (double)part[(size_t)k * C + c];
    red[sl][cl] = t;
    __syncthreads();
    if (sl == 0 && c < C) {
        double u = 0;
        for (int k = 0; k < 16; ++k) u += red[k][cl];
        out[(size_t)blockIdx.y * C + c] = (float)u;
    }
}
// rows -> db in one launch (few rows) or two (row ranges of 128 across workgroups first: a 4096-row fold on C / 16 workgroups took 30 us);
// `rows` is overwritten by the intermediate partial rows (its first ceil(nrow / 128) rows)
static void bias_fold_rows(float* rows, int nrow, int C, float* db, hipStream_t stream) {
    if (nrow > 256) {
        const int nr = (nrow + 127) / 128;
        // (the ranges run concurrently: the partial rows go BEHIND the input rows, not over them)
        float* tmp = rows + (size_t)nrow * C;
        hipLaunchKernelGGL(bias_fold_f32_kernel, dim3((C + 15) / 16, nr), dim3(256), 0, stream, rows, nrow, 128, C, tmp);
        hipLaunchKernelGGL(bias_fold_f32_kernel, dim3((C + 15) / 16, 1), dim3(256), 0, stream, tmp, nr, nr, C, db);
    } else {
        hipLaunchKernelGGL(bias_fold_f32_kernel, dim3((C + 15) / 16, 1), dim3(256), 0, stream, rows, nrow, nrow, C, db);
    }
}
// scratch: nchunk * C doubles, nchunk = bias_grad_chunks(npix, C, scratch floats available)
int bias_grad_chunks(long long npix, int C, size_t scratch_floats) {
    long long n = (long long)(scratch_floats / 2 / (size_t)C);
    if (n > 512) n = 512;
    if (n > npix) n = npix;
    return n < 1 ? 1 : (int)n;
}
static void bias_grad_run(const float* dy, long long npix, int C, float* db, double* scratch, int nchunk, hipStream_t stream) {
    hipLaunchKernelGGL(bias_grad_partial_kernel, dim3(nchunk), dim3(256), 0, stream, dy, npix, C, nchunk, scratch);
    hipLaunchKernelGGL(bias_grad_fold_kernel, dim3((C + 15) / 16), dim3(256), 0, stream, scratch, nchunk, C, db);
}

int conv_wgrad_parts(int B, int H, int W, int Cin, int Cout, int taps) {
    // enough workgroups for two per CU, at most one tile each, at least 1
    // mirrors launch_conv_wgrad's choice of kernel: the split families tile (8 samples) x (2 x 8 pixels); chunks of 32 input channels
    // (3x3, and the single-pass 1x1 fallback) or 64 (1x1 over images, and the fp32 family's 1x1)
    const bool f32k = wgrad_mode() == 0;
    const bool img = !f32k && (taps == 9 || Cin % 64 == 0);
    const int ntile = !f32k ? ((B + 7) / 8) * ((H + 1) / 2) * ((W + 7) / 8) : B * (H / 4) * ((W + 31) / 32);
    const int per = (Cout / 64) * (Cin / ((taps == 9 || (!f32k && !img)) ? 32 : 64));
    int P = (512 + per - 1) / per;
    if (P > ntile) P = ntile;
    if (P > 64) P = 64;
    if (P < 1) P = 1;
    if (img) P = (P + 7) & ~7;      // the two-pass kernel spreads its pixel ranges over the 8 XCDs
    return P;
}

// scratch of the two-pass 3x3 family beyond the partial tiles: the two k-images, in 16-byte units (0: this call does not use them)
size_t conv_wgrad_image_units(int B, int H, int W, int Cin, int Cout, int taps) {
    const int mode = wgrad_mode();
    if (mode == 0 || (taps == 1 && Cin % 64)) return 0;
    // + the bias gradient's partial rows written by the dy image pass: [(HW / 8 rounded up) * G][Cout] floats
    const size_t bias_rows = (size_t)((H * W + 7) / 8) * ((B + 7) / 8);
    const size_t bias_units = ((bias_rows + (bias_rows + 127) / 128) * Cout + 3) / 4;       // + the second-level partial rows behind them
    return (size_t)(mode == 2 ? 2 : 1) * ((B + 7) / 8) * (size_t)(Cin + Cout) * H * W + bias_units;
}

void launch_conv_wgrad(const float* x0, int C0, const float* x1, int C1, const float* coef, int silu, int up, const float* dy, int B, int H,
                       int W, int Cout, int taps, float* part, int P, void* images, float* dw, float* db, hipStream_t stream) {
    WgradArgs a;
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1; a.coef = coef; a.silu = silu; a.up = up; a.dy = dy; a.B = B; a.H = H; a.W = W; a.Cout = Cout;
    a.part = part; a.P = P;
    const int Cin = C0 + C1, mode = wgrad_mode();
    const int G = (B + 7) / 8;
    int CK = 32;
    if (mode != 0 && (taps == 9 || Cin % 64 == 0)) {
        // pass 1: the two k-images; pass 2: the GEMM over them
        const int NS = mode == 2 ? 2 : 1;
        uint4* aimg = static_cast<uint4*>(images);
        uint4* dimg = aimg + (size_t)NS * G * Cin * H * W;
        ImageArgs ia;
        ia.x0 = x0; ia.x1 = x1; ia.C0 = C0; ia.C1 = C1; ia.coef = coef; ia.silu = silu; ia.up = up; ia.B = B; ia.H = H; ia.W = W; ia.G = G;
        ia.img = aimg; ia.NS = NS; ia.bsum = nullptr;
        hipLaunchKernelGGL(wgrad_image_kernel, dim3((H * W + 7) / 8, (Cin + 127) / 128, G), dim3(256), 0, stream, ia);
        ia.x0 = dy; ia.x1 = nullptr; ia.C0 = Cout; ia.C1 = 0; ia.coef = nullptr; ia.silu = 0; ia.up = 0; ia.img = dimg;
        float* bsum = reinterpret_cast<float*>(dimg + (size_t)NS * G * Cout * H * W);       // behind the two images (conv_wgrad_image_units)
        ia.bsum = db ? bsum : nullptr;
        hipLaunchKernelGGL(wgrad_image_kernel, dim3((H * W + 7) / 8, (Cout + 127) / 128, G), dim3(256), 0, stream, ia);
        if (db) {       // the bias gradient from the rows the pass just wrote: dy is not read again
            bias_fold_rows(bsum, ((H * W + 7) / 8) * G, Cout, db, stream);
            db = nullptr;
        }
        WgradImgArgs w;
        w.act = aimg; w.dy = dimg; w.Cin = Cin; w.Cout = Cout; w.H = H; w.W = W; w.G = G; w.part = part; w.P = P;
        CK = taps == 9 ? 32 : 64;
        const unsigned grid = (unsigned)(P * (Cout / 64) * (Cin / CK));
        if (taps == 9) {
            if (NS == 2) hipLaunchKernelGGL((conv_wgrad_img_kernel<9, 2>), dim3(grid), dim3(256), 0, stream, w);
            else         hipLaunchKernelGGL((conv_wgrad_img_kernel<9, 1>), dim3(grid), dim3(256), 0, stream, w);
        } else {
            if (NS == 2) hipLaunchKernelGGL((conv_wgrad_img_kernel<1, 2>), dim3(grid), dim3(256), 0, stream, w);
            else         hipLaunchKernelGGL((conv_wgrad_img_kernel<1, 1>), dim3(grid), dim3(256), 0, stream, w);
        }
    } else if (mode == 0) {
        CK = taps == 9 ? 32 : 64;
        const unsigned grid = (unsigned)(P * (Cout / 64) * (Cin / CK));
        if (taps == 9) hipLaunchKernelGGL(conv_wgrad_kernel<9>, dim3(grid), dim3(256), 0, stream, a);
        else           hipLaunchKernelGGL(conv_wgrad_kernel<1>, dim3(grid), dim3(256), 0, stream, a);
    } else {
        const unsigned grid = (unsigned)(P * (Cout / 64) * (Cin / 32));
        if (mode == 1) hipLaunchKernelGGL((conv_wgrad_x_kernel<1, 1>), dim3(grid), dim3(256), 0, stream, a, G);
        else           hipLaunchKernelGGL((conv_wgrad_x_kernel<1, 2>), dim3(grid), dim3(256), 0, stream, a, G);
    }
    const long long n = (long long)Cout * Cin * taps;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, part, P, Cout, Cin, taps, CK, dw);
    if (db) {   // the partial tiles are folded by now (stream order): their memory serves as the bias sum's scratch
        const size_t part_floats = (size_t)P * Cout * Cin * taps;
        bias_grad_run(dy, (long long)B * H * W, Cout, db, reinterpret_cast<double*>(part), bias_grad_chunks((long long)B * H * W, Cout, part_floats), stream);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Batched fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 products): C[z] = alpha * op(A[z]) . op(B[z]), arbitrary M, N, K and
// strides, two-level batch index z = (z0, z1). Used by the attention backward (five N x N x 64 products per head) and by the
// backward of the embedding linears. 64 x 64 tiles, 4 waves (2 x 2 of 32 x 32), K steps of 32 through k-major LDS tiles
// (operand fragment = one ds_read_b32 of 32 consecutive rows / columns at one k). Correctness-first: these products are < 2 % of a
// training step's FLOPs.
// ------------------------------------------------------------------------------------------------------------------
struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    long long lda, ldb, ldc;          // row strides of the STORED matrices
    int transA, transB;               // transA = 0: A stored [M][K];  1: stored [K][M].  transB = 0: B stored [K][N];  1: stored [N][K]
    int nz1;                          // inner batch extent: z = z0 * nz1 + z1
    long long sA0, sA1, sB0, sB1, sC0, sC1;
    float alpha;
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
    constexpr int BM = 64, BN = 64, BK = 32, LDT = BM + 4;      // +4: the transposing stores spread over banks
    __shared__ float As[BK * LDT], Bs[BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int z = blockIdx.z, z0 = z / g.nz1, z1 = z - z0 * g.nz1;
    const float* A = g.A + z0 * g.sA0 + z1 * g.sA1;
    const float* B = g.B + z0 * g.sB0 + z1 * g.sB1;
    float* C = g.C + z0 * g.sC0 + z1 * g.sC1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    wg_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        __syncthreads();
        // A tile -> As[k][m], B tile -> Bs[k][n]: 2048 elements each, 8 per thread; the index that is contiguous in memory runs fastest
        for (int e = tid; e < BM * BK; e += 256) {
            int m, k;
            if (g.transA) { m = e % BM; k = e / BM; } else { k = e % BK; m = e / BK; }
            const int gm = m0 + m, gk = k0 + k;
            float v = 0.f;
            if (gm < g.M && gk < g.K) v = g.transA ? A[(long long)gk * g.lda + gm] : A[(long long)gm * g.lda + gk];
            As[k * LDT + m] = v;
        }
        for (int e = tid; e < BN * BK; e += 256) {
            int n, k;
            if (g.transB) { k = e % BK; n = e / BK; } else { n = e % BN; k = e / BN; }
            const int gn = n0 + n, gk = k0 + k;
            float v = 0.f;
            if (gn < g.N && gk < g.K) v = g.transB ? B[(long long)gn * g.ldb + gk] : B[(long long)gk * g.ldb + gn];
            Bs[k * LDT + n] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float av = As[(kk + lh) * LDT + wm * 32 + li];
            const float bv = Bs[(kk + lh) * LDT + wn * 32 + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + wn * 32 + li;
        if (m < g.M && n < g.N) C[(long long)m * g.ldc + n] = g.alpha * acc[r];
    }
}

void launch_gemm_f32(const GemmArgs& g, int nz, hipStream_t stream) {
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((g.N + 63) / 64, (g.M + 63) / 64, nz), dim3(256), 0, stream, g);
}

// row softmax in place: one workgroup per row of an [rows][n] matrix
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int n) {
    __shared__ float red[256];
    float* row = s + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x;
    float mx = -3.0e38f;
    for (int i = tid; i < n; i += 256) mx = fmaxf(mx, row[i]);
    red[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) { const float e = __expf(row[i] - mx); row[i] = e; sum += e; }
    red[tid] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float inv = 1.0f / red[0];
    for (int i = tid; i < n; i += 256) row[i] *= inv;
}

// dS = P o (dP - rowsum(dP o P)) in place of dP: one workgroup per row
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ p, float* __restrict__ dp, int n) {
    __shared__ float red[256];
    const float* pr = p + (size_t)blockIdx.x * n;
    float* dr = dp + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x;
    float d = 0.f;
    for (int i = tid; i < n; i += 256) d += pr[i] * dr[i];
    red[tid] = d;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    d = red[0];
    for (int i = tid; i < n; i += 256) dr[i] = pr[i] * (dr[i] - d);
}

// QKVAttention backward (src/models/modules/OpenAI_Unet.py:457-476, new attention order): qkv NHWC [B][N][3C] = (q | k | v), heads =
// contiguous groups of 64 channels; w = softmax((q s)^T (k s)), s = 64^-1/4; a = w v. Given da [B][N][C] writes dqkv [B][N][3C].
// p, dp: scratch [B * heads][N][N] each (the probabilities are recomputed, not stored by the forward).
void launch_attention_backward(const float* qkv, const float* da, float* dqkv, float* p, float* dp, int B, int N, int C,
                               hipStream_t stream) {
    const int heads = C / 64, nz = B * heads;
    const long long row = 3LL * C, NN = (long long)N * N;
    GemmArgs g;
    g.nz1 = heads;
    // S = Q K^T / 8
    g.A = qkv; g.lda = row; g.transA = 0; g.sA0 = (long long)N * row; g.sA1 = 64;
    g.B = qkv + C; g.ldb = row; g.transB = 1; g.sB0 = (long long)N * row; g.sB1 = 64;
    g.C = p; g.ldc = N; g.sC0 = heads * NN; g.sC1 = NN;
    g.M = N; g.N = N; g.K = 64; g.alpha = 0.125f;
    launch_gemm_f32(g, nz, stream);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((long long)nz * N)), dim3(256), 0, stream, p, N);
    // dV = P^T dA
    g.A = p; g.lda = N; g.transA = 1; g.sA0 = heads * NN; g.sA1 = NN;
    g.B = da; g.ldb = C; g.transB = 0; g.sB0 = (long long)N * C; g.sB1 = 64;
    g.C = dqkv + 2 * C; g.ldc = row; g.sC0 = (long long)N * row; g.sC1 = 64;
    g.M = N; g.N = 64; g.K = N; g.alpha = 1.0f;
    launch_gemm_f32(g, nz, stream);
    // dP = dA V^T
    g.A = da; g.lda = C; g.transA = 0; g.sA0 = (long long)N * C; g.sA1 = 64;
    g.B = qkv + 2 * C; g.ldb = row; g.transB = 1; g.sB0 = (long long)N * row; g.sB1 = 64;
    g.C = dp; g.ldc = N; g.sC0 = heads * NN; g.sC1 = NN;
    g.M = N; g.N = N; g.K = 64; g.alpha = 1.0f;
    launch_gemm_f32(g, nz, stream);
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((long long)nz * N)), dim3(256), 0, stream, p, dp, N);
    // dQ = dS K / 8
    g.A = dp; g.lda = N; g.transA = 0; g.sA0 = heads * NN; g.sA1 = NN;
    g.B = qkv + C; g.ldb = row; g.transB = 0; g.sB0 = (long long)N * row; g.sB1 = 64;
    g.C = dqkv; g.ldc = row; g.sC0 = (long long)N * row; g.sC1 = 64;
    g.M = N; g.N = 64; g.K = N; g.alpha = 0.125f;
    launch_gemm_f32(g, nz, stream);
    // dK = dS^T Q / 8
    g.A = dp; g.lda = N; g.transA = 1;
    g.B = qkv; g.ldb = row; g.transB = 0; g.sB0 = (long long)N * row; g.sB1 = 64;
    g.C = dqkv + C;
    launch_gemm_f32(g, nz, stream);
}

// y = act(x) W^T + b (torch.nn.Linear, optional SiLU on the input: emb_layers = Sequential(SiLU, Linear), OpenAI_Unet.py:201-207):
// given dy [M][N]: dW [N][K] = dy^T act(x), db [N] = column sums of dy, dx [M][K] = (dy W) o act'(x).  a_scratch: [M][K] (act(x))
__global__ void silu_rows_kernel(const float* __restrict__ x, float* __restrict__ a, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = silu_t(x[i]);
}
__global__ void silu_bwd_mul_kernel(const float* __restrict__ x, float* __restrict__ dx, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = dact(x[i], dx[i], 1);
}
__global__ void colsum_kernel(const float* __restrict__ dy, int M, int N, float* __restrict__ db) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0;
    for (int m = 0; m < M; ++m) s += dy[(size_t)m * N + n];
    db[n] = (float)s;
}

// dx [M][K] = dy [M][N] . W [N][K] for a SKINNY M (the embedding MLPs: M = batch): the 64 x 64-tile GEMM would walk N serially in K / 64
// workgroups (16 workgroups x 368 steps for the 27 emb_layers as one [11776][1024] matrix). Here: workgroup = (64 columns of K, one of
// LIN_NSPLIT ranges of N), thread = (column, quarter of the M rows); dy of the range sits in LDS (broadcast reads), W rows are read
// once, coalesced; the partial sums are folded in the order of the ranges.
constexpr int LIN_NSPLIT = 32;
__global__ __launch_bounds__(256) void skinny_dx_partial_kernel(const float* __restrict__ dy, const float* __restrict__ W, int M, int N, int K,
                                                                float* __restrict__ part /*[LIN_NSPLIT][M][K]*/) {
    __shared__ float dys[64][65];                 // [m][n within the 64-row step]
    const int tid = threadIdx.x, kc = blockIdx.x * 64 + (tid & 63), mg = tid >> 6;
    const int per = (N + LIN_NSPLIT - 1) / LIN_NSPLIT, n0 = blockIdx.y * per, n1 = min(N, n0 + per);
    const int mq = (M + 3) / 4, m0 = mg * mq;     // this thread's rows [m0, m0 + mq), mq <= 16
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int ns = n0; ns < n1; ns += 64) {
        __syncthreads();
        for (int e = tid; e < M * 64; e += 256) {
            const int m = e >> 6, nn = e & 63;
            dys[m][nn] = (ns + nn < n1) ? dy[(size_t)m * N + ns + nn] : 0.f;
        }
        __syncthreads();
        const int lim = min(64, n1 - ns);
        for (int nn = 0; nn < lim; ++nn) {
            const float w = (kc < K) ? W[(size_t)(ns + nn) * K + kc] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i < mq && m0 + i < M) acc[i] = fmaf(dys[m0 + i][nn], w, acc[i]);
        }
    }
    if (kc < K)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < mq && m0 + i < M) part[((size_t)blockIdx.y * M + m0 + i) * K + kc] = acc[i];
}
__global__ __launch_bounds__(256) void skinny_dx_fold_kernel(const float* __restrict__ part, int M, int K, float* __restrict__ dx) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)M * K) return;
    float sum = 0.f;
    for (int sp = 0; sp < LIN_NSPLIT; ++sp) sum += part[(size_t)sp * M * K + e];
    dx[e] = sum;
}
size_t linear_backward_scratch_floats(int M, int N, int K, int silu_in) {
    return (size_t)(silu_in ? M * (size_t)K : 0) + ((M <= 64 && N >= 1024) ? (size_t)LIN_NSPLIT * M * K : 0);
}

void launch_linear_backward(const float* x, const float* W, const float* dy, int M, int N, int K, int silu_in, float* a_scratch,
                            float* dW, float* db, float* dx, hipStream_t stream) {
    const float* act = x;
    if (silu_in) {
        const long long n = (long long)M * K;
        hipLaunchKernelGGL(silu_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, a_scratch, n);
        act = a_scratch;
    }
    GemmArgs g;
    g.nz1 = 1; g.sA0 = g.sA1 = g.sB0 = g.sB1 = g.sC0 = g.sC1 = 0; g.alpha = 1.0f;
    // dW [N][K] = dy^T [N][M] . act [M][K]
    g.A = dy; g.lda = N; g.transA = 1; g.B = act; g.ldb = K; g.transB = 0; g.C = dW; g.ldc = K; g.M = N; g.N = K; g.K = M;
    launch_gemm_f32(g, 1, stream);
    if (db) hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, dy, M, N, db);
    if (dx) {
        // dx [M][K] = dy [M][N] . W [N][K]
        if (M <= 64 && N >= 1024) {
            float* part = a_scratch + (silu_in ? (size_t)M * K : 0);
            hipLaunchKernelGGL(skinny_dx_partial_kernel, dim3((K + 63) / 64, LIN_NSPLIT), dim3(256), 0, stream, dy, W, M, N, K, part);
            hipLaunchKernelGGL(skinny_dx_fold_kernel, dim3((unsigned)(((long long)M * K + 255) / 256)), dim3(256), 0, stream, part, M, K, dx);
        } else {
            g.A = dy; g.lda = N; g.transA = 0; g.B = W; g.ldb = K; g.transB = 0; g.C = dx; g.ldc = K; g.M = M; g.N = K; g.K = N;
            launch_gemm_f32(g, 1, stream);
        }
        if (silu_in) {
            const long long n = (long long)M * K;
            hipLaunchKernelGGL(silu_bwd_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, dx, n);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Elementwise / resampling backward pieces, NHWC fp32, 16 B per lane
// ------------------------------------------------------------------------------------------------------------------
// dx [B,H,W,C] (+)= dyp [B,H/2,W/2,C] at (y >> 1, x >> 1) * scale    (AvgPool2d(2) backward: scale = 1/4)
__global__ __launch_bounds__(256) void unpool2_kernel(const float* __restrict__ dyp, float* __restrict__ dx, int B, int H, int W, int C,
                                                      float scale, int accumulate) {
    const int ncq = C >> 2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * H * W * ncq) return;
    const int cq = (int)(e % ncq);
    const long long pix = e / ncq;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    const float4 v = *reinterpret_cast<const float4*>(dyp + (((size_t)b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * C + 4 * cq);
    float4* o = reinterpret_cast<float4*>(dx + (size_t)pix * C + 4 * cq);
    float4 r = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    if (accumulate) { const float4 t = *o; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    *o = r;
}
// dxp [B,H/2,W/2,C] (+)= sum of the 2x2 block of dy [B,H,W,C]    (nearest x2 upsample backward)
__global__ __launch_bounds__(256) void sumpool2_kernel(const float* __restrict__ dy, float* __restrict__ dxp, int B, int H, int W, int C,
                                                       int accumulate) {
    const int ncq = C >> 2, h2 = H >> 1, w2 = W >> 1;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * h2 * w2 * ncq) return;
    const int cq = (int)(e % ncq);
    const long long pix = e / ncq;
    const int x = (int)(pix % w2), y = (int)((pix / w2) % h2), b = (int)(pix / ((long long)w2 * h2));
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
        for (int dx_ = 0; dx_ < 2; ++dx_) {
            const float4 v = *reinterpret_cast<const float4*>(dy + (((size_t)b * H + 2 * y + dy_) * W + 2 * x + dx_) * C + 4 * cq);
            r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
        }
    float4* o = reinterpret_cast<float4*>(dxp + (size_t)pix * C + 4 * cq);
    if (accumulate) { const float4 t = *o; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    *o = r;
}
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 x = reinterpret_cast<float4*>(a)[i];
    const float4 y = reinterpret_cast<const float4*>(b)[i];
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    reinterpret_cast<float4*>(a)[i] = x;
}
void launch_unpool2(const float* dyp, float* dx, int B, int H, int W, int C, float scale, int accumulate, hipStream_t stream) {
    const long long n = (long long)B * H * W * (C / 4);
    hipLaunchKernelGGL(unpool2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dyp, dx, B, H, W, C, scale, accumulate);
}
void launch_sumpool2(const float* dy, float* dxp, int B, int H, int W, int C, int accumulate, hipStream_t stream) {
    const long long n = (long long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(sumpool2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dy, dxp, B, H, W, C, accumulate);
}
void launch_add_inplace(float* a, const float* b, long long n, hipStream_t stream) {
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, a, b, n / 4);
}

// ------------------------------------------------------------------------------------------------------------------
// The two single-channel convolutions at the ends of the UNet (input_blocks.0: Conv2d(1 -> C); out.2: Conv2d(C -> 1), OpenAI_Unet.py
// :606-612, :793-797). Both weight gradients are a correlation of a C-channel tensor T with a one-channel image s:
//     dW[c][tap] = sum_{b,q} T'[b,q,c] * s[b, q + sign * tap],   T' = act(T) (coefficient planes + SiLU) or T itself
//   input conv : T = dL/d(output) [B,HW,C], s = the image x, sign = +1;   head conv: T = its input tensor, s = dL/d(out), sign = -1
// and the head's input gradient is  d act[b,q,c] = sum_tap w[c][tap] * dout[b, q - tap].
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chan_image_corr_kernel(const float* __restrict__ T, const float* __restrict__ coef, int silu,
                                                              const float* __restrict__ simg, int sign, int B, int H, int W, int C,
                                                              int nsplit, double* __restrict__ part /*[nsplit][C][9]*/) {
    // workgroup = (pixel split, 64-channel block): thread = (channel quad of 16, pixel lane of 16)
    __shared__ double red[16][16][37];
    const int tid = threadIdx.x, cq = tid & 15, pl = tid >> 4;
    const int split = blockIdx.x, cblk = blockIdx.y;
    const int c0 = cblk * 64 + 4 * cq;
    const long long npix = (long long)B * H * W;
    const long long per = (npix + nsplit - 1) / nsplit, p0 = split * per, p1 = min(npix, p0 + per);
    double acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = 0;
    const size_t plane = (size_t)B * C;
    for (long long p = p0 + pl; p < p1; p += 16) {
        const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((long long)W * H));
        float4 v = *reinterpret_cast<const float4*>(T + (size_t)p * C + c0);
        if (coef) {
            const size_t bc = (size_t)b * C + c0;
            const float4 m = *reinterpret_cast<const float4*>(coef + bc), g = *reinterpret_cast<const float4*>(coef + plane + bc);
            const float4 d = *reinterpret_cast<const float4*>(coef + 2 * plane + bc);
            v.x = (v.x - m.x) * g.x + d.x; v.y = (v.y - m.y) * g.y + d.y; v.z = (v.z - m.z) * g.z + d.z; v.w = (v.w - m.w) * g.w + d.w;
        }
        if (silu) { v.x = silu_t(v.x); v.y = silu_t(v.y); v.z = silu_t(v.z); v.w = silu_t(v.w); }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + sign * (t / 3 - 1), xx = x + sign * (t % 3 - 1);
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const double sv = simg[((size_t)b * H + yy) * W + xx];
            acc[0][t] += sv * v.x; acc[1][t] += sv * v.y; acc[2][t] += sv * v.z; acc[3][t] += sv * v.w;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) red[pl][cq][i * 9 + t] = acc[i][t];
    __syncthreads();
    for (int e = tid; e < 16 * 36; e += 256) {
        const int q = e / 36, it = e % 36;
        double sum = 0;
        for (int l = 0; l < 16; ++l) sum += red[l][q][it];
        part[((size_t)split * C + cblk * 64 + 4 * q + it / 9) * 9 + it % 9] = sum;
    }
}
__global__ void corr_reduce_kernel(const double* __restrict__ part, int nsplit, int n, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0;
    for (int p = 0; p < nsplit; ++p) s += part[(size_t)p * n + i];
    dw[i] = (float)s;
}
// part: scratch of 64 * C * 9 doubles;  dw [C][9]
void launch_chan_image_corr(const float* T, const float* coef, int silu, const float* simg, int sign, int B, int H, int W, int C,
                            double* part, float* dw, hipStream_t stream) {
    const int nsplit = 256;      // x C / 64 workgroups (128 of them took 0.5 ms per launch at 16 x 128 x 128)
    hipLaunchKernelGGL(chan_image_corr_kernel, dim3(nsplit, C / 64), dim3(256), 0, stream, T, coef, silu, simg, sign, B, H, W, C, nsplit, part);
    hipLaunchKernelGGL(corr_reduce_kernel, dim3((C * 9 + 255) / 256), dim3(256), 0, stream, part, nsplit, C * 9, dw);
}

// head dgrad: dact[b,q,c] = sum_tap w9[tap][c] * dout[b, q - tap]    (w9: the head's [9][C] weight image)
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ w9, float* __restrict__ dact,
                                                         int B, int H, int W, int C) {
    const int ncq = C >> 2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * H * W * ncq) return;
    const int cq = (int)(e % ncq);
    const long long pix = e / ncq;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y - (t / 3 - 1), xx = x - (t % 3 - 1);
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const float d = dout[((size_t)b * H + yy) * W + xx];
        const float4 w = *reinterpret_cast<const float4*>(w9 + (size_t)t * C + 4 * cq);
        r.x += w.x * d; r.y += w.y * d; r.z += w.z * d; r.w += w.w * d;
    }
    *reinterpret_cast<float4*>(dact + (size_t)pix * C + 4 * cq) = r;
}
void launch_head_dgrad(const float* dout, const float* w9, float* dact, int B, int H, int W, int C, hipStream_t stream) {
    const long long n = (long long)B * H * W * (C / 4);
    hipLaunchKernelGGL(head_dgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dout, w9, dact, B, H, W, C);
}

// loss of p_losses (cond_DDPM.py:636-645): per-sample mean of |out - target| (l1) or (out - target)^2 (l2), times p2_loss_weight[t_b],
// mean over the batch; writes dL/d(out) and the B per-sample terms (their mean is the loss)
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ out, const float* __restrict__ target, const float* __restrict__ w_b,
                                                   int l2, int B, int HW, float grad_scale, float* __restrict__ dout, float* __restrict__ loss_b) {
    __shared__ double red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float wb = w_b ? w_b[b] : 1.0f;
    const float gscale = grad_scale * wb / ((float)B * (float)HW);
    double s = 0;
    for (int p = tid; p < HW; p += 256) {
        const float d = out[(size_t)b * HW + p] - target[(size_t)b * HW + p];
        s += l2 ? (double)d * d : fabs((double)d);
        dout[(size_t)b * HW + p] = l2 ? 2.0f * d * gscale : ((d > 0.f) - (d < 0.f)) * gscale;
    }
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) loss_b[b] = (float)(red[0] / HW * wb);
}
void launch_bias_grad(const float* dy, long long npix, int C, float* db, double* scratch /* 512 * C doubles */, hipStream_t stream) {
    bias_grad_run(dy, npix, C, db, scratch, bias_grad_chunks(npix, C, (size_t)1024 * C), stream);
}
void launch_loss(const float* out, const float* target, const float* w_b, int l2, int B, int HW, float grad_scale, float* dout, float* loss_b,
                 hipStream_t stream) {
    hipLaunchKernelGGL(loss_kernel, dim3(B), dim3(256), 0, stream, out, target, w_b, l2, B, HW, grad_scale, dout, loss_b);
}

// Adam (torch.optim.Adam defaults of DDPM_2D.configure_optimizers, DDPM_2D.py:305-306: lr 1e-4, betas (0.9, 0.999), eps 1e-8, no weight
// decay) on a flat parameter vector: m, v fp32 state; bias corrections passed in
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   long long n, float lr, float b1, float b2, float eps, float bc1, float bc2, float unscale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * unscale;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}
void launch_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, int step,
                 float grad_unscale, hipStream_t stream) {
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2, grad_unscale);
}

// ---- guarded update: a non-finite gradient (an fp16 operand overflow in the precision-16 arithmetic, a diverging loss) must not reach the
// parameters or Adam's moments -- what torch's GradScaler does for the reference trainer (`precision: 16`, configs/trainer/default.yaml:7):
// the step is skipped and the optimizer's step count does not advance. All on the device, no read-back: ctrl = int32[8] =
// {non-finite flag, step, skip this update, updates skipped so far, bits of 1 - beta1^step, bits of 1 - beta2^step, -, -}.
__global__ __launch_bounds__(256) void grad_check_kernel(const float* __restrict__ g, long long n, int* __restrict__ ctrl) {
    bool bad = false;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        // (x - x) is 0 for finite x, NaN for inf / NaN
        bad |= !((v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w) == 0.0f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float x = g[(n & ~3LL) + threadIdx.x]; bad |= !(x - x == 0.0f); }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(ctrl, 1);
}
__global__ void guard_commit_kernel(int* __restrict__ ctrl, float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (ctrl[0]) { ctrl[2] = 1; ctrl[3] += 1; }
    else {
        ctrl[2] = 0;
        const int step = ++ctrl[1];
        reinterpret_cast<float*>(ctrl)[4] = 1.0f - powf(b1, (float)step);
        reinterpret_cast<float*>(ctrl)[5] = 1.0f - powf(b2, (float)step);
    }
    ctrl[0] = 0;
}
__global__ __launch_bounds__(256) void adam_guarded_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                           float unscale, const int* __restrict__ ctrl) {
    if (ctrl[2]) return;                                   // uniform over the grid
    const float bc1 = reinterpret_cast<const float*>(ctrl)[4], bc2 = reinterpret_cast<const float*>(ctrl)[5];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * unscale;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}
void launch_grad_check(const float* g, long long n, int* ctrl, hipStream_t stream) {
    const unsigned blocks = (unsigned)((n / 4 + 255) / 256 < 2048 ? ((n / 4 + 255) / 256 > 0 ? (n / 4 + 255) / 256 : 1) : 2048);
    hipLaunchKernelGGL(grad_check_kernel, dim3(blocks), dim3(256), 0, stream, g, n, ctrl);
}
void launch_guard_commit(int* ctrl, float b1, float b2, hipStream_t stream) {
    hipLaunchKernelGGL(guard_commit_kernel, dim3(1), dim3(64), 0, stream, ctrl, b1, b2);
}
void launch_adam_guarded(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                         float grad_unscale, const int* ctrl, hipStream_t stream) {
    hipLaunchKernelGGL(adam_guarded_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, g, m, v, n, lr, b1, b2, eps, grad_unscale, ctrl);
}

}  // namespace cddpm
