// Training-mode kernels of the context encoder (SURVEY.md section 8 rows f2 + f4): timm's ResNet-50 (in_chans = 1) as the reference
// builds it (src/models/modules/DDPM_encoder.py:21-23; spark/models.py:89-109) and trains it jointly with the UNet
// (src/models/DDPM_2D.py:114-135: features = self(input) carries a gradient, :305-306 Adam over self.parameters()).
//
// Everything NHWC fp32, plain FMA tiles: the encoder is ~5 % of a training step's FLOPs (0.45 GFLOP per 128 x 128 slice forward) and
// its layers are small (64 ... 2048 channels over 64 x 64 ... 4 x 4 pixels), so these kernels are written for correctness and
// coalescing, not for the matrix pipe. Pieces:
//   enc_gemm_conv_kernel   one 64 x 64-tile direct convolution in two roles: forward (rows = output pixels, gathered input rows
//                          yo * stride + ky - pad) and input gradient (rows = INPUT pixels, gathered dz rows (yi + pad - ky) / stride
//                          where that division is exact): K in {1, 3}, stride in {1, 2}, weights as [taps][K-dim][N-dim] images
//   enc_wgrad_kernel       dW: contraction over output pixels, 64 ci x 64 co per workgroup and tap, P pixel ranges, fixed-order fold
//   enc_stem_*             the 7 x 7 / 2 single-channel stem: forward and dW (its input gradient is never needed)
//   enc_chan_partial / fold   per-channel reductions in fp64: BatchNorm batch statistics (+ running-statistics update, momentum 0.1,
//                          unbiased running variance as torch) and the two sums of its backward
//   enc_bn_act / enc_bn_bwd_apply   y = relu(((z - mean) rstd gamma + beta) s[b] + res) and its backward (s[b]: stochastic-depth scale
//                          of the residual branch per sample, 1 when unused)
//   max-pool 3x3/2 backward (gather form: deterministic), global average pool forward / backward, weight image packing
#include "kernels.h"
#include <hip/hip_runtime.h>

namespace cddpm {

// ---------------------------------------------------------------------------------------------------------------- convolution
struct EncGemmConv {
    const float* src;      // forward: x [B,H,W,Kdim]; input gradient: dz [B,Ho,Wo,Kdim]
    const float* w;        // [taps][Kdim][Ndim]
    float* dst;            // forward: z [B,Ho,Wo,Ndim]; input gradient: dx [B,H,W,Ndim]
    int B, H, W, Ho, Wo;   // H, W: the convolution's input size; Ho, Wo: its output size
    int Kdim, Ndim, K, stride, transposed;
    int Z;                 // contraction split: blockIdx.z multiplies steps [nstep z / Z, nstep (z + 1) / Z) into plane z of `part`
    float* part;           // [Z][rows][Ndim] when Z > 1 (folded in the order of z by enc_fold_planes_kernel)
};
__global__ __launch_bounds__(256) void enc_gemm_conv_kernel(const EncGemmConv a) {
    __shared__ float As[16][64 + 4];
    __shared__ float Bs[16][64 + 4];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int Hr = a.transposed ? a.H : a.Ho, Wr = a.transposed ? a.W : a.Wo;       // geometry of the rows this launch produces
    const int Hs = a.transposed ? a.Ho : a.H, Ws = a.transposed ? a.Wo : a.W;       // ... and of the tensor it gathers from
    const long long M = (long long)a.B * Hr * Wr;
    const long long p0 = (long long)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    const int lp = tid >> 2, lc4 = tid & 3;
    const long long pl = p0 + lp;
    int lb = 0, ly = 0, lx = 0;
    const bool lvalid = pl < M;
    if (lvalid) { long long t = pl; lx = (int)(t % Wr); t /= Wr; ly = (int)(t % Hr); lb = (int)(t / Hr); }
    const int bk = tid >> 4, bc4 = tid & 15;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int pad = a.K / 2, taps = a.K * a.K;
    const int nck = a.Kdim / 16, nstep = taps * nck;
    // one step = 16 contraction channels of one tap; the next step's two global loads are in flight while this step multiplies
    auto fetch = [&](int step, float4& av, float4& bv) {
        const int t = step / nck, c0 = (step - t * nck) * 16;
        const int ky = t / a.K, kx = t - ky * a.K;
        int ys, xs;
        bool inb = lvalid;
        if (!a.transposed) { ys = ly * a.stride + ky - pad; xs = lx * a.stride + kx - pad; }
        else {
            const int ny = ly + pad - ky, nx = lx + pad - kx;
            inb = inb && ny >= 0 && nx >= 0 && (ny % a.stride) == 0 && (nx % a.stride) == 0;
            ys = ny / a.stride; xs = nx / a.stride;
        }
        inb = inb && ys >= 0 && ys < Hs && xs >= 0 && xs < Ws;
        av = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inb) av = *reinterpret_cast<const float4*>(a.src + (((size_t)lb * Hs + ys) * Ws + xs) * a.Kdim + c0 + 4 * lc4);
        bv = *reinterpret_cast<const float4*>(a.w + ((size_t)t * a.Kdim + c0 + bk) * a.Ndim + n0 + 4 * bc4);
    };
    const int s0 = (int)((long long)nstep * blockIdx.z / a.Z), s1 = (int)((long long)nstep * (blockIdx.z + 1) / a.Z);
    float4 av, bv;
    fetch(s0, av, bv);
    for (int step = s0; step < s1; ++step) {
        __syncthreads();
        As[4 * lc4 + 0][lp] = av.x; As[4 * lc4 + 1][lp] = av.y; As[4 * lc4 + 2][lp] = av.z; As[4 * lc4 + 3][lp] = av.w;
        *reinterpret_cast<float4*>(&Bs[bk][4 * bc4]) = bv;
        __syncthreads();
        if (step + 1 < s1) fetch(step + 1, av, bv);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 a4 = *reinterpret_cast<const float4*>(&As[k][4 * ty]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Bs[k][4 * tx]);
            const float aa[4] = {a4.x, a4.y, a4.z, a4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(aa[i], bb[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long p = p0 + 4 * ty + i;
        if (p >= M) continue;
        float* o = a.Z > 1 ? a.part + (size_t)blockIdx.z * M * a.Ndim : a.dst;
        *reinterpret_cast<float4*>(o + (size_t)p * a.Ndim + n0 + 4 * tx) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
}
__global__ __launch_bounds__(256) void enc_fold_planes_kernel(const float* __restrict__ part, int Z, long long n4, float* __restrict__ dst) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n4) return;
    float4 s = reinterpret_cast<const float4*>(part)[e];
    for (int z = 1; z < Z; ++z) {
        const float4 v = reinterpret_cast<const float4*>(part)[(size_t)z * n4 + e];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(dst)[e] = s;
}
// layers that give the chip fewer than 128 workgroups (the 8 x 8 and 4 x 4 stages at batch 16) split their contraction
int enc_conv_split(int B, int H, int W, int Cin, int Cout, int K, int stride, int transposed) {
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const long long M = (long long)B * (transposed ? H * W : Ho * Wo);
    const long long wgs = ((M + 63) / 64) * ((transposed ? Cin : Cout) / 64);
    const int nstep = K * K * (transposed ? Cout : Cin) / 16;
    int Z = 1;
    while (Z < 8 && wgs * Z * 2 <= 256 && nstep / (Z * 2) >= 8) Z *= 2;
    return Z;
}
void launch_enc_conv(const float* src, const float* w, float* dst, int B, int H, int W, int Cin, int Cout, int K, int stride, int transposed,
                     float* part, hipStream_t s) {
    EncGemmConv a;
    a.Z = part ? enc_conv_split(B, H, W, Cin, Cout, K, stride, transposed) : 1;
    a.part = part;
    a.src = src; a.w = w; a.dst = dst; a.B = B; a.H = H; a.W = W; a.Ho = (H + stride - 1) / stride; a.Wo = (W + stride - 1) / stride;
    a.Kdim = transposed ? Cout : Cin; a.Ndim = transposed ? Cin : Cout; a.K = K; a.stride = stride; a.transposed = transposed;
    const long long M = (long long)B * (transposed ? H * W : a.Ho * a.Wo);
    hipLaunchKernelGGL(enc_gemm_conv_kernel, dim3((unsigned)((M + 63) / 64), (unsigned)(a.Ndim / 64), a.Z), dim3(256), 0, s, a);
    if (a.Z > 1) {
        const long long n4 = M * a.Ndim / 4;
        hipLaunchKernelGGL(enc_fold_planes_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, part, a.Z, n4, dst);
    }
}

// weight images of one convolution from the PyTorch tensor w [Cout][Cin][K][K]: wf [taps][Cin][Cout] (forward), wd [taps][Cout][Cin]
__global__ __launch_bounds__(256) void enc_pack_w_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, float* __restrict__ wf,
                                                         float* __restrict__ wd) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)Cout * Cin * taps) return;
    const int t = (int)(e % taps), ci = (int)((e / taps) % Cin), co = (int)(e / ((long long)taps * Cin));
    const float v = w[e];
    wf[((size_t)t * Cin + ci) * Cout + co] = v;
    if (wd) wd[((size_t)t * Cout + co) * Cin + ci] = v;
}
void launch_enc_pack_w(const float* w, int Cout, int Cin, int taps, float* wf, float* wd, hipStream_t s) {
    const long long n = (long long)Cout * Cin * taps;
    hipLaunchKernelGGL(enc_pack_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, Cout, Cin, taps, wf, wd);
}

// dW [Cout][Cin][K][K]: workgroup = (64 ci, 64 co, tap, pixel range); part [P][taps][Cin][Cout]
struct EncWgrad { const float* x; const float* dz; float* part; int B, H, W, Ho, Wo, Cin, Cout, K, stride, P; };
__global__ __launch_bounds__(256) void enc_wgrad_kernel(const EncWgrad a) {
    __shared__ float As[16][64 + 4];     // [pixel][ci]
    __shared__ float Bs[16][64 + 4];     // [pixel][co]
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int taps = a.K * a.K, pad = a.K / 2;
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
    const int t = blockIdx.z % taps, part = blockIdx.z / taps;
    const int ky = t / a.K, kx = t - ky * a.K;
    const long long M = (long long)a.B * a.Ho * a.Wo;
    const long long m0 = M * part / a.P, m1 = M * (part + 1) / a.P;
    const int lp = tid >> 4, lc4 = tid & 15;          // loader: pixel lp of the 16-pixel step, 4 channels
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (long long q0 = m0; q0 < m1; q0 += 16) {
        const long long q = q0 + lp;
        float4 av = make_float4(0.f, 0.f, 0.f, 0.f), bv = av;
        if (q < m1) {
            long long r = q;
            const int xo = (int)(r % a.Wo); r /= a.Wo;
            const int yo = (int)(r % a.Ho);
            const int b = (int)(r / a.Ho);
            const int yi = yo * a.stride + ky - pad, xi = xo * a.stride + kx - pad;
            if (yi >= 0 && yi < a.H && xi >= 0 && xi < a.W)
                av = *reinterpret_cast<const float4*>(a.x + (((size_t)b * a.H + yi) * a.W + xi) * a.Cin + ci0 + 4 * lc4);
            bv = *reinterpret_cast<const float4*>(a.dz + (size_t)q * a.Cout + co0 + 4 * lc4);
        }
        __syncthreads();
        *reinterpret_cast<float4*>(&As[lp][4 * lc4]) = av;
        *reinterpret_cast<float4*>(&Bs[lp][4 * lc4]) = bv;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 a4 = *reinterpret_cast<const float4*>(&As[k][4 * ty]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Bs[k][4 * tx]);
            const float aa[4] = {a4.x, a4.y, a4.z, a4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(aa[i], bb[j], acc[i][j]);
        }
    }
    float* o = a.part + (((size_t)part * taps + t) * a.Cin + ci0) * a.Cout + co0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(o + (size_t)(4 * ty + i) * a.Cout + 4 * tx) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
}
__global__ __launch_bounds__(256) void enc_wgrad_fold_kernel(const float* __restrict__ part, int P, int Cout, int Cin, int taps, float* __restrict__ dw) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)Cout * Cin * taps) return;
    const int t = (int)(e % taps), ci = (int)((e / taps) % Cin), co = (int)(e / ((long long)taps * Cin));
    const size_t plane = (size_t)taps * Cin * Cout, idx = ((size_t)t * Cin + ci) * Cout + co;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += part[(size_t)p * plane + idx];
    dw[e] = s;
}
int enc_wgrad_parts(int B, int Ho, int Wo, int Cin, int Cout, int K) {
    const long long M = (long long)B * Ho * Wo;
    const int per = (Cin / 64) * (Cout / 64) * K * K;
    long long P = (1024 + per - 1) / per;
    if (P > M / 64) P = M / 64;
    if (P > 16) P = 16;
    return P < 1 ? 1 : (int)P;
}
void launch_enc_wgrad(const float* x, const float* dz, float* part, int P, float* dw, int B, int H, int W, int Cin, int Cout, int K, int stride,
                      hipStream_t s) {
    EncWgrad a;
    a.x = x; a.dz = dz; a.part = part; a.B = B; a.H = H; a.W = W; a.Ho = (H + stride - 1) / stride; a.Wo = (W + stride - 1) / stride;
    a.Cin = Cin; a.Cout = Cout; a.K = K; a.stride = stride; a.P = P;
    hipLaunchKernelGGL(enc_wgrad_kernel, dim3(Cin / 64, Cout / 64, K * K * P), dim3(256), 0, s, a);
    const long long n = (long long)Cout * Cin * K * K;
    hipLaunchKernelGGL(enc_wgrad_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, P, Cout, Cin, K * K, dw);
}

// ---------------------------------------------------------------------------------------------------------------- stem 7x7 / 2, 1 -> 64
__global__ __launch_bounds__(256) void enc_stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w /*[64][49]*/,
                                                           float* __restrict__ z, int B, int H, int W, int Ho, int Wo) {
    __shared__ float ws[49][64];
    for (int i = threadIdx.x; i < 49 * 64; i += 256) ws[i % 49][i / 49] = w[i];
    __syncthreads();
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;       // one thread per (pixel, 4 channels)
    if (e >= (long long)B * Ho * Wo * 16) return;
    const int c4 = (int)(e & 15);
    long long p = e >> 4;
    const int xo = (int)(p % Wo); p /= Wo;
    const int yo = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < 7; ++ky) {
        const int yi = 2 * yo + ky - 3;
        if (yi < 0 || yi >= H) continue;
        for (int kx = 0; kx < 7; ++kx) {
            const int xi = 2 * xo + kx - 3;
            if (xi < 0 || xi >= W) continue;
            const float v = x[((size_t)b * H + yi) * W + xi];
            const float4 wv = *reinterpret_cast<const float4*>(&ws[ky * 7 + kx][4 * c4]);
            acc[0] = fmaf(v, wv.x, acc[0]); acc[1] = fmaf(v, wv.y, acc[1]);
            acc[2] = fmaf(v, wv.z, acc[2]); acc[3] = fmaf(v, wv.w, acc[3]);
        }
    }
    *reinterpret_cast<float4*>(z + (((size_t)b * Ho + yo) * Wo + xo) * 64 + 4 * c4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}
// dW[co][tap] = sum_p x[b, 2 yo + ky - 3, 2 xo + kx - 3] dz[p][co]: workgroup = (tap, pixel range), thread = (co, pixel lane of 4)
__global__ __launch_bounds__(256) void enc_stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz, double* __restrict__ part,
                                                             int B, int H, int W, int Ho, int Wo, int P) {
    __shared__ double red[4][64];
    const int t = blockIdx.x, pr = blockIdx.y, ky = t / 7, kx = t - 7 * ky;
    const int co = threadIdx.x & 63, ln = threadIdx.x >> 6;
    const long long M = (long long)B * Ho * Wo, m0 = M * pr / P, m1 = M * (pr + 1) / P;
    double s = 0.0;
    for (long long q = m0 + ln; q < m1; q += 4) {
        long long r = q;
        const int xo = (int)(r % Wo); r /= Wo;
        const int yo = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const int yi = 2 * yo + ky - 3, xi = 2 * xo + kx - 3;
        if (yi < 0 || yi >= H || xi < 0 || xi >= W) continue;
        s += (double)x[((size_t)b * H + yi) * W + xi] * (double)dz[(size_t)q * 64 + co];
    }
    red[ln][co] = s;
    __syncthreads();
    if (ln == 0) part[((size_t)pr * 49 + t) * 64 + co] = red[0][co] + red[1][co] + red[2][co] + red[3][co];
}
__global__ void enc_stem_wgrad_fold_kernel(const double* __restrict__ part, int P, float* __restrict__ dw /*[64][49]*/) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 49 * 64) return;
    const int t = e / 64, co = e % 64;
    double s = 0.0;
    for (int p = 0; p < P; ++p) s += part[((size_t)p * 49 + t) * 64 + co];
    dw[co * 49 + t] = (float)s;
}
void launch_enc_stem_fwd(const float* x, const float* w, float* z, int B, int H, int W, hipStream_t s) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long total = (long long)B * Ho * Wo * 16;
    hipLaunchKernelGGL(enc_stem_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, z, B, H, W, Ho, Wo);
}
void launch_enc_stem_wgrad(const float* x, const float* dz, double* part /* 32 * 49 * 64 */, float* dw, int B, int H, int W, hipStream_t s) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, P = 32;
    hipLaunchKernelGGL(enc_stem_wgrad_kernel, dim3(49, P), dim3(256), 0, s, x, dz, part, B, H, W, Ho, Wo, P);
    hipLaunchKernelGGL(enc_stem_wgrad_fold_kernel, dim3((49 * 64 + 255) / 256), dim3(256), 0, s, part, P, dw);
}

// ---------------------------------------------------------------------------------------------------------------- BatchNorm (+ ReLU, shortcut)
// per-channel pair of sums over the N = B * HW pixels, in fp64: workgroup = (pixel chunk, 64 channels), thread = (channel quad, pixel lane)
//   MODE 0: (sum z, sum z^2)      MODE 1: g = dy [y > 0 if relu] s[b]; (sum g, sum g zhat), zhat = (z - mean) rstd
template <int MODE>
__global__ __launch_bounds__(256) void enc_chan_partial_kernel(const float* __restrict__ z, const float* __restrict__ y, const float* __restrict__ dy,
                                                               const float* __restrict__ mr, const float* __restrict__ sscale, int relu,
                                                               long long N, int HW, int C, int nchunk, double* __restrict__ part) {
    __shared__ double red[2][16][64 + 1];
    const int tid = threadIdx.x, q = tid & 15, r = tid >> 4, c = blockIdx.y * 64 + 4 * q;
    const long long per = (N + nchunk - 1) / nchunk, p0 = per * blockIdx.x, p1 = p0 + per < N ? p0 + per : N;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    float mean[4] = {0, 0, 0, 0}, rstd[4] = {1, 1, 1, 1};
    if (MODE == 1)
        for (int i = 0; i < 4; ++i) { mean[i] = mr[c + i]; rstd[i] = mr[C + c + i]; }
    for (long long p = p0 + r; p < p1; p += 16) {
        const float4 zv = *reinterpret_cast<const float4*>(z + (size_t)p * C + c);
        const float zz[4] = {zv.x, zv.y, zv.z, zv.w};
        if (MODE == 0) {
            for (int i = 0; i < 4; ++i) { s0[i] += zz[i]; s1[i] += (double)zz[i] * zz[i]; }
        } else {
            const float4 dv = *reinterpret_cast<const float4*>(dy + (size_t)p * C + c);
            float gg[4] = {dv.x, dv.y, dv.z, dv.w};
            if (relu) {
                const float4 yv = *reinterpret_cast<const float4*>(y + (size_t)p * C + c);
                const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
                for (int i = 0; i < 4; ++i) gg[i] = yy[i] > 0.f ? gg[i] : 0.f;
            }
            const float sc = sscale ? sscale[p / HW] : 1.f;
            for (int i = 0; i < 4; ++i) {
                const float g = gg[i] * sc;
                s0[i] += g;
                s1[i] += (double)g * ((zz[i] - mean[i]) * rstd[i]);
            }
        }
    }
    for (int i = 0; i < 4; ++i) { red[0][r][4 * q + i] = s0[i]; red[1][r][4 * q + i] = s1[i]; }
    __syncthreads();
    if (tid < 128) {
        const int w = tid >> 6, cc = tid & 63;
        double t = 0;
        for (int k = 0; k < 16; ++k) t += red[w][k][cc];
        part[((size_t)blockIdx.x * 2 + w) * C + blockIdx.y * 64 + cc] = t;
    }
}
// batch statistics: mr = (mean | rstd); running statistics as torch.nn.BatchNorm2d in training mode (momentum, unbiased variance)
__global__ __launch_bounds__(256) void enc_bn_stats_fold_kernel(const double* __restrict__ part, int nchunk, int C, long long N, float eps, float momentum,
                                                                float* __restrict__ mr, float* __restrict__ run_mean, float* __restrict__ run_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s0 = 0, s1 = 0;
    for (int k = 0; k < nchunk; ++k) { s0 += part[((size_t)k * 2) * C + c]; s1 += part[((size_t)k * 2 + 1) * C + c]; }
    const double mean = s0 / (double)N;
    double var = s1 / (double)N - mean * mean;
    if (var < 0) var = 0;
    mr[c] = (float)mean;
    mr[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
        const double unb = N > 1 ? var * (double)N / (double)(N - 1) : var;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
}
// backward sums -> dgamma, dbeta and the two per-channel means the elementwise pass needs (k = (sum g / N | sum g zhat / N))
__global__ __launch_bounds__(256) void enc_bn_bwd_fold_kernel(const double* __restrict__ part, int nchunk, int C, long long N, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ k) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s0 = 0, s1 = 0;
    for (int j = 0; j < nchunk; ++j) { s0 += part[((size_t)j * 2) * C + c]; s1 += part[((size_t)j * 2 + 1) * C + c]; }
    dbeta[c] = (float)s0;
    dgamma[c] = (float)s1;
    k[c] = (float)(s0 / (double)N);
    k[C + c] = (float)(s1 / (double)N);
}
__global__ __launch_bounds__(256) void enc_bn_act_kernel(const float* __restrict__ z, const float* __restrict__ mr, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ sscale, const float* __restrict__ res,
                                                         int relu, float* __restrict__ y, long long N, int HW, int C) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nq = C >> 2;
    if (e >= N * nq) return;
    const int c = 4 * (int)(e % nq);
    const long long p = e / nq;
    const float4 zv = *reinterpret_cast<const float4*>(z + (size_t)p * C + c);
    const float4 m = *reinterpret_cast<const float4*>(mr + c), rs = *reinterpret_cast<const float4*>(mr + C + c);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
    const float sc = sscale ? sscale[p / HW] : 1.f;
    float4 o;
    o.x = ((zv.x - m.x) * rs.x * g.x + b.x) * sc; o.y = ((zv.y - m.y) * rs.y * g.y + b.y) * sc;
    o.z = ((zv.z - m.z) * rs.z * g.z + b.z) * sc; o.w = ((zv.w - m.w) * rs.w * g.w + b.w) * sc;
    if (res) {
        const float4 rv = *reinterpret_cast<const float4*>(res + (size_t)p * C + c);
        o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
    }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *reinterpret_cast<float4*>(y + (size_t)p * C + c) = o;
}
// dz = gamma rstd (g - k0 - zhat k1), g = dy [y > 0] s[b]; dres (optional) = dy [y > 0]: the gradient of the shortcut operand
__global__ __launch_bounds__(256) void enc_bn_bwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ y, const float* __restrict__ dy,
                                                               const float* __restrict__ mr, const float* __restrict__ gamma, const float* __restrict__ k,
                                                               const float* __restrict__ sscale, int relu, float* __restrict__ dz,
                                                               float* __restrict__ dres, long long N, int HW, int C) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nq = C >> 2;
    if (e >= N * nq) return;
    const int c = 4 * (int)(e % nq);
    const long long p = e / nq;
    const float4 zv = *reinterpret_cast<const float4*>(z + (size_t)p * C + c);
    float4 dv = *reinterpret_cast<const float4*>(dy + (size_t)p * C + c);
    if (relu) {
        const float4 yv = *reinterpret_cast<const float4*>(y + (size_t)p * C + c);
        dv.x = yv.x > 0.f ? dv.x : 0.f; dv.y = yv.y > 0.f ? dv.y : 0.f; dv.z = yv.z > 0.f ? dv.z : 0.f; dv.w = yv.w > 0.f ? dv.w : 0.f;
    }
    if (dres) *reinterpret_cast<float4*>(dres + (size_t)p * C + c) = dv;
    const float sc = sscale ? sscale[p / HW] : 1.f;
    const float4 m = *reinterpret_cast<const float4*>(mr + c), rs = *reinterpret_cast<const float4*>(mr + C + c);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    const float4 k0 = *reinterpret_cast<const float4*>(k + c), k1 = *reinterpret_cast<const float4*>(k + C + c);
    float4 o;
    o.x = g.x * rs.x * (dv.x * sc - k0.x - (zv.x - m.x) * rs.x * k1.x);
    o.y = g.y * rs.y * (dv.y * sc - k0.y - (zv.y - m.y) * rs.y * k1.y);
    o.z = g.z * rs.z * (dv.z * sc - k0.z - (zv.z - m.z) * rs.z * k1.z);
    o.w = g.w * rs.w * (dv.w * sc - k0.w - (zv.w - m.w) * rs.w * k1.w);
    *reinterpret_cast<float4*>(dz + (size_t)p * C + c) = o;
}
int enc_bn_chunks(long long N) {
    long long n = N / 256;            // >= 16 pixels per pixel lane of a chunk; the folds read the chunks serially
    if (n > 32) n = 32;
    return n < 1 ? 1 : (int)n;
}
void launch_enc_bn_forward(const float* z, const float* gamma, const float* beta, const float* sscale, const float* res, int relu, float eps,
                           float momentum, float* run_mean, float* run_var, float* mr, float* y, double* part, long long N, int HW, int C,
                           hipStream_t s) {
    const int nchunk = enc_bn_chunks(N);
    hipLaunchKernelGGL(enc_chan_partial_kernel<0>, dim3(nchunk, C / 64), dim3(256), 0, s, z, nullptr, nullptr, nullptr, nullptr, 0, N, HW, C, nchunk, part);
    hipLaunchKernelGGL(enc_bn_stats_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, s, part, nchunk, C, N, eps, momentum, mr, run_mean, run_var);
    const long long total = N * (C / 4);
    hipLaunchKernelGGL(enc_bn_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, mr, gamma, beta, sscale, res, relu, y, N, HW, C);
}
void launch_enc_bn_backward(const float* z, const float* y, const float* dy, const float* mr, const float* gamma, const float* sscale, int relu,
                            float* dz, float* dres, float* dgamma, float* dbeta, float* k /* 2 C */, double* part, long long N, int HW, int C,
                            hipStream_t s) {
    const int nchunk = enc_bn_chunks(N);
    hipLaunchKernelGGL(enc_chan_partial_kernel<1>, dim3(nchunk, C / 64), dim3(256), 0, s, z, y, dy, mr, sscale, relu, N, HW, C, nchunk, part);
    hipLaunchKernelGGL(enc_bn_bwd_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, s, part, nchunk, C, N, dgamma, dbeta, k);
    const long long total = N * (C / 4);
    hipLaunchKernelGGL(enc_bn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, y, dy, mr, gamma, k, sscale, relu, dz, dres,
                       N, HW, C);
}

// ---------------------------------------------------------------------------------------------------------------- pooling
__global__ __launch_bounds__(256) void enc_maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int Ho,
                                                              int Wo) {
    const int nq = C >> 2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * Ho * Wo * nq) return;
    const int c4 = (int)(e % nq);
    long long p = e / nq;
    const int xo = (int)(p % Wo); p /= Wo;
    const int yo = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int ky = 0; ky < 3; ++ky) {
        const int yi = 2 * yo + ky - 1;
        if (yi < 0 || yi >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int xi = 2 * xo + kx - 1;
            if (xi < 0 || xi >= W) continue;
            const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)b * H + yi) * W + xi) * C + 4 * c4);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    *reinterpret_cast<float4*>(y + (((size_t)b * Ho + yo) * Wo + xo) * C + 4 * c4) = m;
}
// gather form of the max-pool backward: an input element receives dy of every window whose FIRST maximum (row-major scan, as torch) it is
__global__ __launch_bounds__(256) void enc_maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int B,
                                                              int H, int W, int C, int Ho, int Wo) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * H * W * C) return;
    const int c = (int)(e % C);
    long long p = e / C;
    const int xi = (int)(p % W); p /= W;
    const int yi = (int)(p % H);
    const int b = (int)(p / H);
    const float v = x[e];
    float s = 0.f;
    for (int yo = (yi + 1 - 2 + 1) / 2; yo <= (yi + 1) / 2; ++yo) {           // windows with 2 yo - 1 <= yi <= 2 yo + 1
        if (yo < 0 || yo >= Ho) continue;
        for (int xo = (xi + 1 - 2 + 1) / 2; xo <= (xi + 1) / 2; ++xo) {
            if (xo < 0 || xo >= Wo) continue;
            bool first = true;                                                  // is (yi, xi) the first maximum of window (yo, xo)?
            for (int ky = 0; ky < 3 && first; ++ky) {
                const int y2 = 2 * yo + ky - 1;
                if (y2 < 0 || y2 >= H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int x2 = 2 * xo + kx - 1;
                    if (x2 < 0 || x2 >= W) continue;
                    const float u = x[(((size_t)b * H + y2) * W + x2) * C + c];
                    const bool before = (y2 < yi) || (y2 == yi && x2 < xi);
                    if (u > v || (before && u == v)) { first = false; break; }
                }
            }
            if (first) s += dy[(((size_t)b * Ho + yo) * Wo + xo) * C + c];
        }
    }
    dx[e] = s;
}
__global__ __launch_bounds__(256) void enc_avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ g, int B, int HW, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += x[((size_t)b * HW + p) * C + c];
    g[i] = s / (float)HW;
}
__global__ __launch_bounds__(256) void enc_avgpool_bwd_kernel(const float* __restrict__ dg, float* __restrict__ dx, int B, int HW, int C) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * HW * C) return;
    const int c = (int)(e % C), b = (int)(e / ((long long)HW * C));
    dx[e] = dg[(size_t)b * C + c] / (float)HW;
}
void launch_enc_maxpool(const float* x, float* y, int B, int H, int W, int C, hipStream_t s) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long total = (long long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(enc_maxpool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, B, H, W, C, Ho, Wo);
}
void launch_enc_maxpool_backward(const float* x, const float* dy, float* dx, int B, int H, int W, int C, hipStream_t s) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long total = (long long)B * H * W * C;
    hipLaunchKernelGGL(enc_maxpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, dy, dx, B, H, W, C, Ho, Wo);
}
void launch_enc_avgpool(const float* x, float* g, int B, int HW, int C, hipStream_t s) {
    hipLaunchKernelGGL(enc_avgpool_fwd_kernel, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, s, x, g, B, HW, C);
}
void launch_enc_avgpool_backward(const float* dg, float* dx, int B, int HW, int C, hipStream_t s) {
    const long long total = (long long)B * HW * C;
    hipLaunchKernelGGL(enc_avgpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dg, dx, B, HW, C);
}

}  // namespace cddpm
