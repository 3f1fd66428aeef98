// Memory-bound kernels of the cDDPM reverse path for gfx950: the two single-channel convolutions at the
// ends of the UNet, the down-ResBlock pooling front end, the embedding linears, the posterior step with
// its counter RNG. All NHWC fp32, 16-B accesses per lane, wave64.
#include "kernels.h"

namespace cddpm {

__device__ __forceinline__ float silu_s(float v) {
    // same evaluation as conv_mfma.hip::silu_f (split-product exp2, ~1.5 ulp)
    // v * sigmoid(v). exp(-v) = 2^t with t = -v log2(e) carried as (t, tl): the rounding error of the product is
    // recovered with two fmas and applied as a first-order correction, so the result is good to ~1.5 ulp on the
    // v_exp_f32 / v_rcp_f32 pair at 9 VALU ops (ocml expf: ~20). t is clamped so 2^t stays finite (no inf * 0).
    const float t = fminf(-v * 1.44269502162933349609375f, 126.0f);
    float tl = __builtin_fmaf(-v, 1.44269502162933349609375f, -t);
    tl = __builtin_fmaf(-v, 1.925963033500011e-08f, tl);
    tl = (t < 126.0f) ? tl : 0.0f;
    float e = __builtin_amdgcn_exp2f(t);
    e = __builtin_fmaf(e, tl * 0.693147180559945f, e);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// ------------------------------------------------------------------------------------------------
// input_blocks.0: Conv2d(1 -> C, 3x3, pad 1)  (src/models/modules/OpenAI_Unet.py:606-612)
// thread = (pixel lane, channel quad); the 36 weights of the quad stay in registers.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_in1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int B, int H, int W, int C) {
    const int ncq = C >> 2, npl = 256 / ncq;
    const int tid = threadIdx.x;
    const int cq = tid % ncq, pl = tid / ncq;
    float wr[4][9], br[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        br[i] = bias[4 * cq + i];
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[i][t] = w[(4 * cq + i) * 9 + t];
    }
    const unsigned total = (unsigned)B * H * W;     // < 2^31 (checked by the launcher): 32-bit divisions, not 64-bit ones
    const unsigned ppb = 16u * npl;   // pixels per block: 16 per thread row (64 made a small batch wait ~45 us on one block's serial loads)
    const unsigned pend = min(total, (blockIdx.x + 1u) * ppb);
    for (unsigned p = blockIdx.x * ppb + pl; p < pend; p += npl) {
        const unsigned row = p / (unsigned)W;
        const int xx = (int)(p - row * W);
        const unsigned bb = row / (unsigned)H;
        const int yy = (int)(row - bb * H);
        const float* img = x + (size_t)bb * H * W;
        float v[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int y2 = yy + ky - 1, x2 = xx + kx - 1;
                v[ky * 3 + kx] = (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) ? img[y2 * W + x2] : 0.f;
            }
        float4 o;
        float* op = &o.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float acc = br[i];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = fmaf(v[t], wr[i][t], acc);
            op[i] = acc;
        }
        *reinterpret_cast<float4*>(out + (size_t)p * C + 4 * cq) = o;
    }
}

void launch_conv_in1(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int C,
                     hipStream_t stream) {
    const int npl = 256 / (C / 4);
    const long long total = (long long)B * H * W;
    const int ppb = 16 * npl;
    hipLaunchKernelGGL(conv_in1_kernel, dim3((unsigned)((total + ppb - 1) / ppb)), dim3(256), 0, stream, x, w, bias, out,
                       B, H, W, C);
}

// ------------------------------------------------------------------------------------------------
// output head: GroupNorm -> SiLU -> Conv2d(C -> 1, 3x3, pad 1)  (OpenAI_Unet.py:793-797, :991)
// split as: P[pixel][tap] = sum_c w[tap][c] * act(x[pixel][c])   (a C -> 9 pointwise map, x read once)
//           out[y][x]     = bias + sum_tap P[(y+ky-1, x+kx-1)][tap]  (zero padding = skipped neighbours)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_dots_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                        const float* __restrict__ w9, float* __restrict__ P, int B,
                                                        int HW, int C) {
    extern __shared__ float hl[];
    float* lx = hl;                    // [64][C+4]: 16-B aligned rows, 64 lanes x b128 conflict-free (row stride = 4 banks mod 64)
    float* lw = hl + 64 * (C + 4);     // [9][C]
    const int tid = threadIdx.x;
    const long long total = (long long)B * HW;
    const long long pix0 = (long long)blockIdx.x * 64;
    const int ncq = C >> 2;
    const size_t plane = (size_t)B * C;
    for (int i = tid; i < 9 * C; i += 256) lw[i] = w9[i];
    for (int e = tid; e < 64 * ncq; e += 256) {
        const int pl = e / ncq, cq = e - pl * ncq;
        const long long p = pix0 + pl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < total) {
            const int b = (int)(p / HW);
            v = *reinterpret_cast<const float4*>(x + p * C + 4 * cq);
            const size_t ci = (size_t)b * C + 4 * cq;
            const float4 m = *reinterpret_cast<const float4*>(coef + ci);
            const float4 a = *reinterpret_cast<const float4*>(coef + plane + ci);
            const float4 d = *reinterpret_cast<const float4*>(coef + 2 * plane + ci);
            v.x = silu_s((v.x - m.x) * a.x + d.x);
            v.y = silu_s((v.y - m.y) * a.y + d.y);
            v.z = silu_s((v.z - m.z) * a.z + d.z);
            v.w = silu_s((v.w - m.w) * a.w + d.w);
        }
        *reinterpret_cast<float4*>(lx + pl * (C + 4) + 4 * cq) = v;
    }
    __syncthreads();
    const int pl = tid & 63;
    const int tg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long p = pix0 + pl;
    for (int tap = tg; tap < 9; tap += 4) {
        float acc = 0.f;
        const float4* xr = reinterpret_cast<const float4*>(lx + pl * (C + 4));
        const float4* wr = reinterpret_cast<const float4*>(lw + tap * C);      // wave-uniform: broadcast reads
        for (int c = 0; c < (C >> 2); ++c) {       // same summation order as one fma per channel
            const float4 xv = xr[c], wv = wr[c];
            acc = fmaf(xv.x, wv.x, acc);
            acc = fmaf(xv.y, wv.y, acc);
            acc = fmaf(xv.z, wv.z, acc);
            acc = fmaf(xv.w, wv.w, acc);
        }
        if (p < total) P[p * 9 + tap] = acc;
    }
}

void launch_head_dots(const float* x, const float* coef, const float* w9, float* P, int B, int HW, int C,
                      hipStream_t stream) {
    const long long total = (long long)B * HW;
    const size_t lds = (size_t)(64 * (C + 4) + 9 * C) * sizeof(float);
    hipLaunchKernelGGL(head_dots_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), lds, stream, x, coef, w9, P, B,
                       HW, C);
}

__global__ __launch_bounds__(256) void head_gather_kernel(const float* __restrict__ P, float bias, const float* __restrict__ bias_ptr,
                                                          float* __restrict__ out, int B, int H, int W) {
    const long long total = (long long)B * H * W;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const int xx = (int)(p % W);
    const int yy = (int)((p / W) % H);
    float acc = bias_ptr ? bias_ptr[0] : bias;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int y2 = yy + ky - 1, x2 = xx + kx - 1;
            if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) acc += P[(p + (ky - 1) * W + (kx - 1)) * 9 + ky * 3 + kx];
        }
    out[p] = acc;
}

void launch_head_gather(const float* P, float bias, const float* bias_ptr, float* out, int B, int H, int W, hipStream_t stream) {
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(head_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, P, bias, bias_ptr, out, B,
                       H, W);
}

// ------------------------------------------------------------------------------------------------
// down ResBlock front end (OpenAI_Unet.py:287-293 with Downsample(use_conv=False) = AvgPool2d(2), :166-177):
//   hp = avgpool2(silu(groupnorm(x))),  xp = avgpool2(x)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_act_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                       float* __restrict__ hp, float* __restrict__ xp, int B, int H,
                                                       int W, int C) {
    const int ncq = C >> 2;
    const int Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)B * Ho * Wo * ncq;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int cq = (int)(e % ncq);
    const long long po = e / ncq;
    const int xo = (int)(po % Wo);
    const int yo = (int)((po / Wo) % Ho);
    const int b = (int)(po / ((long long)Wo * Ho));
    const size_t plane = (size_t)B * C;
    const size_t ci = (size_t)b * C + 4 * cq;
    const float4 m = *reinterpret_cast<const float4*>(coef + ci);
    const float4 a = *reinterpret_cast<const float4*>(coef + plane + ci);
    const float4 d = *reinterpret_cast<const float4*>(coef + 2 * plane + ci);
    float4 sh = make_float4(0.f, 0.f, 0.f, 0.f), sx = sh;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const size_t p = ((size_t)b * H + 2 * yo + dy) * W + 2 * xo + dx;
            const float4 v = *reinterpret_cast<const float4*>(x + p * C + 4 * cq);
            sx.x += v.x; sx.y += v.y; sx.z += v.z; sx.w += v.w;
            sh.x += silu_s((v.x - m.x) * a.x + d.x);
            sh.y += silu_s((v.y - m.y) * a.y + d.y);
            sh.z += silu_s((v.z - m.z) * a.z + d.z);
            sh.w += silu_s((v.w - m.w) * a.w + d.w);
        }
    sh.x *= 0.25f; sh.y *= 0.25f; sh.z *= 0.25f; sh.w *= 0.25f;
    sx.x *= 0.25f; sx.y *= 0.25f; sx.z *= 0.25f; sx.w *= 0.25f;
    *reinterpret_cast<float4*>(hp + (size_t)po * C + 4 * cq) = sh;
    *reinterpret_cast<float4*>(xp + (size_t)po * C + 4 * cq) = sx;
}

void launch_pool_act(const float* x, const float* coef, float* hp, float* xp, int B, int H, int W, int C,
                     hipStream_t stream) {
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(pool_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, coef, hp, xp, B,
                       H, W, C);
}

// ------------------------------------------------------------------------------------------------
// embedding linears (time_embed, label_emb, emb_layers: OpenAI_Unet.py:583-602, :201-207):
// one wave per output element, 16-B loads, xor-shuffle reduction. Off the per-step path (tables).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ Wt,
                                                     int ldw, int koff, const float* __restrict__ bias,
                                                     float* __restrict__ y, int ldy, int M, int N, int K, int silu_in) {
    const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wv >= (long long)M * N) return;
    const int m = (int)(wv / N), n = (int)(wv % N);
    const float* xr = x + (size_t)m * ldx;
    const float* wr = Wt + (size_t)n * ldw + koff;
    float acc = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
        float4 xv = *reinterpret_cast<const float4*>(xr + k);
        const float4 wv4 = *reinterpret_cast<const float4*>(wr + k);
        if (silu_in) { xv.x = silu_s(xv.x); xv.y = silu_s(xv.y); xv.z = silu_s(xv.z); xv.w = silu_s(xv.w); }
        acc = fmaf(xv.x, wv4.x, acc);
        acc = fmaf(xv.y, wv4.y, acc);
        acc = fmaf(xv.z, wv4.z, acc);
        acc = fmaf(xv.w, wv4.w, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) y[(size_t)m * ldy + n] = acc + (bias ? bias[n] : 0.f);
}

void launch_linear(const float* x, int ldx, const float* W, int ldw, int koff, const float* bias, float* y, int ldy,
                   int M, int N, int K, int silu_in, hipStream_t stream) {
    const long long waves = (long long)M * N;
    hipLaunchKernelGGL(linear_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, x, ldx, W, ldw, koff, bias,
                       y, ldy, M, N, K, silu_in);
}

__global__ void fill_int_kernel(int* p, int n, int v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void add_int_kernel(int* p, int n, int v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] += v;
}
// ---- split-K combine (small batches): out = ((plane 0 + plane 1) + ... ) + bias + residual, fixed order -> deterministic and
//      independent of the batch; optional GroupNorm statistics records of `out` (one per 64 consecutive pixels).
// One workgroup = 64 pixels x 64 channels (blockIdx.z = channel slice): thread = (channel quad cq of 16, pixel lane pl of 16), four
// pixels per thread -- a 32 x 32 layer of a 4-slice batch is 256 workgroups with 4 x ksplit loads in flight per thread (the first
// version gave a workgroup all channels of its 64 pixels: 64 workgroups, 16 x ksplit dependent loads per thread, 23 us per launch
// and 19 % of a B = 4 reverse step).
__global__ __launch_bounds__(256) void conv_reduce_kernel(const float* __restrict__ planes, int ksplit, const float* __restrict__ bias,
                                                          const float* __restrict__ res, int res_up, float* __restrict__ out,
                                                          float* __restrict__ stats, int H, int W, int Cout, size_t plane_elems) {
    __shared__ float4 red_s[256], red_q[256];
    const int tid = threadIdx.x, b = blockIdx.y, HW = H * W;
    const int p0 = blockIdx.x * 64;
    const int nrec = gridDim.x;
    const int cq = blockIdx.z * 16 + (tid & 15), pl = tid >> 4;      // Cout is a multiple of 64
    float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), ssq = ssum;
    const float4 bs = bias ? *reinterpret_cast<const float4*>(bias + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + pl + 16 * i;
        if (p >= HW) break;
        const size_t idx = ((size_t)b * HW + p) * Cout + 4 * cq;
        float4 v = *reinterpret_cast<const float4*>(planes + idx);
        for (int j = 1; j < ksplit; ++j) {
            const float4 w = *reinterpret_cast<const float4*>(planes + (size_t)j * plane_elems + idx);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        v.x += bs.x; v.y += bs.y; v.z += bs.z; v.w += bs.w;
        if (res) {
            const int y = p / W, x = p - y * W;
            const size_t rp = res_up ? ((size_t)(b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) : ((size_t)b * HW + p);
            const float4 r = *reinterpret_cast<const float4*>(res + rp * Cout + 4 * cq);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        *reinterpret_cast<float4*>(out + idx) = v;
        ssum.x += v.x; ssum.y += v.y; ssum.z += v.z; ssum.w += v.w;
        ssq.x += v.x * v.x; ssq.y += v.y * v.y; ssq.z += v.z * v.z; ssq.w += v.w * v.w;
    }
    if (stats) {
        red_s[tid] = ssum; red_q[tid] = ssq;
        __syncthreads();
        if (tid < 16) {
            float4 s4 = red_s[tid], q4 = red_q[tid];
            for (int l = 1; l < 16; ++l) {
                const float4 a = red_s[l * 16 + tid], q = red_q[l * 16 + tid];
                s4.x += a.x; s4.y += a.y; s4.z += a.z; s4.w += a.w;
                q4.x += q.x; q4.y += q.y; q4.z += q.z; q4.w += q.w;
            }
            float* o = stats + (((size_t)b * nrec + blockIdx.x) * Cout + 4 * cq) * 2;
            *reinterpret_cast<float4*>(o) = make_float4(s4.x, q4.x, s4.y, q4.y);
            *reinterpret_cast<float4*>(o + 4) = make_float4(s4.z, q4.z, s4.w, q4.w);
        }
    }
}
void launch_conv_reduce(const float* planes, int ksplit, const float* bias, const float* res, int res_up, float* out, float* stats,
                        int B, int H, int W, int Cout, hipStream_t stream) {
    hipLaunchKernelGGL(conv_reduce_kernel, dim3((H * W + 63) / 64, B, Cout / 64), dim3(256), 0, stream, planes, ksplit, bias, res, res_up, out,
                       stats, H, W, Cout, (size_t)B * H * W * Cout);
}

__global__ void copy_clamp_int_kernel(int* dst, const int* src, int n, int lo, int hi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = min(max(src[i], lo), hi);
}
void launch_copy_clamp_int(int* dst, const int* src, int n, int lo, int hi, hipStream_t stream) {
    hipLaunchKernelGGL(copy_clamp_int_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dst, src, n, lo, hi);
}
void launch_add_int(int* p, int n, int v, hipStream_t stream) {
    hipLaunchKernelGGL(add_int_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, n, v);
}
void launch_fill_int(int* p, int n, int v, hipStream_t stream) {
    hipLaunchKernelGGL(fill_int_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, n, v);
}

// ------------------------------------------------------------------------------------------------
// counter RNG: Philox4x32-10, key = seed, counter = (quad, t, slice, stream) -- must match synth.py
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }

__device__ __forceinline__ float4 normal4(uint32_t quad, uint32_t t, uint32_t slice, uint32_t stream, uint64_t seed) {
    uint32_t r[4];
    philox4x32_10(quad, t, slice, stream, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float ra = sqrtf(-2.0f * logf(u01(r[0])));
    const float ta = 6.28318530717958647692f * u01(r[1]);
    const float rb = sqrtf(-2.0f * logf(u01(r[2])));
    const float tb = 6.28318530717958647692f * u01(r[3]);
    return make_float4(ra * cosf(ta), ra * sinf(ta), rb * cosf(tb), rb * sinf(tb));
}

__global__ __launch_bounds__(256) void noise_fill_kernel(float* __restrict__ out, uint64_t seed, uint32_t stream_id,
                                                         int t, uint64_t slice0, int B, int HW) {
    const int nq = HW >> 2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * nq) return;
    const int q = (int)(e % nq);
    const int b = (int)(e / nq);
    const float4 z = normal4((uint32_t)q, (uint32_t)t, (uint32_t)(slice0 + b), stream_id, seed);
    *reinterpret_cast<float4*>(out + (size_t)b * HW + 4 * q) = z;
}

void launch_noise_fill(float* out, uint64_t seed, uint32_t stream_id, int t, uint64_t slice0, int B, int HW,
                       hipStream_t stream) {
    const long long total = (long long)B * (HW / 4);
    hipLaunchKernelGGL(noise_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, out, seed,
                       stream_id, t, slice0, B, HW);
}

// ------------------------------------------------------------------------------------------------
// posterior step: p_sample -> p_mean_variance -> model_predictions -> q_posterior
// (src/models/modules/cond_DDPM.py:432-444, :422-430, :400-420, :391-398); final map to [0,1] (:463)
// ------------------------------------------------------------------------------------------------
// torch.clamp(x, -1, 1): NaN stays NaN (fminf/fmaxf would turn it into -1 and hide an overflow upstream)
__device__ __forceinline__ float clamp11(float v) { return (v != v) ? v : fminf(fmaxf(v, -1.f), 1.f); }

__global__ __launch_bounds__(256) void step_kernel(const StepArgs a) {
    const int nq = a.HW >> 2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)a.B * nq) return;
    const int q = (int)(e % nq);
    const int b = (int)(e / nq);
    const int t = a.t_dev[b];
    const size_t off = (size_t)b * a.HW + 4 * q;
    float4 x = *reinterpret_cast<const float4*>(a.x + off);
    const float4 mo = *reinterpret_cast<const float4*>(a.model_out + off);
    float4 x0;
    if (a.objective == 0) {
        x0 = mo;
    } else {
        const float sr = a.sqrt_recip[t], srm1 = a.sqrt_recipm1[t];
        x0.x = sr * x.x - srm1 * mo.x;
        x0.y = sr * x.y - srm1 * mo.y;
        x0.z = sr * x.z - srm1 * mo.z;
        x0.w = sr * x.w - srm1 * mo.w;
    }
    if (a.clip) {     // clip_denoised (cond_DDPM.py:416-419, :426-427); off: the raw prediction enters the posterior mean
        x0.x = clamp11(x0.x);
        x0.y = clamp11(x0.y);
        x0.z = clamp11(x0.z);
        x0.w = clamp11(x0.w);
    }
    const float c1 = a.coef1[t], c2 = a.coef2[t];
    // separate roundings of the two products, as the reference's tensor expression does
    float4 r;
    r.x = __fadd_rn(__fmul_rn(c1, x0.x), __fmul_rn(c2, x.x));
    r.y = __fadd_rn(__fmul_rn(c1, x0.y), __fmul_rn(c2, x.y));
    r.z = __fadd_rn(__fmul_rn(c1, x0.z), __fmul_rn(c2, x.z));
    r.w = __fadd_rn(__fmul_rn(c1, x0.w), __fmul_rn(c2, x.w));
    if (t > 0) {
        const float sigma = expf(0.5f * a.logvar[t]);
        float4 z;
        if (a.noise) z = *reinterpret_cast<const float4*>(a.noise + (size_t)t * a.noise_t_stride + off);
        else z = normal4((uint32_t)q, (uint32_t)t, (uint32_t)(a.slice0 + b), 0x1002u, a.seed);
        r.x = __fadd_rn(r.x, __fmul_rn(sigma, z.x));
        r.y = __fadd_rn(r.y, __fmul_rn(sigma, z.y));
        r.z = __fadd_rn(r.z, __fmul_rn(sigma, z.z));
        r.w = __fadd_rn(r.w, __fmul_rn(sigma, z.w));
    }
    if (a.finalize < 0 ? (t == 0) : (a.finalize != 0)) {
        r.x = (r.x + 1.f) * 0.5f; r.y = (r.y + 1.f) * 0.5f; r.z = (r.z + 1.f) * 0.5f; r.w = (r.w + 1.f) * 0.5f;
    }
    *reinterpret_cast<float4*>(a.x + off) = r;
}

void launch_step(const StepArgs& a, hipStream_t stream) {
    const long long total = (long long)a.B * (a.HW / 4);
    hipLaunchKernelGGL(step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
}

// ------------------------------------------------------------------------------------------------
// DDIM update (src/models/modules/cond_DDPM.py:487-513; model_predictions :400-420 with clip_x_start = False, so the
// noise estimate uses the unclipped x0; the clamp of :496 comes after). Products and sums rounded separately, in the
// order of the reference's tensor expression.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ddim_step_kernel(const DdimArgs a) {
    const int nq = a.HW >> 2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)a.B * nq) return;
    const int q = (int)(e % nq);
    const int b = (int)(e / nq);
    const int t = a.t_dev[b];
    const size_t off = (size_t)b * a.HW + 4 * q;
    const float4 x4 = *reinterpret_cast<const float4*>(a.x + off);
    const float4 mo4 = *reinterpret_cast<const float4*>(a.model_out + off);
    float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.add_noise) {
        if (a.noise) z4 = *reinterpret_cast<const float4*>(a.noise + off);
        else z4 = normal4((uint32_t)q, (uint32_t)t, (uint32_t)(a.slice0 + b), 0x1002u, a.seed);
    }
    const float sr = a.sqrt_recip[t], srm1 = a.sqrt_recipm1[t];
    const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ms[4] = {mo4.x, mo4.y, mo4.z, mo4.w}, zs[4] = {z4.x, z4.y, z4.z, z4.w};
    float rs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float x0, eps;
        if (a.objective == 0) {      // pred_x0
            x0 = ms[i];
            eps = __fdiv_rn(__fsub_rn(__fmul_rn(sr, xs[i]), x0), srm1);
        } else {                     // pred_noise
            eps = ms[i];
            x0 = __fsub_rn(__fmul_rn(sr, xs[i]), __fmul_rn(srm1, eps));
        }
        if (a.clip) x0 = clamp11(x0);          // clip_denoised (cond_DDPM.py:493-494)
        float r = __fadd_rn(__fmul_rn(x0, a.coef_x0), __fmul_rn(a.coef_eps, eps));
        r = __fadd_rn(r, __fmul_rn(a.sigma, zs[i]));
        if (a.finalize) r = (r + 1.f) * 0.5f;
        rs[i] = r;
    }
    *reinterpret_cast<float4*>(a.x + off) = make_float4(rs[0], rs[1], rs[2], rs[3]);
}

void launch_ddim_step(const DdimArgs& a, hipStream_t stream) {
    const long long total = (long long)a.B * (a.HW / 4);
    hipLaunchKernelGGL(ddim_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
}

// q_sample fused with normalize_to_neg_one_to_one (cond_DDPM.py:548-554, :75, :653)
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x01, const float* __restrict__ noise,
                                                       const int* __restrict__ t_dev, const float* __restrict__ sa,
                                                       const float* __restrict__ s1ma, float* __restrict__ out, int B,
                                                       int HW) {
    const int nq = HW >> 2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * nq) return;
    const int b = (int)(e / nq);
    const int t = t_dev[b];
    const float ca = sa[t], cb = s1ma[t];
    const size_t off = (size_t)e * 4;
    const float4 x = *reinterpret_cast<const float4*>(x01 + off);
    const float4 n = *reinterpret_cast<const float4*>(noise + off);
    float4 r;
    r.x = __fadd_rn(__fmul_rn(ca, x.x * 2.f - 1.f), __fmul_rn(cb, n.x));
    r.y = __fadd_rn(__fmul_rn(ca, x.y * 2.f - 1.f), __fmul_rn(cb, n.y));
    r.z = __fadd_rn(__fmul_rn(ca, x.z * 2.f - 1.f), __fmul_rn(cb, n.z));
    r.w = __fadd_rn(__fmul_rn(ca, x.w * 2.f - 1.f), __fmul_rn(cb, n.w));
    *reinterpret_cast<float4*>(out + off) = r;
}

void launch_q_sample(const float* x01, const float* noise, const int* t_dev, const float* sa, const float* s1ma,
                     float* out, int B, int HW, hipStream_t stream) {
    const long long total = (long long)B * (HW / 4);
    hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x01, noise, t_dev, sa,
                       s1ma, out, B, HW);
}

}  // namespace cddpm
