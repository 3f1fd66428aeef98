// Fused implicit-GEMM convolution for gfx950 (MI355X), exact fp32 on v_mfma_f32_32x32x2_f32.
//
// Replaces, inside every ResBlock / AttentionBlock of the reference UNet
// (src/models/modules/OpenAI_Unet.py:284-338, :386-394), the chain
//     GroupNorm32 -> [FiLM] -> SiLU -> [nearest x2] -> [torch.cat] -> Conv2d(3x3 | 1x1) -> [+ skip]
// with ONE kernel: normalisation/FiLM arrive as per-(sample, channel) coefficients (mean, a, d) computed by
// norm_kernels.hip and are applied while the input patch is staged into LDS; the channel concat of the
// up path (OpenAI_Unet.py:948) is two source pointers; the nearest-neighbour upsample (:118-128) is an
// index shift; the residual add / 1x1 skip_connection (:261-268, :336) is the epilogue / a second K segment.
//
// GEMM view:  D[pixel][cout] = sum_{tap, ci} act(X)[pixel + tap][ci] * Wt[tap][ci][cout]
//   M = 128 pixels  (4 image rows x 32 columns),  N = 128 output channels,  K step = 32 input channels x 1 tap.
//   4 waves, each owns a 64 x 64 sub-tile = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator VGPRs).
// LDS (58.9 KB for 3x3 -> 2 workgroups per CU):
//   act patch   : (4+2) x (32+2) pixels x 32 channels, one 128-B row per pixel, 16-B slot s stored at
//                 s ^ ((pixel>>1)&7)  -> ds_read_b128 of 16 consecutive pixels is bank-conflict free
//   weight slab : 2 buffers x [128 cout][32 ci], same swizzle; the packed global image IS the LDS image, so
//                 staging is a linear 16-B copy.
// Each lane fetches 4 consecutive channels per ds_read_b128 and feeds them to 4 successive MFMAs; the K order
// inside a 8-channel group is therefore {c, c+4} pairs -- identical for A and B, so the sum is unchanged.
#include "kernels.h"
#include <cstdlib>
#include <vector>

namespace cddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));   // native vector: stays in registers (HIP's float4 struct arrays may not)

#ifdef CDDPM_STAMPS
// phase accounting for diagnostic builds: 0 prologue, 1 patch stage (barrier + transform + ds_write), 2 weight stage
// (ds_write + prefetch issue + barrier), 3 MFMA compute, 4 chunk fold, 5 epilogue
#define STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_[i] += now_ - last_; last_ = now_; }
#else
#define STAMP(i)
#endif

__device__ __forceinline__ float silu_f(float v) {
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

// NWV = waves per workgroup: 4 -> each wave owns 64 pixels x 64 couts (2 x 2 MFMA tiles), 2 workgroups per CU;
//                            8 -> each wave owns 64 pixels x 32 couts (2 x 1 MFMA tiles), 2 workgroups per CU = 4 waves
//                                 per SIMD: with four MFMA streams per SIMD the matrix pipe stays fed while some
//                                 waves stage, wait at the per-tap barrier or run their epilogue.
template <int TAPS, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV / 2) void conv_mfma_kernel(const ConvArgs a) {
    constexpr int THREADS = 64 * NWV;
    constexpr int NT = 8 / NWV;                 // 32-cout MFMA tiles per wave: 2 | 1
    // TAPS == 4 is the folded form of "nearest x2 upsample -> 3x3 conv": an output pixel (2y+a, 2x+b) sees only
    // 2 x 2 distinct source pixels, so each of the four parity classes (a, b) is a 2x2-tap convolution of the
    // LOW-resolution input with pre-summed weights (pack_conv_weights_up2): 4/9 of the multiplies, same result up to
    // the rounding of the weight sums. Tiles then walk the low-resolution grid of one class.
    constexpr bool UP2 = (TAPS == 4);
    constexpr int PAD = (TAPS == 9) ? 1 : 0;
    constexpr int PW = UP2 ? 33 : 32 + 2 * PAD;     // patch width  (pixels)
    constexpr int PH = UP2 ? 5 : 4 + 2 * PAD;       // patch height (pixels)
    constexpr int NPIX = PW * PH;                   // 204 | 128 | 165
    constexpr int NK = (NPIX * 8 + THREADS - 1) / THREADS;  // v4f patch entries per thread: 7 | 4 (4 waves), 4 | 2 (8 waves)
    constexpr int WK = 1024 / THREADS;          // v4f of a weight slab per thread: 4 | 2

    extern __shared__ v4f lds[];
    v4f* ldsA = lds;                // NPIX * 8 v4f
    v4f* ldsW = lds + NPIX * 8;     // 2 * 1024 v4f

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
#ifdef CDDPM_STAMPS
    unsigned long long st_[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
    const unsigned long long t0c_ = last_, t0r_ = __builtin_amdgcn_s_memrealtime();   // in-kernel clock = dc / dr * 100 MHz
#endif
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int wm = wave & 1;    // pixel rows {0,1} | {2,3}
    const int wn = wave >> 1;   // cout block of 32 * NT within the 128

    const int ncb = a.Cout >> 7;
    // tile grid: the output image, or (UP2) the low-resolution grid of one parity class
    const int gridH = UP2 ? (a.H >> 1) : a.H, gridW = UP2 ? (a.W >> 1) : a.W;
    const int tilesX = (gridW + 31) >> 5;
    const int tilesY = (gridH + 3) >> 2;
    int bid = blockIdx.x;
    const int cb = bid % ncb;
    bid /= ncb;
    const int tx = bid % tilesX;
    bid /= tilesX;
    const int ty = bid % tilesY;
    bid /= tilesY;
    const int cls = UP2 ? (bid & 3) : 0;        // parity class: a = cls >> 1 (row), bb = cls & 1 (column)
    const int b = UP2 ? (bid >> 2) : bid;
    const int pa = cls >> 1, pb = cls & 1;
    const int y0 = ty * 4, x0 = tx * 32;

    const int Cin = a.C0 + a.C1;
    const int nch_main = Cin >> 5;
    const int nch_skip = (a.S0 + a.S1) >> 5;
    const int nch = nch_main + nch_skip;

    // ---- per-thread patch entries: slot s is fixed per thread, pixel q = (tid>>3) + (THREADS/8) k
    const int s = tid & 7;
    int psrc[NK];           // source pixel index (main segment), -1 = zero padding / outside
    unsigned centre = 0;    // bit k: entry k is a centre (non-halo) pixel inside the image -> read by the skip segment,
                            // whose sources live at the output resolution (same index: such convs never upsample)
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int q = (tid >> 3) + (THREADS / 8) * k;
        const int pr = q / PW, pc = q - pr * PW;
        // UP2: the patch covers low-res rows y0 + a - 1 .. y0 + a + 3 and columns x0 + b - 1 .. x0 + b + 31
        const int y = UP2 ? (y0 + pr + pa - 1) : (y0 + pr - PAD), x = UP2 ? (x0 + pc + pb - 1) : (x0 + pc - PAD);
        const bool valid = (q < NPIX) && (y >= 0) && (y < gridH) && (x >= 0) && (x < gridW);
        const int sy = (!UP2 && a.upsample) ? (y >> 1) : y, sx = (!UP2 && a.upsample) ? (x >> 1) : x;
        psrc[k] = valid ? ((b * a.srcH + sy) * a.srcW + sx) : -1;
        if (valid && (pr >= PAD) && (pr < PH - PAD) && (pc >= PAD) && (pc < PW - PAD)) centre |= 1u << k;
    }

    const v4f* wmain = reinterpret_cast<const v4f*>(a.wpk) + (size_t)(cls * ncb + cb) * nch_main * TAPS * 1024;
    const v4f* wskip = reinterpret_cast<const v4f*>(a.skip_wpk) + (size_t)cb * nch_skip * 1024;

    v4f wreg[WK];
    v4f areg[NK];
    // GroupNorm/FiLM coefficients of this sample (3 x Cin floats), cached in LDS by the prologue
    v4f* ldsC = lds + NPIX * 8 + 2048;
    const bool have_coef = (a.coef != nullptr);

    // stage pointer: slab of (chunk, tap); past the end it wraps to stage 0 so the prefetch stays unconditional
    auto wslab = [&](int chunk, int tap) -> const v4f* {
        if (chunk >= nch) { chunk = 0; tap = 0; }
        return (chunk < nch_main) ? (wmain + ((size_t)chunk * TAPS + tap) * 1024)
                                  : (wskip + (size_t)(chunk - nch_main) * 1024);
    };
    auto load_act = [&](int chunk) {
        const float* base;
        int Cs, c0;
        const bool main_seg = chunk < nch_main;
        if (main_seg) {
            const int ch = chunk << 5;
            if (ch < a.C0) { base = a.src0; Cs = a.C0; c0 = ch; }
            else           { base = a.src1; Cs = a.C1; c0 = ch - a.C0; }
        } else {
            const int ch = (chunk - nch_main) << 5;
            if (ch < a.S0) { base = a.skip0; Cs = a.S0; c0 = ch; }
            else           { base = a.skip1; Cs = a.S1; c0 = ch - a.S0; }
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int p = (main_seg || ((centre >> k) & 1u)) ? psrc[k] : -1;
            v4f v = v4f{0.f, 0.f, 0.f, 0.f};
            if (p >= 0) v = *reinterpret_cast<const v4f*>(base + (size_t)p * Cs + c0 + 4 * s);
            areg[k] = v;
        }
    };
    auto store_act = [&](int chunk) {
        const bool main_seg = chunk < nch_main;
        const bool do_silu = main_seg && a.silu;
        v4f cm = v4f{0.f, 0.f, 0.f, 0.f}, ca = v4f{1.f, 1.f, 1.f, 1.f}, cd = cm;
        const bool aff = main_seg && have_coef;
        if (aff) {
            const int ci = (chunk << 3) + s;
            cm = ldsC[ci];
            ca = ldsC[(Cin >> 2) + ci];
            cd = ldsC[2 * (Cin >> 2) + ci];
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            v4f v = areg[k];
            const int p = (main_seg || ((centre >> k) & 1u)) ? psrc[k] : -1;
            if (p >= 0) {   // zero padding stays exactly zero: the conv pads AFTER the activation
                if (aff) v = (v - cm) * ca + cd;
                if (do_silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            }
            const int q = (tid >> 3) + (THREADS / 8) * k;
            if (q < NPIX) ldsA[q * 8 + (s ^ ((q >> 1) & 7))] = v;
        }
    };

    // Two-level accumulation: `acc` collects one 32-channel chunk (<= 9 taps x 32 = 288 products per chain),
    // `tot` sums the chunks. A single K-long fp32 fmaf chain (K up to 4608 + 512) carries ~sqrt(K/2) ulp of
    // rounding noise, about 3x what the reference's blocked CPU convolution shows against fp64; splitting the
    // chain brings this kernel to the same level (measured in tests/test_gpu_unet.py, fp64 yardstick).
    f32x16 acc[2][NT], tot[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    // B operand (weights) LDS offsets: row j = cout within the 128 block
    int boff[NT], bsw[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int j = 32 * NT * wn + 32 * nt + li;
        boff[nt] = j * 8;
        bsw[nt] = (j >> 1) & 7;
    }

    auto compute = [&](int tap, int buf) {
        const int ky = (TAPS == 9) ? (tap / 3) : (UP2 ? (tap >> 1) : 0);
        const int kx = (TAPS == 9) ? (tap - 3 * ky) : (UP2 ? (tap & 1) : 0);
        int aoff[2], asw[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int q = (2 * wm + mt + ky) * PW + li + kx;
            aoff[mt] = q * 8;
            asw[mt] = (q >> 1) & 7;
        }
        const v4f* wb = ldsW + buf * 1024;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int sl = 2 * g + lh;
            const v4f a0 = ldsA[aoff[0] + (sl ^ asw[0])];
            const v4f a1 = ldsA[aoff[1] + (sl ^ asw[1])];
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w};
            const float av1[4] = {a1.x, a1.y, a1.z, a1.w};
            v4f bq[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[nt] = wb[boff[nt] + (sl ^ bsw[nt])];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float bv = (m == 0) ? bq[nt].x : (m == 1) ? bq[nt].y : (m == 2) ? bq[nt].z : bq[nt].w;
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[m], bv, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[m], bv, acc[1][nt], 0, 0, 0);
                }
            }
        }
    };

    // ---- main loop: weights double-buffered in LDS and prefetched through registers one stage ahead;
    //      the next chunk's patch is fetched into registers behind the last tap's MFMAs.
    {
        const v4f* p0 = wslab(0, 0);
#pragma unroll
        for (int i = 0; i < WK; ++i) wreg[i] = p0[tid + THREADS * i];
    }
    load_act(0);
    if (have_coef) {
        const int nq = Cin >> 2;
        const size_t plane = (size_t)a.B * Cin;
        for (int i = tid; i < 3 * nq; i += THREADS) {
            const int pl = i / nq, c4 = i - pl * nq;
            ldsC[i] = *reinterpret_cast<const v4f*>(a.coef + pl * plane + (size_t)b * Cin + 4 * c4);
        }
    }
    int buf = 0;
    STAMP(0)
    for (int chunk = 0; chunk < nch; ++chunk) {
        const bool main_seg = chunk < nch_main;
        const int ntap = main_seg ? TAPS : 1;
        __syncthreads();   // every wave is done reading the previous patch
        store_act(chunk);
        STAMP(1)
        for (int t = 0; t < ntap; ++t) {
#pragma unroll
            for (int i = 0; i < WK; ++i) ldsW[buf * 1024 + tid + THREADS * i] = wreg[i];
            const bool last_tap = (t == ntap - 1);
            const v4f* pn = last_tap ? wslab(chunk + 1, 0) : wslab(chunk, t + 1);
#pragma unroll
            for (int i = 0; i < WK; ++i) wreg[i] = pn[tid + THREADS * i];
            if (last_tap && chunk + 1 < nch) load_act(chunk + 1);
            __syncthreads();
            STAMP(2)
            compute(main_seg ? t : (TAPS / 2), buf);   // skip segment: centre tap
            buf ^= 1;
            STAMP(3)
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                tot[i][j] += acc[i][j];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        STAMP(4)
    }

    __syncthreads();   // every wave is done with the patch / weight buffers before they become transpose space
    // ---- epilogue. The accumulator layout (D row = pixel (r&3) + 8 (r>>2) + 4 lh, D col = cout li) would give
    //      4-byte stores and one dependent residual load per store; instead each wave transposes its 64 x 64 tile
    //      through a private 8-KB LDS region (patch buffers are dead after the last barrier), one 32-channel half
    //      at a time, so that every lane moves 16 B and every wave instruction covers eight full 128-B lines:
    //      all residual loads of a half are in flight before the first add.
    {
        float* tr = reinterpret_cast<float*>(lds) + wave * 2048;      // [64 pixels][32 channels]
        const int cq = lane & 7;                                      // channel quad of this lane in the read phase
        const int prow = lane >> 3;                                   // pixel row within a group of 8
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = cb * 128 + 32 * NT * wn + 32 * nt + 4 * cq;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tr[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = tot[mt][nt][r];
            __builtin_amdgcn_wave_barrier();
            const v4f bias = a.bias ? *reinterpret_cast<const v4f*>(a.bias + co) : v4f{0.f, 0.f, 0.f, 0.f};
            v4f ssum = v4f{0.f, 0.f, 0.f, 0.f}, ssq = ssum;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {          // two batches of 4 pixel rows: bounds the live registers
                v4f val[4], rsd[4];
                size_t oidx[4];
                bool ok[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = 8 * (4 * hb + i) + prow;                // 0..63: tile pixel (row p>>5, column p&31)
                    const int gy = y0 + 2 * wm + (p >> 5), gx = x0 + (p & 31);      // tile-grid coordinates
                    ok[i] = (gy < gridH) && (gx < gridW);
                    const int y = UP2 ? (2 * gy + pa) : gy, x = UP2 ? (2 * gx + pb) : gx;   // output pixel
                    oidx[i] = ((size_t)(b * a.H + y) * a.W + x) * a.Cout + co;
                    rsd[i] = v4f{0.f, 0.f, 0.f, 0.f};
                    if (a.res && ok[i]) {
                        const size_t rp = a.res_up ? ((size_t)(b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1))
                                                   : ((size_t)(b * a.H + y) * a.W + x);
                        rsd[i] = *reinterpret_cast<const v4f*>(a.res + rp * a.Cout + co);
                    }
                    val[i] = *reinterpret_cast<const v4f*>(tr + p * 32 + 4 * cq);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (ok[i]) {
                        const v4f o = val[i] + bias + rsd[i];
                        *reinterpret_cast<v4f*>(a.out + oidx[i]) = o;
                        ssum += o;
                        ssq += o * o;
                    }
            }
            // GroupNorm statistics of the tensor just written, for the NEXT GroupNorm: per-channel sums over this
            // wave's 64 pixels (8 in-lane values, then the 8 lanes sharing a channel quad), one record per wave tile
            if (a.stats) {
#pragma unroll
                for (int m = 8; m < 64; m <<= 1) {
                    ssum.x += __shfl_xor(ssum.x, m, 64); ssum.y += __shfl_xor(ssum.y, m, 64);
                    ssum.z += __shfl_xor(ssum.z, m, 64); ssum.w += __shfl_xor(ssum.w, m, 64);
                    ssq.x += __shfl_xor(ssq.x, m, 64); ssq.y += __shfl_xor(ssq.y, m, 64);
                    ssq.z += __shfl_xor(ssq.z, m, 64); ssq.w += __shfl_xor(ssq.w, m, 64);
                }
                if (prow == 0) {
                    const int nrec = (UP2 ? 8 : 2) * tilesX * tilesY;
                    const int rec = 2 * ((cls * tilesY + ty) * tilesX + tx) + wm;
                    float* o = a.stats + (((size_t)b * nrec + rec) * a.Cout + co) * 2;
                    *reinterpret_cast<v4f*>(o) = v4f{ssum.x, ssq.x, ssum.y, ssq.y};
                    *reinterpret_cast<v4f*>(o + 4) = v4f{ssum.z, ssq.z, ssum.w, ssq.w};
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#ifdef CDDPM_STAMPS
    STAMP(5)
    if (a.stamps && lane == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&a.stamps[wave * 8 + i], st_[i]);
        if (wave == 0) {
            atomicAdd(&a.stamps[40], __builtin_amdgcn_s_memtime() - t0c_);
            atomicAdd(&a.stamps[41], __builtin_amdgcn_s_memrealtime() - t0r_);
        }
    }
#endif
}

void launch_conv(const ConvArgs& a, hipStream_t stream) {
    if (conv_mode() != 0) { launch_conv_split(a, stream); return; }
    // 4 waves (64 x 64 per wave, 2 waves per SIMD) is the default; CDDPM_CONV_WAVES=8 selects the 8-wave split
    // (64 x 32 per wave, 4 waves per SIMD), which measures the same throughput (tools/conv_ab.py, profiles/)
    static const int nwv = [] { const char* e = getenv("CDDPM_CONV_WAVES"); return (e && e[0] == '8') ? 8 : 4; }();
    const bool up2 = (a.taps == 4);
    const int gh = up2 ? a.H / 2 : a.H, gw = up2 ? a.W / 2 : a.W;
    const int tilesX = (gw + 31) / 32, tilesY = (gh + 3) / 4;
    const unsigned grid = (unsigned)(a.B * (up2 ? 4 : 1) * tilesX * tilesY * (a.Cout / 128));
    // patch + 2 weight slabs + coefficient cache; the epilogue reuses the space as NWV private 8-KB transpose regions
    const size_t coef_lds = a.coef ? (size_t)3 * (a.C0 + a.C1) * sizeof(float) : 0;
    const size_t need9 = (size_t)(6 * 34 * 8 + 2048) * 16 + coef_lds, need1 = (size_t)(4 * 32 * 8 + 2048) * 16 + coef_lds;
    const size_t need4 = (size_t)(5 * 33 * 8 + 2048) * 16 + coef_lds;
    static bool attr = false;
    if (!attr) {   // > 64 KB of dynamic LDS needs the opt-in; 80 KB still leaves two workgroups per CU
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<9, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<9, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr = true;
    }
    if (up2) {
        hipLaunchKernelGGL((conv_mfma_kernel<4, 4>), dim3(grid), dim3(256), need4, stream, a);
    } else if (nwv == 8) {
        const size_t tr = 8 * 2048 * sizeof(float);
        if (a.taps == 9) hipLaunchKernelGGL((conv_mfma_kernel<9, 8>), dim3(grid), dim3(512), need9 > tr ? need9 : tr, stream, a);
        else hipLaunchKernelGGL((conv_mfma_kernel<1, 8>), dim3(grid), dim3(512), need1 > tr ? need1 : tr, stream, a);
    } else {
        if (a.taps == 9) hipLaunchKernelGGL((conv_mfma_kernel<9, 4>), dim3(grid), dim3(256), need9, stream, a);
        else hipLaunchKernelGGL((conv_mfma_kernel<1, 4>), dim3(grid), dim3(256), need1, stream, a);
    }
}

size_t packed_conv_floats(int Cout, int Cin, int taps) {
    const size_t n = (size_t)Cout * Cin * taps;
    return conv_mode() == 1 ? n + n / 2 : n;      // x6: three bf16 per weight; h3: two fp16; f32: one float
}

// w: PyTorch [Cout][Cin][k][k] (taps = k*k, tap = ky*3+kx) -> [Cout/128][Cin/32][taps][128][8 slots][4]
void pack_conv_weights(const float* w, int Cout, int Cin, int taps, float* dst, int wexp) {
    if (conv_mode() != 0) { pack_conv_weights_split(w, Cout, Cin, taps, dst, wexp); return; }
    const int ncb = Cout / 128, nch = Cin / 32;
    for (int cb = 0; cb < ncb; ++cb)
        for (int ch = 0; ch < nch; ++ch)
            for (int t = 0; t < taps; ++t) {
                float* img = dst + (((size_t)cb * nch + ch) * taps + t) * 4096;
                for (int j = 0; j < 128; ++j)
                    for (int sl = 0; sl < 8; ++sl) {
                        float* d4 = img + (size_t)(j * 8 + (sl ^ ((j >> 1) & 7))) * 4;
                        for (int e = 0; e < 4; ++e) {
                            const int co = cb * 128 + j, ci = ch * 32 + sl * 4 + e;
                            d4[e] = w[((size_t)co * Cin + ci) * taps + t];
                        }
                    }
            }
}

// Folded weights of "nearest x2 upsample -> 3x3 conv": class (a, b), tap (ty, tx) of the 2x2 low-resolution stencil
// collects the original taps that land on the same source pixel:
//   rows  a = 0: ty 0 <- ky {0},    ty 1 <- ky {1, 2};   a = 1: ty 0 <- ky {0, 1}, ty 1 <- ky {2}   (columns alike)
// dst: [4 classes][Cout/128][Cin/32][4 taps][image], each class packed like pack_conv_weights with taps = 4
// (4 * packed_conv_floats(Cout, Cin, 4) floats).
int pack_conv_weights_up2(const float* w /*[Cout][Cin][3][3]*/, int Cout, int Cin, float* dst) {
    std::vector<float> wf((size_t)Cout * Cin * 4);
    int wexp = 24;      // one exponent for the four classes: they share the bias / epilogue of one launch
    for (int pass = 0; pass < 2; ++pass)
    for (int cls = 0; cls < 4; ++cls) {
        const int pa = cls >> 1, pb = cls & 1;
        for (size_t oc = 0; oc < (size_t)Cout * Cin; ++oc)
            for (int ty = 0; ty < 2; ++ty)
                for (int tx = 0; tx < 2; ++tx) {
                    const int ky0 = pa == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = pa == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
                    const int kx0 = pb == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = pb == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
                    double acc = 0.0;    // summed in double, rounded once
                    for (int ky = ky0; ky <= ky1; ++ky)
                        for (int kx = kx0; kx <= kx1; ++kx) acc += (double)w[oc * 9 + ky * 3 + kx];
                    wf[oc * 4 + ty * 2 + tx] = (float)acc;
                }
        if (pass == 0) { const int e = conv_weight_exp(wf.data(), wf.size()); if (e < wexp) wexp = e; }
        else pack_conv_weights(wf.data(), Cout, Cin, 4, dst + (size_t)cls * packed_conv_floats(Cout, Cin, 4), wexp);
    }
    return wexp;
}

}  // namespace cddpm
