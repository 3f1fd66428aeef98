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
// LDS (52.2 KB for 3x3 -> 2 workgroups per CU):
//   act patch   : 2 buffers x (4+2) x (32+2) pixels x 32 channels, one 128-B row per pixel, 16-B slot s stored at
//                 s ^ ((pixel>>1)&7)  -> ds_read_b128 of 16 consecutive pixels is bank-conflict free
//   weights     : never in LDS. The packed global image is cut into the 1-KiB pieces one wave consumes per
//                 (N tile, 8-channel group); each piece is one coalesced global_load_dwordx4 into a register ring
//                 (L2/MALL resident: a layer's weights are shared by every workgroup). Without a shared weight
//                 slab the four waves synchronise once per 32-channel chunk instead of once per tap.
// Each lane fetches 4 consecutive channels per ds_read_b128 and feeds them to 4 successive MFMAs; the K order
// inside a 8-channel group is therefore {c, c+4} pairs -- identical for A and B, so the sum is unchanged.
#include "kernels.h"

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
    // v * sigmoid(v); exp(+large) = inf -> rcp = 0, no NaN
#ifdef CDDPM_ACCURATE_SILU
    return v * __builtin_amdgcn_rcpf(1.0f + expf(-v));       // ocml expf (<= 1 ulp), ~20 VALU
#else
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));     // v_exp_f32 path (~3 ulp on the exp), 6 VALU
#endif
}

template <int TAPS>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs a) {
    constexpr int PAD = (TAPS == 9) ? 1 : 0;
    constexpr int PW = 32 + 2 * PAD;            // patch width  (pixels)
    constexpr int PH = 4 + 2 * PAD;             // patch height (pixels)
    constexpr int NPIX = PW * PH;               // 204 | 128
    constexpr int NK = (NPIX * 8 + 255) / 256;  // v4f patch entries per thread: 7 | 4

    extern __shared__ v4f lds[];                // two patch buffers of NPIX * 8 v4f

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
#ifdef CDDPM_STAMPS
    unsigned long long st_[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int wm = wave & 1;    // pixel rows {0,1} | {2,3}
    const int wn = wave >> 1;   // cout 0..63 | 64..127

    const int ncb = a.Cout >> 7;
    const int tilesX = (a.W + 31) >> 5;
    const int tilesY = (a.H + 3) >> 2;
    int bid = blockIdx.x;
    const int cb = bid % ncb;
    bid /= ncb;
    const int tx = bid % tilesX;
    bid /= tilesX;
    const int ty = bid % tilesY;
    const int b = bid / tilesY;
    const int y0 = ty * 4, x0 = tx * 32;

    const int Cin = a.C0 + a.C1;
    const int nch_main = Cin >> 5;
    const int nch_skip = (a.S0 + a.S1) >> 5;
    const int nch = nch_main + nch_skip;
#ifdef CDDPM_STAGGER
    // Break the lockstep of the two co-resident workgroups of a CU: they are dispatched together, run equally long
    // programs and would reach every non-MFMA phase (prologue, patch staging, epilogue) at the same time, leaving
    // the matrix pipe idle. First-round workgroups start after a pseudo-random delay of up to ~7/8 of a workgroup's
    // run time; the offset then persists for the rest of the launch because every later workgroup is equally long.
    if (blockIdx.x < 512) {
        const unsigned hsh = (blockIdx.x * 2654435761u) >> 29;                       // 0..7
        const int units = (int)((hsh * (unsigned)(nch_main * TAPS + nch_skip)) >> 3);  // 1 unit ~ 8k cycles ~ 1 shared stage
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif

    // ---- patch staging. Thread -> (16-B channel slot s, pixel): s = tid & 7 is fixed per thread.
    //   9-tap chunks: NK entries per thread, pixel q = (tid>>3) + 32 k of the haloed patch;
    //   single-tap chunks (1x1 kernel, fused skip_connection segment): 4 entries, centre pixels only.
    const int s = tid & 7;
    const size_t coef_plane = (size_t)a.B * Cin;

    // source description of a chunk
    struct Src { const float* base; int Cs; int c0; bool main_seg; };
    auto chunk_src = [&](int chunk) -> Src {
        Src r;
        r.main_seg = chunk < nch_main;
        if (r.main_seg) {
            const int ch = chunk << 5;
            if (ch < a.C0) { r.base = a.src0; r.Cs = a.C0; r.c0 = ch; }
            else           { r.base = a.src1; r.Cs = a.C1; r.c0 = ch - a.C0; }
        } else {
            const int ch = (chunk - nch_main) << 5;
            if (ch < a.S0) { r.base = a.skip0; r.Cs = a.S0; r.c0 = ch; }
            else           { r.base = a.skip1; r.Cs = a.S1; r.c0 = ch - a.S0; }
        }
        return r;
    };
    // haloed-patch entry k of a main chunk: returns the value (zero outside the image) and whether it is a real pixel
    auto fetch9 = [&](const Src& sc, int k, bool& inside) -> v4f {
        const int q = (tid >> 3) + 32 * k;
        const int pr = q / PW, pc = q - pr * PW;
        const int y = y0 + pr - PAD, x = x0 + pc - PAD;
        inside = (q < NPIX) && (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
        v4f v = v4f{0.f, 0.f, 0.f, 0.f};
        if (inside) {
            const int sy = a.upsample ? (y >> 1) : y, sx = a.upsample ? (x >> 1) : x;
            v = *reinterpret_cast<const v4f*>(sc.base + (size_t)((b * a.srcH + sy) * a.srcW + sx) * sc.Cs + sc.c0 + 4 * s);
        }
        return v;
    };
    // centre entry j (0..3) of a single-tap chunk; sources of such chunks are never upsampled
    auto fetch1 = [&](const Src& sc, int j, bool& inside) -> v4f {
        const int cp = (tid >> 3) + 32 * j;          // 0..127
        const int y = y0 + (cp >> 5), x = x0 + (cp & 31);
        inside = (y < a.H) && (x < a.W);
        v4f v = v4f{0.f, 0.f, 0.f, 0.f};
        if (inside) {
            const int sy = (TAPS == 1 && a.upsample) ? (y >> 1) : y, sx = (TAPS == 1 && a.upsample) ? (x >> 1) : x;
            const int sH = (TAPS == 1) ? a.srcH : a.H, sW = (TAPS == 1) ? a.srcW : a.W;
            v = *reinterpret_cast<const v4f*>(sc.base + (size_t)((b * sH + sy) * sW + sx) * sc.Cs + sc.c0 + 4 * s);
        }
        return v;
    };
    // GroupNorm/FiLM affine + SiLU of one entry. The sample's coefficients (3 x Cin floats) are cached in LDS by
    // the prologue: re-reading them from global here would make the in-order vmcnt wait drain the weight prefetch.
    v4f* ldsC = lds + 2 * NPIX * 8;    // [3][Cin/4]
    auto transform = [&](v4f v, int chunk, bool inside) -> v4f {
        if (inside && chunk < nch_main) {   // zero padding stays exactly zero: the conv pads AFTER the activation
            if (a.coef) {
                const int ci = (chunk << 3) + s;
                const v4f cm = ldsC[ci];
                const v4f ca = ldsC[(Cin >> 2) + ci];
                const v4f cd = ldsC[2 * (Cin >> 2) + ci];
                v = (v - cm) * ca + cd;
            }
            if (a.silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
        }
        return v;
    };
    auto put9 = [&](v4f* pbuf, int k, v4f v) {
        const int q = (tid >> 3) + 32 * k;
        if (q < NPIX) pbuf[q * 8 + (s ^ ((q >> 1) & 7))] = v;
    };
    auto put1 = [&](v4f* pbuf, int j, v4f v) {
        const int cp = (tid >> 3) + 32 * j;
        const int q = ((cp >> 5) + PAD) * PW + (cp & 31) + PAD;
        pbuf[q * 8 + (s ^ ((q >> 1) & 7))] = v;
    };

    // weight stage images: 1024 v4f per (cout block, chunk, tap), laid out [wn][nt][g][lane] (see pack_conv_weights)
    const v4f* wmain = reinterpret_cast<const v4f*>(a.wpk) + (size_t)cb * nch_main * TAPS * 1024 + wn * 512 + lane;
    const v4f* wskip = reinterpret_cast<const v4f*>(a.skip_wpk) + (size_t)cb * nch_skip * 1024 + wn * 512 + lane;
    // past the end the pointer wraps to stage 0 so the prefetch stays unconditional
    auto wslab = [&](int chunk, int tap) -> const v4f* {
        if (chunk >= nch) { chunk = 0; tap = 0; }
        return (chunk < nch_main) ? (wmain + ((size_t)chunk * TAPS + tap) * 1024)
                                  : (wskip + (size_t)(chunk - nch_main) * 1024);
    };

    // Two-level accumulation: `acc` collects one 32-channel chunk (<= 9 taps x 32 = 288 products per chain),
    // `tot` sums the chunks. A single K-long fp32 fmaf chain (K up to 4608 + 512) carries ~sqrt(K/2) ulp of
    // rounding noise, about 3x what the reference's blocked CPU convolution shows against fp64; splitting the
    // chain brings this kernel to the same level (measured in tests/test_gpu_unet.py, fp64 yardstick).
    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    // B operand (weights): straight from global/L2 into a register ring, one 1-KiB wave-load per (N tile, group);
    // group g of the NEXT stage is requested as soon as group g of this stage has been handed to the MFMAs.
    v4f breg[4][2];
    // A operand (patch) fragments, double-buffered by hand: while group g multiplies, group g+1 is in flight
    v4f fa0, fa1, fb0, fb1;

    auto a_off = [&](int tap, int mt) -> int {   // v4f offset of this lane's pixel row for a tap, swizzle key in bits 28..30
        const int ky = (TAPS == 9) ? (tap / 3) : 0;
        const int kx = (TAPS == 9) ? (tap - 3 * ky) : 0;
        const int q = (2 * wm + mt + ky) * PW + li + kx;
        return (q * 8) | (((q >> 1) & 7) << 28);
    };
    auto a_read = [&](const v4f* pbuf, int off, int g) -> v4f {
        return pbuf[(off & 0x0fffffff) + ((2 * g + lh) ^ (off >> 28))];
    };

#define MFMA16(A0, A1, G)                                                                                   \
    {                                                                                                       \
        const float av0_[4] = {A0.x, A0.y, A0.z, A0.w};                                                     \
        const float av1_[4] = {A1.x, A1.y, A1.z, A1.w};                                                     \
        const float bv0_[4] = {breg[G][0].x, breg[G][0].y, breg[G][0].z, breg[G][0].w};                     \
        const float bv1_[4] = {breg[G][1].x, breg[G][1].y, breg[G][1].z, breg[G][1].w};                     \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                     \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0_[m], bv0_[m], acc[0][0], 0, 0, 0);         \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0_[m], bv1_[m], acc[0][1], 0, 0, 0);         \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1_[m], bv0_[m], acc[1][0], 0, 0, 0);         \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1_[m], bv1_[m], acc[1][1], 0, 0, 0);         \
        }                                                                                                   \
        breg[G][0] = wnext[G * 64];                                                                         \
        breg[G][1] = wnext[256 + G * 64];                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }

    // One stage = one tap x 32 channels = 64 MFMAs per wave. On entry (fa0, fa1) hold the A fragments of group 0.
    // next_off0/1 >= 0: offsets of the NEXT stage's patch rows in the same buffer -> its group 0 is prefetched here.
    auto stage = [&](int off0, int off1, const v4f* wnext, const v4f* pbuf, int next_off0, int next_off1) {
        // (sched_barrier after each read pair: hipcc otherwise sinks the reads below the MFMAs they should overlap)
        fb0 = a_read(pbuf, off0, 1); fb1 = a_read(pbuf, off1, 1);
        __builtin_amdgcn_sched_barrier(0);
        MFMA16(fa0, fa1, 0)
        fa0 = a_read(pbuf, off0, 2); fa1 = a_read(pbuf, off1, 2);
        __builtin_amdgcn_sched_barrier(0);
        MFMA16(fb0, fb1, 1)
        fb0 = a_read(pbuf, off0, 3); fb1 = a_read(pbuf, off1, 3);
        __builtin_amdgcn_sched_barrier(0);
        MFMA16(fa0, fa1, 2)
        if (next_off0 >= 0) { fa0 = a_read(pbuf, next_off0, 0); fa1 = a_read(pbuf, next_off1, 0); }
        __builtin_amdgcn_sched_barrier(0);
        MFMA16(fb0, fb1, 3)
    };

    // ---- prologue: coefficient cache, first patch (all entries at once), first weight pieces
    if (a.coef) {
        const int nq = Cin >> 2;
        for (int i = tid; i < 3 * nq; i += 256) {
            const int pl = i / nq, c4 = i - pl * nq;
            ldsC[i] = *reinterpret_cast<const v4f*>(a.coef + pl * coef_plane + (size_t)b * Cin + 4 * c4);
        }
        __syncthreads();
    }
    {
        const v4f* p0 = wslab(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) { breg[g][0] = p0[g * 64]; breg[g][1] = p0[256 + g * 64]; }
        const Src sc = chunk_src(0);
        if (TAPS == 9) {
            v4f e[NK];
            bool in[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) e[k] = fetch9(sc, k, in[k]);
#pragma unroll
            for (int k = 0; k < NK; ++k) put9(lds, k, transform(e[k], 0, in[k]));
        } else {
            v4f e[4];
            bool in[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = fetch1(sc, j, in[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) put1(lds, j, transform(e[j], 0, in[j]));
        }
    }
    __syncthreads();
    STAMP(0)

    // ---- main loop. The patch is double-buffered in LDS and the weights never touch LDS, so the four waves
    //      meet at ONE barrier per 32-channel chunk (9 taps = 576 MFMAs per wave), not one per tap. The next
    //      chunk's patch is staged one entry per tap behind the MFMAs (9-tap chunks) or around the last stage.
    for (int chunk = 0; chunk < nch; ++chunk) {
        const bool main_seg = chunk < nch_main;
        const int ntap = main_seg ? TAPS : 1;
        const v4f* pbuf = lds + (chunk & 1) * (NPIX * 8);
        v4f* nbuf = lds + ((chunk + 1) & 1) * (NPIX * 8);
        const bool have_next = chunk + 1 < nch;
        const bool next_is9 = have_next && (TAPS == 9) && (chunk + 1 < nch_main);
        const Src nsc = chunk_src(have_next ? chunk + 1 : chunk);
        {
            const int t0 = main_seg ? 0 : (TAPS / 2);     // single-tap segment: centre tap
            fa0 = a_read(pbuf, a_off(t0, 0), 0);
            fa1 = a_read(pbuf, a_off(t0, 1), 0);
        }
        for (int t = 0; t < ntap; ++t) {
            const bool last_tap = (t == ntap - 1);
            const int tap = main_seg ? t : (TAPS / 2);
            const v4f* wnext = last_tap ? wslab(chunk + 1, 0) : wslab(chunk, t + 1);
            const int n0 = last_tap ? -1 : a_off(t + 1, 0), n1 = last_tap ? -1 : a_off(t + 1, 1);
            STAMP(2)
            // staging of the next chunk's patch around this stage (one call site: one copy of the MFMA stream):
            //   next is a 9-tap chunk -> entry t rides behind tap t's MFMAs (t < NK);
            //   next is single-tap    -> its 4 centre entries travel around this chunk's last stage.
            const bool st9 = next_is9 && (t < NK);
            const bool st1 = have_next && !next_is9 && last_tap;
            v4f e0 = v4f{0.f, 0.f, 0.f, 0.f}, e1 = e0, e2 = e0, e3 = e0;
            bool i0 = false, i1 = false, i2 = false, i3 = false;
            if (st9) e0 = fetch9(nsc, t, i0);
            if (st1) { e0 = fetch1(nsc, 0, i0); e1 = fetch1(nsc, 1, i1); e2 = fetch1(nsc, 2, i2); e3 = fetch1(nsc, 3, i3); }
            stage(a_off(tap, 0), a_off(tap, 1), wnext, pbuf, n0, n1);
            if (st9) put9(nbuf, t, transform(e0, chunk + 1, i0));
            if (st1) {
                put1(nbuf, 0, transform(e0, chunk + 1, i0));
                put1(nbuf, 1, transform(e1, chunk + 1, i1));
                put1(nbuf, 2, transform(e2, chunk + 1, i2));
                put1(nbuf, 3, transform(e3, chunk + 1, i3));
            }
            STAMP(3)
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                tot[i][j] += acc[i][j];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        STAMP(4)
        __syncthreads();   // next patch visible; everyone is done with this one before it is overwritten a chunk later
        STAMP(1)
    }
#undef MFMA16

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
        for (int nt = 0; nt < 2; ++nt) {
            const int co = cb * 128 + 64 * wn + 32 * nt + 4 * cq;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tr[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = tot[mt][nt][r];
            __builtin_amdgcn_wave_barrier();
            const v4f bias = a.bias ? *reinterpret_cast<const v4f*>(a.bias + co) : v4f{0.f, 0.f, 0.f, 0.f};
            v4f val[8], rsd[8];
            size_t oidx[8];
            bool ok[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int p = 8 * i + prow;                           // 0..63: tile pixel (row p>>5, column p&31)
                const int y = y0 + 2 * wm + (p >> 5), x = x0 + (p & 31);
                ok[i] = (y < a.H) && (x < a.W);
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
            for (int i = 0; i < 8; ++i)
                if (ok[i]) *reinterpret_cast<v4f*>(a.out + oidx[i]) = val[i] + bias + rsd[i];
            __builtin_amdgcn_wave_barrier();
        }
    }
#ifdef CDDPM_STAMPS
    STAMP(5)
    if (a.stamps && lane == 0)
        for (int i = 0; i < 6; ++i) atomicAdd(&a.stamps[wave * 8 + i], st_[i]);
#endif
}

void launch_conv(const ConvArgs& a, hipStream_t stream) {
    const int tilesX = (a.W + 31) / 32, tilesY = (a.H + 3) / 4;
    const unsigned grid = (unsigned)(a.B * tilesX * tilesY * (a.Cout / 128));
    size_t coef_lds = a.coef ? (size_t)3 * (a.C0 + a.C1) * sizeof(float) : 0;
#ifdef CDDPM_ONE_WG_PER_CU
    coef_lds += 40 * 1024;   // diagnostic builds: force one workgroup per CU (one wave per SIMD)
#endif
    if (a.taps == 9) {
        const size_t lds = (size_t)(2 * 6 * 34 * 8) * 16 + coef_lds;      // 52.2 KB + <= 6 KB -> 2 workgroups per CU
        hipLaunchKernelGGL(conv_mfma_kernel<9>, dim3(grid), dim3(256), lds, stream, a);
    } else {
        const size_t lds = (size_t)(2 * 4 * 32 * 8) * 16 + coef_lds;
        hipLaunchKernelGGL(conv_mfma_kernel<1>, dim3(grid), dim3(256), lds, stream, a);
    }
}

size_t packed_conv_floats(int Cout, int Cin, int taps) { return (size_t)Cout * Cin * taps; }

// w: PyTorch [Cout][Cin][k][k] (taps = k*k, tap = ky*3+kx) -> stage images [Cout/128][Cin/32][taps] of 1024 float4:
// [wn 2][nt 2][g 4][lh 2][li 32][4]  <->  cout = 128 cb + 64 wn + 32 nt + li,  ci = 32 chunk + 8 g + 4 lh + e.
// One (wn, nt, g) piece is the 1 KiB a wave fetches with a single global_load_dwordx4 (lane = 32 lh + li).
void pack_conv_weights(const float* w, int Cout, int Cin, int taps, float* dst) {
    const int ncb = Cout / 128, nch = Cin / 32;
    for (int cb = 0; cb < ncb; ++cb)
        for (int ch = 0; ch < nch; ++ch)
            for (int t = 0; t < taps; ++t) {
                float* img = dst + (((size_t)cb * nch + ch) * taps + t) * 4096;
                for (int wn = 0; wn < 2; ++wn)
                    for (int nt = 0; nt < 2; ++nt)
                        for (int g = 0; g < 4; ++g)
                            for (int lh = 0; lh < 2; ++lh)
                                for (int li = 0; li < 32; ++li) {
                                    float* d4 = img + ((((size_t)(wn * 2 + nt) * 4 + g) * 2 + lh) * 32 + li) * 4;
                                    const int co = cb * 128 + 64 * wn + 32 * nt + li;
                                    for (int e = 0; e < 4; ++e) {
                                        const int ci = ch * 32 + 8 * g + 4 * lh + e;
                                        d4[e] = w[((size_t)co * Cin + ci) * taps + t];
                                    }
                                }
            }
}

}  // namespace cddpm
