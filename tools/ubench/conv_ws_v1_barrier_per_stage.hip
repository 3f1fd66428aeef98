// Wave-specialised, persistent variant of the fused implicit-GEMM convolution (same math, same packed weights and
// the same epilogue as conv_mfma.hip; see that file for the GEMM view and what is fused).
//
// Why: with two equal workgroups per CU, conv_mfma.hip keeps the matrix pipe ~80 % busy -- every wave alternates
// between MFMA work and staging/barrier/epilogue work, and the co-resident wave hides only part of it. A wave that
// does nothing but ds_read_b128 + MFMA sustains 98 % of the fp32 matrix peak (tools/ubench/mfma_ceiling.hip).
//
// Layout: one 512-thread workgroup per CU, persistent over output tiles (tile = blockIdx.x + k * gridDim.x).
//   waves 0-3  "matrix waves": one per SIMD; A fragments from the patch buffer, B fragments from the weight ring,
//              64 MFMAs per stage, chunk fold, and at the end of a tile the transposed 16-B epilogue (+ GroupNorm
//              statistics) through a private LDS region.
//   waves 4-7  "staging waves": one per SIMD; copy the packed weight image of stage s+2 into a 3-slot LDS ring
//              (global loads issued one stage earlier), and stage the NEXT chunk's activation patch (global load ->
//              GroupNorm/FiLM affine -> SiLU -> swizzled ds_write) into the other patch buffer, one entry per tap.
//              They run ahead across tile boundaries, so a tile's prologue is hidden behind the previous tile.
//   one workgroup barrier per stage (= per 64 MFMAs of each matrix wave).
// LDS: 2 x 26.1 KB patch + 3 x 16 KB weights + 32 KB transpose = 131 KB.
#include "kernels.h"

namespace cddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_ws(float v) {
    // identical evaluation to conv_mfma.hip::silu_f (split-product exp2, ~1.5 ulp)
    const float t = fminf(-v * 1.44269502162933349609375f, 126.0f);
    float tl = __builtin_fmaf(-v, 1.44269502162933349609375f, -t);
    tl = __builtin_fmaf(-v, 1.925963033500011e-08f, tl);
    tl = (t < 126.0f) ? tl : 0.0f;
    float e = __builtin_amdgcn_exp2f(t);
    e = __builtin_fmaf(e, tl * 0.693147180559945f, e);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

template <int TAPS>
__global__ __launch_bounds__(512, 2) void conv_ws_kernel(const ConvArgs a, int ntiles) {
    constexpr int PAD = (TAPS == 9) ? 1 : 0;
    constexpr int PW = 32 + 2 * PAD;
    constexpr int PH = 4 + 2 * PAD;
    constexpr int NPIX = PW * PH;                 // 204 | 128
    constexpr int NE9 = (NPIX * 8 + 255) / 256;   // patch entries per staging thread, 9-tap chunk: 7
    constexpr int NE1 = 4;                        // single-tap chunk (centre pixels only): 128 * 8 / 256
    constexpr int NEMAX = (TAPS == 9) ? NE9 : NE1;

    extern __shared__ v4f lds[];
    v4f* ldsP = lds;                                   // 2 patch buffers of NPIX * 8
    v4f* ldsW = lds + 2 * NPIX * 8;                    // 3 weight slots of 1024
    float* ldsT = reinterpret_cast<float*>(ldsW + 3 * 1024);   // 4 x [64 pixels][32 channels]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    const int ncb = a.Cout >> 7;
    const int tilesX = (a.W + 31) >> 5;
    const int tilesY = (a.H + 3) >> 2;
    const int Cin = a.C0 + a.C1;
    const int nch_main = Cin >> 5;
    const int nch_skip = (a.S0 + a.S1) >> 5;
    const int nch = nch_main + nch_skip;
    const int S = nch_main * TAPS + nch_skip;          // stages per tile
    const int ntl = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup
    const int total = ntl * S;                          // stages of this workgroup

    struct Tile { int cb, tx, ty, b; };
    auto decode = [&](int k) -> Tile {
        int t = blockIdx.x + k * gridDim.x;
        Tile r;
        r.cb = t % ncb; t /= ncb;
        r.tx = t % tilesX; t /= tilesX;
        r.ty = t % tilesY;
        r.b = t / tilesY;
        return r;
    };

    if (wave >= 4) {
        // =============================== staging waves ===============================================
        // They have little to do per stage but share a SIMD's issue port with a matrix wave that always has an MFMA
        // ready; at equal priority they are starved and become the critical path of the per-stage barrier.
        __builtin_amdgcn_s_setprio(3);
        const int ltid = tid - 256;
        const int s = ltid & 7;
        const size_t coef_plane = (size_t)a.B * Cin;

        // Stage cursor (tile ordinal k, chunk c, tap t), advanced incrementally: the staging waves share their SIMDs
        // with the matrix waves, so their per-stage bookkeeping must stay at a handful of scalar ops (no divisions).
        struct Cur { int k, c, t; Tile tl; };
        auto cur_init = [&]() -> Cur { Cur r; r.k = 0; r.c = 0; r.t = 0; r.tl = decode(0); return r; };
        auto cur_next = [&](Cur& r) {
            const int ntap = (r.c < nch_main) ? TAPS : 1;
            if (++r.t == ntap) {
                r.t = 0;
                if (++r.c == nch) { r.c = 0; ++r.k; r.tl = decode(r.k); }
            }
        };
        auto wptr = [&](const Cur& r) -> const v4f* {
            return (r.c < nch_main)
                       ? reinterpret_cast<const v4f*>(a.wpk) + (((size_t)r.tl.cb * nch_main + r.c) * TAPS + r.t) * 1024
                       : reinterpret_cast<const v4f*>(a.skip_wpk) + ((size_t)r.tl.cb * nch_skip + (r.c - nch_main)) * 1024;
        };
        // What the staging waves commit to LDS during the stage at cursor r: entries [first, first+count) of the patch
        // of the NEXT chunk (c2 of tile ordinal k2, tile tl2), into patch buffer `buf`.
        struct Plan { int k2, c2, first, count, buf; Tile tl2; };
        auto plan = [&](const Cur& r, const Tile& next_tile) -> Plan {
            Plan p;
            p.k2 = r.k; p.c2 = r.c + 1; p.tl2 = r.tl; p.first = 0; p.count = 0;
            if (p.c2 >= nch) { p.k2 = r.k + 1; p.c2 = 0; p.tl2 = next_tile; }
            p.buf = (r.k * nch + r.c + 1) & 1;
            if (r.k >= ntl || p.k2 >= ntl) return p;
            const bool cur9 = (r.c < nch_main) && (TAPS == 9);
            const int ne_next = ((p.c2 < nch_main) && (TAPS == 9)) ? NE9 : NE1;
            if (cur9) { p.first = r.t; p.count = (r.t < ne_next) ? 1 : 0; }
            else { p.first = 0; p.count = ne_next; }
            return p;
        };
        struct Ent { v4f e[NEMAX]; int q[NEMAX]; bool in[NEMAX]; v4f cm, ca, cd; };   // entries + their GroupNorm coefficients
        // issue the global loads of a plan's entries (no wait): q = patch pixel (-1: nothing), in = real image pixel
        auto fetch = [&](const Plan& p, Ent& r) {
            const Tile tl = p.tl2;
            const int y0 = tl.ty * 4, x0 = tl.tx * 32;
            const bool main_seg = p.c2 < nch_main;
            const bool nine = main_seg && (TAPS == 9);
            const float* base;
            int Cs, c0;
            if (main_seg) {
                const int ch = p.c2 << 5;
                if (ch < a.C0) { base = a.src0; Cs = a.C0; c0 = ch; }
                else           { base = a.src1; Cs = a.C1; c0 = ch - a.C0; }
            } else {
                const int ch = (p.c2 - nch_main) << 5;
                if (ch < a.S0) { base = a.skip0; Cs = a.S0; c0 = ch; }
                else           { base = a.skip1; Cs = a.S1; c0 = ch - a.S0; }
            }
#pragma unroll
            for (int j = 0; j < NEMAX; ++j) {
                const int kk = p.first + j;
                r.q[j] = -1;
                r.in[j] = false;
                r.e[j] = v4f{0.f, 0.f, 0.f, 0.f};
                if (j < p.count) {
                    int y, x, qq;
                    if (nine) {
                        qq = (ltid >> 3) + 32 * kk;
                        const int pr = qq / PW, pc = qq - pr * PW;
                        y = y0 + pr - PAD; x = x0 + pc - PAD;
                        if (qq >= NPIX) qq = -1;
                    } else {
                        const int cp = (ltid >> 3) + 32 * kk;           // 0..127 centre pixel
                        y = y0 + (cp >> 5); x = x0 + (cp & 31);
                        qq = ((cp >> 5) + PAD) * PW + (cp & 31) + PAD;
                    }
                    r.q[j] = qq;
                    r.in[j] = (qq >= 0) && (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
#ifdef CDDPM_WS_NOLOAD
                    if (false) {
#else
                    if (r.in[j]) {
#endif
                        // main segment sources may be upsampled; skip-segment sources live at the output resolution
                        const bool up = main_seg && a.upsample;
                        const int sy = up ? (y >> 1) : y, sx = up ? (x >> 1) : x;
                        const int sH = main_seg ? a.srcH : a.H, sW = main_seg ? a.srcW : a.W;
                        r.e[j] = *reinterpret_cast<const v4f*>(base + (size_t)((tl.b * sH + sy) * sW + sx) * Cs + c0 + 4 * s);
                    }
                }
            }
            // the coefficients travel with the entries: loaded at commit time they would be the youngest loads in
            // flight and the in-order vmcnt wait for them would drain the whole prefetch
            r.cm = v4f{0.f, 0.f, 0.f, 0.f}; r.ca = v4f{1.f, 1.f, 1.f, 1.f}; r.cd = r.cm;
#ifdef CDDPM_WS_NOLOAD
            if (false) {
#else
            if (p.count > 0 && main_seg && a.coef) {
#endif
                const size_t ci = (size_t)tl.b * Cin + (p.c2 << 5) + 4 * s;
                r.cm = *reinterpret_cast<const v4f*>(a.coef + ci);
                r.ca = *reinterpret_cast<const v4f*>(a.coef + coef_plane + ci);
                r.cd = *reinterpret_cast<const v4f*>(a.coef + 2 * coef_plane + ci);
            }
        };
        // GroupNorm/FiLM affine + SiLU, swizzled write of previously fetched entries
        auto commit = [&](const Plan& p, const Ent& r) {
            if (p.count == 0) return;
            const bool main_seg = p.c2 < nch_main;
            const v4f cm = r.cm, ca = r.ca, cd = r.cd;
            const bool aff = main_seg && a.coef;
            const bool do_silu = main_seg && a.silu;
            v4f* pbuf = ldsP + p.buf * (NPIX * 8);
#pragma unroll
            for (int j = 0; j < NEMAX; ++j) {
                if (r.q[j] >= 0) {
                    v4f v = r.e[j];
                    if (r.in[j]) {   // zero padding stays exactly zero: the conv pads AFTER the activation
                        if (aff) v = (v - cm) * ca + cd;
                        if (do_silu) { v.x = silu_ws(v.x); v.y = silu_ws(v.y); v.z = silu_ws(v.z); v.w = silu_ws(v.w); }
                    }
                    pbuf[r.q[j] * 8 + (s ^ ((r.q[j] >> 1) & 7))] = v;
                }
            }
        };

        v4f wreg[4];
        Ent pend;
        // prologue: weight slots 0 and 1, the whole patch of the first chunk; then the requests stage 0 will consume
        Cur cw = cur_init();                  // cursor of the stage whose weights are fetched next
        for (int g = 0; g < 2; ++g) {
            if (g < total) {
                const v4f* p = wptr(cw);
#pragma unroll
                for (int i = 0; i < 4; ++i) ldsW[g * 1024 + ltid + 256 * i] = p[ltid + 256 * i];
                if (g + 1 < total) cur_next(cw);
            }
        }
        Cur cp = cur_init();                  // cursor of the stage whose patch work is requested next
        {
            Plan p0;
            p0.k2 = 0; p0.c2 = 0; p0.first = 0; p0.count = (TAPS == 9) ? NE9 : NE1; p0.buf = 0; p0.tl2 = cp.tl;
            fetch(p0, pend);
            commit(p0, pend);
        }
        // cw now points at stage min(2, total-1); cp at stage 0
        {
            const v4f* p = wptr(cw);
#pragma unroll
            for (int i = 0; i < 4; ++i) wreg[i] = p[ltid + 256 * i];
        }
        Tile ntile = decode(1 < ntl ? 1 : 0);  // tile after cp's
        Plan pcur = plan(cp, ntile);
        fetch(pcur, pend);
        __syncthreads();
        // Steady state, per stage: FIRST consume what was requested one stage ago (its loads had a whole stage of
        // MFMA time to land, so the in-order vmcnt wait costs nothing), THEN request what the next stage consumes.
#ifdef CDDPM_STAMPS
        unsigned long long sw_ = 0, sb_ = 0, sl_ = __builtin_amdgcn_s_memtime();
#endif
        for (int gs = 0; gs < total; ++gs) {
            if (gs + 2 < total) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ldsW[((gs + 2) % 3) * 1024 + ltid + 256 * i] = wreg[i];
            }
            commit(pcur, pend);
            __builtin_amdgcn_sched_barrier(0);
            if (gs + 3 < total) cur_next(cw);   // else: the tail re-reads the last image (harmless)
#ifndef CDDPM_WS_NOLOAD
            {
                const v4f* p = wptr(cw);
#pragma unroll
                for (int i = 0; i < 4; ++i) wreg[i] = p[ltid + 256 * i];
            }
#endif
            if (gs + 1 < total) {
                const int kprev = cp.k;
                cur_next(cp);
                if (cp.k != kprev) ntile = decode(cp.k + 1 < ntl ? cp.k + 1 : cp.k);
                pcur = plan(cp, ntile);
            } else {
                pcur.count = 0;
            }
            fetch(pcur, pend);
#ifdef CDDPM_STAMPS
            { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sw_ += n_ - sl_; sl_ = n_; }
#endif
            __syncthreads();
#ifdef CDDPM_STAMPS
            { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sb_ += n_ - sl_; sl_ = n_; }
#endif
        }
#ifdef CDDPM_STAMPS
        if (a.stamps && lane == 0) { atomicAdd(&a.stamps[2], sw_); atomicAdd(&a.stamps[3], sb_); }
#endif
        return;
    }

    // ==================================== matrix waves ===================================================
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int wm = wave & 1;    // pixel rows {0,1} | {2,3}
    const int wn = wave >> 1;   // cout 0..63 | 64..127

    // weights: row j = cout within the 128 block, slot (2g + lh) ^ ((j >> 1) & 7)   (pack_conv_weights image)
    int boff[2], bsw[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int j = 64 * wn + 32 * nt + li;
        boff[nt] = j * 8;
        bsw[nt] = (j >> 1) & 7;
    }
    auto a_off = [&](int tap, int mt) -> int {   // v4f offset of this lane's pixel row for a tap; swizzle key in bits 28..30
        const int ky = (TAPS == 9) ? (tap / 3) : 0;
        const int kx = (TAPS == 9) ? (tap - 3 * ky) : 0;
        const int q = (2 * wm + mt + ky) * PW + li + kx;
        return (q * 8) | (((q >> 1) & 7) << 28);
    };
    auto a_read = [&](const v4f* pbuf, int off, int g) -> v4f {
        return pbuf[(off & 0x0fffffff) + ((2 * g + lh) ^ (off >> 28))];
    };
    auto b_read = [&](const v4f* wb, int nt, int g) -> v4f { return wb[boff[nt] + ((2 * g + lh) ^ bsw[nt])]; };

    f32x16 acc[2][2], tot[2][2];
    v4f fa0, fa1, fb0, fb1;     // fragment set "x": A rows mt=0,1 ; B tiles nt=0,1
    v4f ga0, ga1, gb0, gb1;     // fragment set "y"

#define WS_MFMA16(A0, A1, B0, B1)                                                                          \
    {                                                                                                      \
        const float av0_[4] = {A0.x, A0.y, A0.z, A0.w};                                                    \
        const float av1_[4] = {A1.x, A1.y, A1.z, A1.w};                                                    \
        const float bv0_[4] = {B0.x, B0.y, B0.z, B0.w};                                                    \
        const float bv1_[4] = {B1.x, B1.y, B1.z, B1.w};                                                    \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                    \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0_[m], bv0_[m], acc[0][0], 0, 0, 0);        \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0_[m], bv1_[m], acc[0][1], 0, 0, 0);        \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1_[m], bv0_[m], acc[1][0], 0, 0, 0);        \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1_[m], bv1_[m], acc[1][1], 0, 0, 0);        \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
#define WS_READ(SET_A0, SET_A1, SET_B0, SET_B1, PB, O0, O1, WB, G)                                          \
    {                                                                                                      \
        SET_A0 = a_read(PB, O0, G); SET_A1 = a_read(PB, O1, G);                                            \
        SET_B0 = b_read(WB, 0, G);  SET_B1 = b_read(WB, 1, G);                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }

    __syncthreads();   // pairs with the staging waves' prologue barrier
#ifdef CDDPM_STAMPS
    unsigned long long mw_ = 0, mb_ = 0, ml_ = __builtin_amdgcn_s_memtime();
#endif
    int gs = 0;
    for (int k = 0; k < ntl; ++k) {
        const Tile tl = decode(k);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
        for (int c = 0; c < nch; ++c) {
            const bool main_seg = c < nch_main;
            const int ntap = main_seg ? TAPS : 1;
            const v4f* pbuf = ldsP + ((k * nch + c) & 1) * (NPIX * 8);
            for (int t = 0; t < ntap; ++t, ++gs) {
                const int tap = main_seg ? t : (TAPS / 2);     // single-tap segment: centre tap
                const v4f* wb = ldsW + (gs % 3) * 1024;
                const int o0 = a_off(tap, 0), o1 = a_off(tap, 1);
                if (t == 0) WS_READ(fa0, fa1, fb0, fb1, pbuf, o0, o1, wb, 0)     // else prefetched by the previous stage
                WS_READ(ga0, ga1, gb0, gb1, pbuf, o0, o1, wb, 1)
                WS_MFMA16(fa0, fa1, fb0, fb1)
                WS_READ(fa0, fa1, fb0, fb1, pbuf, o0, o1, wb, 2)
                WS_MFMA16(ga0, ga1, gb0, gb1)
                WS_READ(ga0, ga1, gb0, gb1, pbuf, o0, o1, wb, 3)
                WS_MFMA16(fa0, fa1, fb0, fb1)
                if (t + 1 < ntap) {
                    // group 0 of the next stage: same patch buffer, next weight slot (written >= 1 barrier ago)
                    const v4f* wbn = ldsW + ((gs + 1) % 3) * 1024;
                    WS_READ(fa0, fa1, fb0, fb1, pbuf, a_off(tap + 1, 0), a_off(tap + 1, 1), wbn, 0)
                }
                WS_MFMA16(ga0, ga1, gb0, gb1)
                if (t == ntap - 1) {
                    // two-level accumulation (see conv_mfma.hip): fold the chunk, restart the inner chains
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            tot[i][j] += acc[i][j];
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                        }
                    if (c == nch - 1) {
                        // ---- epilogue of the tile (same as conv_mfma.hip, private transpose region)
                        float* tr = ldsT + wave * 2048;
                        const int cq = lane & 7, prow = lane >> 3;
                        const int y0 = tl.ty * 4, x0 = tl.tx * 32;
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const int co = tl.cb * 128 + 64 * wn + 32 * nt + 4 * cq;
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                                for (int r = 0; r < 16; ++r)
                                    tr[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = tot[mt][nt][r];
                            __builtin_amdgcn_wave_barrier();
                            const v4f bias = a.bias ? *reinterpret_cast<const v4f*>(a.bias + co) : v4f{0.f, 0.f, 0.f, 0.f};
                            v4f ssum = v4f{0.f, 0.f, 0.f, 0.f}, ssq = ssum;
#pragma unroll
                            for (int hb = 0; hb < 2; ++hb) {
                                v4f val[4], rsd[4];
                                size_t oidx[4];
                                bool ok[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int p = 8 * (4 * hb + i) + prow;
                                    const int y = y0 + 2 * wm + (p >> 5), x = x0 + (p & 31);
                                    ok[i] = (y < a.H) && (x < a.W);
                                    oidx[i] = ((size_t)(tl.b * a.H + y) * a.W + x) * a.Cout + co;
                                    rsd[i] = v4f{0.f, 0.f, 0.f, 0.f};
                                    if (a.res && ok[i]) {
                                        const size_t rp = a.res_up
                                            ? ((size_t)(tl.b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1))
                                            : ((size_t)(tl.b * a.H + y) * a.W + x);
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
                            if (a.stats) {
#pragma unroll
                                for (int m = 8; m < 64; m <<= 1) {
                                    ssum.x += __shfl_xor(ssum.x, m, 64); ssum.y += __shfl_xor(ssum.y, m, 64);
                                    ssum.z += __shfl_xor(ssum.z, m, 64); ssum.w += __shfl_xor(ssum.w, m, 64);
                                    ssq.x += __shfl_xor(ssq.x, m, 64); ssq.y += __shfl_xor(ssq.y, m, 64);
                                    ssq.z += __shfl_xor(ssq.z, m, 64); ssq.w += __shfl_xor(ssq.w, m, 64);
                                }
                                if (prow == 0) {
                                    const int nrec = 2 * tilesX * tilesY;
                                    const int rec = 2 * (tl.ty * tilesX + tl.tx) + wm;
                                    float* o = a.stats + (((size_t)tl.b * nrec + rec) * a.Cout + co) * 2;
                                    *reinterpret_cast<v4f*>(o) = v4f{ssum.x, ssq.x, ssum.y, ssq.y};
                                    *reinterpret_cast<v4f*>(o + 4) = v4f{ssum.z, ssq.z, ssum.w, ssq.w};
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                }
#ifdef CDDPM_STAMPS
                { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); mw_ += n_ - ml_; ml_ = n_; }
#endif
                __syncthreads();   // one per stage, paired with the staging waves' loop
#ifdef CDDPM_STAMPS
                { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); mb_ += n_ - ml_; ml_ = n_; }
#endif
            }
        }
    }
#ifdef CDDPM_STAMPS
    if (a.stamps && lane == 0) { atomicAdd(&a.stamps[0], mw_); atomicAdd(&a.stamps[1], mb_); }
#endif
#undef WS_MFMA16
#undef WS_READ
}

static int ws_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

void launch_conv_ws(const ConvArgs& a, hipStream_t stream) {
    const int tilesX = (a.W + 31) / 32, tilesY = (a.H + 3) / 4;
    const int ntiles = a.B * tilesX * tilesY * (a.Cout / 128);
    const int grid = ntiles < ws_num_cus() ? ntiles : ws_num_cus();
    static bool attr_set = false;
    const size_t lds9 = (size_t)(2 * 6 * 34 * 8 + 3 * 1024) * 16 + 4 * 2048 * 4;
    const size_t lds1 = (size_t)(2 * 4 * 32 * 8 + 3 * 1024) * 16 + 4 * 2048 * 4;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ws_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds9);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ws_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        attr_set = true;
    }
    if (a.taps == 9) hipLaunchKernelGGL(conv_ws_kernel<9>, dim3(grid), dim3(512), lds9, stream, a, ntiles);
    else hipLaunchKernelGGL(conv_ws_kernel<1>, dim3(grid), dim3(512), lds1, stream, a, ntiles);
}

}  // namespace cddpm
