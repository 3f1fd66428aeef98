// Wave-specialised, persistent variant of the fused implicit-GEMM convolution (same math, same packed weights, same
// epilogue as conv_mfma.hip; see that file for the GEMM view and what is fused). EXPERIMENTAL: selected with
// CDDPM_CONV_WS=1, never the default unless it measures faster (tools/conv_ab.py). The first version of this file
// (one workgroup barrier per stage, 94 TF) is kept as tools/ubench/conv_ws_v1_barrier_per_stage.hip.
//
// Why: with two equal workgroups per CU, conv_mfma.hip keeps the matrix pipe ~80 % busy -- every wave alternates
// between MFMA work and staging / barrier / epilogue work and the co-resident wave hides little of it (ablations in
// DESIGN.md section 4). A wave that does nothing but ds_read_b128 + MFMA sustains 92-98 % (tools/ubench/mfma_ceiling.hip).
//
// Layout: one 512-thread workgroup per CU, persistent over output tiles (tile = blockIdx.x + k * gridDim.x).
//   waves 0-3  "matrix waves", one per SIMD: A fragments from the patch buffer, B fragments from the weight ring,
//              64 MFMAs per stage, chunk fold, transposed 16-B epilogue (+ GroupNorm statistics) at the end of a tile.
//   waves 4-7  "staging waves", one per SIMD: weight slab of stage i+2 registers -> LDS ring (global loads issued two
//              stages earlier), patch entries of the NEXT chunk (global load one stage earlier -> GroupNorm/FiLM
//              affine -> SiLU -> swizzled ds_write). They run ahead across tile boundaries.
// Synchronisation: NO workgroup barriers in the steady state. Progress counters in LDS, one word per wave, written by
// lane 0 after `s_waitcnt lgkmcnt(0)` and polled with one ds_read_b128 (the LDS executes a CU's requests in order, so a
// reader that sees a counter value sees every LDS write its owner completed before publishing it):
//   prodW[s] = weight slabs staging wave s has completed      (matrix stage g needs min >= g + 1)
//   prodP[s] = patches (global chunk count) wave s completed  (first stage of global chunk c needs min >= c + 1)
//   cons[m]  = stages matrix wave m has finished reading      (ring-slot / patch-buffer reuse)
// LDS: 3 x 26.1 KB patch + 3 x 16 KB weight ring + 32 KB transpose + counters = 156.6 KB.
#include "kernels.h"

namespace cddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

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

constexpr int WS_NW = 3;    // weight ring slots
constexpr int WS_NP = 3;    // patch buffers
constexpr int WS_LEAD = WS_NW - 1;   // staging iteration i writes the slab of stage i + LEAD
constexpr int WS_PLEAD = 2;          // ... and requests patch entries for the chunk after the one holding stage i + PLEAD (needs NP >= 3)

template <int TAPS>
__global__ __launch_bounds__(512, 2) void conv_ws_kernel(const ConvArgs a, int ntiles) {
    constexpr int PAD = (TAPS == 9) ? 1 : 0;
    constexpr int PW = 32 + 2 * PAD;
    constexpr int PH = 4 + 2 * PAD;
    constexpr int NPIX = PW * PH;                 // 204 | 128
    constexpr int NE9 = (NPIX * 8 + 255) / 256;   // patch entries per staging thread, 9-tap chunk: 7
    constexpr int NE1 = 4;                        // single-tap chunk (centre pixels only): 128 * 8 / 256

    extern __shared__ v4f lds[];
    v4f* ldsP = lds;                                               // WS_NP patch buffers of NPIX * 8
    v4f* ldsW = lds + WS_NP * NPIX * 8;                            // WS_NW weight slots of 1024
    float* ldsT = reinterpret_cast<float*>(ldsW + WS_NW * 1024);   // 4 x [64 pixels][32 channels]
    // progress counters: three groups of four words behind the transpose region. They are accessed with inline
    // ds_read_b128 / ds_write_b32 on their byte address (the kernel has no static LDS, so dynamic LDS starts at 0):
    // `volatile` accesses would make the compiler drain vmcnt around every poll and serialise the staging pipeline.
    constexpr unsigned SYNC_ADDR = (WS_NP * NPIX * 8 + WS_NW * 1024) * 16 + 4 * 2048 * 4 + 64;
    constexpr unsigned prodW = SYNC_ADDR, prodP = SYNC_ADDR + 16, cons = SYNC_ADDR + 32;     // byte addresses
    auto ws_min4 = [&](unsigned grp) -> int {
        v4i v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(grp) : "memory");
        return min(min(v.x, v.y), min(v.z, v.w));
    };
    // wait until all four counters of a group reach `target`; returns the cycles spent when `timed`
    auto ws_wait = [&](unsigned grp, int target, bool timed_) -> unsigned long long {
        if (ws_min4(grp) >= target) return 0;
        const unsigned long long t0 = timed_ ? __builtin_amdgcn_s_memtime() : 0ull;
        while (ws_min4(grp) < target) __builtin_amdgcn_s_sleep(1);
        return timed_ ? __builtin_amdgcn_s_memtime() - t0 : 0ull;
    };
    // publish a counter value after every earlier LDS access of this wave has completed
    auto ws_publish = [&](unsigned addr, int value) {
        if ((threadIdx.x & 63) == 0)
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" : : "v"(addr), "v"(value) : "memory");
    };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // kernel arguments used in the loops, copied once (keeps them out of the steady-state scalar loads)
    const int H = a.H, W = a.W, Cout = a.Cout, C0 = a.C0, C1 = a.C1, S0 = a.S0, S1 = a.S1;
    const int ncb = Cout >> 7;
    const int tilesX = (W + 31) >> 5;
    const int tilesY = (H + 3) >> 2;
    const int Cin = C0 + C1;
    const int nch_main = Cin >> 5;
    const int nch_skip = (S0 + S1) >> 5;
    const int nch = nch_main + nch_skip;
    const int S = nch_main * TAPS + nch_skip;          // stages per tile
    const int ntl = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup
    const int total = ntl * S;                          // stages of this workgroup
    const int nchunks = ntl * nch;                      // patches of this workgroup
    const bool timed = a.stamps != nullptr && blockIdx.x == 0;
    unsigned long long t_wait = 0;
    const unsigned long long t_begin = timed ? __builtin_amdgcn_s_memtime() : 0ull;

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

    if (tid < 12) reinterpret_cast<int*>(lds)[SYNC_ADDR / 4 + tid] = 0;
    __syncthreads();     // the only workgroup barrier

    if (wave >= 4) {
        // =============================== staging waves ===============================================
#ifndef WS_PRIO_S
#define WS_PRIO_S 3
#endif
#ifndef WS_PRIO_M
#define WS_PRIO_M 0
#endif
        __builtin_amdgcn_s_setprio(WS_PRIO_S);
        const int sw = wave - 4;
        const int ltid = tid - 256;
        const int s = ltid & 7;
        const int prow = ltid >> 3;                    // 0..31
        const size_t coef_plane = (size_t)a.B * Cin;
        const float* src0 = a.src0; const float* src1 = a.src1;
        const float* skip0 = a.skip0; const float* skip1 = a.skip1;
        const float* coefp = a.coef;
        const v4f* wpk = reinterpret_cast<const v4f*>(a.wpk);
        const v4f* swpk = reinterpret_cast<const v4f*>(a.skip_wpk);
        const bool do_up = a.upsample != 0, do_silu = a.silu != 0;
        const int srcH = a.srcH, srcW = a.srcW;

        // ---- weight cursor: stage -> packed slab pointer, advanced incrementally; sticks at the last stage
        struct WCur { int i, k, c, t, cb; };
        auto wc_ptr = [&](const WCur& r) -> const v4f* {
            return (r.c < nch_main) ? wpk + (((size_t)r.cb * nch_main + r.c) * TAPS + r.t) * 1024
                                    : swpk + ((size_t)r.cb * nch_skip + (r.c - nch_main)) * 1024;
        };
        auto wc_next = [&](WCur& r) {
            if (r.i + 1 >= total) return;
            ++r.i;
            const int ntap = (r.c < nch_main) ? TAPS : 1;
            if (++r.t == ntap) {
                r.t = 0;
                if (++r.c == nch) { r.c = 0; ++r.k; r.cb = decode(r.k).cb; }
            }
        };
        // ---- patch cursor: the patch being requested = global chunk gc (tile ordinal k, chunk c), next entry e of ne
        struct PCur { int gc, k, c, e, ne; Tile tl; const float* base; int Cs, c0; bool main_seg; };
        auto pc_setup = [&](PCur& r) {
            r.main_seg = r.c < nch_main;
            if (r.main_seg) {
                const int ch = r.c << 5;
                if (ch < C0) { r.base = src0; r.Cs = C0; r.c0 = ch; }
                else         { r.base = src1; r.Cs = C1; r.c0 = ch - C0; }
            } else {
                const int ch = (r.c - nch_main) << 5;
                if (ch < S0) { r.base = skip0; r.Cs = S0; r.c0 = ch; }
                else         { r.base = skip1; r.Cs = S1; r.c0 = ch - S0; }
            }
            r.ne = (r.main_seg && TAPS == 9) ? NE9 : NE1;
            r.e = 0;
        };
        auto pc_next_chunk = [&](PCur& r) {
            ++r.gc;
            if (r.gc >= nchunks) return;
            if (++r.c == nch) { r.c = 0; ++r.k; r.tl = decode(r.k); }
            pc_setup(r);
        };
        // stages before global chunk g (chunks are TAPS-stage main chunks followed by 1-stage skip chunks)
        auto stages_before = [&](int g) -> int {
            const int k = g / nch, c = g - k * nch;
            return k * S + (c <= nch_main ? c * TAPS : nch_main * TAPS + (c - nch_main));
        };

        // one patch entry in flight: raw values + where they go
        struct Ent { v4f v; int q; bool in; bool act; int gc; bool last; };
        auto fetch = [&](const PCur& r) -> Ent {
            Ent o;
            const int e = r.e;
            int y, x;
            if (r.main_seg && TAPS == 9) {
                o.q = prow + 32 * e;
                const int pr = o.q / PW, pc = o.q - pr * PW;
                y = r.tl.ty * 4 + pr - PAD; x = r.tl.tx * 32 + pc - PAD;
                if (o.q >= NPIX) o.q = -1;
            } else {
                const int cp = prow + 32 * e;                   // 0..127 centre pixel
                y = r.tl.ty * 4 + (cp >> 5); x = r.tl.tx * 32 + (cp & 31);
                o.q = ((cp >> 5) + PAD) * PW + (cp & 31) + PAD;
            }
            o.in = (o.q >= 0) && (y >= 0) && (y < H) && (x >= 0) && (x < W);
            // main-segment sources may be upsampled; skip-segment sources live at the output resolution
            const bool up = r.main_seg && do_up;
            const int sy = up ? (y >> 1) : y, sx = up ? (x >> 1) : x;
            const int sH = r.main_seg ? srcH : H, sWd = r.main_seg ? srcW : W;
            const size_t off = o.in ? ((size_t)((r.tl.b * sH + sy) * sWd + sx) * r.Cs + r.c0 + 4 * s) : (size_t)(4 * s);
            o.v = *reinterpret_cast<const v4f*>(r.base + off);      // unconditional load (clamped address)
            o.act = r.main_seg;
            o.gc = r.gc;
            o.last = (e == r.ne - 1);
            return o;
        };
        auto load_coef = [&](const PCur& r, v4f& cm, v4f& ca, v4f& cd) {
            cm = v4f{0.f, 0.f, 0.f, 0.f}; ca = v4f{1.f, 1.f, 1.f, 1.f}; cd = cm;
            if (coefp && r.main_seg) {
                const size_t ci = (size_t)r.tl.b * Cin + (r.c << 5) + 4 * s;
                cm = *reinterpret_cast<const v4f*>(coefp + ci);
                ca = *reinterpret_cast<const v4f*>(coefp + coef_plane + ci);
                cd = *reinterpret_cast<const v4f*>(coefp + 2 * coef_plane + ci);
            }
        };
        // transform + swizzled write of one entry into its patch buffer
        auto commit = [&](const Ent& en, const v4f& cm, const v4f& ca, const v4f& cd) {
            if (en.q < 0) return;
            v4f v = v4f{0.f, 0.f, 0.f, 0.f};
            if (en.in) {   // zero padding stays exactly zero: the conv pads AFTER the activation
                v = en.v;
                if (en.act) {
                    if (coefp) v = (v - cm) * ca + cd;
                    if (do_silu) { v.x = silu_ws(v.x); v.y = silu_ws(v.y); v.z = silu_ws(v.z); v.w = silu_ws(v.w); }
                }
            }
            ldsP[(en.gc % WS_NP) * (NPIX * 8) + en.q * 8 + (s ^ ((en.q >> 1) & 7))] = v;
        };

        if (TAPS == 9 && nch_skip == 0) {
            // ---- fast path: every chunk is nine stages. The loop body is straight-line in its global loads (slab: 4 per
            // stage; patch entry: 1 per stage at taps 0..6; coefficients: 3 at tap 0), so the compiler's vmcnt waits are
            // exact and every load has two full stages to land: the slab written at stage st was requested at st - 2,
            // the patch entry committed at tap t was requested at tap t - 2 of the same chunk.
            PCur pcur;
            pcur.gc = 0; pcur.k = 0; pcur.c = 0; pcur.tl = decode(0);
            pc_setup(pcur);
            const float* cbase = coefp ? coefp : src0;      // never dereferenced for values when coefp is null
            auto load_coef_u = [&](const PCur& r, v4f& m_, v4f& a_, v4f& d_) {     // unconditional loads
                const size_t ci = coefp ? ((size_t)r.tl.b * Cin + (r.c << 5) + 4 * s) : (size_t)(4 * s);
                const size_t pl = coefp ? coef_plane : 0;
                m_ = *reinterpret_cast<const v4f*>(cbase + ci);
                a_ = *reinterpret_cast<const v4f*>(cbase + pl + ci);
                d_ = *reinterpret_cast<const v4f*>(cbase + 2 * pl + ci);
            };
            v4f cm, ca, cd;
            load_coef_u(pcur, cm, ca, cd);
            for (; pcur.e < pcur.ne; ++pcur.e) commit(fetch(pcur), cm, ca, cd);
            ws_publish(prodP + 4 * sw, 1);
            pc_next_chunk(pcur);

            WCur wcur;
            wcur.i = 0; wcur.k = 0; wcur.c = 0; wcur.t = 0; wcur.cb = decode(0).cb;
            v4f wA[4], wB[4];
            {
                const v4f* p = wc_ptr(wcur);
#pragma unroll
                for (int j = 0; j < 4; ++j) wA[j] = p[ltid + 256 * j];
                wc_next(wcur);
                const v4f* p2 = wc_ptr(wcur);
#pragma unroll
                for (int j = 0; j < 4; ++j) wB[j] = p2[ltid + 256 * j];
            }
            Ent pOld, pNew;         // requested two / one stage(s) ago
            pOld.q = -1; pOld.in = false; pOld.v = v4f{0.f, 0.f, 0.f, 0.f}; pOld.act = true; pOld.gc = 0; pOld.last = false;
            pNew = pOld;
            for (int g = 0; g < nchunks; ++g) {
                const bool more = (g + 1 < nchunks);     // a next patch exists (pcur points at it)
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int st = 9 * g + t;
                    if (st >= WS_NW) t_wait += ws_wait(cons, st - WS_NW + 1, timed);     // slot last read by stage st - NW
#ifndef WS_ABL_NOW
#pragma unroll
                    for (int j = 0; j < 4; ++j) ldsW[(st % WS_NW) * 1024 + ltid + 256 * j] = wA[j];
#endif
                    // the slab wait above also covers the patch buffer: at tap 2 it guarantees every stage before chunk g
                    // is consumed, and buffer (g + 1) % NP was last read by chunk g + 1 - NP <= g - 1
#ifndef WS_ABL_NOPATCH
                    if (t >= 2) commit(pOld, cm, ca, cd);
#endif
                    if (t == 8 && more) ws_publish(prodP + 4 * sw, g + 2);
                    ws_publish(prodW + 4 * sw, st + 1);
                    // ---- requests
#ifndef WS_ABL_NOW
#pragma unroll
                    for (int j = 0; j < 4; ++j) wA[j] = wB[j];
                    wc_next(wcur);
                    {
                        const v4f* p = wc_ptr(wcur);
#pragma unroll
                        for (int j = 0; j < 4; ++j) wB[j] = p[ltid + 256 * j];
                    }
#endif
#ifndef WS_ABL_NOPATCH
                    pOld = pNew;
                    if (t < NE9) {
                        pcur.e = t;
                        pNew = fetch(pcur);
                        if (!more) pNew.q = -1;
                    }
                    if (t == 0) load_coef_u(pcur, cm, ca, cd);     // the previous chunk's commits ended at its tap 8
#endif
                }
                if (more) pc_next_chunk(pcur);
            }
            if (timed && lane == 0) {
                a.stamps[2 + 4 * sw] = __builtin_amdgcn_s_memtime() - t_begin - t_wait;
                a.stamps[3 + 4 * sw] = t_wait;
            }
            return;
        }

        // ---- prologue: patch 0 completely, weight slabs of stages 0 .. LEAD-1, registers <- stages LEAD and LEAD+1
        PCur pcur;
        pcur.gc = 0; pcur.k = 0; pcur.c = 0; pcur.tl = decode(0);
        pc_setup(pcur);
        v4f cm, ca, cd;                 // coefficients of the patch being requested
        load_coef(pcur, cm, ca, cd);
        for (; pcur.e < pcur.ne; ++pcur.e) commit(fetch(pcur), cm, ca, cd);
        ws_publish(prodP + 4 * sw, 1);
        pc_next_chunk(pcur);
        if (pcur.gc < nchunks) load_coef(pcur, cm, ca, cd);

        WCur wcur;
        wcur.i = 0; wcur.k = 0; wcur.c = 0; wcur.t = 0; wcur.cb = decode(0).cb;
        const int npre = total < WS_LEAD ? total : WS_LEAD;
        for (int st = 0; st < npre; ++st) {
            const v4f* p = wc_ptr(wcur);
#pragma unroll
            for (int j = 0; j < 4; ++j) ldsW[st * 1024 + ltid + 256 * j] = p[ltid + 256 * j];
            wc_next(wcur);
        }
        ws_publish(prodW + 4 * sw, npre);
        v4f wregA[4], wregB[4];         // slabs in flight: even iterations drain A, odd iterations drain B
        {
            const v4f* p = wc_ptr(wcur);        // stage LEAD (or the last stage)
#pragma unroll
            for (int j = 0; j < 4; ++j) wregA[j] = p[ltid + 256 * j];
            wc_next(wcur);
            const v4f* p2 = wc_ptr(wcur);       // stage LEAD + 1 (or the last stage)
#pragma unroll
            for (int j = 0; j < 4; ++j) wregB[j] = p2[ltid + 256 * j];
        }

        // patch pipeline: entries fetched in one iteration are committed in the next. Requests are paced by the stage
        // cursor `sc` = stage i + WS_PLEAD: while that stage lies in global chunk g, the entries of chunk g + 1 are
        // requested -- one per stage inside a 9-tap chunk, all of them when the chunk is a single stage (or when the
        // request cursor is behind), so single-stage chunks do not make the matrix waves wait.
        Ent pend[NE9];
        int npend = 0;
        v4f pcm = cm, pca = ca, pcd = cd;      // coefficients belonging to the pending entries
        struct SCur { int gc, c, t; } sc = {0, 0, 0};
        auto sc_next = [&]() {
            const int ntap = (sc.c < nch_main) ? TAPS : 1;
            if (++sc.t == ntap) { sc.t = 0; ++sc.gc; if (++sc.c == nch) sc.c = 0; }
        };
        for (int j = 0; j < WS_PLEAD; ++j) sc_next();

        auto iteration = [&](int i, v4f (&wreg)[4]) {
            const int st = i + WS_LEAD;         // slab written in this iteration
            if (st < total) {
                if (st >= WS_NW) t_wait += ws_wait(cons, st - WS_NW + 1, timed);      // slot last read by stage st - NW
#pragma unroll
                for (int j = 0; j < 4; ++j) ldsW[(st % WS_NW) * 1024 + ltid + 256 * j] = wreg[j];
            }
            if (npend) {
                // buffer gc % NP was last read during global chunk gc - NP: every stage before chunk gc - NP + 1 consumed
                const int g = pend[0].gc;
                if (g >= WS_NP) t_wait += ws_wait(cons, stages_before(g - WS_NP + 1), timed);
                bool done = false;
#pragma unroll
                for (int j = 0; j < NE9; ++j)
                    if (j < npend) { commit(pend[j], pcm, pca, pcd); done = done || pend[j].last; }
                if (done) ws_publish(prodP + 4 * sw, g + 1);
            }
            ws_publish(prodW + 4 * sw, (st < total ? st + 1 : total));
            // requests: slab of stage i + LEAD + 2 into the register set just drained, next patch entries
            wc_next(wcur);
            {
                const v4f* p = wc_ptr(wcur);
#pragma unroll
                for (int j = 0; j < 4; ++j) wreg[j] = p[ltid + 256 * j];
            }
            npend = 0;
            if (pcur.gc < nchunks && pcur.gc <= sc.gc + 1) {
                pcm = cm; pca = ca; pcd = cd;
                const bool single = (sc.c >= nch_main) || TAPS != 9 || pcur.gc <= sc.gc;
                const int n = single ? (pcur.ne - pcur.e) : 1;
#pragma unroll
                for (int j = 0; j < NE9; ++j)
                    if (j < n) { pend[j] = fetch(pcur); ++pcur.e; }
                npend = n;
                if (pcur.e == pcur.ne) {
                    pc_next_chunk(pcur);
                    if (pcur.gc < nchunks) load_coef(pcur, cm, ca, cd);
                }
            }
            sc_next();
        };
        int i = 0;
        for (; i + 1 < total; i += 2) { iteration(i, wregA); iteration(i + 1, wregB); }
        if (i < total) { iteration(i, wregA); ++i; }
        // drain (only when the tail is made of single-stage chunks): commit what is pending, request what is left
        while (npend || pcur.gc < nchunks) {
            if (npend) {
                const int g = pend[0].gc;
                if (g >= WS_NP) t_wait += ws_wait(cons, stages_before(g - WS_NP + 1), timed);
                bool done = false;
#pragma unroll
                for (int j = 0; j < NE9; ++j)
                    if (j < npend) { commit(pend[j], pcm, pca, pcd); done = done || pend[j].last; }
                if (done) ws_publish(prodP + 4 * sw, g + 1);
            }
            npend = 0;
            if (pcur.gc < nchunks) {
                pcm = cm; pca = ca; pcd = cd;
                const int n = pcur.ne - pcur.e;
#pragma unroll
                for (int j = 0; j < NE9; ++j)
                    if (j < n) { pend[j] = fetch(pcur); ++pcur.e; }
                npend = n;
                pc_next_chunk(pcur);
                if (pcur.gc < nchunks) load_coef(pcur, cm, ca, cd);
            }
        }
        if (timed && lane == 0) {
            a.stamps[2 + 4 * sw] = __builtin_amdgcn_s_memtime() - t_begin - t_wait;
            a.stamps[3 + 4 * sw] = t_wait;
        }
        return;
    }

    // ==================================== matrix waves ===================================================
    __builtin_amdgcn_s_setprio(WS_PRIO_M);
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
#define WS_READ(SA0, SA1, SB0, SB1, PB, O0, O1, WB, G)                                                      \
    {                                                                                                      \
        SA0 = a_read(PB, O0, G); SA1 = a_read(PB, O1, G);                                                  \
        SB0 = b_read(WB, 0, G);  SB1 = b_read(WB, 1, G);                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }

    // counter poll split in two, so the LDS round trip hides behind MFMAs: issue now, consume later
    auto poll_issue = [&](unsigned grp) -> v4i {
        v4i v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(grp) : "memory");
        return v;
    };
    auto poll_min = [&](v4i& v) -> int {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory");
        return min(min(v.x, v.y), min(v.z, v.w));
    };
    // this wave's consumption counter: no wait needed -- a wave's LDS instructions execute in issue order, so the
    // counter write lands after the fragment reads issued before it
    const unsigned cons_addr = cons + 4 * wave;
    auto publish_cons = [&](int value) {
        if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(cons_addr), "v"(value) : "memory");
    };

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
            const int gc = k * nch + c;
            const v4f* pbuf = ldsP + (gc % WS_NP) * (NPIX * 8);
            t_wait += ws_wait(prodP, gc + 1, timed);     // this chunk's patch is complete
            t_wait += ws_wait(prodW, gs + 1, timed);     // the first stage's slab is complete
            {
                const int tap0 = main_seg ? 0 : (TAPS / 2);
                WS_READ(fa0, fa1, fb0, fb1, pbuf, a_off(tap0, 0), a_off(tap0, 1), ldsW + (gs % WS_NW) * 1024, 0)
            }
            // stage loop: every fragment read below is unconditional, so the compiler's lgkmcnt waits stay exact
            for (int t = 0; t < ntap; ++t, ++gs) {
                const int tap = main_seg ? t : (TAPS / 2);     // single-tap segment: centre tap
                const v4f* wb = ldsW + (gs % WS_NW) * 1024;
                const int o0 = a_off(tap, 0), o1 = a_off(tap, 1);
                const bool more = t + 1 < ntap;
                v4i pv = poll_issue(prodW);
                WS_READ(ga0, ga1, gb0, gb1, pbuf, o0, o1, wb, 1)
                WS_MFMA16(fa0, fa1, fb0, fb1)
                WS_READ(fa0, fa1, fb0, fb1, pbuf, o0, o1, wb, 2)
                WS_MFMA16(ga0, ga1, gb0, gb1)
                WS_READ(ga0, ga1, gb0, gb1, pbuf, o0, o1, wb, 3)
                publish_cons(gs + 1);                           // the stage's last reads are queued
                WS_MFMA16(fa0, fa1, fb0, fb1)
                // next stage's slab: normally there already (polled a stage ago); the group-0 fragments of the next
                // stage are requested now and land behind the 16 MFMAs below. On the chunk's last stage the same
                // addresses are re-read as a dummy, so that the read count per iteration is constant.
                const int seen = poll_min(pv);      // always consumed: the poll's destination registers stay reserved until it lands
                if (more && seen < gs + 2) t_wait += ws_wait(prodW, gs + 2, timed);
                {
                    const int tn = more ? tap + 1 : tap;
                    const v4f* wbn = ldsW + ((more ? gs + 1 : gs) % WS_NW) * 1024;
                    WS_READ(fa0, fa1, fb0, fb1, pbuf, a_off(tn, 0), a_off(tn, 1), wbn, 0)
                }
                WS_MFMA16(ga0, ga1, gb0, gb1)
            }
            // two-level accumulation (see conv_mfma.hip): fold the chunk, restart the inner chains
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    tot[i][j] += acc[i][j];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
        }
        // ---- epilogue of the tile (same as conv_mfma.hip, private transpose region per wave)
        {
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
                        ok[i] = (y < H) && (x < W);
                        oidx[i] = ((size_t)(tl.b * H + y) * W + x) * Cout + co;
                        rsd[i] = v4f{0.f, 0.f, 0.f, 0.f};
                        if (a.res && ok[i]) {
                            const size_t rp = a.res_up
                                ? ((size_t)(tl.b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1))
                                : ((size_t)(tl.b * H + y) * W + x);
                            rsd[i] = *reinterpret_cast<const v4f*>(a.res + rp * Cout + co);
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
                        float* o = a.stats + (((size_t)tl.b * nrec + rec) * Cout + co) * 2;
                        *reinterpret_cast<v4f*>(o) = v4f{ssum.x, ssq.x, ssum.y, ssq.y};
                        *reinterpret_cast<v4f*>(o + 4) = v4f{ssum.z, ssq.z, ssum.w, ssq.w};
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (timed && lane == 0 && wave == 0) {
        a.stamps[0] = __builtin_amdgcn_s_memtime() - t_begin - t_wait;
        a.stamps[1] = t_wait;
    }
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
    const size_t lds9 = (size_t)(WS_NP * 6 * 34 * 8 + WS_NW * 1024) * 16 + 4 * 2048 * 4 + 64;
    const size_t lds1 = (size_t)(WS_NP * 4 * 32 * 8 + WS_NW * 1024) * 16 + 4 * 2048 * 4 + 64;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ws_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds9);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ws_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        attr_set = true;
    }
    if (a.taps == 9) hipLaunchKernelGGL(conv_ws_kernel<9>, dim3(grid), dim3(512), lds9, stream, a, ntiles);
    else hipLaunchKernelGGL(conv_ws_kernel<1>, dim3(grid), dim3(512), lds1, stream, a, ntiles);
}

}  // namespace cddpm
