// "Ping-pong" schedule of the fp16-split 3x3 convolution (conv_x6.hip, NS = 2): same operator, same packed weights, same
// arithmetic in the same order -- results are bit-identical -- but the two waves of every SIMD never do the same thing at the
// same time.
//
// In conv_x6.hip all eight waves of the workgroup walk through a stage together: stage weights, barrier, read fragments,
// MFMAs, fold. The matrix pipe is idle whenever they stage or wait, and when they compute, the two waves of a SIMD queue for
// the one pipe: 46 % busy. Here the workgroup is split into two groups of four waves (one wave per SIMD each; group = cout
// half), and time into PHASES separated by one workgroup barrier:
//
//      phase 2s     : group 0 issues the 24 MFMAs of stage s (tap s % 9 of chunk s / 9)   | group 1 stages
//      phase 2s + 1 : group 1 issues the 24 MFMAs of stage s                              | group 0 stages
//
// "Staging" is everything that is not an MFMA: folding the accumulators (after taps 2, 5, 8), this thread's two 16-B pieces of
// the weight slab of stage s + 2 (registers -> ring slot (s + 2) % 3; the global loads for s + 3 leave right after), one
// 16-B entry of the NEXT chunk's patch (load at taps 0..5, normalise / activate / split / store two taps later into the
// other patch buffer), and -- last -- the k-step-0 fragments of the stage this wave computes next. A computing wave therefore
// starts its MFMAs straight after the barrier, and the pipe sees one uncontended MFMA stream at a time.
// Live registers are fewer than in the lock-step kernel: the patch prefetch is 2 entries (8 VGPRs) instead of 6.
//
// STATUS: experimental, opt-in with CDDPM_CONV_PP=1. Parity-green (bit-identical to the lock-step kernel), but 12-15 % SLOWER
// than conv_split_kernel with three-tap weight stages: a phase lasts ~1900 cycles where its 24 MFMAs need 768 -- the
// non-MFMA instructions of either kind of phase (fragment addressing, the patch-entry transform the compiler keeps in
// its own basic blocks instead of between the MFMAs, 64-bit address arithmetic of the requests, LDS round trips) still
// sit on the critical path of every phase, and there are two barriers per tap instead of one per three taps.
// Restrictions (launch_conv_split falls back to conv_split_kernel otherwise): 3x3 taps, no fused skip segment.
// LDS (one workgroup per CU): 2 patches x 43.5 KB + 3 weight slabs x 16 KB + 12 KB source-pixel table + coefficient cache
// <= 157 KB.
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

namespace cddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_pp(float v) {
    // identical evaluation to conv_mfma.hip::silu_f / conv_x6.hip::silu_x6 (split-product exp2, ~1.5 ulp)
    const float t = fminf(-v * 1.44269502162933349609375f, 126.0f);
    float tl = __builtin_fmaf(-v, 1.44269502162933349609375f, -t);
    tl = __builtin_fmaf(-v, 1.925963033500011e-08f, tl);
    tl = (t < 126.0f) ? tl : 0.0f;
    float e = __builtin_amdgcn_exp2f(t);
    e = __builtin_fmaf(e, tl * 0.693147180559945f, e);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

__global__ __launch_bounds__(512) void conv_pp_kernel(const ConvArgs a) {
    constexpr int ROWS = 8, THREADS = 512, SP = 8, TAPS = 9;
    constexpr int PW = 34, PH = ROWS + 2, NPIX = PW * PH;          // 340
    constexpr int NK = (NPIX * 8 + THREADS - 1) / THREADS;         // 6 entries per thread and chunk
    constexpr int WSLOTS = 128 * SP;                               // 1024 16-B pieces per slab
    constexpr int WK = WSLOTS / THREADS;                           // 2 per thread
    constexpr int NWS = 3;

    extern __shared__ v4f lds[];
    v4f* const ldsW = lds + 2 * NPIX * SP;          // NWS * WSLOTS
    int* const ldsPS = reinterpret_cast<int*>(ldsW + NWS * WSLOTS);     // source pixel of each thread's NK patch entries (12 KB)
    v4f* const ldsC = ldsW + NWS * WSLOTS + (NK * THREADS) / 4;         // GroupNorm/FiLM coefficients of this sample (3 x Cin floats)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int wm = wave & 3;        // pixel rows {2 wm, 2 wm + 1}
    const int grp = wave >> 2;      // group = cout half (wn); group 0 = first wave of each SIMD, group 1 = second

    const int ncb = a.Cout >> 7;
    const int gridH = a.H, gridW = a.W;
    const int tilesX = (gridW + 31) >> 5;
    const int tilesY = (gridH + ROWS - 1) / ROWS;
    int bid = blockIdx.x;
    const int cb = bid % ncb;
    bid /= ncb;
    const int tx = bid % tilesX;
    bid /= tilesX;
    const int ty = bid % tilesY;
    const int b = bid / tilesY;
    const int y0 = ty * ROWS, x0 = tx * 32;

    auto slot_of = [](int row, int sp, int u) -> int { return row * 8 + ((4 * sp + u) ^ ((row >> 1) & 7)); };

    const int Cin = a.C0 + a.C1;
    const int nch = Cin >> 5;
    const int N = nch * TAPS;                       // stages

    // ---- per-thread patch entries: channel quad c4 fixed per thread, pixel q = (tid>>3) + 64 k
    const int c4 = tid & 7;
    int psrc[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int q = (tid >> 3) + (THREADS / 8) * k;
        const int pr = q / PW, pc = q - pr * PW;
        const int y = y0 + pr - 1, x = x0 + pc - 1;
        const bool valid = (q < NPIX) && (y >= 0) && (y < gridH) && (x >= 0) && (x < gridW);
        const int sy = a.upsample ? (y >> 1) : y, sx = a.upsample ? (x >> 1) : x;
        psrc[k] = valid ? ((b * a.srcH + sy) * a.srcW + sx) : -1;
        ldsPS[k * THREADS + tid] = psrc[k];     // read back by this thread only (no barrier needed): keeps 6 VGPRs free in the loop
    }
    const v4f* const wmain = reinterpret_cast<const v4f*>(a.wpk) + (size_t)cb * nch * TAPS * WSLOTS;
    const bool have_coef = (a.coef != nullptr);

    auto entry_ptr = [&](int chunk, int p) -> const v4f* {
        const int ch = chunk << 5;
        const float* base = (ch < a.C0) ? (a.src0 + (size_t)p * a.C0 + ch) : (a.src1 + (size_t)p * a.C1 + (ch - a.C0));
        return reinterpret_cast<const v4f*>(base + 4 * c4);
    };
    // unconditional load (address clamped to pixel 0 when the entry is padding): no exec-masked load, no merge copies,
    // so the compiler's vmcnt bookkeeping stays exact; padding is zeroed when the entry is stored
    auto load_entry = [&](int chunk, int p) -> v4f { return *entry_ptr(chunk, p >= 0 ? p : 0); };
    // normalise / activate / split one entry and write it into patch buffer `pb` (k: entry number, p = psrc[k])
    auto store_entry = [&](int k, int p, v4f v, v4f cm, v4f ca, v4f cd, v4f* pb) {
        const int q = (tid >> 3) + (THREADS / 8) * k;
        if (q >= NPIX) return;
        if (p < 0) v = v4f{0.f, 0.f, 0.f, 0.f};
        if (p >= 0) {   // zero padding stays exactly zero: the conv pads AFTER the activation
            if (have_coef) v = (v - cm) * ca + cd;
            if (a.silu) { v.x = silu_pp(v.x); v.y = silu_pp(v.y); v.z = silu_pp(v.z); v.w = silu_pp(v.w); }
        }
        const f16x4 h = __builtin_convertvector(v, f16x4);
        const v4f r = v - __builtin_convertvector(h, v4f);
        const f16x4 m = __builtin_convertvector(r, f16x4);
        v2f* dst = reinterpret_cast<v2f*>(pb);
        dst[slot_of(q, 0, c4 >> 1) * 2 + (c4 & 1)] = __builtin_bit_cast(v2f, h);
        dst[slot_of(q, 1, c4 >> 1) * 2 + (c4 & 1)] = __builtin_bit_cast(v2f, m);
    };
    auto coefs_of = [&](int chunk, v4f& cm, v4f& ca, v4f& cd) {
        cm = v4f{0.f, 0.f, 0.f, 0.f}; ca = v4f{1.f, 1.f, 1.f, 1.f}; cd = cm;
        if (have_coef) {
            const int ci = (chunk << 3) + c4;
            cm = ldsC[ci]; ca = ldsC[(Cin >> 2) + ci]; cd = ldsC[2 * (Cin >> 2) + ci];
        }
    };

    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    int brow[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) brow[nt] = 64 * grp + 32 * nt + li;

    // fragments of (stage s, k-step jk): A from the patch of chunk s / 9, B from ring slot s % 3
    auto load_frags = [&](int s, int jk, f16x8 (&fa)[2][2], f16x8 (&fb)[2][2]) {
        const int chunk = s / TAPS, tap = s - chunk * TAPS;
        const int ky = tap / 3, kx = tap - 3 * ky;
        const v4f* pa = lds + (chunk & 1) * (NPIX * SP);
        const v4f* wb = ldsW + (s % NWS) * WSLOTS;
        const int u = 2 * jk + lh;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int arow = (2 * wm + i + ky) * PW + li + kx;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                fa[sp][i] = __builtin_bit_cast(f16x8, pa[slot_of(arow, sp, u)]);
                fb[sp][i] = __builtin_bit_cast(f16x8, wb[slot_of(brow[i], sp, u)]);
            }
        }
    };
    // products kept, smallest first: mid*hi, hi*mid, hi*hi (same order as conv_split_kernel<.., 2>)
    auto products = [&](const f16x8 (&fa)[2][2], const f16x8 (&fb)[2][2], bool restart) {
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int sa = (p == 0) ? 1 : 0, sb = (p == 1) ? 1 : 0;
            if (p == 0 && restart) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sa][i], fb[sb][j], zero16, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sa][i], fb[sb][j], acc[i][j], 0, 0, 0);
            }
        }
    };
    auto slab_ptr = [&](int s) -> const v4f* { return wmain + (size_t)((s < N) ? s : 0) * WSLOTS; };   // wraps: prefetch stays unconditional

    // ---- prologue (lock-step): coefficient cache, patch of chunk 0, slabs of stages 0 and 1, registers <- slab 2
    v4f wreg[WK];
    {
        v4f w0[WK], w1[WK], e0[NK];
#pragma unroll
        for (int i = 0; i < WK; ++i) { w0[i] = slab_ptr(0)[tid + THREADS * i]; w1[i] = slab_ptr(1)[tid + THREADS * i]; }
#pragma unroll
        for (int k = 0; k < NK; ++k) e0[k] = load_entry(0, psrc[k]);
        if (have_coef) {
            const int nq = Cin >> 2;
            const size_t plane = (size_t)a.B * Cin;
            for (int i = tid; i < 3 * nq; i += THREADS) {
                const int pl = i / nq, cq = i - pl * nq;
                ldsC[i] = *reinterpret_cast<const v4f*>(a.coef + pl * plane + (size_t)b * Cin + 4 * cq);
            }
        }
#pragma unroll
        for (int i = 0; i < WK; ++i) { ldsW[tid + THREADS * i] = w0[i]; ldsW[WSLOTS + tid + THREADS * i] = w1[i]; }
#pragma unroll
        for (int i = 0; i < WK; ++i) wreg[i] = slab_ptr(2)[tid + THREADS * i];
        __syncthreads();       // coefficient cache visible
        v4f cm0, ca0, cd0;
        coefs_of(0, cm0, ca0, cd0);
#pragma unroll
        for (int k = 0; k < NK; ++k) store_entry(k, psrc[k], e0[k], cm0, ca0, cd0, lds);
        __syncthreads();       // patch 0, slabs 0 and 1 visible
    }
    f16x8 pa[2][2], pb[2][2];
#ifdef CDDPM_STAMPS
    unsigned long long st_[4] = {0, 0, 0, 0};      // compute phase work, staging phase work, barrier wait after compute / after staging
    unsigned long long fine_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // segments inside the phases (group 0 only)
    unsigned long long tl_ = 0;
#define PST(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); fine_[i] += n_ - tl_; tl_ = n_; }
#else
#define PST(i)
#endif
    if (grp == 0) load_frags(0, 0, pa, pb);        // group 0 computes first
    v4f entA = v4f{0.f, 0.f, 0.f, 0.f}, entB = entA;   // patch entries in flight: even / odd taps

    // ---- phases. The body is instantiated for both parities of the stage index, four phases per loop trip, so that the
    // patch entry in flight (requested at stage s, stored at stage s + 2) lives in a FIXED register set: a run-time
    // choice between two sets makes the compiler copy the freshly loaded registers, i.e. wait for the load right away.
    auto phase = [&](int phi, auto parity_c) {
        constexpr int PARITY = decltype(parity_c)::value;      // == (phi >> 1) & 1
        v4f& ent = PARITY ? entB : entA;
        const int s = phi >> 1;                     // stage both groups work on during this pair of phases
        const int chunk = s / TAPS, tap = s - chunk * TAPS;
#ifdef CDDPM_STAMPS
        const unsigned long long tp0_ = __builtin_amdgcn_s_memtime();
        const bool was_compute_ = ((phi & 1) == grp);
        tl_ = tp0_;
#endif
        const bool more = chunk + 1 < nch;
        v4f& ent_other = PARITY ? entA : entB;
        if ((phi & 1) == grp) {
            // ============ compute phase: 24 MFMAs; k-step 0 fragments were fetched in the previous (staging) phase.
            // (k-step 1 reuses the fragment registers: its reads are issued behind the 12 MFMAs of k-step 0, which cover
            // the LDS latency -- the other group is not reading fragments now -- and 32 VGPRs stay free.)
            products(pa, pb, (tap % 3) == 0);
            PST(0)
            load_frags(s, 1, pa, pb);
            PST(1)
            // One entry of the NEXT chunk's patch is normalised / activated / split / written here, in the issue slots the
            // MFMAs leave free (an MFMA holds the vector issue port 8 of its 32 cycles): the entry this wave requested three
            // phases ago -- for group 0 that was stage s - 2 (same parity set), for group 1 stage s - 1 (the other set),
            // because group 1's staging phase of a stage comes BEFORE its compute phase.
            {
                const int k = tap - (grp ? 1 : 2);
                if (more && k >= 0 && k < NK) {
                    const int p_st = ldsPS[k * THREADS + tid];
                    v4f cm, ca, cd;
                    coefs_of(chunk + 1, cm, ca, cd);
                    store_entry(k, p_st, grp ? ent_other : ent, cm, ca, cd, lds + ((chunk + 1) & 1) * (NPIX * SP));
                }
            }
            PST(2)
            products(pa, pb, false);
            PST(3)
        } else {
            // ============ staging phase
            // the stage this wave computed last (group 1 runs one phase behind group 0) and the one it computes next
            const int s_done = (grp == 0) ? s : s - 1;
            const bool do_ld = more && (tap < NK);
            const int p_ld = do_ld ? ldsPS[tap * THREADS + tid] : -1;
            // k-step 0 fragments of the stage this wave computes next: requested FIRST -- the barrier at the end of the phase
            // waits for every LDS operation of the wave (lgkmcnt(0)), so their round trip has to hide behind the rest
            const int s_next = s_done + 1;
            if (s_next < N) load_frags(s_next, 0, pa, pb);
            if (s_done >= 0 && ((s_done % TAPS) % 3) == 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) tot[i][j] += acc[i][j];
            }
            PST(4)
            // weight slab of stage s + 2: this thread's pieces into the slot stage s - 1 used (both groups are done with it)
            if (s + 2 < N) {
#pragma unroll
                for (int i = 0; i < WK; ++i) ldsW[((s + 2) % NWS) * WSLOTS + tid + THREADS * i] = wreg[i];
            }
            PST(5)
            // ---- requests: next slab pieces, patch entry `tap` of the next chunk (stored from a compute phase, see above)
#pragma unroll
            for (int i = 0; i < WK; ++i) wreg[i] = slab_ptr(s + 3)[tid + THREADS * i];
            if (do_ld) ent = load_entry(chunk + 1, p_ld);
            PST(6)
            PST(7)
        }
#ifdef CDDPM_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long tp1_ = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef CDDPM_STAMPS
        const unsigned long long tp2_ = __builtin_amdgcn_s_memtime();
        st_[was_compute_ ? 0 : 1] += tp1_ - tp0_;
        st_[was_compute_ ? 2 : 3] += tp2_ - tp1_;
#endif
    };
    {
        int phi = 0;
        for (; phi + 3 < 2 * N; phi += 4) {
            phase(phi, std::integral_constant<int, 0>{});
            phase(phi + 1, std::integral_constant<int, 0>{});
            phase(phi + 2, std::integral_constant<int, 1>{});
            phase(phi + 3, std::integral_constant<int, 1>{});
        }
        if (phi < 2 * N) {          // odd number of stages: the last stage has an even index
            phase(phi, std::integral_constant<int, 0>{});
            phase(phi + 1, std::integral_constant<int, 0>{});
        }
    }
#ifdef CDDPM_STAMPS
    if (a.stamps && lane == 0 && (wave & 3) == 0)
        for (int i = 0; i < 4; ++i) atomicAdd(&a.stamps[48 + grp * 4 + i], st_[i]);
    if (a.stamps && lane == 0 && wave == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&a.stamps[i], fine_[i]);
#endif
    // group 1 computed the last stage in the last phase: its fold is still due (group 0 folded in that phase)
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) tot[i][j] += acc[i][j];
    }

    // ---- residual prefetch + epilogue: identical to conv_split_kernel (conv_x6.hip)
    const int wn = grp;
    v4f rsd_all[2][2][4];
    {
        const int cq = lane & 7, prow = lane >> 3;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = cb * 128 + 64 * wn + 32 * nt + 4 * cq;
                    const int p = 8 * (4 * hb + i) + prow;
                    const int gy = y0 + 2 * wm + (p >> 5), gx = x0 + (p & 31);
                    v4f r = v4f{0.f, 0.f, 0.f, 0.f};
                    if (a.res && (gy < gridH) && (gx < gridW)) {
                        const size_t rp = a.res_up ? ((size_t)(b * (a.H >> 1) + (gy >> 1)) * (a.W >> 1) + (gx >> 1))
                                                   : ((size_t)(b * a.H + gy) * a.W + gx);
                        r = *reinterpret_cast<const v4f*>(a.res + rp * a.Cout + co);
                    }
                    rsd_all[nt][hb][i] = r;
                }
    }
    __syncthreads();   // (already synchronised by the loop's last barrier; keeps the aliasing below obviously safe)
    {
        float* tr = reinterpret_cast<float*>(lds) + wave * 2048;      // [64 pixels][32 channels]
        const int cq = lane & 7;
        const int prow = lane >> 3;
        const int tilesY4 = (gridH + 3) >> 2;
        const int ty4 = (y0 >> 2) + (wm >> 1);
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
            const float wsc = a.wscale_inv;
            v4f ssum = v4f{0.f, 0.f, 0.f, 0.f}, ssq = ssum;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                v4f val[4];
                size_t oidx[4];
                bool ok[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = 8 * (4 * hb + i) + prow;
                    const int gy = y0 + 2 * wm + (p >> 5), gx = x0 + (p & 31);
                    ok[i] = (gy < gridH) && (gx < gridW);
                    oidx[i] = ((size_t)(b * a.H + gy) * a.W + gx) * a.Cout + co;
                    val[i] = *reinterpret_cast<const v4f*>(tr + p * 32 + 4 * cq);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (ok[i]) {
                        const v4f o = val[i] * wsc + bias + rsd_all[nt][hb][i];
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
                if (prow == 0 && ty4 < tilesY4) {
                    const int nrec = 2 * tilesX * tilesY4;
                    const int rec = 2 * (ty4 * tilesX + tx) + (wm & 1);
                    float* o = a.stats + (((size_t)b * nrec + rec) * a.Cout + co) * 2;
                    *reinterpret_cast<v4f*>(o) = v4f{ssum.x, ssq.x, ssum.y, ssq.y};
                    *reinterpret_cast<v4f*>(o + 4) = v4f{ssum.z, ssq.z, ssum.w, ssq.w};
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

bool conv_pp_applicable(const ConvArgs& a) {
    static const bool on = [] { const char* e = getenv("CDDPM_CONV_PP"); return e && e[0] == '1'; }();     // opt-in: measured slower, see above
    return on && a.taps == 9 && a.S0 + a.S1 == 0 && (a.C0 + a.C1) >= 32;
}

void launch_conv_pp(const ConvArgs& a, hipStream_t stream) {
    const int tilesX = (a.W + 31) / 32, tilesY = (a.H + 7) / 8;
    const unsigned grid = (unsigned)(a.B * tilesX * tilesY * (a.Cout / 128));
    const size_t coef_lds = a.coef ? (size_t)3 * (a.C0 + a.C1) * sizeof(float) : 0;
    const size_t need = (size_t)(2 * 340 * 8 + 3 * 1024) * 16 + 6 * 512 * 4 + coef_lds;       // >= 8 x 8 KB of epilogue transpose space
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(conv_pp_kernel, dim3(grid), dim3(512), need, stream, a);
}

}  // namespace cddpm
