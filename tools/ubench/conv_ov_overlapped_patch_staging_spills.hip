// ARCHIVED EXPERIMENT (round 2) -- not part of libcddpm_hip.so. Results correct (tests/test_gpu_kernels.py::test_conv passed on the
// 3x3 shapes with CDDPM_CONV_OV=1), speed 0.47x of conv_split_kernel<9,8,2,true> on every layer shape (gpurun_out/r2_ab_ov.log):
// hipcc (ROCm 7.2) allocates 256 VGPRs and spills ~140 more (`tot`/`acc` tiles, fragment base addresses) inside the tap loop as
// soon as transform_entry sits in the main-chunk loop -- wherever it is placed (behind the MFMAs, behind the fold where `acc` is
// dead, between sched_barriers, one entry per tap or three per row) -- while the same loop without it, or with the transform
// only in the prologue / skip path, compiles to 256 VGPRs and no scratch. Every scratch reload also draws a compiler-inserted
// s_waitcnt vmcnt(0), which drains the counted LDS-DMA pipeline. The schedule itself (second patch buffer from a 3-slab weight
// ring, raw patch by LDS-DMA, in-place transform, counted vmcnt, one barrier per tap) is what DESIGN.md section 7 item 2 asks for;
// the register allocation is the open problem (candidates: `tot` pinned in AGPRs by inline asm, the transform as hand-written asm).
// The 3x3 convolution of the fp16-split family with the patch staging OVERLAPPED with the matrix work (gfx950).
//
// Same operator, arguments, packed weight image and results as conv_split_kernel<9,8,2,true> of conv_x6.hip (see there for the
// arithmetic and for the reference lines it replaces: src/models/modules/OpenAI_Unet.py:284-338, :948). What differs is the
// schedule. In conv_x6.hip every 32-channel chunk starts with a phase in which all eight waves transform the next patch
// (GroupNorm/FiLM affine, SiLU, fp16 split) between two workgroup barriers: 17-21 % of a workgroup's time in which no MFMA
// issues (in-kernel accounting). Here
//   * the raw fp32 patch of chunk c + 1 is copied global -> LDS by LDS-DMA (global_load_lds_dwordx4, lane-linear: lane = (pixel,
//     channel quad) in the [pixel][32 channels] order of NHWC) into the SECOND patch buffer at the first tap of chunk c -- no
//     prefetch registers -- and
//   * transformed IN PLACE, one 16-B entry per thread and tap, behind the MFMAs of taps 3..8 of chunk c: the eight lanes that own
//     the eight quads of a pixel sit in one wave, read their quads with one ds_read_b128 and write the two 8-B halves of the hi
//     and mid slots of the same pixel with two ds_write_b64 afterwards (LDS operations of a wave execute in order);
//   * LDS for the second patch comes from the weights: a ring of three one-tap slabs (48 KB) instead of two three-tap stages
//     (96 KB), refilled two taps ahead by LDS-DMA; one workgroup barrier per tap, behind a COUNTED s_waitcnt vmcnt that leaves
//     the youngest slab (and, at taps 0 and 1, the patch copy) in flight. All loads of the loop are LDS-DMA, so the counts are
//     exact (cdna_hip_programming.md, "Pipelining across barriers"); the barrier is the raw s_barrier.
// The skip segment (1x1, one tap per chunk) has no MFMA time to hide a transform behind: its patch is transformed right after
// the tap, as before. LDS: 2 x 43.5 KB patches + 48 KB ring + coefficient cache (<= 18 KB) <= 155 KB.
#include "kernels.h"
#include "conv_split.h"

namespace cddpm {

#define OV_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define OV_BARRIER() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }

__global__ __launch_bounds__(512) void conv_ov_kernel(const ConvArgs a) {
    constexpr int ROWS = 8, THREADS = 512;
    constexpr int PW = 34, PH = ROWS + 2, NPIX = PW * PH;      // 340 patch pixels
    constexpr int NK = 6;                                      // 16-B patch entries per thread (NPIX * 8 / 512, rounded up)
    constexpr int WSLOTS = 1024;                               // 16-B slots of a one-tap weight slab (128 rows x 8 slots)
    constexpr int NSLAB = 3;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    extern __shared__ v4f lds[];
    v4f* ldsP = lds;                              // 2 patch buffers of NPIX * 8 slots
    v4f* ldsW = lds + 2 * NPIX * 8;               // NSLAB slabs
    v4f* ldsC = ldsW + NSLAB * WSLOTS;            // GroupNorm/FiLM coefficients of this sample (3 x Cin floats)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3;            // pixel rows {2 wm, 2 wm + 1} of the tile
    const int wn = wave >> 2;           // cout half

    const int ncb = a.Cout >> 7;
    const int tilesX = (a.W + 31) >> 5;
    const int tilesY = (a.H + ROWS - 1) / ROWS;
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous tile ranges (conv_x6.hip)
    const int cb = bid % ncb;
    bid /= ncb;
    const int nks = a.ksplit > 1 ? a.ksplit : 1;
    const int ks = bid % nks;
    bid /= nks;
    const int tx = bid % tilesX;
    bid /= tilesX;
    const int ty = bid % tilesY;
    const int b = bid / tilesY;
    const int y0 = ty * ROWS, x0 = tx * 32;

    auto swz16 = [](int pc) -> int { return (2 * ((pc >> 1) & 3)) ^ (4 * (pc & 1)); };   // patch swizzle by pixel column (conv_x6.hip)

    const int Cin = a.C0 + a.C1;
    const int nch_main = Cin >> 5;
    const int nch = nch_main + ((a.S0 + a.S1) >> 5);
    const int kc0 = a.ksplit > 1 ? a.kbound[ks] : 0;
    const int kc1 = a.ksplit > 1 ? a.kbound[ks + 1] : nch;
    const int nmain = max(0, min(kc1, nch_main) - kc0);                 // main-segment chunks of this workgroup
    const int nsteps = nmain * 9 + (kc1 - kc0 - nmain);                 // weight slabs it multiplies: 9 per main chunk, 1 per skip chunk

    // ---- per-thread patch entries: channel quad c4 fixed per thread, pixel q = (tid >> 3) + 64 k
    const int c4 = tid & 7;
    int psrc[NK];
    unsigned centre = 0, colswz = 0, inimg = 0;       // per entry k: bit k of centre / inimg, bits 3k..3k+2 of colswz
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int q = (tid >> 3) + (THREADS / 8) * k;
        const int pr = q / PW, pc = q - pr * PW;
        colswz |= (unsigned)swz16(pc) << (3 * k);
        const int y = y0 + pr - 1, x = x0 + pc - 1;
        const bool valid = (q < NPIX) && (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
        const int sy = a.upsample ? (y >> 1) : y, sx = a.upsample ? (x >> 1) : x;
        psrc[k] = valid ? (sy * a.srcW + sx) : -1;
        if (valid) inimg |= 1u << k;
        if (valid && (pr >= 1) && (pr < PH - 1) && (pc >= 1) && (pc < PW - 1)) centre |= 1u << k;
    }

    const v4f* wmain = reinterpret_cast<const v4f*>(a.wpk) + (size_t)cb * nch_main * 9 * WSLOTS;
    const v4f* wskip = reinterpret_cast<const v4f*>(a.skip_wpk) + (size_t)cb * (nch - nch_main) * WSLOTS;
    const bool have_coef = (a.coef != nullptr);

    // weight slab of step s -> ring slot s % 3: each wave copies one contiguous eighth (2 KB = two 1-KB wave-instructions)
    auto dma_w = [&](int s, int slot) {
        const v4f* p = (s < nmain * 9) ? (wmain + ((size_t)kc0 * 9 + s) * WSLOTS)
                                       : (wskip + ((size_t)(max(kc0, nch_main) - nch_main) + (s - nmain * 9)) * WSLOTS);
        const v4f* g0 = p + wave * 128 + lane;
        v4f* d0 = ldsW + slot * WSLOTS + wave * 128;
        __builtin_amdgcn_global_load_lds((gptr_t)g0, (lptr_t)d0, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)g0, (lptr_t)d0, 16, 1024, 0);
    };
    // raw fp32 patch of `chunk` -> patch buffer pb: six wave-instructions per wave (lane = (pixel, quad), lane-linear in LDS).
    // Lanes without a source pixel (zero padding, the halo of a skip chunk) copy a valid dummy address: transform_entry never uses
    // the value. Every wave issues exactly six pieces (the vmcnt counts below rely on it): where a wave's last piece lies beyond
    // the patch it repeats its fifth.
    auto dma_p = [&](int chunk, int pb) {
        const float* base;
        int Cs, c0;
        if (chunk < nch_main) {
            const int ch = chunk << 5;
            if (ch < a.C0) { base = a.src0; Cs = a.C0; c0 = ch; }
            else           { base = a.src1; Cs = a.C1; c0 = ch - a.C0; }
        } else {
            const int ch = (chunk - nch_main) << 5;
            if (ch < a.S0) { base = a.skip0; Cs = a.S0; c0 = ch; }
            else           { base = a.skip1; Cs = a.S1; c0 = ch - a.S0; }
        }
        const float* ubase = base + ((size_t)b * a.srcH * a.srcW * Cs + c0);
        v4f* dbase = ldsP + pb * (NPIX * 8) + wave * 64;           // slot of (pixel 8 wave, quad 0)
        auto src_of = [&](int p) -> const float* {
            asm volatile("" : "+v"(p));      // opaque: keeps the six offsets of each possible source from being precomputed (and kept) outside the chunk loop
            return ubase + ((p >= 0) ? (__umul24((unsigned)p, (unsigned)Cs) + 4u * (unsigned)c4) : 0u);
        };
#pragma unroll
        for (int k = 0; k < NK - 1; ++k)
            __builtin_amdgcn_global_load_lds((gptr_t)src_of(psrc[k]), (lptr_t)(dbase + k * 512), 16, 0, 0);
        if (wave >= 3) {         // (wave-uniform) the sixth piece would lie beyond the patch: repeat the fifth
            __builtin_amdgcn_global_load_lds((gptr_t)src_of(psrc[NK - 2]), (lptr_t)(dbase + (NK - 2) * 512), 16, 0, 0);
        } else if ((tid >> 3) + (THREADS / 8) * (NK - 1) < NPIX) {      // waves 0..2: the piece that ends inside the last patch rows
            __builtin_amdgcn_global_load_lds((gptr_t)src_of(psrc[NK - 1]), (lptr_t)(dbase + (NK - 1) * 512), 16, 0, 0);
        }
    };
    // entry k of `chunk` in patch buffer pb: raw fp32 quad -> activation -> (hi, mid) fp16 quads, written over the raw bytes
    auto transform_entry = [&](int chunk, int pb, int k) {        // k may be a run-time value: only bit masks are indexed by it
        const int q = (tid >> 3) + (THREADS / 8) * k;
        if (q >= NPIX) return;
        const bool main_seg = chunk < nch_main;
        v4f* pix = ldsP + pb * (NPIX * 8) + q * 8;
        v4f v = pix[c4];
        const bool live = (((main_seg ? inimg : (inimg & centre)) >> k) & 1u) != 0;
        if (live) {
            if (main_seg && have_coef) {
                const int ci = (chunk << 3) + c4;
                v = (v - ldsC[ci]) * ldsC[(Cin >> 2) + ci] + ldsC[2 * (Cin >> 2) + ci];
            }
            if (main_seg && a.silu) v = silu_x6_v4(v);
        } else {
            v = v4f{0.f, 0.f, 0.f, 0.f};        // zero padding stays exactly zero: the conv pads AFTER the activation
        }
        f16x4 t[2];
        split_x4<2>(v, t);
        v2f* dst = reinterpret_cast<v2f*>(pix);
        const int sw = (int)((colswz >> (3 * k)) & 7u);
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) dst[(((4 * sp + (c4 >> 1)) ^ sw) << 1) + (c4 & 1)] = __builtin_bit_cast(v2f, t[sp]);
    };

    f32x4 acc16[4][4], tot16[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; tot16[i][j] = acc16[i][j]; }

    // fragment slots (conv_x6.hip): lane (r = lane & 15, g = lane >> 4) holds row r, channels 8 g .. 8 g + 7 of a 16 x 32 tile
    int a_hi[3], b_hi;
    {
        const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int pc = r16 + kx;
            a_hi[kx] = ((2 * wm) * PW + pc) * 8 + (g ^ swz16(pc));
        }
        const int row = 64 * wn + r16;
        b_hi = row * 8 + (g ^ ((row >> 1) & 7));
    }
    // one tap: 48 MFMAs per wave (4 pixel groups x 4 cout groups x {mid*hi, hi*mid, hi*hi}); `first` restarts the chains on C = 0
    auto compute = [&](int ky, int kx, const v4f* pa, const v4f* wb, bool first) {
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        f16x8 fa[2][4];
        const v4f* ah = pa + (a_hi[kx] + ky * (PW * 8));
        const v4f* am = pa + ((a_hi[kx] ^ 4) + ky * (PW * 8));
        const v4f* bh = wb + b_hi;
        const v4f* bm = wb + (b_hi ^ 4);
#pragma unroll
        for (int t16 = 0; t16 < 4; ++t16) {
            const int off = ((t16 >> 1) * PW + (t16 & 1) * 16) * 8;
            fa[0][t16] = __builtin_bit_cast(f16x8, ah[off]);
            fa[1][t16] = __builtin_bit_cast(f16x8, am[off]);
        }
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            f16x8 fb[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fb[0][j] = __builtin_bit_cast(f16x8, bh[128 * (2 * nh + j)]);
                fb[1][j] = __builtin_bit_cast(f16x8, bm[128 * (2 * nh + j)]);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int sa = (p == 0) ? 1 : 0, sb = (p == 1) ? 1 : 0;
                if (p == 0 && first) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc16[i][2 * nh + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[sa][i], fb[sb][j], zero4, 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc16[i][2 * nh + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[sa][i], fb[sb][j], acc16[i][2 * nh + j], 0, 0, 0);
                }
            }
        }
    };
    auto fold_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) tot16[i][j] += acc16[i][j];
    };

    // ---- prologue: slabs 0 and 1, the first patch, the coefficient cache; transform the first patch
    dma_w(0, 0);
    dma_w(min(1, nsteps - 1), 1);
    dma_p(kc0, 0);
    if (have_coef) {
        const int nq = Cin >> 2;
        const size_t plane = (size_t)a.B * Cin;
        for (int i = tid; i < 3 * nq; i += THREADS) {
            const int pl = i / nq, cq = i - pl * nq;
            ldsC[i] = *reinterpret_cast<const v4f*>(a.coef + pl * plane + (size_t)b * Cin + 4 * cq);
        }
    }
    OV_WAIT_VM(0);
    OV_BARRIER();                       // the coefficient cache is complete (every thread's own patch entries have landed: vmcnt)
#pragma unroll 1
    for (int k = 0; k < NK; ++k) transform_entry(kc0, 0, k);      // (one entry at a time: six in flight would cost 100 registers)
    OV_BARRIER();

    // ---- main loop. Invariant at the top of step s: slabs s and s + 1 are requested (s landed), patch buffer `pb` holds chunk's
    //      split patch, every wave is past step s - 1 (so ring slot (s + 2) % 3 and, at the first step of a chunk, the other patch
    //      buffer are free).
    int s = 0;
    for (int chunk = kc0; chunk < kc1; ++chunk) {
        const int pb = (chunk - kc0) & 1;
        const v4f* pa = ldsP + pb * (NPIX * 8);
        const bool more = chunk + 1 < kc1;
        if (chunk < nch_main) {
            // Branch-free body: every tap requests a slab and waits with the same count. Past the last slab the request repeats
            // the last one into the free ring slot, and in the last chunk the patch copy / transform run on the current chunk into
            // the idle patch buffer -- a few KB of redundant traffic per workgroup instead of divergent wait counts.
            const int cn = more ? chunk + 1 : chunk;
            for (int ky = 0; ky < 3; ++ky) {                   // (a real loop: three unrolled taps per iteration keep the code and the
#pragma unroll                                                 //  register pressure of a row of taps, not of nine)
                for (int kx = 0; kx < 3; ++kx) {
                    dma_w(min(s + 2, nsteps - 1), (s + 2) % NSLAB);
                    if (ky == 0 && kx == 0) dma_p(cn, pb ^ 1);
                    compute(ky, kx, pa, ldsW + (s % NSLAB) * WSLOTS, kx == 0);
                    if (kx == 2) fold_acc();
                    if (ky >= 1) transform_entry(cn, pb ^ 1, 3 * (ky - 1) + kx);
                    // slab s + 1 must have landed; younger requests stay in flight: the 2 pieces of slab s + 2 and, at taps 0 / 1, the
                    // 6 pieces of the patch copy issued behind slab s + 2 / s + 1 (from tap 2 on the patch has landed: transform_entry
                    // may read it)
                    if (ky == 0 && kx < 2) { OV_WAIT_VM(8); } else { OV_WAIT_VM(2); }
                    OV_BARRIER();
                    ++s;
                }
            }
        } else {
            // skip segment: one centre tap per chunk
            dma_w(min(s + 2, nsteps - 1), (s + 2) % NSLAB);
            if (more) dma_p(chunk + 1, pb ^ 1);
            compute(1, 1, pa, ldsW + (s % NSLAB) * WSLOTS, true);
            fold_acc();
            OV_WAIT_VM(0);
            if (more) {
#pragma unroll 1
                for (int k = 0; k < NK; ++k) transform_entry(chunk + 1, pb ^ 1, k);
            }
            OV_BARRIER();
            ++s;
        }
    }

    OV_WAIT_VM(0);                      // (the redundant requests of the last steps)
    OV_BARRIER();
    // ---- epilogue (as conv_x6.hip): residual requests, per-wave LDS transpose, bias / scale / residual, 16-B stores, statistics
    // (the lane index is made opaque here so that the epilogue's address arithmetic is not computed -- and kept live, or spilled --
    // in front of the main loop)
    int elane = lane;
    asm volatile("" : "+v"(elane));
    v4f rsd_all[2][2][4];
    {
        const int cq = elane & 7, prow = elane >> 3;
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
                    if (a.res && (gy < a.H) && (gx < a.W)) {
                        const size_t rp = a.res_up ? ((size_t)(b * (a.H >> 1) + (gy >> 1)) * (a.W >> 1) + (gx >> 1))
                                                   : ((size_t)(b * a.H + gy) * a.W + gx);
                        r = *reinterpret_cast<const v4f*>(a.res + rp * a.Cout + co);
                    }
                    rsd_all[nt][hb][i] = r;
                }
    }
    {
        constexpr int TRS = 36;
        float* tr = reinterpret_cast<float*>(lds) + wave * (64 * TRS);     // aliases the two patch buffers (73.7 KB <= 87 KB)
        const int cq = elane & 7;
        const int prow = elane >> 3;
        const int tilesY4 = (a.H + 3) >> 2;
        const int ty4 = (y0 >> 2) + (wm >> 1);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int co = cb * 128 + 64 * wn + 32 * nt + 4 * cq;
#pragma unroll
            for (int t16 = 0; t16 < 4; ++t16)
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        tr[(16 * t16 + 4 * (elane >> 4) + r) * TRS + 16 * n2 + (elane & 15)] = tot16[t16][2 * nt + n2][r];
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
                    ok[i] = (gy < a.H) && (gx < a.W);
                    oidx[i] = ((size_t)((ks * a.B + b) * a.H + gy) * a.W + gx) * a.Cout + co;
                    val[i] = *reinterpret_cast<const v4f*>(tr + p * TRS + 4 * cq);
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

bool conv_ov_applicable(const ConvArgs& a) {
    // 3x3, fp16-split family, and the coefficient cache must fit beside two patches and the slab ring
    const size_t coef_lds = a.coef ? (size_t)3 * (a.C0 + a.C1) * sizeof(float) : 0;
    return a.taps == 9 && (size_t)(2 * 340 * 8 + 3 * 1024) * 16 + coef_lds <= 160 * 1024;
}

void launch_conv_ov(const ConvArgs& a, hipStream_t stream) {
    const int tilesX = (a.W + 31) / 32, tilesY = (a.H + 7) / 8;
    const unsigned grid = (unsigned)(a.B * tilesX * tilesY * (a.Cout / 128) * (a.ksplit > 1 ? a.ksplit : 1));
    const size_t coef_lds = a.coef ? (size_t)3 * (a.C0 + a.C1) * sizeof(float) : 0;
    size_t need = (size_t)(2 * 340 * 8 + 3 * 1024) * 16 + coef_lds;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ov_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(conv_ov_kernel, dim3(grid), dim3(512), need, stream, a);
}

}  // namespace cddpm
