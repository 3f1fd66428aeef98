// Fused implicit-GEMM convolution for gfx950 (MI355X): fp32 accuracy on the 16-bit matrix pipe.
//
// Same operator as conv_mfma.hip (see there for what is fused and the reference lines it replaces:
// src/models/modules/OpenAI_Unet.py:284-338, :386-394, :118-128, :948), same arguments, same results to fp32
// rounding. What changes is how a product of two fp32 numbers reaches the accumulator: every fp32 operand is split
// into a few 16-bit terms whose sum reproduces it to fp32 precision, the partial products are exact in fp32 and the
// 16-bit MFMA (16x the rate of the fp32 MFMA: 2.5 PFLOP/s vs 157 TFLOP/s dense) accumulates them in fp32.
//
//   NS = 2, fp16 (default, CDDPM_CONV=h3):  hi = fp16(x), mid = fp16(x - hi), round-to-nearest: two 11-bit terms,
//       |x - hi - mid| <= 2^-23 |x| (one fp32 ulp at worst; 75 % of fp32 values are reproduced exactly, rms error
//       0.73 x 2^-24 |x|) while mid is a normal fp16, i.e. |x| >= 2^-2; below that the error is ABSOLUTE, <= 2^-25.
//       a*b = hi*hi + hi*mid + mid*hi (+ mid*mid, dropped: |mid| <= 2^-11 |x|, so <= 2^-22 |ab| at worst, 2^-24 typical): 3 MFMAs
//       (v_mfma_f32_32x32x16_f16) = 3/16 of the fp32-MFMA cost. fp16's exponent range needs care: weights are
//       pre-scaled by a power of two chosen per convolution so that max|w| lands in [2^13, 2^14) (exact; the epilogue
//       multiplies by the inverse), activations are used as they are -- after GroupNorm/FiLM/SiLU they are O(1), the
//       domain is |act| < 65504 (beyond it fp16 overflows to inf and the result is NaN: loud, not silently wrong).
//   NS = 3, bf16 (CDDPM_CONV=x6): hi, mid, lo = three 8-bit terms, exact over the whole fp32 range;
//       a*b = hh + hm + mh + mm + hl + lh (+ terms <= 2^-24 |ab|): 6 MFMAs (v_mfma_f32_32x32x16_bf16) = 6/16.
//
// Against fp64 (tools/ubench/bf16_split_accuracy.hip, K = 4608, SiLU-distributed activations, folded every 96
// products): fp16x3 1.9e-7, bf16x6 2.0e-7 of rms(C); the fp32 MFMA chain folded per 288: 3.2e-7; the reference's CPU
// fmaf chain 1.2e-6. On the whole reverse chain the rounding noise against float64 is below the reference's own
// (tools/chain_noise.py).
//
// GEMM view:  D[pixel][cout] = sum_{tap, ci} act(X)[pixel + tap][ci] * Wt[tap][ci][cout]
//   M = 256 pixels (8 image rows x 32 columns), N = 128 output channels, K step = 32 input channels x 1 tap
//   = 2 MFMA k-steps of 16. 8 waves (2 per SIMD), each owns 64 pixels x 64 couts = 2 x 2 MFMA tiles.
// LDS (one workgroup per CU): per pixel / per cout row SP = 4 NS slots of 16 B (slot = split s, u = channel / 8):
//   NS = 3: stride 12 slots, slot (s, u) at 4 s + (u ^ ((row >> 2) & 3));  NS = 2: stride 8, (4 s + u) ^ ((row >> 1) & 7)
//   -> 16 consecutive rows x one slot cover all 16 four-bank groups: ds_read_b128 of a fragment is conflict free.
//   act patch (8+2) x (32+2) pixels (65 | 43.5 KB) + 2 weight stages (bf16: one slab [128 cout][SP] = 24.6 KB each;
//   fp16: a row of three taps = 3 x 16 KB each) + coefficient cache;
//   the packed global weight image (host-split, pack_conv_weights_split) IS the LDS image: staging is a 16-B copy.
// A lane's 16-B fragment = 8 consecutive channels = its K elements of one k-step (lane>>5 selects the half).
#include "kernels.h"
#include "conv_split.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace cddpm {

#if defined(CDDPM_STAMPS_EPI)
// accounting of the fixed per-workgroup cost (tools/conv_ab.py, AB_EPI=1): 0 prologue (kernel start .. first chunk), 1 main loop,
// 2 residual requests, 3 the barrier in front of the epilogue, 4 / 6 transpose of cout half 0 / 1 through LDS, 5 / 7 its stores + statistics
#define STAMP(i)
#define FSTAMP(i)
#define ESTAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_[i] += now_ - last_; last_ = now_; }
#elif defined(CDDPM_STAMPS_FINE)
// finer accounting of the LDS-DMA main loop (tools/conv_ab.py, AB_FINE=1): 0 wait for the chunk's first weight stage and patch
// registers (vmcnt), 1 the barrier behind it, 2 patch transform + store, 3 the barrier behind it, 4 wait for the next weight stage at
// the end of a stage (vmcnt), 5 the barrier behind it, 6 request issue + MFMA compute + fold, 7 prologue + epilogue
#define STAMP(i)
#define FSTAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_[i] += now_ - last_; last_ = now_; }
#define ESTAMP(i)
#elif defined(CDDPM_STAMPS)
// phase accounting for diagnostic builds: 0 prologue, 1 patch stage (barrier + transform + split + ds_write), 2 weight
// stage (ds_write + prefetch issue + barrier), 3 MFMA compute, 4 fold, 5 epilogue
#define STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_[i] += now_ - last_; last_ = now_; }
#define FSTAMP(i)
#define ESTAMP(i)
#else
#define STAMP(i)
#define FSTAMP(i)
#define ESTAMP(i)
#endif

// M16X: the fp16 form on v_mfma_f32_16x16x32_f16 (one MFMA spans a tap's whole 32-channel chunk; 16 tiles of 16 x 16 per
// wave) instead of 32x32x16: the same cycle count, but the chip holds a higher clock on this shape -- the default of the
// fp16 form (the 32x32x16 instances stay for CDDPM_M16=0); half as many accumulate roundings per product.
// HI1: multiply the hi terms only (plain fp16 operands, fp32 accumulation: the training operators under CDDPM_TRAIN_PRECISION=16); its own
// instantiation, so that the reconstruction path's kernel is unchanged by it
// NB: cout blocks of 128 per workgroup. NB = 2 (16 x 16 form, LDS-DMA loop only; a.nb2): the workgroup's tile is 256 pixels x 256 couts,
// multiplied as two passes over N per 32-channel chunk -- the chunk's patch (GroupNorm/FiLM/SiLU transform, fp16 split, two barriers) is
// produced ONCE and read by both passes, where two NB = 1 workgroups each produce it. The two 64 x 64 wave tiles' accumulators are the
// registers `acc` and `tot` take at NB = 1, so the three-level accumulation becomes two-level: every MFMA accumulates into the output's
// one long-lived chain (rounding noise of a K = 4608 dot product 7.6e-7 of rms(C) instead of 1.9e-7; the reference's CPU fmaf chain:
// 1.2e-6 -- tools/ubench/bf16_split_accuracy.hip, rows "h3" / "h3 fold96" / "cpu fma32").
template <int TAPS, int ROWS, int NS, bool M16X = false, bool HI1 = false, int NB = 1>
__global__ __launch_bounds__(64 * ROWS) void conv_split_kernel(const ConvArgs a) {
    static_assert(NB == 1 || (NB == 2 && M16X && NS == 2 && TAPS != 1), "two cout blocks per workgroup: 16 x 16 fp16 form, LDS-DMA loop (3x3 and folded 2x2) only");
    typedef typename SplitT<NS>::v8 frag;
    constexpr int SP = 4 * NS;                          // 16-B slots per pixel / per cout row
    constexpr int THREADS = 64 * ROWS;
    // TAPS == 4: folded "nearest x2 upsample -> 3x3 conv", one parity class of the output per tile (see conv_mfma.hip)
    constexpr bool UP2 = (TAPS == 4);
    constexpr int PAD = (TAPS == 9) ? 1 : 0;
    constexpr int PW = UP2 ? 33 : 32 + 2 * PAD;         // patch width  (pixels)
    constexpr int PH = UP2 ? ROWS + 1 : ROWS + 2 * PAD; // patch height (pixels)
    constexpr int NPIX = PW * PH;                       // 340 | 256 | 297 at ROWS = 8
    constexpr int NK = (NPIX * 8 + THREADS - 1) / THREADS;   // 16-B (4-channel) patch entries per thread: 6 | 4 | 5
#ifndef CDDPM_X6_FOLD
#define CDDPM_X6_FOLD 3
#endif
#ifndef CDDPM_GLDS
#define CDDPM_GLDS 1
#endif
// (the timing-only ablation builds of rounds 1-2 -- NOBARRIER / NOFOLD / NOFRAG / NODMA, wrong results by construction -- are no longer
// switches of this file: tools/ubench/conv_x6_ablation_switches.patch re-creates them for tools/conv_ab.py)
#define LOOP_BARRIER() __syncthreads()
    constexpr int FOLD = (TAPS == 9) ? CDDPM_X6_FOLD : (TAPS == 4 ? 2 : 1);   // taps per accumulation group
    constexpr int WSLOTS = 128 * SP;                    // 16-B slots of a weight slab
    // taps per weight stage: the fp16 form stages a whole row of taps (3 of the 3x3, 2 of the folded 2x2) per workgroup
    // barrier -- one barrier (and one burst of fragment reads behind it) per 72 MFMAs of a wave instead of per 24, and the
    // stage is exactly one accumulation group (FOLD taps). The bf16 form's slabs are 1.5x larger: one tap per stage.
    constexpr int TPS = (NS == 2) ? FOLD : 1;
    constexpr int WSTAGE = TPS * WSLOTS;                // 16-B slots of one weight stage
    constexpr int WK = WSLOTS / THREADS;                // 16-B pieces of ONE slab per thread: 3 | 2
    static_assert(WSLOTS % THREADS == 0, "weight slab must divide evenly");

    extern __shared__ v4f lds[];
    v4f* ldsA = lds;                    // NPIX * SP slots
    v4f* ldsW = lds + NPIX * SP;        // 2 * WSTAGE
    v4f* ldsC = ldsW + 2 * WSTAGE;      // GroupNorm/FiLM coefficients of this sample (3 x Cin floats)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
#if defined(CDDPM_STAMPS) || defined(CDDPM_STAMPS_FINE) || defined(CDDPM_STAMPS_EPI)
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
    const unsigned long long t0c_ = last_, t0r_ = __builtin_amdgcn_s_memrealtime();
#endif
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int wm = wave % (ROWS / 2);   // pixel rows {2 wm, 2 wm + 1} of the tile
    const int wn = wave / (ROWS / 2);   // cout half

    const int ncb = a.Cout >> 7;              // cout blocks of the weight image
    const int ncbw = ncb / NB;                // ... per workgroup column: this workgroup multiplies blocks NB cb .. NB cb + NB - 1
    const int gridH = UP2 ? (a.H >> 1) : a.H, gridW = UP2 ? (a.W >> 1) : a.W;
    const int tilesX = (gridW + 31) >> 5;
    const int tilesY = (gridH + ROWS - 1) / ROWS;
    // Workgroups go to the 8 XCDs round-robin (blockIdx % 8), each with its own L2. Give every XCD a CONTIGUOUS range of
    // tiles instead: the cout blocks of one tile and the tiles above / below it (shared halo rows) then run on one XCD at
    // about the same time, and the second reader of a patch row finds it in that L2.
    int bid = blockIdx.x;
#ifndef CDDPM_NO_XCD_REMAP
    if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#endif
    const int cb = bid % ncbw;
    bid /= ncbw;
    // split-K (small batches, cddpm_api.hip::conv_launch): this workgroup multiplies the chunks kbound[ks] .. kbound[ks + 1] only
    // and stores its raw sums to plane ks of `out`; conv_reduce_kernel adds the planes in the order of ks
    const int nks = a.ksplit > 1 ? a.ksplit : 1;
    const int ks = bid % nks;
    bid /= nks;
    const int tx = bid % tilesX;
    bid /= tilesX;
    const int ty = bid % tilesY;
    bid /= tilesY;
    const int cls = UP2 ? (bid & 3) : 0;
    const int b = UP2 ? (bid >> 2) : bid;
    const int pa = cls >> 1, pb = cls & 1;
    const int y0 = ty * ROWS, x0 = tx * 32;

    // 16-B slot of (row = pixel or cout row, split s, u = channel / 8)
    auto slot_of = [](int row, int sp, int u) -> int {
        return (NS == 3) ? (row * 12 + 4 * sp + (u ^ ((row >> 2) & 3))) : (row * 8 + ((4 * sp + u) ^ ((row >> 1) & 7)));
    };
    // ... of the patch in the fp16 form. The weight image above is fixed by the host packer; the patch is written here, so
    // its swizzle follows the reads: a ds_read_b128 is served in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,
    // 28-31}, +32) over 64 banks of 4 B, a ds_write_b64 in 16 consecutive lanes over 32 banks (MI355X_MICROARCH.md, LDS).
    //   32x32x16: a group reads 16 consecutive pixels at one u  -> pixel bits 1..3 into the slot (as the weights), and pixel
    //             bit 0 flips the split half so that the two pixels one store instruction covers do not share banks;
    //   16x16x32: a group reads 8 + 8 pixels at u and u ^ 1, from ANY start pixel (tap shifts) -> slot bit 0 must stay u's,
    //             so only bits 1, 2 are swizzled -- by the pixel's COLUMN in the patch (bits 1, 2 -> slot bit 1, 2; bit 0 ->
    //             slot bit 2; the patch width is even, so column and pixel parity agree): a tap's row shift ky is then a
    //             plain address offset and the fragment addresses are computed once per kernel, not once per tap.
    // Both are conflict-free for every tap; with the weights' swizzle the 16x16 reads took 7 LDS cycles instead of 4.
    constexpr bool M16S = M16X && (NS == 2);
    auto swz16 = [](int pc) -> int { return (2 * ((pc >> 1) & 3)) ^ (4 * (pc & 1)); };
    auto slot_a = [](int row, int sp, int u) -> int {      // (not the 16x16 form)
        if (NS == 3) return row * 12 + 4 * sp + (u ^ ((row >> 2) & 3));
        return row * 8 + ((4 * sp + u) ^ ((row >> 1) & 7) ^ (4 * (row & 1)));
    };

    const int Cin = a.C0 + a.C1;
    const int nch_main = Cin >> 5;
    const int nch_skip = (a.S0 + a.S1) >> 5;
    const int nch = nch_main + nch_skip;
    const int kc0 = a.ksplit > 1 ? a.kbound[ks] : 0;          // first chunk of this workgroup
    const int kc1 = a.ksplit > 1 ? a.kbound[ks + 1] : nch;    // one past its last chunk

    // ---- per-thread patch entries: channel quad c4 is fixed per thread, pixel q = (tid>>3) + (THREADS/8) k
    const int c4 = tid & 7;
    int psrc[NK];
    unsigned centre = 0;
    unsigned colswz = 0;                 // 16x16 form: swz16(column) of entry k in bits 3k .. 3k+2
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int q = (tid >> 3) + (THREADS / 8) * k;
        const int pr = q / PW, pc = q - pr * PW;
        if constexpr (M16S) colswz |= (unsigned)swz16(pc) << (3 * k);
        const int y = UP2 ? (y0 + pr + pa - 1) : (y0 + pr - PAD), x = UP2 ? (x0 + pc + pb - 1) : (x0 + pc - PAD);
        const bool valid = (q < NPIX) && (y >= 0) && (y < gridH) && (x >= 0) && (x < gridW);
        const int sy = (!UP2 && a.upsample) ? (y >> 1) : y, sx = (!UP2 && a.upsample) ? (x >> 1) : x;
        psrc[k] = valid ? (sy * a.srcW + sx) : -1;           // pixel index inside sample b (< 2^24)
        if (valid && (pr >= PAD) && (pr < PH - PAD) && (pc >= PAD) && (pc < PW - PAD)) centre |= 1u << k;
    }

    const v4f* wmain = reinterpret_cast<const v4f*>(a.wpk) + (size_t)(cls * ncb + NB * cb) * nch_main * TAPS * WSLOTS;
    const v4f* wskip = reinterpret_cast<const v4f*>(a.skip_wpk) + (size_t)(NB * cb) * nch_skip * WSLOTS;
    const size_t wmain_blk = (size_t)nch_main * TAPS * WSLOTS, wskip_blk = (size_t)nch_skip * WSLOTS;    // next cout block of the image

    v4f wreg[TPS * WK];
    v4f areg[NK];
    const bool have_coef = (a.coef != nullptr);

    auto wslab = [&](int chunk, int tap) -> const v4f* {
        if (chunk >= nch) { chunk = 0; tap = 0; }      // past the end: wrap, so the prefetch stays unconditional
        return (chunk < nch_main) ? (wmain + ((size_t)chunk * TAPS + tap) * WSLOTS)
                                  : (wskip + (size_t)(chunk - nch_main) * WSLOTS);
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
        const float* ubase = base + ((size_t)b * a.srcH * a.srcW * Cs + c0);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int p = (main_seg || ((centre >> k) & 1u)) ? psrc[k] : -1;
            v4f v = v4f{0.f, 0.f, 0.f, 0.f};
            // wave-uniform 64-bit base (sample, channel chunk) + a 32-bit per-lane offset: one v_mad_u32_u24 per entry
            if (p >= 0) v = *reinterpret_cast<const v4f*>(ubase + (__umul24((unsigned)p, (unsigned)Cs) + 4u * (unsigned)c4));
            areg[k] = v;
        }
    };
    // The patch of a chunk is produced in two steps so that the arithmetic can run BEFORE the chunk barrier: transform_act turns the
    // prefetched raw values into their 16-bit split terms in registers (GroupNorm/FiLM affine, SiLU, split: no LDS writes, the
    // coefficient cache is read-only), write_act stores them to the patch once every wave is done reading the previous one.
    // A wave that finishes its MFMAs early (the older wave of a SIMD wins the matrix pipe) transforms while its partner still
    // multiplies, instead of waiting at the barrier and transforming afterwards.
    typename SplitT<NS>::v4 sreg[NK][NS];
    auto transform_act = [&](int chunk) {
        const bool main_seg = chunk < nch_main;
        const bool do_silu = main_seg && a.silu;
        v4f cm = v4f{0.f, 0.f, 0.f, 0.f}, ca = v4f{1.f, 1.f, 1.f, 1.f}, cd = cm;
        const bool aff = main_seg && have_coef;
        if (aff) {
            const int ci = (chunk << 3) + c4;
            cm = ldsC[ci];
            ca = ldsC[(Cin >> 2) + ci];
            cd = ldsC[2 * (Cin >> 2) + ci];
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            v4f v = areg[k];
            const int p = (main_seg || ((centre >> k) & 1u)) ? psrc[k] : -1;
            if (p >= 0) {   // zero padding stays exactly zero: the conv pads AFTER the activation
                v = (v - cm) * ca + cd;         // (identity coefficients where there is no GroupNorm: exact, and no select)
                if (do_silu) {
                    if constexpr (NS == 2) v = silu_x6_v4(v);
                    else { v.x = silu_x6(v.x); v.y = silu_x6(v.y); v.z = silu_x6(v.z); v.w = silu_x6(v.w); }
                }
            }
            split_x4<NS>(v, sreg[k]);
        }
    };
    auto write_act = [&]() {
        v2f* dst = reinterpret_cast<v2f*>(ldsA);       // 8-B units
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int q = (tid >> 3) + (THREADS / 8) * k;
            if (q < NPIX) {
                // 4 channels = half a slot: slot u = c4 >> 1 of each split, half c4 & 1
#pragma unroll
                for (int sp = 0; sp < (HI1 ? 1 : NS); ++sp) {        // (hi-only products never read the mid plane: it is neither computed nor stored)
                    const int sl = M16S ? (q * 8 + ((4 * sp + (c4 >> 1)) ^ (int)((colswz >> (3 * k)) & 7u))) : slot_a(q, sp, c4 >> 1);
                    dst[sl * 2 + (c4 & 1)] = __builtin_bit_cast(v2f, sreg[k][sp]);
                }
            }
        }
    };
    auto store_act = [&](int chunk) { transform_act(chunk); write_act(); };

    constexpr bool M16 = M16X && (NS == 2);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x16 acc[2][2], tot[2][2];            // 32 x 32 form: [pixel row][cout 32-tile]
    f32x4 acc16[NB][4][4], tot16[4][4];     // 16 x 16 form: [cout block][pixel 16-group][cout 16-group]  (only one of the two sets is live; NB = 2: no tot16)
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tot16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < NB; ++h) acc16[h][i][j] = tot16[i][j];
            }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
    }
    auto fold_acc = [&]() {
        if constexpr (NB == 2) {
            // two-level accumulation: the MFMAs chain into acc16[h] from the first chunk to the last, nothing to fold
        } else if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) tot16[i][j] += acc16[0][i][j];
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) tot[i][j] += acc[i][j];
        }
    };

    // B operand (weights) LDS offsets: row j = cout within the 128 block
    int brow[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) brow[nt] = 64 * wn + 32 * nt + li;

    // 16 x 16 form: fragment slots, once per kernel. Lane (r = lane & 15, g = lane >> 4) holds row r, channels 8 g .. 8 g + 7
    // of a 16 x 32 operand tile. The column swizzle only looks at column bits 0..2 and the weight swizzle at cout bits 1..3,
    // so the four pixel groups (two rows x two 16-column halves) and the four cout groups of a wave sit at CONSTANT slot
    // offsets from the first one: one hi-term and one mid-term base (4 slots away under XOR) per column shift kx and one
    // pair for the weights; everything else is an instruction offset. A tap's ky adds ky * PW * 8 slots.
    constexpr int NKX = (TAPS == 9) ? 3 : (UP2 ? 2 : 1);
    static_assert(!M16 || TPS == NKX, "16x16 form: a weight stage is one row of taps, so the column shift is the unrolled index");
    int a_hi[NKX], a_mid[NKX], b_hi = 0, b_mid = 0;
    if constexpr (M16) {
        const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int kx = 0; kx < NKX; ++kx) {
            const int pc = r16 + kx;
            a_hi[kx] = ((2 * wm) * PW + pc) * 8 + (g ^ swz16(pc));
            a_mid[kx] = a_hi[kx] ^ 4;
        }
        b_hi = slot_of(64 * wn + r16, 0, g);
        b_mid = b_hi ^ 4;
    }

    // `first`: the accumulators restart here (C = 0 in the first MFMA of each tile, no register clearing).
    // kxc: the tap's column shift where the caller knows it at compile time (16 x 16 form: it selects registers); skip: the
    // 1x1 skip segment (centre tap).
    auto compute = [&](auto hc, int tap, int kxc, bool skip, const v4f* wb, bool first) {
        constexpr int HB = decltype(hc)::value;      // cout block of this pass (selects the accumulator registers)
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int ky = (TAPS == 9) ? (tap / 3) : (UP2 ? (tap >> 1) : 0);
        const int kx = (TAPS == 9) ? (tap - 3 * ky) : (UP2 ? (tap & 1) : 0);
        if constexpr (M16) {
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            f16x8 fa[2][4];
            const v4f* ah = ldsA + (((TAPS == 9 && skip) ? a_hi[NKX / 2] : a_hi[kxc]) + ky * (PW * 8));
            const v4f* am = ldsA + (((TAPS == 9 && skip) ? a_mid[NKX / 2] : a_mid[kxc]) + ky * (PW * 8));
            const v4f* bh = wb + b_hi;
            const v4f* bm = wb + b_mid;
#pragma unroll
            for (int t16 = 0; t16 < 4; ++t16) {
                const int off = ((t16 >> 1) * PW + (t16 & 1) * 16) * 8;      // compile-time: an instruction offset
                fa[0][t16] = __builtin_bit_cast(f16x8, ah[off]);
                fa[1][t16] = __builtin_bit_cast(f16x8, am[off]);
            }
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {            // two cout 16-groups at a time: 12 fragments live instead of 16
                f16x8 fb[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    fb[0][j] = __builtin_bit_cast(f16x8, bh[128 * (2 * nh + j)]);
                    fb[1][j] = __builtin_bit_cast(f16x8, bm[128 * (2 * nh + j)]);
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) {       // mid*hi, hi*mid, hi*hi
                    const int sa = (p == 0) ? 1 : 0, sb = (p == 1) ? 1 : 0;
                    if constexpr (HI1) { if (p < 2) continue; }         // plain fp16 operands: the hi x hi product only
                    if ((HI1 ? p == 2 : p == 0) && first) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc16[HB][i][2 * nh + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[sa][i], fb[sb][j], zero4, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc16[HB][i][2 * nh + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[sa][i], fb[sb][j], acc16[HB][i][2 * nh + j], 0, 0, 0);
                    }
                }
            }
            return;
        }
        int arow[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) arow[mt] = (2 * wm + mt + ky) * PW + li + kx;
        // products kept, smallest first: NS = 3: lh hl mm mh hm hh;  NS = 2: mh hm hh
        constexpr int NP = (NS == 3) ? 6 : 3;
        constexpr int PA[6] = {NS == 3 ? 2 : 1, 0, 1, 1, 0, 0};      // split index of the A term
        constexpr int PB[6] = {0, NS == 3 ? 2 : 1, NS == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
        for (int jk = 0; jk < 2; ++jk) {
            const int u = 2 * jk + lh;
            frag fa[NS][2], fb[NS][2];
#pragma unroll
            for (int sp = 0; sp < NS; ++sp)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[sp][i] = __builtin_bit_cast(frag, ldsA[slot_a(arow[i], sp, u)]);
                    fb[sp][i] = __builtin_bit_cast(frag, wb[slot_of(brow[i], sp, u)]);
                }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int sa = (NS == 3) ? PA[p] : (p == 0 ? 1 : 0);
                const int sb = (NS == 3) ? PB[p] : (p == 1 ? 1 : 0);
                if (jk == 0 && p == 0 && first) {       // uniform branch: restart the chains on an inline C = 0
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = SplitT<NS>::mfma(fa[sa][i], fb[sb][j], zero16);
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = SplitT<NS>::mfma(fa[sa][i], fb[sb][j], acc[i][j]);
                }
            }
        }
    };

    if constexpr (TPS == 1) {
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
            const int pl = i / nq, cq = i - pl * nq;
            ldsC[i] = *reinterpret_cast<const v4f*>(a.coef + pl * plane + (size_t)b * Cin + 4 * cq);
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
            for (int i = 0; i < WK; ++i) ldsW[buf * WSLOTS + tid + THREADS * i] = wreg[i];
            const bool last_tap = (t == ntap - 1);
            const v4f* pn = last_tap ? wslab(chunk + 1, 0) : wslab(chunk, t + 1);
#pragma unroll
            for (int i = 0; i < WK; ++i) wreg[i] = pn[tid + THREADS * i];
            if (last_tap && chunk + 1 < nch) load_act(chunk + 1);
            __syncthreads();
            STAMP(2)
            // accumulation in three levels: an MFMA sums 16 products, `acc` collects FOLD taps of a 32-channel chunk
            // (<= 96 products per chain), `tot` sums those groups. The rounding noise of an fp32 chain grows with the
            // magnitude of its partial sums, so short chains folded into a long-lived total keep it near the
            // storage-rounding level (tools/ubench/bf16_split_accuracy.hip, tools/chain_noise.py).
            compute(std::integral_constant<int, 0>{}, main_seg ? t : (TAPS / 2), 0, !main_seg, ldsW + buf * WSLOTS, (t % FOLD) == 0);   // skip segment: centre tap (16x16 form: 1x1 convolutions only get here)
            buf ^= 1;
            STAMP(3)
            if ((t % FOLD) == FOLD - 1 || last_tap) {
                fold_acc();
                STAMP(4)
            }
        }
    }

    } else if constexpr (CDDPM_GLDS != 0) {
    // ---- main loop, weights by LDS-DMA: a stage (TPS taps of one 32-channel chunk, 48 KB) is copied global -> LDS by
    //      global_load_lds_dwordx4 while the previous stage is multiplied -- the packed image IS the LDS image, so the copy is
    //      lane-linear; no weight registers, no ds_write pass. One barrier per stage, behind a vmcnt(0) that retires the copy.
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto wstage = [&](int chunk, int hb, int st, int& nsl) -> const v4f* {
        if (chunk < nch_main) { nsl = TPS; return wmain + hb * wmain_blk + ((size_t)chunk * TAPS + st * TPS) * WSLOTS; }
        nsl = 1;
        return wskip + hb * wskip_blk + (size_t)(chunk - nch_main) * WSLOTS;
    };
    // Each wave copies one contiguous eighth of a slab (2 KB = two 1-KB wave-instructions) per slab of the stage. The pieces of
    // a slab share one address pair: the instruction's immediate offset advances the global and the LDS address alike.
    static_assert(THREADS == 512 && WSLOTS == 1024, "one slab = 8 waves x 2 KB");
    auto dma_stage = [&](const v4f* p, int nsl, int buf) {
        const v4f* g0 = p + wave * 128 + lane;
        v4f* d0 = ldsW + buf * WSTAGE + wave * 128;          // wave-uniform; lane l lands 16 l bytes further
#pragma unroll
        for (int sl = 0; sl < TPS; ++sl)
            if (sl < nsl) {
                __builtin_amdgcn_global_load_lds((gptr_t)(g0 + sl * WSLOTS), (lptr_t)(d0 + sl * WSLOTS), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(g0 + sl * WSLOTS), (lptr_t)(d0 + sl * WSLOTS), 16, 1024, 0);
            }
    };
    int nsl_cur = 0;
    {
        const v4f* p0 = wstage(kc0, 0, 0, nsl_cur);
        dma_stage(p0, nsl_cur, 0);
    }
    load_act(kc0);
    if (have_coef) {
        const int nq = Cin >> 2;
        const size_t plane = (size_t)a.B * Cin;
        for (int i = tid; i < 3 * nq; i += THREADS) {
            const int pl = i / nq, cq = i - pl * nq;
            ldsC[i] = *reinterpret_cast<const v4f*>(a.coef + pl * plane + (size_t)b * Cin + 4 * cq);
        }
    }
#ifndef CDDPM_TRANSFORM_AFTER_BARRIER
    if (have_coef) __syncthreads();     // the coefficient cache is read by transform_act in FRONT of the first chunk barrier
#endif
    int buf = 0;
#ifdef CDDPM_STAGGER_SLEEP
    // A/B switch (tools/conv_ab.py): the two waves of a SIMD (w and w + 4) run the same program between the same barriers;
    // waves 4-7 start every stage 64 * CDDPM_STAGGER_SLEEP cycles late so that their fragment reads fall behind their partner's
    // (MI355X_MICROARCH.md, Two waves per SIMD, item 9). Results unchanged.
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= 4;
#define STAGGER() { if (late) __builtin_amdgcn_s_sleep(CDDPM_STAGGER_SLEEP); }
#else
#define STAGGER()
#endif
    STAMP(0)
    for (int chunk = kc0; chunk < kc1; ++chunk) {
        const bool main_seg = chunk < nch_main;
        const int nst = main_seg ? TAPS / TPS : 1;       // stages of this chunk
        FSTAMP(chunk == kc0 ? 7 : 6)
        if (chunk == kc0) { ESTAMP(0) }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this chunk's first weight stage (and its patch registers) have landed
        FSTAMP(0)
#ifndef CDDPM_TRANSFORM_AFTER_BARRIER
        transform_act(chunk);   // registers only: in front of the barrier, where the early waves would otherwise wait
        FSTAMP(2)
        LOOP_BARRIER();   // the stage has landed in every wave, and every wave is done reading the previous patch
        FSTAMP(1)
        write_act();
#else
        LOOP_BARRIER();   // ... in every wave, and every wave is done reading the previous patch
        FSTAMP(1)
        store_act(chunk);
#endif
        FSTAMP(2)
        LOOP_BARRIER();
        FSTAMP(3)
        STAGGER();
        STAMP(1)
        if constexpr (NB == 1) {
        for (int st = 0; st < nst; ++st) {
            const int ntaps = nsl_cur;                    // taps of this stage
            const bool last_st = (st == nst - 1);
            int nsl_next = 0;
            // (patch loads first: with a DMA in flight the compiler drains vmcnt to 0 in front of ordinary loads)
            if (last_st && chunk + 1 < kc1) load_act(chunk + 1);
            if (!(last_st && chunk + 1 >= kc1)) {         // (nothing may be in flight when the epilogue reuses the buffers)
                const v4f* pn = last_st ? wstage(chunk + 1, 0, 0, nsl_next) : wstage(chunk, 0, st + 1, nsl_next);
                dma_stage(pn, nsl_next, buf ^ 1);
            }
            STAMP(2)
#pragma unroll
            for (int tt = 0; tt < TPS; ++tt)
                if (tt < ntaps) {
                    const int t = st * TPS + tt;                                  // tap index inside the chunk
                    compute(std::integral_constant<int, 0>{}, main_seg ? t : (TAPS / 2), tt, !main_seg, ldsW + buf * WSTAGE + tt * WSLOTS, (t % FOLD) == 0);   // skip segment: centre tap
                }
            buf ^= 1;
            nsl_cur = nsl_next;
            STAMP(3)
            if (((st + 1) * TPS) % FOLD == 0 || last_st) {
                fold_acc();
                STAMP(4)
            }
            if (!last_st) {
                FSTAMP(6)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                FSTAMP(4)
                LOOP_BARRIER();   // the next stage has landed in every wave; this stage's buffer is free
                FSTAMP(5)
                STAGGER();
            }
        }
        } else {
        // the chunk's stages: NB passes over N (cout block hb of this workgroup), nst weight stages each; the patch is shared
        auto run_block = [&](auto hc) {
            constexpr int HB = decltype(hc)::value;
            for (int st = 0; st < nst; ++st) {
                const int ntaps = nsl_cur;                    // taps of this stage
                const bool last_st = (st == nst - 1) && (HB == NB - 1);      // last stage of the chunk
                int nsl_next = 0;
                // (patch loads first: with a DMA in flight the compiler drains vmcnt to 0 in front of ordinary loads)
                if (last_st && chunk + 1 < kc1) load_act(chunk + 1);
                if (!(last_st && chunk + 1 >= kc1)) {         // (nothing may be in flight when the epilogue reuses the buffers)
                    const v4f* pn = last_st ? wstage(chunk + 1, 0, 0, nsl_next)
                                            : (st == nst - 1 ? wstage(chunk, HB + 1, 0, nsl_next) : wstage(chunk, HB, st + 1, nsl_next));
                    dma_stage(pn, nsl_next, buf ^ 1);
                }
                STAMP(2)
#pragma unroll
                for (int tt = 0; tt < TPS; ++tt)
                    if (tt < ntaps) {
                        const int t = st * TPS + tt;                                  // tap index inside the chunk
                        compute(hc, main_seg ? t : (TAPS / 2), tt, !main_seg, ldsW + buf * WSTAGE + tt * WSLOTS,
                                (NB == 1) && (t % FOLD) == 0);   // skip segment: centre tap; NB = 2: one chain from the zeroed registers
                    }
                buf ^= 1;
                nsl_cur = nsl_next;
                STAMP(3)
                if (((st + 1) * TPS) % FOLD == 0 || st == nst - 1) {
                    fold_acc();
                    STAMP(4)
                }
                if (!last_st) {
                    FSTAMP(6)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    FSTAMP(4)
                    LOOP_BARRIER();   // the next stage has landed in every wave; this stage's buffer is free
                    FSTAMP(5)
                    STAGGER();
                }
            }
        };
        run_block(std::integral_constant<int, 0>{});
        if constexpr (NB == 2) run_block(std::integral_constant<int, 1>{});
        }
    }

    } else {
    // ---- main loop: weight stages (TPS taps of one 32-channel chunk) double-buffered in LDS and prefetched through
    //      registers one stage ahead; the next chunk's patch is fetched into registers behind the last stage's MFMAs.
    // stage pointer / slab count; past the end it wraps to stage 0 so that the prefetch stays unconditional
    auto wstage = [&](int chunk, int st, int& nsl) -> const v4f* {
        if (chunk >= nch) { chunk = 0; st = 0; }
        if (chunk < nch_main) { nsl = TPS; return wmain + ((size_t)chunk * TAPS + st * TPS) * WSLOTS; }
        nsl = 1;
        return wskip + (size_t)(chunk - nch_main) * WSLOTS;
    };
    auto get_stage = [&](const v4f* p, int nsl) {        // global -> registers (nsl slabs of this stage)
#pragma unroll
        for (int sl = 0; sl < TPS; ++sl)
            if (TPS == 1 || sl < nsl) {
#pragma unroll
                for (int i = 0; i < WK; ++i) wreg[sl * WK + i] = p[sl * WSLOTS + tid + THREADS * i];
            }
    };
    auto put_stage = [&](int buf, int nsl) {             // registers -> LDS
#pragma unroll
        for (int sl = 0; sl < TPS; ++sl)
            if (TPS == 1 || sl < nsl) {
#pragma unroll
                for (int i = 0; i < WK; ++i) ldsW[buf * WSTAGE + sl * WSLOTS + tid + THREADS * i] = wreg[sl * WK + i];
            }
    };
    int nsl_cur = 0;
    {
        const v4f* p0 = wstage(0, 0, nsl_cur);
        get_stage(p0, nsl_cur);
    }
    load_act(0);
    if (have_coef) {
        const int nq = Cin >> 2;
        const size_t plane = (size_t)a.B * Cin;
        for (int i = tid; i < 3 * nq; i += THREADS) {
            const int pl = i / nq, cq = i - pl * nq;
            ldsC[i] = *reinterpret_cast<const v4f*>(a.coef + pl * plane + (size_t)b * Cin + 4 * cq);
        }
    }
    int buf = 0;
    STAMP(0)
    for (int chunk = 0; chunk < nch; ++chunk) {
        const bool main_seg = chunk < nch_main;
        const int nst = main_seg ? TAPS / TPS : 1;       // stages of this chunk
        __syncthreads();   // every wave is done reading the previous patch
        store_act(chunk);
        STAMP(1)
        for (int st = 0; st < nst; ++st) {
            put_stage(buf, nsl_cur);
            const int ntaps = nsl_cur;                    // taps of this stage
            const bool last_st = (st == nst - 1);
            int nsl_next = 0;
            const v4f* pn = last_st ? wstage(chunk + 1, 0, nsl_next) : wstage(chunk, st + 1, nsl_next);
            get_stage(pn, nsl_next);
            nsl_cur = nsl_next;
            if (last_st && chunk + 1 < nch) load_act(chunk + 1);
            __syncthreads();
            STAMP(2)
            // accumulation in three levels: an MFMA sums 16 products, `acc` collects FOLD taps of a 32-channel chunk
            // (<= 96 products per chain), `tot` sums those groups. The rounding noise of an fp32 chain grows with the
            // magnitude of its partial sums, so short chains folded into a long-lived total keep it near the
            // storage-rounding level (tools/ubench/bf16_split_accuracy.hip, tools/chain_noise.py).
#pragma unroll
            for (int tt = 0; tt < TPS; ++tt)
                if (TPS == 1 || tt < ntaps) {
                    const int t = st * TPS + tt;                                  // tap index inside the chunk
                    compute(std::integral_constant<int, 0>{}, main_seg ? t : (TAPS / 2), tt, !main_seg, ldsW + buf * WSTAGE + tt * WSLOTS, (t % FOLD) == 0);   // skip segment: centre tap
                }
            buf ^= 1;
            STAMP(3)
            if (((st + 1) * TPS) % FOLD == 0 || last_st) {
                fold_acc();
                STAMP(4)
            }
        }
    }

    }

    // residual tile of this wave (16 x 16 B per lane): requested here, in one go, so that the loads fly while the waves
    // meet at the barrier and transpose; the accumulator / fragment registers are dead by now. (Loading each batch right
    // before its add exposed the global latency four times per wave: 12 % of the kernel on the +residual layers.)
    ESTAMP(1)
    // the epilogue handles 2 NB quarter-tiles q = 2 hb + nt (cout block hb of the workgroup, 32-cout half nt of the wave's 64): the
    // residual of quarter q sits in buffer q & 1, requested two quarters ahead (q = 0, 1 here, q + 2 when quarter q's registers are free)
    v4f rsd_all[2][2][4];
    auto load_rsd = [&](int q) {
        const int cq = lane & 7, prow = lane >> 3;
        const int hb = q >> 1, nt = q & 1;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = (NB * cb + hb) * 128 + 64 * wn + 32 * nt + 4 * cq;
                const int p = 8 * (4 * hf + i) + prow;
                const int gy = y0 + 2 * wm + (p >> 5), gx = x0 + (p & 31);
                const int y = UP2 ? (2 * gy + pa) : gy, x = UP2 ? (2 * gx + pb) : gx;
                v4f r = v4f{0.f, 0.f, 0.f, 0.f};
                if (a.res && (gy < gridH) && (gx < gridW)) {
                    const size_t rp = a.res_up ? ((size_t)(b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1))
                                               : ((size_t)(b * a.H + y) * a.W + x);
                    r = *reinterpret_cast<const v4f*>(a.res + rp * a.Cout + co);
                }
                rsd_all[q & 1][hf][i] = r;
            }
    };
    load_rsd(0);
    load_rsd(1);
    ESTAMP(2)
    __syncthreads();   // every wave is done with the patch / weight buffers before they become transpose space
    ESTAMP(3)
    // ---- epilogue: as conv_mfma.hip -- each wave transposes its 64 x 64 tile through a private 8-KB LDS region so
    //      that every lane moves 16 B; bias, residual and the GroupNorm statistics of the output are applied here.
    {
#ifndef CDDPM_TRS16
#define CDDPM_TRS16 36
#endif
        constexpr int TRS = M16 ? CDDPM_TRS16 : 32;                            // row stride of the transpose region (36: conflict-free for the 16 x 16 C layout)
        float* tr = reinterpret_cast<float*>(lds) + wave * (64 * TRS);   // [64 pixels][32 channels]
        const int cq = lane & 7;
        const int prow = lane >> 3;
        const int tilesY4 = (gridH + 3) >> 2;                         // statistics records are per 4-row band (kernels.h)
        const int ty4 = (y0 >> 2) + (wm >> 1);
#pragma unroll
        for (int q = 0; q < 2 * NB; ++q) {
            const int hb = q >> 1, nt = q & 1;
            const int co = (NB * cb + hb) * 128 + 64 * wn + 32 * nt + 4 * cq;
            if constexpr (M16) {
                // C tile layout of the 16 x 16 form: register r of lane l = row 4 (l >> 4) + r (pixel), column l & 15 (cout)
#pragma unroll
                for (int t16 = 0; t16 < 4; ++t16)
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            tr[(16 * t16 + 4 * (lane >> 4) + r) * TRS + 16 * n2 + (lane & 15)] =
                                (NB == 2) ? acc16[hb][t16][2 * nt + n2][r] : tot16[t16][2 * nt + n2][r];
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        tr[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = tot[mt][nt][r];
            }
            __builtin_amdgcn_wave_barrier();
            if (nt == 0) { ESTAMP(4) } else { ESTAMP(6) }
            const v4f bias = a.bias ? *reinterpret_cast<const v4f*>(a.bias + co) : v4f{0.f, 0.f, 0.f, 0.f};
            const float wsc = (NS == 2) ? a.wscale_inv : 1.0f;      // fp16 weights were pre-scaled by a power of two
            v4f ssum = v4f{0.f, 0.f, 0.f, 0.f}, ssq = ssum;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                v4f val[4];
                size_t oidx[4];
                bool ok[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = 8 * (4 * hf + i) + prow;
                    const int gy = y0 + 2 * wm + (p >> 5), gx = x0 + (p & 31);
                    ok[i] = (gy < gridH) && (gx < gridW);
                    const int y = UP2 ? (2 * gy + pa) : gy, x = UP2 ? (2 * gx + pb) : gx;
                    oidx[i] = ((size_t)((ks * a.B + b) * a.H + y) * a.W + x) * a.Cout + co;
                    val[i] = *reinterpret_cast<const v4f*>(tr + p * TRS + 4 * cq);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (ok[i]) {
                        const v4f o = val[i] * wsc + bias + rsd_all[q & 1][hf][i];
                        *reinterpret_cast<v4f*>(a.out + oidx[i]) = o;
                        ssum += o;
                        ssq += o * o;
                    }
            }
            if (q + 2 < 2 * NB) load_rsd(q + 2);       // this quarter's residual registers are free: request the one after next
            if (a.stats) {
#pragma unroll
                for (int m = 8; m < 64; m <<= 1) {
                    ssum.x += __shfl_xor(ssum.x, m, 64); ssum.y += __shfl_xor(ssum.y, m, 64);
                    ssum.z += __shfl_xor(ssum.z, m, 64); ssum.w += __shfl_xor(ssum.w, m, 64);
                    ssq.x += __shfl_xor(ssq.x, m, 64); ssq.y += __shfl_xor(ssq.y, m, 64);
                    ssq.z += __shfl_xor(ssq.z, m, 64); ssq.w += __shfl_xor(ssq.w, m, 64);
                }
                if (prow == 0 && ty4 < tilesY4) {
                    const int nrec = (UP2 ? 8 : 2) * tilesX * tilesY4;
                    const int rec = 2 * ((cls * tilesY4 + ty4) * tilesX + tx) + (wm & 1);
                    float* o = a.stats + (((size_t)b * nrec + rec) * a.Cout + co) * 2;
                    *reinterpret_cast<v4f*>(o) = v4f{ssum.x, ssq.x, ssum.y, ssq.y};
                    *reinterpret_cast<v4f*>(o + 4) = v4f{ssum.z, ssq.z, ssum.w, ssq.w};
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (nt == 0) { ESTAMP(5) } else { ESTAMP(7) }
        }
    }
#if defined(CDDPM_STAMPS) || defined(CDDPM_STAMPS_FINE) || defined(CDDPM_STAMPS_EPI)
    STAMP(5)
    FSTAMP(7)
    if (a.stamps && lane == 0 && wave < 4) {
        for (int i = 0; i < 8; ++i) atomicAdd(&a.stamps[wave * 8 + i], st_[i]);
        if (wave == 0) {
            atomicAdd(&a.stamps[40], __builtin_amdgcn_s_memtime() - t0c_);
            atomicAdd(&a.stamps[41], __builtin_amdgcn_s_memrealtime() - t0r_);
        }
    }
#endif
}

int conv_mode() {
    static const int mode = [] {
        const char* e = getenv("CDDPM_CONV");
        if (e && strcmp(e, "f32") == 0) return 0;
        if (e && strcmp(e, "x6") == 0) return 1;
        return 2;                                   // "h3" / unset: fp16 two-term split, three products
    }();
    return mode;
}

template <int NS, int ROWS>
static void launch_split(const ConvArgs& a, hipStream_t stream) {
    const bool up2 = (a.taps == 4);
    const int gh = up2 ? a.H / 2 : a.H, gw = up2 ? a.W / 2 : a.W;
    const int tilesX = (gw + 31) / 32, tilesY = (gh + ROWS - 1) / ROWS;
    // a.nb2 (set by the caller's plan: conv_nb2_ok): 256 couts per workgroup, the chunk's patch produced once for both cout blocks
    static const int m16_env = [] { const char* e = getenv("CDDPM_M16"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    const bool m16 = (NS == 2) && (m16_env != 0);
    const bool nb2 = (NS == 2) && m16 && a.nb2 && a.ksplit <= 1 && (a.Cout % 256) == 0 && a.taps != 1;
    const unsigned grid = (unsigned)(a.B * (up2 ? 4 : 1) * tilesX * tilesY * (a.Cout / (nb2 ? 256 : 128)) * (a.ksplit > 1 ? a.ksplit : 1));
    const size_t coef_lds = a.coef ? (size_t)3 * (a.C0 + a.C1) * sizeof(float) : 0;
    auto need = [&](int npix) {
        const int tps = (NS == 2) ? (a.taps == 9 ? 3 : (a.taps == 4 ? 2 : 1)) : 1;     // taps per weight stage (kernel: TPS)
        const size_t main = (size_t)(npix + 2 * tps * 128) * (4 * NS) * 16 + coef_lds;
        const size_t tr = (size_t)ROWS * 64 * 36 * sizeof(float);   // epilogue transpose regions alias the buffers (stride <= 36)
        return main > tr ? main : tr;
    };
    // MFMA shape of the fp16 form: 16x16x32 (CDDPM_M16=0 forces 32x32x16). The chip holds a higher clock on it
    // (MI355X_MICROARCH.md, DVFS give-back item 7): 3-9 % faster on every layer shape, at the same cycle count.
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<9, ROWS, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<9, ROWS, NS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<1, ROWS, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<1, ROWS, NS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<4, ROWS, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<4, ROWS, NS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if constexpr (NS == 2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<9, ROWS, NS, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<1, ROWS, NS, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<4, ROWS, NS, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<9, ROWS, NS, true, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<4, ROWS, NS, true, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<9, ROWS, NS, true, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<4, ROWS, NS, true, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        attr = true;
    }
    const dim3 g(grid), blk(64 * ROWS);
    if constexpr (NS == 2) {
        if (nb2) {
            if (a.hi_only) {       // plain fp16 operands (the training operators under precision 16)
                if (a.taps == 9) hipLaunchKernelGGL((conv_split_kernel<9, ROWS, NS, true, true, 2>), g, blk, need((ROWS + 2) * 34), stream, a);
                else hipLaunchKernelGGL((conv_split_kernel<4, ROWS, NS, true, true, 2>), g, blk, need((ROWS + 1) * 33), stream, a);
            } else if (a.taps == 9) hipLaunchKernelGGL((conv_split_kernel<9, ROWS, NS, true, false, 2>), g, blk, need((ROWS + 2) * 34), stream, a);
            else hipLaunchKernelGGL((conv_split_kernel<4, ROWS, NS, true, false, 2>), g, blk, need((ROWS + 1) * 33), stream, a);
            return;
        }
    }
    if (a.taps == 9) {
        if (m16 && a.hi_only) hipLaunchKernelGGL((conv_split_kernel<9, ROWS, NS, true, true>), g, blk, need((ROWS + 2) * 34), stream, a);
        else if (m16) hipLaunchKernelGGL((conv_split_kernel<9, ROWS, NS, true>), g, blk, need((ROWS + 2) * 34), stream, a);
        else     hipLaunchKernelGGL((conv_split_kernel<9, ROWS, NS>), g, blk, need((ROWS + 2) * 34), stream, a);
    } else if (a.taps == 1) {
        if (m16 && a.hi_only) hipLaunchKernelGGL((conv_split_kernel<1, ROWS, NS, true, true>), g, blk, need(ROWS * 32), stream, a);
        else if (m16) hipLaunchKernelGGL((conv_split_kernel<1, ROWS, NS, true>), g, blk, need(ROWS * 32), stream, a);
        else     hipLaunchKernelGGL((conv_split_kernel<1, ROWS, NS>), g, blk, need(ROWS * 32), stream, a);
    } else {
        if (m16 && a.hi_only) hipLaunchKernelGGL((conv_split_kernel<4, ROWS, NS, true, true>), g, blk, need((ROWS + 1) * 33), stream, a);
        else if (m16) hipLaunchKernelGGL((conv_split_kernel<4, ROWS, NS, true>), g, blk, need((ROWS + 1) * 33), stream, a);
        else     hipLaunchKernelGGL((conv_split_kernel<4, ROWS, NS>), g, blk, need((ROWS + 1) * 33), stream, a);
    }
}

// May a launch use 256-cout workgroups (ConvArgs::nb2)? Feasibility only -- WHEN it is used is the caller's plan (cddpm_api.hip: the
// reverse loop uses it for the steps t >= the handle's switch step, see cddpm_set_accumulation_switch): the fp16 16 x 16 family, unsplit
// K, Cout a multiple of 256, 3x3 / folded 2x2 taps, and `workgroups128` -- the workgroup count of the 128-cout form at the geometry the
// CALLER plans for (the handle's maximum geometry on the reconstruction path, so that a slice's bits do not depend on the batch it is
// computed in) -- still fills the chip once halved. Environment CDDPM_NB2: "0" = never; "force" = wherever the kernel can, any geometry,
// any step and in the operator entry points (how the parity tests run their small shapes through it).
int conv_nb2_env() {
    static const int v = [] { const char* e = getenv("CDDPM_NB2"); return (e && e[0] == '0') ? 0 : (e && !strcmp(e, "force")) ? 2 : 1; }();
    return v;
}
bool conv_nb2_ok(int Cout, long long workgroups128, int ksplit, int hi_only) {
    static const bool m16 = [] { const char* e = getenv("CDDPM_M16"); return !(e && e[0] == '0'); }();
    const int on = conv_nb2_env();
    (void)hi_only;       // both operand forms have the 256-cout instantiation
    return on && m16 && conv_mode() == 2 && ksplit <= 1 && (Cout % 256) == 0 && (on == 2 || workgroups128 >= 512);
}

void launch_conv_split(const ConvArgs& a, hipStream_t stream) {
    if (conv_mode() == 1) launch_split<3, 8>(a, stream);
    else launch_split<2, 8>(a, stream);
}

// ---- host side: 16-bit round-to-nearest-even conversions and the splits
static inline uint16_t bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);     // inf / nan: truncate (weights are finite)
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f16_rne(float x) { const _Float16 h = (_Float16)x; uint16_t b; memcpy(&b, &h, 2); return b; }   // compiler RNE, subnormals kept
static inline float f16_to_f(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }

// power-of-two pre-scale of a weight tensor for the fp16 split: the largest e in [0, 24] with max|w| * 2^e < 2^14
int conv_weight_exp(const float* w, size_t n) {
    if (conv_mode() != 2) return 0;
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) { const float v = w[i] < 0 ? -w[i] : w[i]; if (v > mx) mx = v; }
    int e = 24;
    while (e > 0 && ldexpf(mx, e) >= 16384.0f) --e;
    return e;
}

// w: PyTorch [Cout][Cin][k][k] (taps = k*k) -> [Cout/128][Cin/32][taps][128 rows][4 NS slots][8 x 16 bit]
//   NS = 3 (bf16): slot (split s, u = channel/8 in the chunk) of row j at 4 s + (u ^ ((j>>2)&3)); 6 bytes per weight
//   NS = 2 (fp16): w * 2^wexp is split; slot at (4 s + u) ^ ((j>>1)&7); 4 bytes per weight
void pack_conv_weights_split(const float* w, int Cout, int Cin, int taps, void* dst_, int wexp) {
    uint16_t* dst = static_cast<uint16_t*>(dst_);
    const int ncb = Cout / 128, nch = Cin / 32;
    const int ns = (conv_mode() == 1) ? 3 : 2, sp = 4 * ns;
    for (int cb = 0; cb < ncb; ++cb)
        for (int ch = 0; ch < nch; ++ch)
            for (int t = 0; t < taps; ++t) {
                uint16_t* img = dst + (((size_t)cb * nch + ch) * taps + t) * (128 * sp * 8);
                for (int j = 0; j < 128; ++j)
                    for (int u = 0; u < 4; ++u)
                        for (int e = 0; e < 8; ++e) {
                            const int co = cb * 128 + j, ci = ch * 32 + 8 * u + e;
                            float r = w[((size_t)co * Cin + ci) * taps + t];
                            if (ns == 3) {
                                for (int s3 = 0; s3 < 3; ++s3) {
                                    const uint16_t q = bf16_rne(r);
                                    r -= bf16_to_f(q);
                                    img[(size_t)(j * 12 + 4 * s3 + (u ^ ((j >> 2) & 3))) * 8 + e] = q;
                                }
                            } else {
                                r = ldexpf(r, wexp);
                                for (int s2 = 0; s2 < 2; ++s2) {
                                    const uint16_t q = f16_rne(r);
                                    r -= f16_to_f(q);
                                    img[(size_t)(j * 8 + ((4 * s2 + u) ^ ((j >> 1) & 7))) * 8 + e] = q;
                                }
                            }
                        }
            }
}

// ---- device twin of pack_conv_weights_split for the fp16 two-term family: the training step re-packs the updated weights each step
// (same image, bit for bit, as the host packer for the same exponent). Element (o, i, t) of the packed tensor [O][I][taps]:
//   mode 0 (forward)          w[(o I + i) taps + t]                      w = [O][I][k][k]
//   mode 1 (input gradient)   w[(i O + o) taps + taps - 1 - t]           w = [I][O][k][k]: transposed and flipped
//   mode 2 (folded upsample)  the class sums of pack_conv_weights_up2 (summed in double, rounded once); w = [O][I][3][3], taps = 4,
//                             blockIdx.y = class
__device__ __forceinline__ void pack_split_unit(const float* __restrict__ w, int O, int I, int taps, int mode, int wexp,
                                                uint16_t* __restrict__ dst, long long id, int cls_in) {
    const int ncb = O / 128, nch = I / 32;
    if (id >= (long long)ncb * nch * taps * 512) return;
    const int u = (int)(id & 3), j = (int)((id >> 2) & 127);
    long long rest = id >> 9;
    const int t = (int)(rest % taps); rest /= taps;
    const int ch = (int)(rest % nch);
    const int cb = (int)(rest / nch);
    const int cls = cls_in;
    uint16_t* img = dst + ((((size_t)cls * ncb + cb) * nch + ch) * taps + t) * (size_t)(128 * 8 * 8);
    const int o = cb * 128 + j;
    union { uint16_t q[8]; uint4 v; } hi, mid;
    for (int e = 0; e < 8; ++e) {
        const int i = ch * 32 + 8 * u + e;
        float r;
        if (mode == 0) r = w[((size_t)o * I + i) * taps + t];
        else if (mode == 1) r = w[((size_t)i * O + o) * taps + (taps - 1 - t)];
        else {
            const int pa = cls >> 1, pb = cls & 1, ty = t >> 1, tx = t & 1;
            const int ky0 = pa == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = pa == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
            const int kx0 = pb == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = pb == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
            double acc = 0.0;
            for (int ky = ky0; ky <= ky1; ++ky)
                for (int kx = kx0; kx <= kx1; ++kx) acc += (double)w[((size_t)o * I + i) * 9 + ky * 3 + kx];
            r = (float)acc;
        }
        r = ldexpf(r, wexp);
        const _Float16 h = (_Float16)r;
        r -= (float)h;
        const _Float16 m = (_Float16)r;
        hi.q[e] = __builtin_bit_cast(uint16_t, h);
        mid.q[e] = __builtin_bit_cast(uint16_t, m);
    }
    const int sw = (j >> 1) & 7;
    *reinterpret_cast<uint4*>(img + (size_t)(j * 8 + ((0 + u) ^ sw)) * 8) = hi.v;
    *reinterpret_cast<uint4*>(img + (size_t)(j * 8 + ((4 + u) ^ sw)) * 8) = mid.v;
}
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ w, int O, int I, int taps, int mode, int wexp,
                                                         uint16_t* __restrict__ dst) {
    pack_split_unit(w, O, I, taps, mode, wexp, dst, (long long)blockIdx.x * 256 + threadIdx.x, (int)blockIdx.y);
}
// every image of a network in ONE launch: blockIdx.y walks a device table of jobs (a folded-upsample image is four jobs, one per class),
// blockIdx.x the job's 16-byte units; a job shorter than the grid's x extent leaves its extra workgroups idle
__global__ __launch_bounds__(256) void pack_split_batch_kernel(const PackJob* __restrict__ jobs) {
    const PackJob j = jobs[blockIdx.y];
    pack_split_unit(j.w, j.O, j.I, j.taps, j.mode, j.wexp, static_cast<uint16_t*>(j.dst), (long long)blockIdx.x * 256 + threadIdx.x, j.cls);
}
void launch_pack_conv_split_batch(const PackJob* jobs_dev, int njobs, long long max_units, hipStream_t stream) {
    hipLaunchKernelGGL(pack_split_batch_kernel, dim3((unsigned)((max_units + 255) / 256), (unsigned)njobs), dim3(256), 0, stream, jobs_dev);
}
void launch_pack_conv_split(const float* w, int O, int I, int taps, int mode, int wexp, void* dst, hipStream_t stream) {
    const long long n = (long long)(O / 128) * (I / 32) * taps * 512;
    hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)((n + 255) / 256), mode == 2 ? 4 : 1), dim3(256), 0, stream, w, O, I, taps, mode, wexp,
                       static_cast<uint16_t*>(dst));
}

// max |x| over n floats into out[0] (out zeroed by the caller's memset; |x| compared as its bit pattern)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
void launch_absmax(const float* x, long long n, float* out, hipStream_t stream) {
    (void)hipMemsetAsync(out, 0, sizeof(float), stream);
    const unsigned g = (unsigned)std::min<long long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(absmax_kernel, dim3(g), dim3(256), 0, stream, x, n, reinterpret_cast<unsigned*>(out));
}

}  // namespace cddpm
