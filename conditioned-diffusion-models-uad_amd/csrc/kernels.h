// Internal launcher interface of libcddpm_hip.so (gfx950 only). Each launcher enqueues on `stream`
// and returns; shape preconditions are checked by the callers in cddpm_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cddpm {

// ------------------------------------------------------------------------------------------------
// Fused implicit-GEMM convolution (conv_mfma.hip). NHWC activations, M = pixels, N = output channels,
// K = taps x input channels, v_mfma_f32_32x32x2_f32 (exact fp32).
// ------------------------------------------------------------------------------------------------
#define CDDPM_MAX_KSPLIT 8
struct ConvArgs {
    const float* src0;   // NHWC [B, srcH, srcW, C0]
    const float* src1;   // NHWC [B, srcH, srcW, C1] or nullptr: channels C0.. of the concatenation
    int C0, C1;          // Cin = C0 + C1, both multiples of 32
    int srcH, srcW;      // source extent (H/2, W/2 when upsample)
    int upsample;        // 1: nearest x2, source pixel (y>>1, x>>1)
    const float* coef;   // [3][B][Cin] (mean, a, d) or nullptr: v -> (v - mean) * a + d
    int silu;            // 1: v -> v * sigmoid(v) after the affine map
    const float* wpk;    // packed weights, see pack_conv_weights()
    const float* bias;   // [Cout]
    const float* res;    // residual NHWC [B, resH, resW, Cout] or nullptr
    int res_up;          // 1: residual lives at half resolution, read at (y>>1, x>>1)
    float* out;          // NHWC [B, H, W, Cout]
    int B, H, W, Cout;   // Cout multiple of 128
    int taps;            // 9 (3x3, zero pad 1), 1, or 4 = folded 'nearest x2 upsample -> 3x3' (src at half resolution,
                         // weights from pack_conv_weights_up2, upsample flag unused)
    // second K segment (optional): an un-normalised 1x1 convolution accumulated into the same tile,
    // used for ResBlock skip_connection convs: out += conv1x1(skip0|skip1) (bias folded by the caller)
    const float* skip0; const float* skip1; int S0, S1;   // NHWC at the OUTPUT resolution
    const float* skip_wpk;
    unsigned long long* stamps;   // diagnostic builds (-DCDDPM_STAMPS) only: per-wave phase cycle sums, else nullptr
    // optional GroupNorm statistics of the OUTPUT tensor, produced by the epilogue: fp32 records
    // [B][nrec = 2 * tilesX * tilesY][Cout][2] = per-channel (sum, sum of squares) over the 64 pixels a wave owns
    float* stats;
    float wscale_inv;    // conv_mode() == 2: 2^-wexp of the packed weights (main and skip segment share it); else unused
    // split-K (fp16-split family only; 0 / 1 = off): the K loop (32-channel chunks of the main segment, then of the skip segment)
    // is cut into `ksplit` consecutive ranges kbound[j] .. kbound[j + 1]; workgroup (tile, j) stores its scaled raw sums to plane j
    // of `out` ([ksplit][B][H][W][Cout]; the caller passes no bias / residual / stats) and launch_conv_reduce combines the planes.
    int ksplit;
    short kbound[CDDPM_MAX_KSPLIT + 1];
    // fp16-split family, 16 x 16 form only: 1 = multiply the hi terms only (plain fp16 operands, fp32 accumulation -- the arithmetic of the
    // reference trainer's `precision: 16`; a third of the MFMAs). Set by the training operators under CDDPM_TRAIN_PRECISION=16, never by
    // the reconstruction path.
    int hi_only;
    // fp16-split family, 16 x 16 form, unsplit K: 1 = workgroups of 256 pixels x 256 couts (two cout blocks per workgroup sharing the
    // chunk's transformed patch; two-level accumulation, see conv_x6.hip). Decided by the caller with conv_nb2_ok.
    int nb2;
};
bool conv_nb2_ok(int Cout, long long workgroups128, int ksplit, int hi_only);
int conv_nb2_env();      // CDDPM_NB2: 0 never, 1 by plan (default), 2 forced
// out[b][p][c] = ((plane 0 + plane 1) + ...) + bias[c] + residual, in this fixed order; optional GroupNorm statistics records of
// `out`: one record per 64 consecutive pixels, [B][ceil(HW / 64)][Cout][2]
void launch_conv_reduce(const float* planes, int ksplit, const float* bias, const float* res, int res_up, float* out, float* stats,
                        int B, int H, int W, int Cout, hipStream_t stream);
inline int conv_reduce_stat_records(int H, int W) { return (H * W + 63) / 64; }
#define CDDPM_STAT_FLOATS 2      // floats per (record, channel): (sum, sum of squares)
inline int conv_stat_records(int H, int W) { return 2 * ((W + 31) / 32) * ((H + 3) / 4); }
inline int conv_stat_records_up2(int H, int W) { return 8 * ((W / 2 + 31) / 32) * ((H / 2 + 3) / 4); }
void launch_conv(const ConvArgs& a, hipStream_t stream);
// fp32-accurate variants on the 16-bit matrix pipe (conv_x6.hip): operands split into 16-bit terms whose partial
// products are exact in fp32. conv_mode(): 2 = fp16 two-term split, three MFMAs per product group (default,
// CDDPM_CONV=h3 or unset); 1 = bf16 three-term split, six MFMAs (CDDPM_CONV=x6); 0 = the fp32-MFMA kernels of
// conv_mfma.hip (CDDPM_CONV=f32). Chosen once per process; it also selects the packed weight format.
int conv_mode();
void launch_conv_split(const ConvArgs& a, hipStream_t stream);
// mode 2 only (else 0): power-of-two pre-scale exponent of a weight tensor, max|w| * 2^e in [2^13, 2^14)
int conv_weight_exp(const float* w, size_t n);
void pack_conv_weights_split(const float* w /*[Cout][Cin][k][k]*/, int Cout, int Cin, int taps, void* dst, int wexp);

// packed weight image sizes / packing (host side, cddpm_api.hip)
// fp32 layout: [Cout/128][Cin/32][taps][128 rows x 8 slots of float4], slot s of row j stored at s ^ ((j>>1)&7)
// x6 layout  : [Cout/128][Cin/32][taps][128 rows x 12 slots of 8 bf16] (conv_x6.hip), 1.5 floats per weight
// split layouts (conv_x6.hip): [..][128 rows x 4 NS slots of 8 x 16 bit]: 1.5 (bf16 x 3) | 1 (fp16 x 2) floats per weight
size_t packed_conv_floats(int Cout, int Cin, int taps);
// wexp: conv_weight_exp() of the tensor (and of every tensor accumulated into the same output tile); ignored unless mode 2
// device packer of the fp16 two-term family (mode 0 forward, 1 transposed + flipped for the input gradient, 2 folded upsample classes) and
// max |x| (the power-of-two pre-scale is chosen from it on the host)
void launch_pack_conv_split(const float* w_dev, int O, int I, int taps, int mode, int wexp, void* dst_dev, hipStream_t stream);
// one job of the batched device packer: the arguments of launch_pack_conv_split + the folded-upsample class (mode 2: four jobs per image).
// Layout shared with cddpm_pack_job of include/cddpm.h (40 bytes).
struct PackJob { const float* w; void* dst; int O, I, taps, mode, wexp, cls; };
void launch_pack_conv_split_batch(const PackJob* jobs_dev, int njobs, long long max_units, hipStream_t stream);
void launch_absmax(const float* x, long long n, float* out, hipStream_t stream);
void pack_conv_weights(const float* w /*[Cout][Cin][k][k]*/, int Cout, int Cin, int taps, float* dst, int wexp);
// folded weights of nearest-x2-upsample + 3x3: 4 parity classes x 4 taps = 4 * packed_conv_floats(Cout, Cin, 4) floats
// returns the pre-scale exponent it chose for the folded weights (0 unless mode 2)
int pack_conv_weights_up2(const float* w /*[Cout][Cin][3][3]*/, int Cout, int Cin, float* dst);

// ------------------------------------------------------------------------------------------------
// GroupNorm(32) statistics and per-(sample, channel) coefficients (norm_kernels.hip)
// ------------------------------------------------------------------------------------------------
// statistics records of a tensor: rec[b][r][C][2] fp32 (sum, sum of squares of a pixel subset per channel).
// Producers: the conv epilogue (r = wave tile) or this sweep (r = pixel-range split, fp64 inside, rounded once).
void launch_gn_partial(const float* src, int C, int B, int HW, int nsplit, float* rec, hipStream_t stream);
int gn_nsplit(int B, int HW);
// coef[0][b][c] = mean_g, coef[1] = rstd*gamma*(1+scale), coef[2] = beta*(1+scale)+shift
// scale/shift = tab[t_b][eoff + c] + cpart[b][eoff + c] (scale) and [.. + C + c] (shift) when tab != nullptr
// rec0 / rec1: records of the (up to two, channel-concatenated) sources with C0 / C1 channels and n0 / n1 records
void launch_gn_finalize(const float* rec0, int C0, int n0, const float* rec1, int C1, int n1, int B, int HW,
                        const float* gamma, const float* beta, const float* tab, const float* cpart, int sumE, int eoff,
                        const int* t_dev, const float* film_direct, float* coef, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// small memory-bound kernels (small_kernels.hip)
// ------------------------------------------------------------------------------------------------
// 3x3 conv with one input channel: x [B,H,W] -> out NHWC [B,H,W,C]; w [C][9], bias [C]
void launch_conv_in1(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int C,
                     hipStream_t stream);
// head: per pixel 9 partial dot products of act(x) with w[tap][C] -> P[B,HW,9]; then gather to out [B,H,W]
void launch_head_dots(const float* x, const float* coef, const float* w9 /*[9][C]*/, float* P, int B, int HW, int C,
                      hipStream_t stream);
void launch_head_gather(const float* P, float bias, const float* bias_ptr /* device, overrides bias */, float* out, int B, int H, int W,
                        hipStream_t stream);
// down ResBlock front end: hp = avgpool2(silu(affine(x))), xp = avgpool2(x); NHWC
void launch_pool_act(const float* x, const float* coef, float* hp, float* xp, int B, int H, int W, int C,
                     hipStream_t stream);
// y[M][N] = act(x)[M][K] . W[N][ldw (cols koff..koff+K)]^T + bias ; silu_in applies SiLU to x first
void launch_linear(const float* x, int ldx, const float* W, int ldw, int koff, const float* bias, float* y, int ldy,
                   int M, int N, int K, int silu_in, hipStream_t stream);
void launch_fill_int(int* p, int n, int v, hipStream_t stream);
void launch_add_int(int* p, int n, int v, hipStream_t stream);
// dst[i] = clamp(src[i], lo, hi): per-sample timesteps handed in by the caller index device tables
void launch_copy_clamp_int(int* dst, const int* src, int n, int lo, int hi, hipStream_t stream);

// posterior step (cond_DDPM.py:391-444): x <- c1[t] * x0hat + c2[t] * x + exp(0.5 logvar[t]) * z (t > 0)
struct StepArgs {
    float* x; const float* model_out; const int* t_dev;
    const float* coef1; const float* coef2; const float* logvar; const float* sqrt_recip; const float* sqrt_recipm1;
    int objective;
    const float* noise;          // explicit z for this step [B,HW] or nullptr -> Philox
    size_t noise_t_stride;       // 0, or (graph replay: `noise` is the base of a [T][B][HW] array) the stride of its t axis
    uint64_t seed, slice0; int t_for_rng;
    int B, HW;
    int finalize;                // 1: also map to [0,1]: (x+1)/2 (cond_DDPM.py:463); -1: exactly when t == 0 (graph replay)
    int clip;                    // clip_denoised: clamp the x0 estimate to [-1,1] (the reference's default, :433)
};
void launch_step(const StepArgs& a, hipStream_t stream);
// DDIM step (cond_DDPM.py:487-511): eps = (sqrt_recip[t] x - x0) / sqrt_recipm1[t] from the UNCLIPPED x0 (pred_x0 objective;
// pred_noise: eps = model output, x0 = sqrt_recip[t] x - sqrt_recipm1[t] eps), x0 clamped to [-1,1], then
// x <- x0 * coef_x0 + coef_eps * eps + sigma * z   (z only when add_noise)
struct DdimArgs {
    float* x; const float* model_out; const int* t_dev; const float* sqrt_recip; const float* sqrt_recipm1;
    int objective;
    float coef_x0, coef_eps, sigma;   // sqrt(alpha_next), sqrt(1 - alpha_next - sigma^2), sigma: computed by the caller in fp32
    int add_noise;                    // time_next > 0
    const float* noise;               // explicit z [B,HW] or nullptr -> Philox keyed by t
    uint64_t seed, slice0;
    int B, HW;
    int finalize;                     // 1: also map to [0,1] (cond_DDPM.py:513)
    int clip;                         // clip_denoised (:467, :493-494)
};
void launch_ddim_step(const DdimArgs& a, hipStream_t stream);
void launch_noise_fill(float* out, uint64_t seed, uint32_t stream_id, int t, uint64_t slice0, int B, int HW,
                       hipStream_t stream);
void launch_q_sample(const float* x01, const float* noise, const int* t_dev, const float* sa, const float* s1ma,
                     float* out, int B, int HW, hipStream_t stream);

// 2-D OpenSimplex fractal noise, fp16 bits, the same field for every batch item (simplex.hip)
void launch_simplex(unsigned short* out, long long seed, int B, int H, int W, int octaves, double persistence,
                    double frequency, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// training pieces (train_kernels.hip): GroupNorm/FiLM/SiLU backward. planes [4][B][C] = mean_g, rstd_g, g', b'; part: fp64 scratch
// [B][nsplit][C][2]; out_bc [4][B][C] scratch (S1, S2, m1, m2)
// ------------------------------------------------------------------------------------------------
void launch_gn_bwd_planes(const float* rec, int nrec, const float* gamma, const float* beta, const float* film, int B, int C, int HW,
                          float* planes, hipStream_t stream);
// conv wgrad (3x3 pad 1, or 1x1): dw [Cout][Cin][k][k] (PyTorch layout), db [Cout] or nullptr; input = cat[x0 (C0), x1 (C1)];
// part: scratch of conv_wgrad_parts() * Cout * Cin * taps floats. Cin multiple of 32 (3x3) / 64 (1x1), C0 of 64, Cout of 64, H of 4
int train_precision();                 // 32 (default) or 16: see train_kernels.hip
int set_train_precision(int bits);    // returns the previous value
int wgrad_mode();      // CDDPM_WGRAD: 2 = h3 (default), 1 = h1, 0 = f32
int conv_wgrad_parts(int B, int H, int W, int Cin, int Cout, int taps);
size_t conv_wgrad_image_units(int B, int H, int W, int Cin, int Cout, int taps);   // 16-byte units of `images` (0: not used by this call)
void launch_conv_wgrad(const float* x0, int C0, const float* x1, int C1, const float* coef, int silu, int up, const float* dy, int B, int H,
                       int W, int Cout, int taps, float* part, int P, void* images, float* dw, float* db, hipStream_t stream);
void launch_bias_grad(const float* dy, long long npix, int C, float* db, double* scratch /* 512 * C doubles */, hipStream_t stream);
// QKVAttention backward: qkv [B][N][3C] (q | k | v), da [B][N][C] -> dqkv [B][N][3C]; p, dp: scratch [B * C / 64][N][N] floats each
// flash-style (default): stats = B * heads * N * 2 floats of scratch; the GEMM form below (CDDPM_ATTN_BWD=gemm) materialises p, dp [B heads][N][N]
void launch_attention_backward_flash(const float* qkv, const float* da, float* dqkv, float* stats, int B, int N, int C, hipStream_t stream);
void launch_attention_backward(const float* qkv, const float* da, float* dqkv, float* p, float* dp, int B, int N, int C,
                               hipStream_t stream);
// backward of y = [SiLU](x) W^T + b: x [M][K], W [N][K], dy [M][N] -> dW [N][K], db [N] (or nullptr), dx [M][K] (or nullptr);
// a_scratch [M][K] when silu_in
size_t linear_backward_scratch_floats(int M, int N, int K, int silu_in);      // floats of a_scratch
void launch_linear_backward(const float* x, const float* W, const float* dy, int M, int N, int K, int silu_in, float* a_scratch,
                            float* dW, float* db, float* dx, hipStream_t stream);
void launch_unpool2(const float* dyp, float* dx, int B, int H, int W, int C, float scale, int accumulate, hipStream_t stream);
void launch_sumpool2(const float* dy, float* dxp, int B, int H, int W, int C, int accumulate, hipStream_t stream);
void launch_add_inplace(float* a, const float* b, long long n, hipStream_t stream);
// dw [C][9] = sum_{b,q} act(T)[b,q,c] * simg[b, q + sign * tap]; part: 256 * C * 9 doubles of scratch
void launch_chan_image_corr(const float* T, const float* coef, int silu, const float* simg, int sign, int B, int H, int W, int C,
                            double* part, float* dw, hipStream_t stream);
void launch_head_dgrad(const float* dout, const float* w9, float* dact, int B, int H, int W, int C, hipStream_t stream);
void launch_loss(const float* out, const float* target, const float* w_b, int l2, int B, int HW, float grad_scale, float* dout, float* loss_b,
                 hipStream_t stream);
void launch_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, int step, float grad_unscale,
                 hipStream_t stream);
void launch_grad_check(const float* g, long long n, int* ctrl, hipStream_t stream);
void launch_guard_commit(int* ctrl, float b1, float b2, hipStream_t stream);
void launch_adam_guarded(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float grad_unscale,
                         const int* ctrl, hipStream_t stream);
void launch_gn_silu_backward(const float* x, const float* x1 /* second source of a concatenated input or nullptr */, int C0, float* dx1,
                             const float* da, const float* planes, const float* gamma, const float* beta,
                             const float* film, int silu, int B, int C, int HW, int nsplit, double* part, float* out_bc, float* dx,
                             float* dgamma, float* dbeta, float* dfilm, const float* add /* dx += add, or nullptr */, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// attention core (attention.hip): qkv NHWC [B,N,3C] -> out [B,N,C], heads of 64 channels
// ------------------------------------------------------------------------------------------------
void launch_attention(const float* qkv, float* out, int B, int N, int C, hipStream_t stream);

// residual-map post-processing (eval_post.hip; src/utils/utils_eval.py:29-33, :447-464)
void launch_residual_mask(const float* orig, const float* recon, const float* mask, float* out, int S, int H, int W,
                          int squared, int iters, hipStream_t stream);
void launch_median3d(const float* in, float* out, int S, int H, int W, int k, hipStream_t stream);

// ---- training-mode kernels of the context encoder (encoder_train.hip)
int enc_conv_split(int B, int H, int W, int Cin, int Cout, int K, int stride, int transposed);     // planes of `part` (1: not used)
void launch_enc_conv(const float* src, const float* w_img, float* dst, int B, int H, int W, int Cin, int Cout, int K, int stride, int transposed,
                     float* part /* enc_conv_split() x rows x produced channels floats, or nullptr */, hipStream_t s);
void launch_enc_pack_w(const float* w, int Cout, int Cin, int taps, float* wf, float* wd, hipStream_t s);
int enc_wgrad_parts(int B, int Ho, int Wo, int Cin, int Cout, int K);
void launch_enc_wgrad(const float* x, const float* dz, float* part, int P, float* dw, int B, int H, int W, int Cin, int Cout, int K, int stride,
                      hipStream_t s);
void launch_enc_stem_fwd(const float* x, const float* w, float* z, int B, int H, int W, hipStream_t s);
void launch_enc_stem_wgrad(const float* x, const float* dz, double* part /* 32 * 49 * 64 */, float* dw, int B, int H, int W, hipStream_t s);
int enc_bn_chunks(long long N);
void launch_enc_bn_forward(const float* z, const float* gamma, const float* beta, const float* sscale, const float* res, int relu, float eps,
                           float momentum, float* run_mean, float* run_var, float* mr, float* y, double* part, long long N, int HW, int C,
                           hipStream_t s);
void launch_enc_bn_backward(const float* z, const float* y, const float* dy, const float* mr, const float* gamma, const float* sscale, int relu,
                            float* dz, float* dres, float* dgamma, float* dbeta, float* k, double* part, long long N, int HW, int C,
                            hipStream_t s);
void launch_enc_maxpool(const float* x, float* y, int B, int H, int W, int C, hipStream_t s);
void launch_enc_maxpool_backward(const float* x, const float* dy, float* dx, int B, int H, int W, int C, hipStream_t s);
void launch_enc_avgpool(const float* x, float* g, int B, int HW, int C, hipStream_t s);
void launch_enc_avgpool_backward(const float* dg, float* dx, int B, int HW, int C, hipStream_t s);

}  // namespace cddpm
