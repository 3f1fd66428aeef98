/*
 * cddpm.h -- C ABI of libcddpm_hip.so: the MI355X (gfx950) implementation of the cDDPM
 * reverse-diffusion reconstruction path.
 *
 * The reference (raymondfdavey/Conditioned-Diffusion-Models-UAD) is pure Python/PyTorch and has no
 * native interface; each entry point below names the reference Python code it replaces
 * (paths relative to the reference root). Plain pointers and sizes only: no torch types.
 *
 * Conventions
 *   - return value: 0 = ok, negative = error; text via cddpm_last_error().
 *   - "dev" pointers are device (HIP) pointers to contiguous fp32; "host" pointers are host memory.
 *   - images are [B,1,H,W] fp32 (single channel, so NCHW == NHWC); H and W multiples of 4.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is
 *     enqueued on it, nothing synchronises with the host inside forward/reverse calls.
 *   - one handle per device; a handle is not thread-safe.
 */
#ifndef CDDPM_H
#define CDDPM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cddpm_ctx* cddpm_handle;

#define CDDPM_MAX_LEVELS 8

/* Shape contract of the denoiser: the constructor arguments of UNetModel
 * (src/models/modules/OpenAI_Unet.py:513-539) as DDPM_2D passes them (src/models/DDPM_2D.py:37-59):
 * resblock_updown=True, use_scale_shift_norm=True, use_new_attention_order=True, dims=2, dropout=0. */
typedef struct cddpm_unet_desc {
    int32_t in_channels;                               /* 1 */
    int32_t out_channels;                              /* 1 */
    int32_t model_channels;                            /* cfg.unet_dim = 128 (multiple of 128) */
    int32_t num_levels;                                /* len(channel_mult) */
    int32_t channel_mult[CDDPM_MAX_LEVELS];            /* cfg.dim_mults = [1,2,2] */
    int32_t num_res_blocks;                            /* 3 */
    int32_t num_attention_resolutions;                 /* len(attention_resolutions) */
    int32_t attention_resolutions[CDDPM_MAX_LEVELS];   /* (3,6,12): never matches ds in {1,2,4} */
    int32_t head_channels;                             /* num_head_channels = 64 */
    int32_t cond_dim;                                  /* num_classes slot = context width 128; 0 = unconditional */
    int32_t timesteps;                                 /* GaussianDiffusion timesteps T */
    int32_t max_batch;                                 /* largest B of any later call */
    int32_t max_h, max_w;                              /* largest H, W of any later call */
} cddpm_unet_desc;

/* objective values for cddpm_set_schedule (src/models/modules/cond_DDPM.py:318) */
#define CDDPM_PRED_X0 0
#define CDDPM_PRED_NOISE 1

/* Replaces UNetModel.__init__ + GaussianDiffusion.__init__ buffer allocation. Allocates packed-weight
 * storage, embedding tables and the activation workspace once; no hipMalloc happens later. */
int cddpm_create(cddpm_handle* out, const cddpm_unet_desc* desc, int device);
void cddpm_destroy(cddpm_handle h);
/* message of the last failing call on h (h == NULL: last failing cddpm_create) */
const char* cddpm_last_error(cddpm_handle h);
/* device bytes cddpm_create will allocate for this descriptor */
size_t cddpm_workspace_bytes(const cddpm_unet_desc* desc);

/* The state_dict entries the library consumes, in the reference's naming
 * (`diffusion.model.` prefix stripped): time_embed.0.weight, input_blocks.1.0.in_layers.2.weight, ...
 * (SURVEY.md 8a 'State-dict naming'; src/models/modules/OpenAI_Unet.py:583-797). */
int cddpm_num_weights(cddpm_handle h);
const char* cddpm_weight_name(cddpm_handle h, int i);
int64_t cddpm_weight_numel(cddpm_handle h, int i);

/* Replaces model.load_state_dict (src/train.py:161). `host_ptrs[i]` is the fp32 tensor `names[i]` in
 * PyTorch layout (Conv2d [Cout,Cin,kh,kw], Linear [out,in], Conv1d [Cout,Cin,1]); the library re-packs
 * into its own MFMA tile images and uploads. The caller keeps ownership. Every name reported by
 * cddpm_weight_name must be present; extra names are ignored. */
int cddpm_load_weights(cddpm_handle h, const char* const* names, const float* const* host_ptrs,
                       const int64_t* numels, int n);

/* Replaces the registered schedule buffers GaussianDiffusion reads in q_posterior / p_sample
 * (src/models/modules/cond_DDPM.py:366-371, :391-398, :444). Host arrays of length T:
 * posterior_mean_coef1, posterior_mean_coef2, posterior_log_variance_clipped,
 * sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod (the last two are read by CDDPM_PRED_NOISE steps and by every
 * DDIM step, cond_DDPM.py:385-389, :491; all five are required).
 * Also (re)builds the per-ResBlock time-embedding tables, so weights must be loaded first. */
int cddpm_set_schedule(cddpm_handle h, const float* coef1, const float* coef2, const float* logvar,
                       const float* sqrt_recip, const float* sqrt_recipm1, int T, int objective);

/* Replaces label_emb(cond) and the cond half of every ResBlock's emb_layers
 * (src/models/modules/OpenAI_Unet.py:583-590, :849-852, :300): computed once per batch, not per step.
 * cond_dev: [B, cond_dim]. */
int cddpm_prepare_cond(cddpm_handle h, const float* cond_dev, int B, void* stream);

/* Replaces UNetModel.forward / forward_with_cond_scale (src/models/modules/OpenAI_Unet.py:814-1006).
 * x_dev [B,1,H,W] -> out_dev [B,1,H,W]. t_dev: int32 [B] per-sample timesteps on the device, or NULL to
 * use t_uniform for every sample. Uses the context prepared by cddpm_prepare_cond for the same B. */
int cddpm_unet_forward(cddpm_handle h, const float* x_dev, const int32_t* t_dev, int t_uniform,
                       float* out_dev, int B, int H, int W, void* stream);

/* Replaces GaussianDiffusion.p_sample_loop, Gaussian branch (src/models/modules/cond_DDPM.py:446-464)
 * including p_sample / p_mean_variance / model_predictions / q_posterior (:391-444):
 * img_inout_dev holds x_T on entry ([B,1,H,W], N(0,1)) and the reconstruction in [0,1] on return.
 * Runs steps t = t_start-1 .. 0 (the reference's T = num_timesteps if start_t == 0 else start_t, :449,
 * is resolved by the caller). noise_dev: z_t at noise_dev + t*B*H*W for t in [1, t_start) (slot 0 unused),
 * or NULL to draw z_t on the device with Philox4x32-10 keyed (seed; quad, t, slice0 + b, stream)
 * -- see conditioned-diffusion-models-uad_amd/synth.py. */
int cddpm_reverse(cddpm_handle h, float* img_inout_dev, const float* noise_dev, uint64_t seed,
                  uint64_t slice0, int t_start, int B, int H, int W, void* stream);

/* Accumulation plan of the reverse steps (cddpm_p_sample / cddpm_reverse / cddpm_reverse_range; default family, handles whose maximum
 * geometry gives every Cout = 256 convolution >= 512 workgroups -- i.e. large batches). OPT-IN speed / accuracy trade, off by default
 * (t_switch = 2^30). Steps t >= t_switch multiply those convolutions on 256-cout workgroups that share the chunk's transformed patch
 * between two cout blocks and accumulate in two levels: +9...12 % per layer, +6 % per reverse step at B = 64; a convolution's rounding
 * noise ~6e-7 of rms instead of 1.9e-7 (the reference's CPU fmaf chain: 1.2e-6). Steps t < t_switch use the three-level kernel.
 * Measured on the full-length parity chain (B = 2 x 128 x 128 x T = 1000, final image vs the reference; tools/accum_switch_sweep.py):
 *   t_switch   off      500      300      200      0
 *   max        1.49e-4  1.32e-4  1.59e-4  2.69e-4  1.99e-4
 *   rms        6.95e-6  8.02e-6  9.08e-6  1.09e-5  1.02e-5        (the reference against itself: 1.02e-4 / 5.6e-6)
 * -- the extra rounding noise of ANY part of the chain survives to the end (there is no late switch step that hides it), which is why
 * the default keeps three-level accumulation on every step. A function of t alone: a slice's bits do not depend on its batch.
 * Single forwards (cddpm_unet_forward, cddpm_ddim_step, the training operators) always use the three-level kernel. */
int cddpm_set_accumulation_switch(cddpm_handle h, int t_switch);

/* `clip_denoised` of p_sample / ddim_sample (src/models/modules/cond_DDPM.py:433, :467): on (the reference's default, and the
 * handle's) clamps the x0 estimate to [-1,1] before the posterior mean / the DDIM update; off uses it as predicted.
 * Applies to every later cddpm_p_sample / cddpm_reverse / cddpm_reverse_range / cddpm_ddim_step on the handle. */
int cddpm_set_clip_denoised(cddpm_handle h, int on);

/* A segment of the same chain: steps t = t_hi, t_hi - 1, ..., t_lo (0 <= t_lo <= t_hi < T) of p_sample_loop's recurrence
 * (cond_DDPM.py:460-461) on img_inout_dev (values in [-1,1]); the map to [0,1] (:463) is applied exactly when t_lo == 0.
 * cddpm_reverse(t_start) == cddpm_reverse_range(t_start - 1, 0); cutting a chain into consecutive segments gives
 * bit-identical results (progress reporting, checkpoints of x_t, bench segments). noise_dev is indexed by the absolute t. */
int cddpm_reverse_range(cddpm_handle h, float* img_inout_dev, const float* noise_dev, uint64_t seed, uint64_t slice0,
                        int t_hi, int t_lo, int B, int H, int W, void* stream);

/* Replaces one GaussianDiffusion.p_sample call (src/models/modules/cond_DDPM.py:432-444): img <- x_{t-1} from x_t,
 * still in [-1,1]. z_dev: the step's N(0,1) draw [B,1,H,W], or NULL for the device Philox (ignored at t == 0). */
int cddpm_p_sample(cddpm_handle h, float* img_inout_dev, const float* z_dev, uint64_t seed, uint64_t slice0,
                   int t, int B, int H, int W, void* stream);

/* Replaces one iteration of GaussianDiffusion.ddim_sample's loop (src/models/modules/cond_DDPM.py:487-511): UNet at
 * t = `time`, eps from the UNCLIPPED x0 (model_predictions :400-420, clip_x_start = False), x0 clamped to [-1,1] (:496),
 * img <- x0 * coef_x0 + coef_eps * eps + sigma * z. The caller supplies what the reference computes from
 * alphas_cumprod_prev[time], alphas_cumprod_prev[time_next] in fp32 (:489-499): coef_x0 = sqrt(alpha_next),
 * sigma = eta sqrt((1 - alpha/alpha_next)(1 - alpha_next)/(1 - alpha)), coef_eps = sqrt(1 - alpha_next - sigma^2);
 * add_noise = (time_next > 0); z_dev = the step's N(0,1) draw or NULL for the device Philox keyed by t;
 * finalize != 0 on the last pair also maps the result to [0,1] (:513). */
int cddpm_ddim_step(cddpm_handle h, float* img_inout_dev, const float* z_dev, uint64_t seed, uint64_t slice0,
                    int t, float coef_x0, float coef_eps, float sigma, int add_noise, int finalize,
                    int B, int H, int W, void* stream);

/* Replaces torch.randn(shape) / torch.randn_like (src/models/modules/cond_DDPM.py:454, :440) with the
 * counter RNG: out_dev [B,1,H,W] ~ N(0,1); stream_id 0x1001 = x_T, 0x1002 = z_t. */
int cddpm_noise_fill(cddpm_handle h, float* out_dev, uint64_t seed, uint32_t stream_id, int t,
                     uint64_t slice0, int B, int H, int W, void* stream);

/* Replaces gen_noise(cfg, shape) for noisetype 'simplex' (src/utils/generate_noise.py:8-52: Simplex_CLASS, _init :214-232,
 * rand_2d_octaves :97-114, _noise2 :252-352; numba CPU code + H2D copy on every step in the reference). out_f16_dev
 * receives [B,1,H,W] IEEE half bit patterns: the SAME field for every batch item, float64 arithmetic rounded to float16
 * as torch's .half() does -- bit-exact with the reference for a given `seed` (the value Simplex_CLASS.newSeed draws with
 * np.random.randint). Square fields only, like the reference. Reference parameters: octaves 6, persistence 0.8, frequency 64. */
int cddpm_simplex_fill(cddpm_handle h, uint16_t* out_f16_dev, int64_t seed, int B, int H, int W, int octaves,
                       double persistence, double frequency, void* stream);

/* Replaces the residual-map post-processing of _test_step (src/utils/utils_eval.py:29-33 residual, :64-66 +
 * apply_brainmask_volume :447-460, :69-71 + apply_3d_median_filter :462-464), which the reference runs in scipy on the CPU
 * after copying the volume to the host. Volumes are [S][H][W] fp32 on the device (the reference indexes [H][W][S]; the
 * filters are symmetric under that permutation):
 *   out = |orig - recon| (squared = 0) or (orig - recon)^2 (squared = 1); recon_dev == NULL: out = orig (the volume
 *     already is a residual: apply_brainmask_volume / apply_3d_median_filter on their own);
 *   mask_dev != NULL: out *= erosion of (mask > 0), per slice, by the 4-connected cross applied erode_iterations times
 *     (scipy.ndimage.binary_erosion(structure = generate_binary_structure(2, 1), border_value = 0); the reference passes
 *     iterations = W / 25); erode_iterations = 0 multiplies by the mask as it is;
 *   median_k = 3 or 5: out = scipy.ndimage.median_filter(out, (k, k, k)), boundary mode 'reflect'; 0 = no filter.
 * Exact: the results equal scipy's bit for bit (a selection and a product with 0/1). tmp_dev: [S][H][W] scratch, needed
 * only when median_k != 0 (may be NULL otherwise); out_dev must not alias orig/recon/mask. */
int cddpm_residual_postprocess(cddpm_handle h, const float* orig_dev, const float* recon_dev, const float* mask_dev,
                               int S, int H, int W, int squared, int erode_iterations, int median_k, float* tmp_dev,
                               float* out_dev, void* stream);

/* Replaces q_sample (src/models/modules/cond_DDPM.py:548-554) fused with normalize_to_neg_one_to_one (:75):
 * out = sqrt_ac[t_b] * (2 x01 - 1) + sqrt_1mac[t_b] * noise; coefficient tables are host arrays [T]
 * uploaded on first use. Used by the single-step reconstruction (GaussianDiffusion.forward, :647-655). */
int cddpm_q_sample(cddpm_handle h, const float* x01_dev, const float* noise_dev, const int32_t* t_dev, int t_uniform,
                   const float* sqrt_ac_host, const float* sqrt_1mac_host, int T,
                   float* out_dev, int B, int H, int W, void* stream);

/* ---- measurement ------------------------------------------------------------------------------- */
/* Per-kernel-class timing with HIP events recorded on the launch stream around every kernel (bench.py's
 * roofline leg). Classes: 0 = fused 3x3 conv (MFMA), 1 = 1x1 conv (MFMA), 2 = attention core,
 * 3 = GroupNorm statistics+coefficients, 4 = other (input conv, head, pooling). cddpm_get_profile
 * synchronises, sums elapsed ms / algorithmic FLOPs / algorithmic bytes / launch counts per class since the
 * last call, and clears the records. */
#define CDDPM_PROF_CLASSES 5
int cddpm_set_profiling(cddpm_handle h, int on);
int cddpm_get_profile(cddpm_handle h, int ncls, double* ms, double* flops, double* bytes, int64_t* launches);

/* ---- test / debug surface (used by tests/ only) ------------------------------------------------- */

/* number of blocks in the forward program and their names ("input_blocks.3", "middle_block.1", ...) */
int cddpm_num_blocks(cddpm_handle h);
const char* cddpm_block_name(cddpm_handle h, int i);
/* ask the next cddpm_unet_forward calls to copy block i's output (NHWC [B,h,w,C] fp32) to dst_dev
 * (NULL clears); channels/height/width of that output for the given input size via cddpm_block_shape */
int cddpm_set_tap(cddpm_handle h, int block, float* dst_dev);
int cddpm_block_shape(cddpm_handle h, int block, int H, int W, int* C, int* h_out, int* w_out);

/* standalone fused convolution on NHWC tensors (the kernel behind every ResBlock conv), for kernel tests:
 * out[B,H,W,Cout] = conv_k(act(cat[src0,src1])) + bias (+ res), k in {1,3}, zero padding k/2, optional
 * nearest x2 upsampling of the sources (srcs are then [B,H/2,W/2,*]; upsample = 1: gather form, 2: the folded form the
 * UNet uses -- four 2x2-tap convolutions of the low-resolution input with pre-summed weights) and of the residual.
 * act(v) = silu?( (v - mean[b,c]) * a[b,c] + d[b,c] ) when coef_dev != NULL (coef_dev = [3][B][Cin]: mean, a, d).
 * w_host is PyTorch layout [Cout,Cin,k,k]. */
int cddpm_op_conv(cddpm_handle h, const float* src0_dev, int C0, const float* src1_dev, int C1,
                  const float* coef_dev, int silu, int upsample,
                  const float* w_host, const float* bias_host, int Cout, int ksize,
                  const float* res_dev, int res_upsample,
                  float* out_dev, int B, int H, int W, void* stream);

/* the ResBlock output convolution with its fused 1x1 skip_connection (OpenAI_Unet.py:261-268, :338): out = conv3x3(
 * act(src)) + conv1x1(skip) + bias, the skip operand RAW (un-normalised residual stream), both weight tensors packed
 * under one pre-scale exponent as cddpm_load_weights does. skip_dev [B,H,W,S0], wskip_host [Cout,S0,1,1]. Kernel tests. */
int cddpm_op_conv_skip(cddpm_handle h, const float* src0_dev, int C0, const float* coef_dev, int silu,
                       const float* w_host, const float* bias_host, int Cout, const float* skip_dev, int S0,
                       const float* wskip_host, float* out_dev, int B, int H, int W, void* stream);

/* a 3x3 convolution whose epilogue writes the GroupNorm statistics records of its output, followed by gn_finalize:
 * out_dev [B,H,W,Cout] = conv3x3(src) + bias and coef_dev [3][B][Cout] = (mean, a, d) of GroupNorm32(out) with
 * gamma/beta -- the statistics path every ResBlock uses (records from the producer, never a re-read). Kernel tests. */
int cddpm_op_conv_gn(cddpm_handle h, const float* src0_dev, int C0, const float* w_host, const float* bias_host, int Cout,
                     const float* gamma_host, const float* beta_host, float* out_dev, float* coef_dev, int B, int H, int W,
                     void* stream);

/* micro-benchmark of the fused conv kernel on device-generated N(0,1) data (no result check): average ms per
 * launch over `iters` launches; res_mode 0 none, 1 same resolution, 2 half resolution; skipC = channels of a fused
 * 1x1 skip_connection segment (0 = none). stamps_out (64 x uint64, may be NULL) receives per-wave phase cycle sums
 * in diagnostic builds (-DCDDPM_STAMPS), zeros otherwise. */
int cddpm_op_conv_bench(cddpm_handle h, int C0, int C1, int Cout, int ksize, int B, int H, int W, int use_coef,
                        int silu, int upsample, int res_mode, int skipC, int iters, double* ms_out,
                        uint64_t* stamps_out);

/* standalone GroupNorm(32) statistics + coefficient kernel pair: coef_dev [3][B][C] (mean, a, d) with
 * a = rstd*gamma*(1+scale), d = beta*(1+scale)+shift; film_dev = [B][2C] (scale | shift) or NULL. */
int cddpm_op_gn_coef(cddpm_handle h, const float* src0_dev, int C0, const float* src1_dev, int C1,
                     const float* gamma_host, const float* beta_host, const float* film_dev,
                     float* coef_dev, int B, int HW, void* stream);

/* host-only: the packed weight image of one convolution as the active kernel family expects it (no GPU needed).
 * format = 2: conv_x6.hip, fp16 split (default) -- w * 2^e (e = *scale_exp_out, the largest exponent <= 24 that keeps
 *   max|w| * 2^e below 2^14) split into two fp16 terms, hi + mid == w * 2^e to within 2^-23 relative (rms 0.73 x 2^-24);
 *   [Cout/128][Cin/32][taps][128 rows][8 slots of 8 fp16], slot (split s, u = channel/8 in the 32-channel chunk) of
 *   row j at (4 s + u) ^ ((j >> 1) & 7); 4 bytes per weight.
 * format = 1: conv_x6.hip, bf16 split (CDDPM_CONV=x6) -- three bf16 terms, hi + mid + lo == w exactly in fp32;
 *   [..][128 rows][12 slots of 8 bf16], slot (s, u) of row j at 4 s + (u ^ ((j >> 2) & 3)); 6 bytes per weight.
 * format = 0: conv_mfma.hip (CDDPM_CONV=f32) -- fp32, [..][128 rows][8 slots of 4 floats], slot s of row j at
 *   s ^ ((j >> 1) & 7); 4 bytes per weight.
 * cddpm_packed_conv_bytes returns the image size; cddpm_pack_conv_weights fills dst_host (that many bytes) from
 * PyTorch-layout w_host [Cout][Cin][k][k] (taps = k*k in {1, 9}; the size query also takes 4, one folded-upsample class) and returns
 * the format, or -1 on a bad shape. */
size_t cddpm_packed_conv_bytes(int Cout, int Cin, int taps);
int cddpm_pack_conv_weights(const float* w_host, int Cout, int Cin, int taps, void* dst_host, int* scale_exp_out);

/* ---- the training step (csrc/train_kernels.hip, csrc/encoder_train.hip): the operators that the host
 *      sequencing of the package (training.py, encoder_training.py) strings into src/models/DDPM_2D.py:114-135 -> cond_DDPM.py:565-645
 *      with gradients, Adam (:305-306) and the gradient all-reduce. Host-pointer variants (w_host ...) serve the kernel tests; the step
 *      itself runs on the device-resident calls further down (cddpm_op_pack_conv, cddpm_op_conv_packed, scratch arena). */
/* dL/d(input) of Conv2d(k in {1,3}, padding k/2): dx[B,H,W,Cin] = conv_k(dy[B,H,W,Cout], w transposed in (Cout,Cin) and flipped in
 * (ky,kx)) -- the fused forward convolution kernel on host-repacked weights. w_host is the FORWARD weight [Cout,Cin,k,k];
 * Cin must be a multiple of 128 and Cout of 32 (true for every convolution inside the UNet). */
int cddpm_op_conv_dgrad(cddpm_handle h, const float* dy_dev, int Cout, const float* w_host, int Cin, int ksize, float* dx_dev,
                        int B, int H, int W, void* stream);
/* dL/d(weight), dL/d(bias) of y = Conv2d(k in {1,3}, padding k/2) applied to a = act(cat[x0, x1]) with act(v) = silu?((v - mean) *
 * a + d) as in cddpm_op_conv (coef_dev [3][B][C0 + C1] or NULL): dw_dev [Cout,Cin,k,k] (PyTorch layout) = sum over batch and pixels of
 * dy (x) a, db_dev [Cout] = sum of dy (may be NULL). x0_dev [B,H,W,C0], x1_dev [B,H,W,C1] or NULL (C1 = 0), dy_dev [B,H,W,Cout];
 * C0 + C1 a multiple of 32, C0 of 32 when C1 > 0, Cout of 64 (the CDDPM_WGRAD=f32 family: 64 for k = 1, C0 of 64, H of 4). upsample != 0:
 * the conv input is the nearest x2 upsampling of act(x0) (up ResBlocks, OpenAI_Unet.py:289-293): x0_dev is [B,H/2,W/2,C0], H and W even.
 * Arithmetic (environment CDDPM_WGRAD): h3 (default) products from two-term fp16 splits of both operands on v_mfma_f32_16x16x32_f16,
 * fp32 accumulation -- the forward kernel's arithmetic; h1 plain fp16 operands; f32 v_mfma_f32_32x32x2_f32. */
int cddpm_op_conv_wgrad(cddpm_handle h, const float* x0_dev, int C0, const float* x1_dev, int C1, const float* coef_dev, int silu,
                        int upsample, const float* dy_dev, int Cout, int ksize, float* dw_dev, float* db_dev, int B, int H, int W,
                        void* stream);
/* db_dev[C] = sum over the npix rows of dy_dev [npix][C] (bias gradient of a convolution) */
int cddpm_op_bias_grad(cddpm_handle h, const float* dy_dev, int64_t npix, int C, float* db_dev, void* stream);
/* backward of the attention core QKVAttention (src/models/modules/OpenAI_Unet.py:457-476): qkv_dev [B,N,3C] (q | k | v, heads of 64
 * channels), da_dev [B,N,C] = dL/d(output) -> dqkv_dev [B,N,3C]. The probabilities are recomputed (2 x B x C/64 x N x N floats of
 * scratch are allocated for the call). */
int cddpm_op_attention_backward(cddpm_handle h, const float* qkv_dev, const float* da_dev, float* dqkv_dev, int B, int N, int C,
                                void* stream);
/* backward of torch.nn.Linear behind an optional SiLU, y = [SiLU](x) W^T + b (emb_layers / time_embed / label_emb,
 * OpenAI_Unet.py:201-207, :583-602): x_dev [M,K], w_dev [N,K], dy_dev [M,N] -> dw_dev [N,K], db_dev [N] (may be NULL), dx_dev [M,K]
 * (may be NULL). */
int cddpm_op_linear_backward(cddpm_handle h, const float* x_dev, const float* w_dev, const float* dy_dev, int M, int N, int K,
                             int silu_in, float* dw_dev, float* db_dev, float* dx_dev, void* stream);
/* small forward operators of the UNet as stand-alone entry points (the training step's host code sequences them; every pointer is a
 * device pointer, NHWC fp32): y[M,N] = [SiLU](x[M,K]) w[N,K]^T + b (torch.nn.Linear); input_blocks.0 Conv2d(1 -> C, 3x3) with w [C][9];
 * the head out.2 Conv2d(C -> 1, 3x3) on act(x) with w9 [9][C]; the down-ResBlock front end (hp = avgpool2(act(x)), xp = avgpool2(x)). */
int cddpm_op_linear(cddpm_handle h, const float* x_dev, const float* w_dev, const float* b_dev, int M, int N, int K, int silu_in,
                    float* y_dev, void* stream);
int cddpm_op_conv_in1(cddpm_handle h, const float* x_dev, const float* w_dev, const float* b_dev, float* out_dev, int B, int H, int W,
                      int C, void* stream);
int cddpm_op_head(cddpm_handle h, const float* x_dev, const float* coef_dev, const float* w9_dev, float bias, const float* bias_dev,
                  float* out_dev, int B, int H, int W, int C, void* stream);   /* bias_dev (device, 1 float) overrides bias when given */
int cddpm_op_pool_act(cddpm_handle h, const float* x_dev, const float* coef_dev, float* hp_dev, float* xp_dev, int B, int H, int W, int C,
                      void* stream);
/* resampling backward and accumulation: dx[B,H,W,C] (+)= scale * dyp[B,H/2,W/2,C] at (y/2, x/2) (AvgPool2d(2) backward: scale 1/4);
 * dxp[B,H/2,W/2,C] (+)= 2x2 block sums of dy (nearest x2 upsample backward); a += b (n floats, multiple of 4). */
int cddpm_op_unpool2(cddpm_handle h, const float* dyp_dev, float* dx_dev, int B, int H, int W, int C, float scale, int accumulate, void* stream);
int cddpm_op_sumpool2(cddpm_handle h, const float* dy_dev, float* dxp_dev, int B, int H, int W, int C, int accumulate, void* stream);
int cddpm_op_add_inplace(cddpm_handle h, float* a_dev, const float* b_dev, int64_t n, void* stream);
/* weight gradients of the two single-channel convolutions: dw_dev [C][9] = sum_{b,q} act(T)[b,q,c] * s[b, q + sign * tap]
 * (input conv: T = dL/d(output), s = the image, sign +1; head conv: T = the head's input, act = its GroupNorm + SiLU, s = dL/d(out),
 * sign -1), and the head's input gradient dact[b,q,c] = sum_tap w9[tap][c] dout[b, q - tap]. */
int cddpm_op_chan_image_corr(cddpm_handle h, const float* t_dev, const float* coef_dev, int silu, const float* s_dev, int sign, float* dw_dev,
                             int B, int H, int W, int C, void* stream);
int cddpm_op_head_dgrad(cddpm_handle h, const float* dout_dev, const float* w9_dev, float* dact_dev, int B, int H, int W, int C, void* stream);
/* p_losses' loss (cond_DDPM.py:636-645): loss_b_dev[b] = p2w[b] * mean_p |out - target|^(1|2) (their mean is the loss), dout_dev =
 * grad_scale * dL/d(out). grad_scale (a power of two, e.g. B * HW rounded) is the loss scale of the backward pass: the convolution
 * input-gradient kernels form their products from fp16 operand splits whose absolute floor is 2^-25, so the gradient tensors are carried
 * at O(1) magnitude (every backward operator is linear in them, the scaling is exact) and cddpm_op_adam divides it out again. */
int cddpm_op_loss(cddpm_handle h, const float* out_dev, const float* target_dev, const float* w_b_dev, int l2, int B, int HW, float grad_scale,
                  float* dout_dev, float* loss_b_dev, void* stream);
/* one Adam update (DDPM_2D.py:305-306: lr 1e-4, torch defaults) of a flat parameter vector; step counts from 1; the gradient used is
 * g_dev * grad_unscale (1 / the loss scale) */
int cddpm_op_adam(cddpm_handle h, float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n, float lr, float beta1, float beta2,
                  float eps, int step, float grad_unscale, void* stream);
/* The guarded form of the update (what torch's GradScaler does for the reference trainer, `precision: 16` in
 * configs/trainer/default.yaml:7: a step whose gradients hold inf / NaN is skipped and does not count), entirely on the device.
 * ctrl_dev: int32[8], zero-initialised by the caller once = {non-finite flag, optimizer step, skip, skipped so far, bits of
 * 1 - beta1^step, bits of 1 - beta2^step, 0, 0}. Per optimisation step: cddpm_op_grad_check on every gradient buffer (ORs the flag),
 * ONE cddpm_op_guard_commit (flag set: skip = 1, skipped += 1; else skip = 0, step += 1, bias corrections refreshed; flag cleared),
 * then cddpm_op_adam_guarded on every parameter buffer (a no-op when skip is set; step and bias corrections come from ctrl_dev). */
/* Arithmetic of the training operators, process-wide: 32 (default) = fp32-grade products from two-term fp16 splits; 16 = plain fp16
 * operands with fp32 accumulation in cddpm_op_conv_packed and cddpm_op_conv_wgrad -- what the reference trainer's `precision: 16`
 * (configs/trainer/default.yaml:7) computes under autocast; GroupNorm, attention, embeddings, Adam and the master weights stay fp32 in both.
 * Initial value: environment CDDPM_TRAIN_PRECISION (16 | unset). set returns the previous value, -1 for an unsupported `bits`.
 * The reconstruction entry points are not affected. */
int cddpm_set_train_precision(int bits);
int cddpm_get_train_precision(void);
int cddpm_op_grad_check(cddpm_handle h, const float* g_dev, int64_t n, int32_t* ctrl_dev, void* stream);
int cddpm_op_guard_commit(cddpm_handle h, int32_t* ctrl_dev, float beta1, float beta2, void* stream);
int cddpm_op_adam_guarded(cddpm_handle h, float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n, float lr, float beta1,
                          float beta2, float eps, float grad_unscale, const int32_t* ctrl_dev, void* stream);
/* ---- device-resident operator calls (what the training step runs on: no host staging, no synchronisation) ----
 * cddpm_op_set_scratch gives the handle an arena of `bytes` (0: release it) from which the operators of this header take their
 * temporaries instead of a hipMalloc / synchronise / hipFree per call; calls then only enqueue work on `stream` (one stream).
 * An operator that needs more than the arena holds fails with the size in cddpm_last_error.
 * Parameter vectors documented as *_host (GroupNorm gamma / beta) may be device pointers: they are then used where they lie.
 * cddpm_op_absmax: out_dev[0] = max |x| (the caller derives a tensor's power-of-two pre-scale from it: the largest e in [0, 24] with
 * max|w| 2^e < 2^14). cddpm_op_pack_conv: the packed image of cddpm_pack_conv_weights (format 2), written by a kernel from the device
 * tensor w_dev [Cout][Cin][k][k]; mode 0 = forward operator, 1 = the input-gradient operator (transposed, taps flipped: Cin outputs,
 * Cout inputs; cddpm_packed_conv_bytes(Cin, Cout, k*k) bytes), 2 = the four folded classes of "nearest x2 upsample -> conv3x3"
 * (4 * cddpm_packed_conv_bytes(Cout, Cin, 4) bytes; the class sums can reach 4 max|w|). Bit-identical to the host packer for the same
 * exponent. Default convolution family only.
 * Environment CDDPM_TRAIN_PRECISION=16 (read once per process): cddpm_op_conv_packed multiplies plain fp16 operands (the hi terms of the
 * images only, fp32 accumulation) and cddpm_op_conv_wgrad defaults to its h1 family -- the arithmetic of the reference trainer's precision 16.
 * The reconstruction entry points (cddpm_reverse, cddpm_unet_forward, ...) are not affected.
 * cddpm_op_conv_packed: cddpm_op_conv / cddpm_op_conv_skip on such images: out = conv_k(act(cat[src0, src1])) [+ conv1x1(skip)] + bias
 * [+ res]; bias_dev NULL = none; skip_dev NULL = no skip segment (its image shares scale_exp); folded_up: src0 is at H/2 x W/2. */
int cddpm_op_set_scratch(cddpm_handle h, size_t bytes);
int cddpm_op_absmax(cddpm_handle h, const float* x_dev, int64_t n, float* out_dev, void* stream);
int cddpm_op_pack_conv(cddpm_handle h, const float* w_dev, int Cout, int Cin, int ksize, int mode, int scale_exp, void* packed_dev,
                       void* stream);
/* The same packer for MANY images in one launch (the training step re-packs 69 convolutions x 2 operators after every update): a device
 * table of jobs, each the arguments of cddpm_op_pack_conv with the operator's dimensions resolved -- O / I = output / input channels of
 * the PACKED operator (mode 1: the forward tensor's Cin / Cout), taps = 1 | 9 (mode 2: 4), cls = the folded-upsample class 0..3 (mode 2:
 * one job per class, dst the image base; else 0). max_units = the largest O / 128 * I / 32 * taps * 512 over the jobs. */
typedef struct cddpm_pack_job { const float* w_dev; void* packed_dev; int32_t O, I, taps, mode, scale_exp, cls; } cddpm_pack_job;
int cddpm_op_pack_conv_batch(cddpm_handle h, const cddpm_pack_job* jobs_dev, int njobs, int64_t max_units, void* stream);
int cddpm_op_conv_packed(cddpm_handle h, const float* src0_dev, int C0, const float* src1_dev, int C1, const float* coef_dev, int silu,
                         int folded_up, const void* packed_dev, int scale_exp, const float* bias_dev, int Cout, int ksize,
                         const float* res_dev, int res_upsample, const float* skip_dev, int S0, const float* skip1_dev, int S1,
                         const void* skip_packed_dev, float* out_dev, float* stats_dev, int B, int H, int W, void* stream);
/* skip1_dev / S1 (0: none): the 1x1 skip segment's input as the concatenation of two tensors (its image covers S0 + S1 channels). */
/* stats_dev (or NULL): the output's GroupNorm statistics records [B][cddpm_stat_records(H, W, folded_up ? 1 : 0)][Cout][2], written by the
 * convolution's epilogue; cddpm_op_gn_coef_rec is cddpm_op_gn_coef on such records (no sweep over the tensors), and
 * cddpm_op_gn_silu_backward takes them through rec_dev / nrec. */
int cddpm_op_gn_coef_rec(cddpm_handle h, const float* rec0_dev, int n0, int C0, const float* rec1_dev, int n1, int C1, const float* gamma_host,
                         const float* beta_host, const float* film_dev, float* coef_dev, int B, int HW, void* stream);
/* ---- training-mode operators of the context encoder (timm ResNet-50, in_chans = 1; reference src/models/modules/DDPM_encoder.py:21-23,
 * trained jointly with the UNet by src/models/DDPM_2D.py:114-135 / :305-306). NHWC fp32 device tensors, plain fp32 FMA kernels (the encoder is
 * ~5 % of a training step's FLOPs). Strided operators map n -> ceil(n / stride).
 * cddpm_op_enc_pack_w: PyTorch weight [Cout][Cin][K][K] -> the images the convolution reads: wf [K*K][Cin][Cout] (forward) and, when wd_dev is
 *   given, wd [K*K][Cout][Cin] (input gradient).
 * cddpm_op_enc_conv: transposed = 0: z [B,Ho,Wo,Cout] = conv_K,stride,pad K/2 (x [B,H,W,Cin]) with wf; transposed = 1: dx [B,H,W,Cin] from
 *   dz [B,Ho,Wo,Cout] with wd (H, W are always the convolution's INPUT size). No bias (ResNet convolutions have none).
 * cddpm_op_enc_conv_wgrad: dw [Cout][Cin][K][K] (PyTorch layout) from x and dz.  cddpm_op_enc_stem / _stem_wgrad: the 7x7 / 2 single-channel stem
 *   (x [B,H,W] -> z [B,ceil(H/2),ceil(W/2),64], weight [64][49]).
 * cddpm_op_enc_bn_forward: BatchNorm2d in TRAINING mode + optional shortcut + optional ReLU:
 *   y = relu(((z - mean_c) rstd_c gamma_c + beta_c) s_b + res), batch statistics over the N = B * HW pixels (fp64 sums), mean_rstd_dev [2][C] kept
 *   for the backward, running statistics updated as torch does (momentum, unbiased variance) when given; s_b (sample_scale_dev [B] or NULL = 1) is
 *   the stochastic-depth scale of the residual branch (timm drop_path).  cddpm_op_enc_bn_backward: dz, dgamma, dbeta and (dres_dev given) the
 *   shortcut operand's gradient dy [y > 0].
 * cddpm_op_enc_maxpool: 3x3 / 2 pad 1 forward (backward = 0) or its backward in gather form (first maximum in row-major order, as torch;
 *   deterministic).  cddpm_op_enc_avgpool: global average pool x [B,HW,C] -> g [B,C], or (backward = 1) dL/dg [B,C] (first pointer) -> dL/dx (second). */
int cddpm_op_enc_pack_w(cddpm_handle h, const float* w_dev, int Cout, int Cin, int K, float* wf_dev, float* wd_dev, void* stream);
int cddpm_op_enc_conv(cddpm_handle h, const float* src_dev, const float* w_img_dev, float* dst_dev, int B, int H, int W, int Cin, int Cout, int K,
                      int stride, int transposed, void* stream);
int cddpm_op_enc_conv_wgrad(cddpm_handle h, const float* x_dev, const float* dz_dev, float* dw_dev, int B, int H, int W, int Cin, int Cout, int K,
                            int stride, void* stream);
int cddpm_op_enc_stem(cddpm_handle h, const float* x_dev, const float* w_dev, float* z_dev, int B, int H, int W, void* stream);
int cddpm_op_enc_stem_wgrad(cddpm_handle h, const float* x_dev, const float* dz_dev, float* dw_dev, int B, int H, int W, void* stream);
int cddpm_op_enc_bn_forward(cddpm_handle h, const float* z_dev, const float* gamma_dev, const float* beta_dev, const float* sample_scale_dev,
                            const float* res_dev, int relu, float eps, float momentum, float* run_mean_dev, float* run_var_dev, float* mean_rstd_dev,
                            float* y_dev, int64_t N, int HW, int C, void* stream);
int cddpm_op_enc_bn_backward(cddpm_handle h, const float* z_dev, const float* y_dev, const float* dy_dev, const float* mean_rstd_dev,
                             const float* gamma_dev, const float* sample_scale_dev, int relu, float* dz_dev, float* dres_dev, float* dgamma_dev,
                             float* dbeta_dev, int64_t N, int HW, int C, void* stream);
int cddpm_op_enc_maxpool(cddpm_handle h, const float* x_dev, float* y_dev, int B, int H, int W, int C, int backward, const float* dy_dev, float* dx_dev,
                         void* stream);
int cddpm_op_enc_avgpool(cddpm_handle h, const float* x_dev, float* g_dev, int B, int HW, int C, int backward, void* stream);

/* backward of a = act(GroupNorm32(x) * (1 + scale) + shift), act = SiLU (silu != 0) or identity (OpenAI_Unet.py:284-338, :325-330):
 * given da_dev [B,HW,C] writes dx_dev [B,HW,C] (with x1_dev: x = cat[x_dev [.., C - C1], x1_dev [.., C1]], the input gradient split the same
 * way into dx_dev / dx1_dev; needs rec_dev), dgamma_dev / dbeta_dev [C] and, when film_dev ([B][2C] scale | shift) is given,
 * dfilm_dev [B][2C]. The forward statistics are recomputed from x_dev. Everything NHWC fp32. */
int cddpm_op_gn_silu_backward(cddpm_handle h, const float* x_dev, const float* x1_dev /* second tensor of a concatenated input, or NULL */, int C1,
                              const float* da_dev, const float* gamma_host,
                              const float* beta_host, const float* film_dev, int silu, float* dx_dev, float* dx1_dev /* [B,HW,C1] */, float* dgamma_dev,
                              float* dbeta_dev, float* dfilm_dev, const float* rec_dev /* x's statistics records or NULL */, int nrec,
                              const float* add_dev /* [B,HW,C] added to dx (the gradient of a parallel skip path), or NULL */, int B, int HW, int C,
                              void* stream);

/* GroupNorm statistics record counts per sample for an H x W tensor (host arithmetic, callable without a GPU):
 * kind 0 = records the fused convolution's epilogue writes, 1 = the folded-upsample convolution's, 2 = the stand-alone
 * sweep's pixel-range split. cddpm_create sizes every records buffer for the largest of the three at max_h x max_w, and
 * every producer launch is refused if its count would not fit (tests/test_host_logic.py checks the counts are monotone). */
int cddpm_stat_records(int H, int W, int kind);

/* standalone attention core on qkv NHWC [B,N,3C] (q | k | v, heads = contiguous groups of head_channels):
 * out [B,N,C] = softmax(q k^T / sqrt(head_channels)) v  (QKVAttention, OpenAI_Unet.py:457-476). */
int cddpm_op_attention(cddpm_handle h, const float* qkv_dev, float* out_dev, int B, int N, int C, void* stream);

/* ---- context encoder ---------------------------------------------------------------
 * Replaces the module get_encoder builds (src/models/modules/DDPM_encoder.py:6-29): timm resnet50(in_chans=1,
 * num_classes=cond_dim) in eval mode, called once per slice batch by DDPM_2D.forward (src/models/DDPM_2D.py:98-104).
 * PARITY UNPINNED: timm is not installed in the build image; the network follows timm's published ResNet-50 v1.5 and
 * its state_dict names and is checked against a torch restatement of that description (oracle/encoder_oracle.py).
 * load_weights takes host pointers under timm's key names (conv1.weight, bn1.{weight,bias,running_mean,running_var},
 * layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.0,downsample.1}.*, fc.{weight,bias}); BatchNorm is folded.
 * forward: x_dev [B,1,H,W] fp32 in [0,1] (H, W >= 32) -> out_dev [B, num_classes] on the caller's stream. */
typedef struct cddpm_encoder_ctx* cddpm_encoder_handle;
int cddpm_encoder_create(cddpm_encoder_handle* out, int num_classes, int max_batch, int max_h, int max_w, int device);
void cddpm_encoder_destroy(cddpm_encoder_handle h);
const char* cddpm_encoder_last_error(cddpm_encoder_handle h);
int cddpm_encoder_num_weights(void);
int cddpm_encoder_load_weights(cddpm_encoder_handle h, const char* const* names, const float* const* host_ptrs,
                               const int64_t* numels, int n);
int cddpm_encoder_forward(cddpm_encoder_handle h, const float* x_dev, float* out_dev, int B, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CDDPM_H */
