"""ORACLE -- test infrastructure, not product code.

CPU restatement (plain torch fp32 functional ops, no nn.Module tree) of the reference's cDDPM
reverse-diffusion path: GaussianDiffusion.p_sample_loop over the conditioned OpenAI-style UNet.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
shipped package never does (it fails loudly without the HIP library instead).

Parity pin: the reference has no tests or golden vectors for this path, so
this restatement is pinned against the REFERENCE ITSELF, imported in the build container by
oracle/ref_harness.py; oracle/make_golden.py stores the reference's outputs under tests/golden/
and tests/test_oracle_golden.py checks this file against them (no reference needed at test time).

All file:line citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# S1: noise schedule   (src/models/modules/cond_DDPM.py:271-287, :328-377)
# ----------------------------------------------------------------------------------------------

def beta_schedule(timesteps: int, kind: str = "cosine") -> Tensor:
    """float64 betas. cosine: cond_DDPM.py:277-287 (s=0.008, clip 0..0.999); linear: :271-275."""
    if kind == "cosine":
        s = 0.008
        x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
        ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
        ac = ac / ac[0]
        betas = 1 - (ac[1:] / ac[:-1])
        return torch.clip(betas, 0, 0.999)
    if kind == "linear":
        scale = 1000 / timesteps
        return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)
    raise ValueError(f"unknown beta schedule {kind}")


def schedule_buffers(timesteps: int, kind: str = "cosine", p2_gamma: float = 0.0, p2_k: float = 1.0) -> Dict[str, Tensor]:
    """The 13 fp32 buffers GaussianDiffusion registers (cond_DDPM.py:336-377), computed in float64
    and rounded to float32 exactly as `register_buffer(name, val.to(torch.float32))` does (:350)."""
    betas = beta_schedule(timesteps, kind)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    b = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": torch.log(post_var.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
        "p2_loss_weight": (p2_k + ac / (1 - ac)) ** -p2_gamma,
    }
    return {k: v.to(torch.float32) for k, v in b.items()}


# ----------------------------------------------------------------------------------------------
# U1: sinusoidal timestep embedding   (src/models/LDM/modules/diffusionmodules/util.py:151-171)
# ----------------------------------------------------------------------------------------------

def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    """cat[cos(t f), sin(t f)], f_i = exp(-ln(max_period) i / half) -- cosine half FIRST (util.py:166)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


# ----------------------------------------------------------------------------------------------
# UNet    (src/models/modules/OpenAI_Unet.py)
# ----------------------------------------------------------------------------------------------

def _gn(x: Tensor, sd, prefix: str) -> Tensor:
    """GroupNorm32(32, C): fp32, eps 1e-5, affine (util.py:199-216). (Computes in the weights' dtype so the
    same restatement run with float64 weights serves as the rounding-free yardstick in tests.)"""
    w = sd[prefix + ".weight"]
    return F.group_norm(x.to(w.dtype), 32, w, sd[prefix + ".bias"], eps=1e-5)


def _conv(x: Tensor, sd, prefix: str, pad: int) -> Tensor:
    return F.conv2d(x, sd[prefix + ".weight"], sd[prefix + ".bias"], padding=pad)


def resblock(x: Tensor, emb: Tensor, sd, prefix: str, up: bool = False, down: bool = False) -> Tensor:
    """ResBlock._forward with use_scale_shift_norm=True, dropout 0 (OpenAI_Unet.py:284-338).

    up/down blocks resample BOTH the activated h and the raw x, before the first conv (:287-293);
    Upsample(use_conv=False) is nearest x2 (:118-128), Downsample(use_conv=False) is AvgPool2d(2) (:166-177).
    FiLM: scale = first half of emb_out, shift = second half; h = GN(h) * (1 + scale) + shift (:325-330).
    """
    h = F.silu(_gn(x, sd, prefix + ".in_layers.0"))
    if up:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif down:
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    h = _conv(h, sd, prefix + ".in_layers.2", 1)
    emb_out = F.linear(F.silu(emb), sd[prefix + ".emb_layers.1.weight"], sd[prefix + ".emb_layers.1.bias"])
    scale, shift = torch.chunk(emb_out[:, :, None, None], 2, dim=1)
    h = _gn(h, sd, prefix + ".out_layers.0") * (1 + scale) + shift
    h = _conv(F.silu(h), sd, prefix + ".out_layers.3", 1)
    if (prefix + ".skip_connection.weight") in sd:
        x = _conv(x, sd, prefix + ".skip_connection", 0)
    return x + h


def attention_block(x: Tensor, sd, prefix: str, head_channels: int = 64) -> Tensor:
    """AttentionBlock._forward + QKVAttention (new order) (OpenAI_Unet.py:386-394, :457-476).

    qkv = Conv1d(GN(x)); q,k,v = chunk(3, dim=1); heads are contiguous channel groups of 64;
    w = softmax_fp32((q s)^T (k s)), s = ch^-1/4; a = w v^T; x + proj_out(a).
    """
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    n = xf.shape[-1]
    wn = sd[prefix + ".norm.weight"]
    qkv = F.conv1d(F.group_norm(xf.to(wn.dtype), 32, wn, sd[prefix + ".norm.bias"], eps=1e-5),
                   sd[prefix + ".qkv.weight"], sd[prefix + ".qkv.bias"])
    heads = c // head_channels
    ch = head_channels
    q, k, v = qkv.chunk(3, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", (q * scale).reshape(b * heads, ch, n), (k * scale).reshape(b * heads, ch, n))
    w = torch.softmax(w.to(wn.dtype), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v.reshape(b * heads, ch, n)).reshape(b, -1, n)
    h = F.conv1d(a, sd[prefix + ".proj_out.weight"], sd[prefix + ".proj_out.bias"])
    return (xf + h).reshape(b, c, hh, ww)


def unet_embedding(t: Tensor, cond: Optional[Tensor], sd, model_channels: int) -> Tensor:
    """emb = cat[time_embed(temb(t)), label_emb(cond)] (OpenAI_Unet.py:583-602, :846-852).
    The concat with label_emb(cond) is the 'Spark-encoder context concat' of the north star."""
    te = timestep_embedding(t, model_channels).to(sd["time_embed.0.weight"].dtype)
    e = F.linear(te, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    e = F.linear(F.silu(e), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    if "label_emb.0.weight" in sd and cond is not None:
        c = F.linear(cond, sd["label_emb.0.weight"], sd["label_emb.0.bias"])
        c = F.linear(F.silu(c), sd["label_emb.2.weight"], sd["label_emb.2.bias"])
        e = torch.cat([e, c], dim=1)
    return e


def unet_forward(x: Tensor, t: Tensor, cond: Optional[Tensor], sd: Dict[str, Tensor], *,
                 model_channels: int = 128, channel_mult: Sequence[int] = (1, 2, 2), num_res_blocks: int = 3,
                 attention_resolutions: Sequence[int] = (3, 6, 12), head_channels: int = 64,
                 taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """UNetModel.forward (OpenAI_Unet.py:823-1006) for resblock_updown=True, use_scale_shift_norm=True,
    use_new_attention_order=True, num_head_channels=64 (ctor arguments at src/models/DDPM_2D.py:37-59).
    The debug clone()/features_info traffic (:305-315, :854-977) has no observable output and is omitted.
    `taps`, if given, receives intermediate activations by block name (for per-kernel GPU tests)."""
    emb = unet_embedding(t, cond, sd, model_channels)
    hs: List[Tensor] = []
    h = _conv(x, sd, "input_blocks.0.0", 1)
    hs.append(h)
    if taps is not None:
        taps["emb"] = emb
        taps["input_blocks.0"] = h
    ds = 1
    idx = 1
    for level, _mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            h = resblock(h, emb, sd, f"input_blocks.{idx}.0")
            if ds in attention_resolutions:
                h = attention_block(h, sd, f"input_blocks.{idx}.1", head_channels)
            hs.append(h)
            if taps is not None:
                taps[f"input_blocks.{idx}"] = h
            idx += 1
        if level != len(channel_mult) - 1:
            h = resblock(h, emb, sd, f"input_blocks.{idx}.0", down=True)
            hs.append(h)
            if taps is not None:
                taps[f"input_blocks.{idx}"] = h
            idx += 1
            ds *= 2
    h = resblock(h, emb, sd, "middle_block.0")
    if taps is not None:
        taps["middle_block.0"] = h
    h = attention_block(h, sd, "middle_block.1", head_channels)
    if taps is not None:
        taps["middle_block.1"] = h
    h = resblock(h, emb, sd, "middle_block.2")
    if taps is not None:
        taps["middle_block.2"] = h
    idx = 0
    for level, _mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            h = torch.cat([h, hs.pop()], dim=1)          # OpenAI_Unet.py:948
            h = resblock(h, emb, sd, f"output_blocks.{idx}.0")
            sub = 1
            if ds in attention_resolutions:
                h = attention_block(h, sd, f"output_blocks.{idx}.{sub}", head_channels)
                sub += 1
            if level and i == num_res_blocks:
                h = resblock(h, emb, sd, f"output_blocks.{idx}.{sub}", up=True)
                ds //= 2
            if taps is not None:
                taps[f"output_blocks.{idx}"] = h
            idx += 1
    h = F.silu(_gn(h, sd, "out.0"))
    return _conv(h, sd, "out.2", 1)


# ----------------------------------------------------------------------------------------------
# S2-S5: diffusion process   (src/models/modules/cond_DDPM.py)
# ----------------------------------------------------------------------------------------------

def q_sample(x0: Tensor, t: Tensor, noise: Tensor, buf) -> Tensor:
    """sqrt(abar_t) x0 + sqrt(1-abar_t) eps (cond_DDPM.py:548-554)."""
    sh = (-1,) + (1,) * (x0.dim() - 1)
    return buf["sqrt_alphas_cumprod"][t].reshape(sh) * x0 + buf["sqrt_one_minus_alphas_cumprod"][t].reshape(sh) * noise


def p_sample(x: Tensor, t: int, cond: Optional[Tensor], sd, buf, z: Optional[Tensor], objective: str = "pred_x0",
             clip_denoised: bool = True, **unet_kw) -> Tensor:
    """One reverse step t -> t-1 (cond_DDPM.py:432-444 -> :422-430 -> :400-420 -> :391-398).
    x0_hat = clamp(model, -1, 1) (pred_x0) or clamp(predict_start_from_noise) (pred_noise);
    mean = coef1[t] x0_hat + coef2[t] x_t; out = mean + exp(0.5 logvar[t]) z, z = 0 at t == 0."""
    b = x.shape[0]
    tt = torch.full((b,), t, dtype=torch.long)
    out = unet_forward(x, tt, cond, sd, **unet_kw)
    if objective == "pred_x0":
        x0 = out
    elif objective == "pred_noise":
        x0 = buf["sqrt_recip_alphas_cumprod"][t] * x - buf["sqrt_recipm1_alphas_cumprod"][t] * out
    else:
        raise ValueError(f"unknown objective {objective}")
    if clip_denoised:          # (:416-419 maybe_clip, :426-427)
        x0 = x0.clamp(-1.0, 1.0)
    mean = buf["posterior_mean_coef1"][t] * x0 + buf["posterior_mean_coef2"][t] * x
    if t > 0:
        return mean + (0.5 * buf["posterior_log_variance_clipped"][t]).exp() * z
    return mean + 0.0


def p_sample_loop(x_T: Tensor, cond: Optional[Tensor], sd, buf, noises, start_t: int = 0,
                  objective: str = "pred_x0", **unet_kw) -> Tensor:
    """p_sample_loop, Gaussian branch (cond_DDPM.py:446-464): T = num_timesteps if start_t == 0 else start_t;
    for t = T-1 .. 0: img = p_sample(img, t); return (img + 1) / 2.
    `noises(t)` returns z for step t (t >= 1): the reference draws randn_like once per step, in this order."""
    T = buf["betas"].shape[0] if start_t == 0 else start_t
    img = x_T
    with torch.no_grad():
        for t in reversed(range(T)):
            z = noises(t) if t > 0 else None
            img = p_sample(img, t, cond, sd, buf, z, objective, **unet_kw)
    return (img + 1) * 0.5


def ddim_time_pairs(num_timesteps: int, sampling_timesteps: int, start_t: int = 0):
    """(time, time_next) pairs of ddim_sample (cond_DDPM.py:468-474)."""
    total = num_timesteps if start_t == 0 else start_t
    times = torch.linspace(0.0, total, steps=sampling_timesteps + 2)[:-1]
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


def ddim_sample(x_T: Tensor, cond: Optional[Tensor], sd, buf, noises, sampling_timesteps: int, eta: float = 1.0,
                start_t: int = 0, x_start: Optional[Tensor] = None, objective: str = "pred_x0", clip_denoised: bool = True,
                **unet_kw) -> Tensor:
    """ddim_sample, Gaussian branch (cond_DDPM.py:466-515). x_T is the N(0,1) draw that is USED (the reference draws one
    more before it and throws it away, :479/:484; with start_t != 0 the used draw is q_sample's noise, :482).
    Per pair: alpha = alphas_cumprod_prev[time], alpha_next = alphas_cumprod_prev[time_next] (:489-490);
    model_predictions with clip_x_start = False (:494, :400-420): eps from the UNCLIPPED x0, then x0.clamp_ (:496);
    sigma = eta sqrt((1 - alpha/alpha_next)(1 - alpha_next)/(1 - alpha)), c = sqrt(1 - alpha_next - sigma^2) (:498-499);
    img = x0 sqrt(alpha_next) + c eps + sigma z, z = randn_like if time_next > 0 else 0 (:501-511); (img + 1)/2 (:513).
    `noises(time)` returns z for the pair whose current step is `time`."""
    T = buf["betas"].shape[0]
    dt = x_T.dtype
    acp = buf["alphas_cumprod_prev"]
    img = x_T
    if start_t != 0:
        tt = torch.tensor([start_t])
        img = q_sample(x_start, tt, x_T, buf)[:, 0].unsqueeze(1)
    b = x_T.shape[0]
    with torch.no_grad():
        for time, time_next in ddim_time_pairs(T, sampling_timesteps, start_t):
            alpha, alpha_next = acp[time], acp[time_next]
            out = unet_forward(img, torch.full((b,), time, dtype=torch.long), cond, sd, **unet_kw)
            if objective == "pred_x0":
                pred_noise = (buf["sqrt_recip_alphas_cumprod"][time] * img - out) / buf["sqrt_recipm1_alphas_cumprod"][time]
                x0 = out
            elif objective == "pred_noise":
                pred_noise = out
                x0 = buf["sqrt_recip_alphas_cumprod"][time] * img - buf["sqrt_recipm1_alphas_cumprod"][time] * out
            else:
                raise ValueError(f"unknown objective {objective}")
            if clip_denoised:
                x0 = x0.clamp(-1.0, 1.0)
            sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            c = ((1 - alpha_next) - sigma ** 2).sqrt()
            z = noises(time).to(dt) if time_next > 0 else 0.0
            img = x0 * alpha_next.sqrt() + c * pred_noise + sigma * z
    return (img + 1) * 0.5


def p_losses_recon(x_start01: Tensor, t: Tensor, cond: Optional[Tensor], noise: Tensor, sd, buf,
                   objective: str = "pred_x0", loss_type: str = "l1", **unet_kw):
    """GaussianDiffusion.forward -> p_losses, no box/mask (cond_DDPM.py:565-655): the single-step
    reconstruction the reference's test_step actually runs. Returns (loss, reco)."""
    x0 = x_start01 * 2 - 1
    x = q_sample(x0, t, noise, buf)
    with torch.no_grad():
        out = unet_forward(x, t, cond, sd, **unet_kw)
    target = noise if objective == "pred_noise" else x0
    loss = (out - target).abs() if loss_type == "l1" else (out - target) ** 2
    loss = loss.reshape(loss.shape[0], -1).mean(dim=1) * buf["p2_loss_weight"][t]
    if objective == "pred_noise":
        sh = (-1, 1, 1, 1)
        reco = (x - buf["sqrt_one_minus_alphas_cumprod"][t].reshape(sh) * out + 1) * 0.5
    else:
        reco = (out + 1) * 0.5
    return loss.mean(), reco


def to_torch_sd(sd_np: Dict[str, np.ndarray], dtype=torch.float32) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in sd_np.items()}


def to_float64(d: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """float64 copies of a state_dict / buffer dict: the same restatement then runs without fp32 rounding
    (yardstick for how far two fp32 implementations may legitimately differ)."""
    return {k: v.double() for k, v in d.items()}


def p_sample_loop_simplex(x_start: Tensor, cond: Optional[Tensor], sd, buf, seeds: Sequence[int], start_t: int,
                          objective: str = "pred_x0", **unet_kw) -> Tensor:
    """p_sample_loop, simplex branch (`noise is not None`; cond_DDPM.py:449-452, :441-443, :460-463):
    img = q_sample(x_start, t=[T], gen_noise())[:, 0:1]; each step t = T-1 .. 0 draws a NEW simplex field with
    gen_noise (float16 -> float32, the same field for every batch item; nothing added at t = 0); (img + 1) / 2.
    `seeds[0]` seeds the start field, `seeds[1 + k]` the field of the k-th step (t = T-1-k), i.e. the values
    Simplex_CLASS.newSeed would draw. T = start_t must index the schedule (1 <= T < num_timesteps)."""
    import simplex_oracle as SX
    T = int(start_t)
    B, C, H, W = x_start.shape
    field = torch.from_numpy(SX.gen_noise(seeds[0], (B, C, H, W))).float()
    img = (buf["sqrt_alphas_cumprod"][T] * x_start + buf["sqrt_one_minus_alphas_cumprod"][T] * field)[:, 0].unsqueeze(1)
    with torch.no_grad():
        for k, t in enumerate(reversed(range(T))):
            z = torch.from_numpy(SX.gen_noise(seeds[1 + k], (B, 1, H, W))).float() if t > 0 else None
            img = p_sample(img, t, cond, sd, buf, z, objective, **unet_kw)
    return (img + 1) * 0.5
