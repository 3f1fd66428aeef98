"""Host-side mirror of the reference diffusion wrapper, `GaussianDiffusion`
(reference src/models/modules/cond_DDPM.py:289-655), running on the HIP engine.

Same constructor arguments, the same 13 registered schedule buffers (they are part of checkpoints), the same
method names and argument meaning for the path in scope:

    p_sample_loop(shape, cond, cond_scale, box, start_t, noise, x_start)   :446-464  -> cddpm_reverse
    p_sample(x, t, clip_denoised, cond, cond_scale, noise)                 :432-444  -> cddpm_p_sample
    sample(batch_size, cond, ...)                                          :517-530
    q_sample(x_start, t, noise)                                            :548-554  -> cddpm_q_sample
    forward(img, t, cond=, noise=) -> (loss, reco)   (single-step reconstruction, :565-655)
    model_predictions / p_mean_variance / q_posterior                      :391-430

Differences, on purpose:
  * no CPU path: tensors must live on the MI355X, otherwise a RuntimeError (never a silent fallback);
  * `use_spatial_transformer` is set to False here -- the reference reads it in model_predictions (:401) but
    never assigns it, so its own p_sample crashes; this is the intended semantics (experiment yaml :31);
  * the simplex branch (`noise is not None`, :441-443, :450-452) draws its fields on the device (csrc/simplex.hip,
    bit-exact with the reference's CPU generator for a given seed) instead of numba + host->device copies per step;
    `box` of p_sample_loop (:455-459) masks x_T as the reference's lines do; `sample(box=...)` raises NotImplementedError (the reference's `ddim_sample_box` does
    not exist, :527); DDIM (`ddim_sample`) is accelerated, with Gaussian or simplex per-step noise as `cfg.noisetype` says;
  * noise: x_T and z_t come from the counter RNG (synth.py / on-device Philox) seeded from torch's global
    generator, so `torch.manual_seed` still makes runs reproducible; pass `seed=`/`slice0=` to pin them.
"""
from __future__ import annotations

from collections import namedtuple

import torch
import torch.nn as nn

from . import schedule as _schedule
from . import synth as _synth
from .generate_noise import gen_noise

ModelPrediction = namedtuple("ModelPrediction", ["pred_noise", "pred_x_start"])


def normalize_to_neg_one_to_one(img):
    return img * 2 - 1


def unnormalize_to_zero_to_one(t):
    return (t + 1) * 0.5


def _extract(a, t, x_shape):
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))


def mask_x_T_to_box(img, box):
    """The `box` lines of the reference's p_sample_loop (cond_DDPM.py:455-459) AS WRITTEN: a zero tensor receives, sample by sample, the
    box [x0:x2) x [y1:y3) of `img` -- but `img = img_patch` sits inside that loop, so from the second sample on the copy reads the zero
    patch itself: sample 0 keeps its x_T inside its box, every other sample starts the chain from all zeros (pinned by a golden made from
    the reference: tests/golden/box_loop_B3_32x32_T1000_start6.npz). box: [B, 4] rows (x0, y1, x2, y3), tensor or nested list."""
    img_patch = torch.zeros_like(img)
    for i in range(img.shape[0]):
        x0, y1, x2, y3 = (int(v) for v in box[i])
        img_patch[i, :, y1:y3, x0:x2] = img[i, :, y1:y3, x0:x2]
        img = img_patch
    return img


class GaussianDiffusion(nn.Module):
    def __init__(self, model, *, image_size, channels=3, timesteps=1000, sampling_timesteps=None, loss_type="l1",
                 objective="pred_noise", beta_schedule="cosine", p2_loss_weight_gamma=0., p2_loss_weight_k=1,
                 ddim_sampling_eta=1., inpaint=False, cfg=None):
        super().__init__()
        assert objective in {"pred_noise", "pred_x0"}, \
            "objective must be either pred_noise (predict noise) or pred_x0 (predict image start)"
        self.cfg = cfg
        self.channels = channels
        self.image_size = image_size
        self.model = model
        self.objective = objective
        self.inpaint = inpaint
        self.loss_type = loss_type
        self.use_spatial_transformer = False
        bufs = _schedule.schedule_buffers(timesteps, beta_schedule, p2_loss_weight_gamma, p2_loss_weight_k)
        self.num_timesteps = int(bufs["betas"].shape[0])
        self.sampling_timesteps = timesteps if sampling_timesteps is None else sampling_timesteps
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        for name in _schedule.BUFFER_NAMES:
            self.register_buffer(name, bufs[name])
        self._sched_token = None

    # ---- engine plumbing --------------------------------------------------------------------------
    def _engine(self, B, H, W, device):
        """the UNet's packed HIP engine with THIS module's schedule installed"""
        token = tuple((getattr(self, n).data_ptr(), getattr(self, n)._version) for n in ("betas", "posterior_mean_coef1"))
        if token != self._sched_token:
            self.model._hip.set_schedule({n: getattr(self, n) for n in _schedule.BUFFER_NAMES}, self.objective)
            self._sched_token = token
        return self.model.hip_engine(B, H, W, device)

    @staticmethod
    def _draw_seed(seed):
        if seed is not None:
            return int(seed)
        return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())   # follows torch.manual_seed

    @property
    def loss_fn(self):
        if self.loss_type == "l1":
            return torch.nn.functional.l1_loss
        if self.loss_type == "l2":
            return torch.nn.functional.mse_loss
        raise ValueError(f"invalid loss type {self.loss_type}")

    # ---- posterior helpers (elementwise; kept for API completeness) -----------------------------------
    def predict_start_from_noise(self, x_t, t, noise):
        return (_extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - _extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise)

    def predict_noise_from_start(self, x_t, t, x0):
        return ((_extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - x0)
                / _extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape))

    def q_posterior(self, x_start, x_t, t):
        mean = (_extract(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + _extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return (mean, _extract(self.posterior_variance, t, x_t.shape),
                _extract(self.posterior_log_variance_clipped, t, x_t.shape))

    @torch.no_grad()
    def model_predictions(self, x, t, cond, cond_scale, clip_x_start=False):
        eng = self._engine(x.shape[0], x.shape[2], x.shape[3], x.device)
        out = eng.unet_forward(x, t, cond)
        clip = (lambda v: v.clamp(-1., 1.)) if clip_x_start else (lambda v: v)
        if self.objective == "pred_noise":
            return ModelPrediction(out, clip(self.predict_start_from_noise(x, t, out)))
        return ModelPrediction(self.predict_noise_from_start(x, t, out), clip(out))

    @torch.no_grad()
    def p_mean_variance(self, x, t, clip_denoised: bool, cond=None, cond_scale=1.):
        x_start = self.model_predictions(x, t, cond, cond_scale, clip_denoised).pred_x_start
        return self.q_posterior(x_start=x_start, x_t=x, t=t)

    # ---- the hot path -----------------------------------------------------------------------------------
    @torch.no_grad()
    def p_sample(self, x, t: int, clip_denoised=True, cond=None, cond_scale=1., noise=None, *, z=None, seed=None, slice0=0):
        """x_t -> x_{t-1} (cond_DDPM.py:432-444). `z`: this step's N(0,1) draw (else drawn on the device)."""
        B, _c, H, W = x.shape
        eng = self._engine(B, H, W, x.device)
        eng.set_clip_denoised(clip_denoised)
        if noise is not None:
            # reference :441-443: the passed tensor only selects the branch; a NEW simplex field is drawn for the step
            z = gen_noise(self.cfg, x.shape, engine=eng).float() if t > 0 else None
        return eng.p_sample(x.float(), int(t), cond.float() if cond is not None else None, z=z, seed=self._draw_seed(seed), slice0=slice0)

    @torch.no_grad()
    def p_sample_loop(self, shape, cond=None, cond_scale=1., box=None, start_t=0, noise=None, x_start=None, *,
                      x_T=None, z_noise=None, seed=None, slice0=0, device=None):
        """Full reverse loop (cond_DDPM.py:446-464): T = num_timesteps if start_t == 0 else start_t, x_T ~ N(0,1),
        T steps, result mapped to [0,1]. Extras: x_T / z_noise ([T,B,1,H,W], z_t at index t) inject given draws,
        seed / slice0 key the counter RNG (slice0 = global index of the first slice, for sharded runs)."""
        B, _c, H, W = shape
        dev = torch.device(device) if device is not None else (cond.device if cond is not None else self.betas.device)
        T = self.num_timesteps if start_t == 0 else int(start_t)
        eng = self._engine(B, H, W, dev)
        if noise is not None:
            # Simplex branch (:450-452, :441-443): start from q_sample(x_start, t=T, simplex field), then one fresh
            # simplex field per step (the same field for every batch item). T indexes the schedule directly, so
            # start_t must be in [1, num_timesteps - 1] (start_t == 0 reads index num_timesteps in the reference: an
            # out-of-range access there, an error here). Fields come from csrc/simplex.hip, seeds from numpy's RNG.
            if x_start is None:
                raise ValueError("the simplex branch of p_sample_loop needs x_start")
            if not (1 <= T < self.num_timesteps):
                raise ValueError(f"simplex branch: start_t must be in [1, {self.num_timesteps - 1}], got {start_t}")
            field = gen_noise(self.cfg, (B, _c, H, W), engine=eng).float()
            tT = torch.full((B,), T, device=dev, dtype=torch.long)
            img = (_extract(self.sqrt_alphas_cumprod, tT, x_start.shape) * x_start.float()
                   + _extract(self.sqrt_one_minus_alphas_cumprod, tT, x_start.shape) * field)[:, 0].unsqueeze(1).contiguous()
            eng.prepare_cond(cond.float() if cond is not None else None, B)
            eng.set_clip_denoised(True)
            for t in reversed(range(0, T)):
                z = gen_noise(self.cfg, (B, 1, H, W), engine=eng).float() if t > 0 else None
                img = eng.p_sample(img, t, None, z=z)
            return unnormalize_to_zero_to_one(img)
        eng.set_clip_denoised(True)                       # p_sample_loop calls p_sample with its default clip_denoised (:461)
        seed = self._draw_seed(seed)
        if x_T is None:
            x_T = eng.noise_fill(B, H, W, seed=seed, stream_id=_synth.STREAM_XT, slice0=slice0)
        if box is not None:
            x_T = mask_x_T_to_box(x_T, box)
        return eng.reverse(x_T, cond.float() if cond is not None else None, T, noise=z_noise, seed=seed, slice0=slice0)

    def ddim_time_pairs(self, start_t=0):
        """(time, time_next) pairs of ddim_sample (cond_DDPM.py:468-474): sampling_timesteps + 1 ints from
        linspace(0, T', sampling_timesteps + 2)[:-1], reversed; T' = start_t or num_timesteps."""
        total = self.num_timesteps if start_t == 0 else int(start_t)
        times = torch.linspace(0., total, steps=self.sampling_timesteps + 2)[:-1]
        times = list(reversed(times.int().tolist()))
        return list(zip(times[:-1], times[1:]))

    def ddim_coefficients(self, time, time_next, eta=None):
        """sqrt(alpha_next), c, sigma of one DDIM update in the reference's fp32 tensor arithmetic (:489-499)"""
        eta = self.ddim_sampling_eta if eta is None else eta
        acp = self.alphas_cumprod_prev.detach().float().cpu()
        alpha, alpha_next = acp[time], acp[time_next]
        sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = ((1 - alpha_next) - sigma ** 2).sqrt()
        return float(alpha_next.sqrt()), float(c), float(sigma)

    @torch.no_grad()
    def ddim_sample(self, shape, clip_denoised=True, cond=None, cond_scale=1., x_start=None, start_t=0, noise=None, *,
                    x_T=None, z_noise=None, seed=None, slice0=0, device=None):
        """DDIM sampling (cond_DDPM.py:466-515) over sampling_timesteps + 1 time pairs, one UNet call per pair.
        start_t != 0 starts from q_sample(x_start, t = start_t, noise) as the reference does (:482): `noise` is then the
        q_sample noise itself (a passed tensor is used as given; None draws N(0,1), i.e. x_T). The per-step z follows
        `cfg.noisetype` exactly like the reference (:501-504): 'simplex' draws a fresh simplex field per pair (device
        generator, generate_noise.py mirror), anything else N(0,1). `clip_denoised=False` skips the clamp of x0 (:493).
        Extras: x_T (the initial N(0,1) draw), z_noise (dict time -> [B,1,H,W] or a [num_timesteps,B,1,H,W] tensor indexed
        by `time`) inject given draws; otherwise the device Philox is keyed by (seed, time, slice0 + b)."""
        B, _c, H, W = shape
        dev = torch.device(device) if device is not None else (cond.device if cond is not None else self.betas.device)
        eng = self._engine(B, H, W, dev)
        eng.set_clip_denoised(clip_denoised)
        seed = self._draw_seed(seed)
        pairs = self.ddim_time_pairs(start_t)
        noisetype = None
        if self.cfg is not None:
            noisetype = self.cfg.get("noisetype", None) if hasattr(self.cfg, "get") else getattr(self.cfg, "noisetype", None)
        if noise is not None and noisetype == "simplex":
            gen_noise(self.cfg, shape, engine=eng)          # the reference draws (and discards) one field here (:477): same seed sequence
        if x_T is None:
            x_T = eng.noise_fill(B, H, W, seed=seed, stream_id=_synth.STREAM_XT, slice0=slice0)
        img = x_T.to(dev).float().contiguous().clone()
        if start_t != 0:
            if x_start is None:
                raise ValueError("ddim_sample with start_t != 0 needs x_start")
            tT = torch.full((B,), int(start_t), device=dev, dtype=torch.long)
            xs = x_start.to(dev).float()
            eps0 = img if noise is None else noise.to(dev).float()
            img = (_extract(self.sqrt_alphas_cumprod, tT, xs.shape) * xs
                   + _extract(self.sqrt_one_minus_alphas_cumprod, tT, xs.shape) * eps0)[:, 0].unsqueeze(1).contiguous()
        eng.prepare_cond(cond.float() if cond is not None else None, B)
        for i, (time, time_next) in enumerate(pairs):
            ca, c, sigma = self.ddim_coefficients(time, time_next)
            z = None
            if time_next > 0:
                if z_noise is not None:
                    z = z_noise[time]
                elif noisetype == "simplex":
                    z = gen_noise(self.cfg, shape, engine=eng)
            eng.ddim_step(img, time, ca, c, sigma, add_noise=time_next > 0, finalize=(i == len(pairs) - 1),
                          z=z.to(dev).float().contiguous() if z is not None else None, seed=seed, slice0=slice0)
        eng.set_clip_denoised(True)
        eng._check_finite(img, "ddim_sample")
        return img

    @torch.no_grad()
    def sample(self, batch_size=16, cond=None, cond_scale=1., box=None, x_start=None, start_t=0, noise=None, **kw):
        """(cond_DDPM.py:517-530) image_size may be an int or an (H, W) pair, as DDPM_2D passes it."""
        hw = self.image_size if isinstance(self.image_size, (tuple, list)) else (self.image_size, self.image_size)
        if box is not None:          # the reference routes EVERY box call of sample() to ddim_sample_box (:526-528), which does not exist
            raise NotImplementedError("sample(box=...): ddim_sample_box is undefined in the reference (cond_DDPM.py:527); "
                                      "p_sample_loop(box=...) is the branch that exists")
        if self.is_ddim_sampling:
            return self.ddim_sample((batch_size, self.channels, int(hw[0]), int(hw[1])), cond=cond, cond_scale=cond_scale,
                                    x_start=x_start, start_t=start_t, noise=noise, **kw)
        return self.p_sample_loop((batch_size, self.channels, int(hw[0]), int(hw[1])), cond=cond, cond_scale=cond_scale,
                                  box=box, start_t=start_t, noise=noise, x_start=x_start, **kw)

    @torch.no_grad()
    def interpolate(self, x1, x2, t=None, lam=0.5, *, cond=None, seed=None):
        """(cond_DDPM.py:532-546) both images noised to step t, mixed (1 - lam) : lam, then t reverse steps; returns the chain's last state
        in [-1, 1] like the reference (no unnormalise). The reference's own loop cannot run -- it hands p_sample a tensor for `t` (:544,
        SURVEY 8a "latent bugs") and no context -- so this follows the intended semantics; `cond` (absent in the reference's signature)
        is the context the conditioned UNet needs."""
        assert x1.shape == x2.shape
        b = x1.shape[0]
        t = self.num_timesteps - 1 if t is None else int(t)
        tb = torch.full((b,), t, device=x1.device, dtype=torch.long)
        img = ((1 - lam) * self.q_sample(x1, tb) + lam * self.q_sample(x2, tb)).contiguous()
        seed = self._draw_seed(seed)
        for i in reversed(range(0, t)):
            img = self.p_sample(img, i, cond=cond, seed=seed)
        return img

    @torch.no_grad()
    def q_sample(self, x_start, t, noise=None):
        """sqrt(abar_t) x0 + sqrt(1 - abar_t) eps on x0 in [-1,1] (cond_DDPM.py:548-554)."""
        if noise is None:
            noise = torch.randn_like(x_start)
        B, _c, H, W = x_start.shape
        eng = self._engine(B, H, W, x_start.device)
        # the kernel fuses the [0,1] -> [-1,1] map; feed it the inverse so callers keep the reference's convention
        return eng.q_sample((x_start.float() + 1) * 0.5, t, noise.float())

    @torch.no_grad()
    def p_losses(self, x_start, t, cond=None, noise=None, box=None, scale_patch=1, onlybox=False, mask=None):
        """single-step reconstruction (cond_DDPM.py:565-645), x_start in [-1,1]; returns (loss, reco in [0,1]). `box` ([B,4] rows
        (x0, y1, x2, y3)): only the box of each slice is noised -- the UNet sees x_start with q_sample's box pasted in (:592-598) -- and
        under pred_noise the target is the noise inside the box, zero outside (:611-615); `inpaint` (constructor flag) pastes the model's
        box back into x_start before the loss (:626-633). `scale_patch` / `onlybox` are accepted and unused, as in the reference."""
        if self.inpaint and box is None:
            raise ValueError("inpaint=True needs a box (the reference indexes box[i] unconditionally, cond_DDPM.py:632)")
        if noise is None:
            noise = torch.randn_like(x_start)
        B, _c, H, W = x_start.shape
        eng = self._engine(B, H, W, x_start.device)
        x = eng.q_sample((x_start.float() + 1) * 0.5, t, noise.float())
        boxes = None
        if box is not None:
            boxes = [tuple(int(v) for v in box[i]) for i in range(B)]
            xb = x_start.float().clone()
            for i, (x0, y1, x2, y3) in enumerate(boxes):
                xb[i, :, y1:y3, x0:x2] = x[i, :, y1:y3, x0:x2]
            x = xb.contiguous()
        out = eng.unet_forward(x, t, cond.float() if cond is not None else None)
        if self.objective == "pred_noise":
            if boxes is not None:
                target = torch.zeros_like(noise)
                for i, (x0, y1, x2, y3) in enumerate(boxes):
                    target[i, :, y1:y3, x0:x2] = noise[i, :, y1:y3, x0:x2]
            else:
                target = noise
        else:
            if mask is not None:
                out = out * mask
            target = x_start
        if self.inpaint:
            pasted = x_start.float().clone()
            if pasted.shape[1] == 2:
                pasted = pasted[:, 0].unsqueeze(1)
            for i, (x0, y1, x2, y3) in enumerate(boxes):
                pasted[i, :, y1:y3, x0:x2] = out[i, :, y1:y3, x0:x2]
            out = pasted
        loss = self.loss_fn(out, target, reduction="none").reshape(B, -1).mean(dim=1)
        loss = loss * self.p2_loss_weight.gather(-1, t.long())
        if self.objective == "pred_noise":
            reco = x - _extract(self.sqrt_one_minus_alphas_cumprod, t.long(), x.shape) * out
        else:
            reco = out
        return loss.mean(), unnormalize_to_zero_to_one(reco)

    def forward(self, img, t=None, *args, **kwargs):
        """(cond_DDPM.py:647-655) img in [0,1]; t: scalar timestep for the whole batch or None for random."""
        b = img.shape[0]
        if t is None:
            t = torch.randint(0, self.num_timesteps, (b,), device=img.device).long()
        else:
            t = (torch.ones([b], device=img.device) * t).long()
        return self.p_losses(normalize_to_neg_one_to_one(img), t, *args, **kwargs)
