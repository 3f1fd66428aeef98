"""Thin Python owner of one libcddpm_hip handle: PyTorch-ROCm tensors in, raw pointers out.

PyTorch is used for device memory, the current HIP stream and (elsewhere) torch.distributed only;
every FLOP of the path runs in csrc/*.hip through the C ABI of include/cddpm.h.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional, Sequence

import numpy as np
import torch

from . import _lib

OBJECTIVES = {"pred_x0": 0, "pred_noise": 1}


def _stream_ptr(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def _check_dev(t: torch.Tensor, name: str, device) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a tensor on a HIP device (got {type(t).__name__} on "
                           f"{getattr(t, 'device', None)}); the HIP path has no CPU fallback")
    if t.device != device:
        raise RuntimeError(f"{name} lives on {t.device}, the engine on {device}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype}); the path computes in fp32 only")
    return t.contiguous()


class CddpmEngine:
    """One device's packed UNet + schedule tables + workspace (cddpm_create .. cddpm_destroy)."""

    def __init__(self, *, model_channels=128, channel_mult=(1, 2, 2), num_res_blocks=3,
                 attention_resolutions=(3, 6, 12), head_channels=64, cond_dim=128, timesteps=1000,
                 max_batch=1, max_h=128, max_w=128, device=None, in_channels=1, out_channels=1):
        self.lib = _lib.load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the cDDPM HIP path needs an MI355X (gfx950); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        d = _lib.UnetDesc()
        d.in_channels, d.out_channels, d.model_channels = in_channels, out_channels, model_channels
        d.num_levels = len(channel_mult)
        for i, m in enumerate(channel_mult):
            d.channel_mult[i] = int(m)
        d.num_res_blocks = num_res_blocks
        d.num_attention_resolutions = len(attention_resolutions)
        for i, a in enumerate(attention_resolutions):
            d.attention_resolutions[i] = int(a)
        d.head_channels, d.cond_dim, d.timesteps = head_channels, int(cond_dim or 0), timesteps
        d.max_batch, d.max_h, d.max_w = max_batch, max_h, max_w
        self.desc = d
        self.timesteps = timesteps
        self.cond_dim = int(cond_dim or 0)
        self.max_batch, self.max_h, self.max_w = max_batch, max_h, max_w
        self._h = C.c_void_p()
        rc = self.lib.cddpm_create(C.byref(self._h), C.byref(d), self.device.index)
        if rc != 0:
            msg = self.lib.cddpm_last_error(None).decode()
            self._h = None
            raise RuntimeError(f"cddpm_create failed: {msg}")
        self._keep = []   # tensors whose pointers the library may still read (taps)

    # ------------------------------------------------------------------ lifetime / errors
    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            self.lib.cddpm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.cddpm_last_error(self._h).decode()}")

    # ------------------------------------------------------------------ setup
    def weight_names(self):
        n = self.lib.cddpm_num_weights(self._h)
        return [(self.lib.cddpm_weight_name(self._h, i).decode(), int(self.lib.cddpm_weight_numel(self._h, i)))
                for i in range(n)]

    def load_weights(self, state_dict: Mapping[str, object], prefix: str = ""):
        """state_dict: reference names (optionally under `prefix`, e.g. 'diffusion.model.') -> tensor/ndarray."""
        names, arrs = [], []
        for name, _numel in self.weight_names():
            key = prefix + name
            if key not in state_dict:
                raise KeyError(f"state_dict has no '{key}'")
            v = state_dict[key]
            if isinstance(v, torch.Tensor):
                v = v.detach().to("cpu", torch.float32).contiguous().numpy()
            v = np.ascontiguousarray(v, dtype=np.float32)
            names.append(name.encode())
            arrs.append(v)
        n = len(names)
        c_names = (C.c_char_p * n)(*names)
        c_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        c_numels = (C.c_int64 * n)(*[a.size for a in arrs])
        self._ck(self.lib.cddpm_load_weights(self._h, c_names, c_ptrs, c_numels, n), "cddpm_load_weights")

    def set_schedule(self, buffers: Mapping[str, object], objective: str = "pred_x0"):
        """buffers: the GaussianDiffusion schedule buffers (host tensors/arrays of length T)."""
        def host(name):
            v = buffers[name]
            if isinstance(v, torch.Tensor):
                v = v.detach().to("cpu", torch.float32).numpy()
            return np.ascontiguousarray(v, dtype=np.float32)
        arrs = [host(k) for k in ("posterior_mean_coef1", "posterior_mean_coef2", "posterior_log_variance_clipped",
                                  "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod")]
        self._qs = (host("sqrt_alphas_cumprod"), host("sqrt_one_minus_alphas_cumprod"))
        T = arrs[0].shape[0]
        self._ck(self.lib.cddpm_set_schedule(self._h, *[a.ctypes.data for a in arrs], T, OBJECTIVES[objective]),
                 "cddpm_set_schedule")

    def prepare_cond(self, cond: Optional[torch.Tensor], B: int):
        if self.cond_dim > 0:
            cond = _check_dev(cond, "cond", self.device)
            if tuple(cond.shape) != (B, self.cond_dim):
                raise RuntimeError(f"cond must be [{B}, {self.cond_dim}], got {tuple(cond.shape)}")
            ptr = cond.data_ptr()
        else:
            ptr = None
        self._ck(self.lib.cddpm_prepare_cond(self._h, ptr, B, _stream_ptr(self.device)), "cddpm_prepare_cond")
        self._cond_keep = cond

    # ------------------------------------------------------------------ the path
    def unet_forward(self, x: torch.Tensor, t, cond: Optional[torch.Tensor] = None) -> torch.Tensor:
        """UNetModel.forward: x [B,1,H,W] fp32 on device, t int or int tensor [B]; cond [B,cond_dim] or None
        to reuse the context of the previous prepare_cond."""
        x = _check_dev(x, "x", self.device)
        B, c, H, W = x.shape
        if c != 1:
            raise RuntimeError("x must be [B,1,H,W]")
        if cond is not None or self.cond_dim == 0:
            self.prepare_cond(cond, B)
        out = torch.empty_like(x)
        if isinstance(t, torch.Tensor):
            tt = self._t_tensor(t, B)
            rc = self.lib.cddpm_unet_forward(self._h, x.data_ptr(), tt.data_ptr(), 0, out.data_ptr(), B, H, W,
                                             _stream_ptr(self.device))
        else:
            rc = self.lib.cddpm_unet_forward(self._h, x.data_ptr(), None, int(t), out.data_ptr(), B, H, W,
                                             _stream_ptr(self.device))
        self._ck(rc, "cddpm_unet_forward")
        return out

    def _t_tensor(self, t: torch.Tensor, B: int) -> torch.Tensor:
        """per-sample timesteps index the device tables (time-embedding table, schedule buffers): range-checked here
        like the reference's `extract` would fail on an out-of-range gather (cond_DDPM.py:266-269)"""
        if t.numel() != B:
            raise RuntimeError("t must have B elements")
        lo, hi = int(t.min()), int(t.max())
        if lo < 0 or hi >= self.timesteps:
            raise IndexError(f"timestep indices must lie in [0, {self.timesteps}), got [{lo}, {hi}]")
        return t.to(self.device, torch.int32).contiguous()

    def reverse(self, x_T: torch.Tensor, cond: Optional[torch.Tensor], t_start: int, *, noise: Optional[torch.Tensor] = None,
                seed: int = 0, slice0: int = 0) -> torch.Tensor:
        """p_sample_loop from x_T: steps t_start-1 .. 0, returns the reconstruction in [0,1] (new tensor).
        noise: [t_start, B, 1, H, W] with z_t at index t (index 0 unused), or None for the device Philox."""
        x = _check_dev(x_T, "x_T", self.device).clone()
        B, c, H, W = x.shape
        if c != 1:
            raise RuntimeError("x_T must be [B,1,H,W]")
        self.prepare_cond(cond, B)
        nptr = None
        if noise is not None:
            noise = _check_dev(noise, "noise", self.device)
            if noise.numel() != t_start * B * H * W:
                raise RuntimeError(f"noise must hold t_start*B*H*W = {t_start * B * H * W} values, got {noise.numel()}")
            nptr = noise.data_ptr()
        self._ck(self.lib.cddpm_reverse(self._h, x.data_ptr(), nptr, seed, slice0, t_start, B, H, W,
                                        _stream_ptr(self.device)), "cddpm_reverse")
        self._check_finite(x, "cddpm_reverse")
        return x

    def reverse_two_streams(self, twin: "CddpmEngine", x_T: torch.Tensor, cond: Optional[torch.Tensor], t_start: int, *, seed: int = 0,
                            slice0: int = 0) -> torch.Tensor:
        """`reverse` for a SMALL batch as two half-batches on two streams: this engine runs the first half, `twin` (an engine created
        with the same geometry and loaded with the same weights and schedule: the same kernel plan, hence the same bits) the second,
        enqueued step by step in alternation, so that one half's small launches (GroupNorm finalizes, split-K combines, attention, the
        posterior step) run behind the other half's convolutions. Same result as `reverse`, bit for bit (slices are independent; noise is
        keyed by the global slice index); measured +4.3 % at B = 4, +2.4 % at B = 2, nothing from B = 8 on (tools/dual_stream_reverse.py).
        Device Philox noise only."""
        x = _check_dev(x_T, "x_T", self.device).clone()
        B, c, H, W = x.shape
        if c != 1 or B < 2:
            raise RuntimeError("reverse_two_streams needs x_T [B,1,H,W] with B >= 2")
        if (twin.max_batch, twin.max_h, twin.max_w, twin.timesteps) != (self.max_batch, self.max_h, self.max_w, self.timesteps):
            raise RuntimeError("the twin engine must be created with the same geometry (the kernel plan is part of a slice's bits)")
        h = (B + 1) // 2
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_two_streams", None) is None:
            self._two_streams = (torch.cuda.Stream(self.device), torch.cuda.Stream(self.device))
        halves = [(self, self._two_streams[0], x[:h], None if cond is None else cond[:h].contiguous(), slice0),
                  (twin, self._two_streams[1], x[h:], None if cond is None else cond[h:].contiguous(), slice0 + h)]
        for e, s, xs, cs, _s0 in halves:
            s.wait_stream(main)
            with torch.cuda.stream(s):
                e.prepare_cond(cs, xs.shape[0])
        for t in range(t_start - 1, -1, -1):
            for e, s, xs, _cs, s0 in halves:
                with torch.cuda.stream(s):
                    e.reverse_range_(xs, t, t, seed=seed, slice0=s0)
        for _e, s, _xs, _cs, _s0 in halves:
            main.wait_stream(s)
        self._check_finite(x, "reverse_two_streams")
        return x

    def reverse_range_(self, x: torch.Tensor, t_hi: int, t_lo: int, *, noise: Optional[torch.Tensor] = None, seed: int = 0,
                       slice0: int = 0) -> torch.Tensor:
        """steps t_hi .. t_lo of the chain IN PLACE on a device tensor (context from the last prepare_cond); the result is
        mapped to [0,1] exactly when t_lo == 0 (cddpm_reverse_range). No host synchronisation: the caller checks the
        final reconstruction (`check_finite`)."""
        x = _check_dev(x, "x", self.device)
        B, _c, H, W = x.shape
        nptr = _check_dev(noise, "noise", self.device).data_ptr() if noise is not None else None
        self._ck(self.lib.cddpm_reverse_range(self._h, x.data_ptr(), nptr, seed, slice0, int(t_hi), int(t_lo), B, H, W,
                                              _stream_ptr(self.device)), "cddpm_reverse_range")
        return x

    def check_finite(self, x: torch.Tensor, what: str = "reconstruction"):
        self._check_finite(x, what)

    @staticmethod
    def _check_finite(x: torch.Tensor, what: str):
        """A non-finite reconstruction is an error, not a result. The default convolution family carries fp32 products on
        the fp16 matrix pipe and needs |activation| < 65504; a model that exceeds it overflows to
        inf/NaN, which the posterior step propagates like torch.clamp does. One reduction + sync per reconstruction."""
        if not bool(torch.isfinite(x).all().item()):
            raise FloatingPointError(f"{what}: non-finite values in the result. If the weights are sane, an activation left the "
                                     "fp16 range of the default convolution family: rerun with CDDPM_CONV=x6 (exact bf16 split, "
                                     "no range limit) or CDDPM_CONV=f32.")

    def p_sample(self, x: torch.Tensor, t: int, cond: Optional[torch.Tensor] = None, *, z: Optional[torch.Tensor] = None,
                 seed: int = 0, slice0: int = 0) -> torch.Tensor:
        """one reverse step x_t -> x_{t-1} (values stay in [-1,1]); cond None reuses the prepared context."""
        x = _check_dev(x, "x", self.device).clone()
        B, _c, H, W = x.shape
        if cond is not None or self.cond_dim == 0:
            self.prepare_cond(cond, B)
        zp = _check_dev(z, "z", self.device).data_ptr() if z is not None else None
        self._ck(self.lib.cddpm_p_sample(self._h, x.data_ptr(), zp, seed, slice0, int(t), B, H, W,
                                         _stream_ptr(self.device)), "cddpm_p_sample")
        return x

    def ddim_step(self, x: torch.Tensor, t: int, coef_x0: float, coef_eps: float, sigma: float, *, add_noise: bool,
                  finalize: bool = False, z: Optional[torch.Tensor] = None, seed: int = 0, slice0: int = 0) -> torch.Tensor:
        """one DDIM update in place on a device tensor (context from the last prepare_cond); see cddpm_ddim_step"""
        x = _check_dev(x, "x", self.device)
        B, _c, H, W = x.shape
        zp = _check_dev(z, "z", self.device).data_ptr() if z is not None else None
        self._ck(self.lib.cddpm_ddim_step(self._h, x.data_ptr(), zp, seed, slice0, int(t), float(coef_x0), float(coef_eps),
                                          float(sigma), int(bool(add_noise)), int(bool(finalize)), B, H, W,
                                          _stream_ptr(self.device)), "cddpm_ddim_step")
        return x

    def p_sample_(self, x: torch.Tensor, t: int, *, seed: int = 0, slice0: int = 0) -> torch.Tensor:
        """in-place reverse step on a device tensor, context from the last prepare_cond (bench loop)"""
        x = _check_dev(x, "x", self.device)
        B, _c, H, W = x.shape
        self._ck(self.lib.cddpm_p_sample(self._h, x.data_ptr(), None, seed, slice0, int(t), B, H, W,
                                         _stream_ptr(self.device)), "cddpm_p_sample")
        return x

    def set_accumulation_switch(self, t_switch: int):
        """reverse steps t >= t_switch use the faster two-level-accumulation convolution plan (include/cddpm.h); default 200"""
        self._ck(self.lib.cddpm_set_accumulation_switch(self._h, int(t_switch)), "cddpm_set_accumulation_switch")

    def set_clip_denoised(self, on: bool):
        """clip_denoised of p_sample / ddim_sample (cond_DDPM.py:433, :467) for every later step on this engine"""
        self._ck(self.lib.cddpm_set_clip_denoised(self._h, int(bool(on))), "cddpm_set_clip_denoised")

    # 0-4: the reconstruction path; 5-8: the training operators (weight gradients, GroupNorm backward, the context encoder, Adam + re-packing)
    PROF_CLASSES = ("conv3x3_mfma", "conv1x1_mfma", "attention", "groupnorm", "other", "wgrad", "groupnorm_backward", "encoder", "optimizer")

    def set_profiling(self, on: bool):
        self._ck(self.lib.cddpm_set_profiling(self._h, int(on)), "cddpm_set_profiling")

    def get_profile(self):
        n = len(self.PROF_CLASSES)
        ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        ln = (C.c_int64 * n)()
        self._ck(self.lib.cddpm_get_profile(self._h, n, ms, fl, by, ln), "cddpm_get_profile")
        return {name: dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=int(ln[i])) for i, name in enumerate(self.PROF_CLASSES)}

    def noise_fill(self, B: int, H: int, W: int, *, seed: int, stream_id: int, t: int = 0, slice0: int = 0) -> torch.Tensor:
        out = torch.empty((B, 1, H, W), dtype=torch.float32, device=self.device)
        self._ck(self.lib.cddpm_noise_fill(self._h, out.data_ptr(), seed, stream_id, t, slice0, B, H, W,
                                           _stream_ptr(self.device)), "cddpm_noise_fill")
        return out

    def residual_postprocess(self, orig: torch.Tensor, recon: Optional[torch.Tensor], mask: Optional[torch.Tensor] = None, *,
                             squared: bool = False, erode_iterations: int = 0, median_k: int = 0) -> torch.Tensor:
        """Residual map of a volume [S,H,W] on the device: |orig - recon| (or squared; recon=None: `orig` as it is), times
        the eroded brain mask, 3-D median filtered -- the CPU/scipy part of the reference's _test_step
        (utils_eval.py:29-33, :64-71), bit-exact."""
        orig = _check_dev(orig, "orig", self.device)
        if recon is not None:
            recon = _check_dev(recon, "recon", self.device)
        if orig.dim() != 3 or (recon is not None and recon.shape != orig.shape):
            raise RuntimeError("orig / recon must be [S,H,W] volumes of the same shape")
        mptr = None
        if mask is not None:
            mask = _check_dev(mask, "mask", self.device)
            if mask.shape != orig.shape:
                raise RuntimeError("mask must have the shape of the volume")
            mptr = mask.data_ptr()
        S, H, W = orig.shape
        out = torch.empty_like(orig)
        tmp = torch.empty_like(orig) if median_k else None
        self._ck(self.lib.cddpm_residual_postprocess(self._h, orig.data_ptr(), recon.data_ptr() if recon is not None else None,
                                                     mptr, S, H, W, int(squared),
                                                     int(erode_iterations), int(median_k),
                                                     tmp.data_ptr() if tmp is not None else None, out.data_ptr(),
                                                     _stream_ptr(self.device)), "cddpm_residual_postprocess")
        return out

    def simplex_noise(self, B: int, H: int, W: int, *, seed: int, octaves: int = 6, persistence: float = 0.8,
                      frequency: float = 64.0) -> torch.Tensor:
        """gen_noise for noisetype 'simplex': float16 [B,1,H,W], the same field for every batch item, bit-exact
        with the reference's CPU generator for the given newSeed value."""
        out = torch.empty((B, 1, H, W), dtype=torch.float16, device=self.device)
        self._ck(self.lib.cddpm_simplex_fill(self._h, out.data_ptr(), int(seed), B, H, W, octaves, float(persistence),
                                             float(frequency), _stream_ptr(self.device)), "cddpm_simplex_fill")
        return out

    def q_sample(self, x01: torch.Tensor, t, noise: torch.Tensor) -> torch.Tensor:
        x01 = _check_dev(x01, "x01", self.device)
        noise = _check_dev(noise, "noise", self.device)
        B, _c, H, W = x01.shape
        out = torch.empty_like(x01)
        sa, s1 = self._qs
        if isinstance(t, torch.Tensor):
            tt = self._t_tensor(t, B)
            tp, tu = tt.data_ptr(), 0
        else:
            tt, tp, tu = None, None, int(t)
        self._ck(self.lib.cddpm_q_sample(self._h, x01.data_ptr(), noise.data_ptr(), tp, tu, sa.ctypes.data, s1.ctypes.data,
                                         sa.shape[0], out.data_ptr(), B, H, W, _stream_ptr(self.device)), "cddpm_q_sample")
        torch.cuda.current_stream(self.device).synchronize()   # host tables were read asynchronously
        return out

    # ------------------------------------------------------------------ test surface
    def block_names(self):
        return [self.lib.cddpm_block_name(self._h, i).decode() for i in range(self.lib.cddpm_num_blocks(self._h))]

    def forward_with_taps(self, x: torch.Tensor, t, cond) -> Dict[str, torch.Tensor]:
        """forward that also returns every block output as NCHW tensors (tests only)."""
        x = _check_dev(x, "x", self.device)
        B, _c, H, W = x.shape
        names = self.block_names()
        bufs = {}
        for i, n in enumerate(names):
            if n == "out":
                continue
            cc, hh, ww = C.c_int(), C.c_int(), C.c_int()
            self._ck(self.lib.cddpm_block_shape(self._h, i, H, W, C.byref(cc), C.byref(hh), C.byref(ww)), "cddpm_block_shape")
            t_ = torch.empty((B, hh.value, ww.value, cc.value), dtype=torch.float32, device=self.device)
            self._ck(self.lib.cddpm_set_tap(self._h, i, t_.data_ptr()), "cddpm_set_tap")
            bufs[n] = t_
        try:
            out = self.unet_forward(x, t, cond)
            torch.cuda.synchronize(self.device)
        finally:
            for i in range(len(names)):
                self.lib.cddpm_set_tap(self._h, i, None)
        res = {n: v.permute(0, 3, 1, 2).contiguous() for n, v in bufs.items()}
        res["out"] = out
        return res

    def op_conv(self, src0, src1, coef, silu, upsample, weight, bias, res, res_upsample, ksize):
        """fused conv on NHWC device tensors (see cddpm_op_conv); weight [Cout,Cin,k,k] host/any tensor."""
        B, h, w, C0 = src0.shape
        H, W = (2 * h, 2 * w) if upsample else (h, w)     # upsample: 0 none, 1 gather form, 2 folded form
        C1 = src1.shape[-1] if src1 is not None else 0
        wt = np.ascontiguousarray(weight.detach().cpu().numpy(), dtype=np.float32)
        bs = np.ascontiguousarray(bias.detach().cpu().numpy(), dtype=np.float32)
        Cout = wt.shape[0]
        out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=self.device)
        self._ck(self.lib.cddpm_op_conv(
            self._h, src0.data_ptr(), C0, src1.data_ptr() if src1 is not None else None, C1,
            coef.data_ptr() if coef is not None else None, int(silu), int(upsample), wt.ctypes.data, bs.ctypes.data,
            Cout, ksize, res.data_ptr() if res is not None else None, int(res_upsample), out.data_ptr(), B, H, W,
            _stream_ptr(self.device)), "cddpm_op_conv")
        return out

    def op_conv_skip(self, src0, coef, silu, weight, bias, skip, wskip):
        """conv3x3(act(src0)) + conv1x1(skip) + bias on NHWC device tensors (cddpm_op_conv_skip)"""
        B, H, W, C0 = src0.shape
        f = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
        wt, bs, ws = f(weight), f(bias), f(wskip)
        Cout = wt.shape[0]
        out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=self.device)
        self._ck(self.lib.cddpm_op_conv_skip(self._h, src0.data_ptr(), C0, coef.data_ptr() if coef is not None else None, int(silu),
                                             wt.ctypes.data, bs.ctypes.data, Cout, skip.data_ptr(), skip.shape[-1], ws.ctypes.data,
                                             out.data_ptr(), B, H, W, _stream_ptr(self.device)), "cddpm_op_conv_skip")
        return out

    def op_conv_gn(self, src0, weight, bias, gamma, beta):
        """conv3x3(src0) + bias with the epilogue's GroupNorm statistics -> (out NHWC, coef [3,B,Cout]) (cddpm_op_conv_gn)"""
        B, H, W, C0 = src0.shape
        f = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
        wt, bs, g, b = f(weight), f(bias), f(gamma), f(beta)
        Cout = wt.shape[0]
        out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=self.device)
        coef = torch.empty((3, B, Cout), dtype=torch.float32, device=self.device)
        self._ck(self.lib.cddpm_op_conv_gn(self._h, src0.data_ptr(), C0, wt.ctypes.data, bs.ctypes.data, Cout, g.ctypes.data,
                                           b.ctypes.data, out.data_ptr(), coef.data_ptr(), B, H, W, _stream_ptr(self.device)),
                 "cddpm_op_conv_gn")
        return out, coef

    # ---- training pieces (row f4, kernel level only) ----------------------------------------------------------------
    def op_conv_dgrad(self, dy, weight):
        """dL/d(input) of Conv2d(k, padding k // 2): dy NHWC [B,H,W,Cout] on the device, weight the FORWARD tensor [Cout,Cin,k,k]"""
        B, H, W, Cout = dy.shape
        wt = np.ascontiguousarray(weight.detach().cpu().numpy(), dtype=np.float32)
        Cin, k = wt.shape[1], wt.shape[2]
        dx = torch.empty((B, H, W, Cin), dtype=torch.float32, device=self.device)
        self._ck(self.lib.cddpm_op_conv_dgrad(self._h, dy.data_ptr(), Cout, wt.ctypes.data, Cin, k, dx.data_ptr(), B, H, W,
                                              _stream_ptr(self.device)), "cddpm_op_conv_dgrad")
        return dx

    def op_conv_wgrad(self, x0, x1, coef, silu, dy, ksize=3, upsample=False):
        """dL/dW [Cout,Cin,k,k] and dL/db [Cout] of y = conv_k(act(cat[x0, x1])) for dy NHWC [B,H,W,Cout]; x0 / x1 NHWC, coef [3,B,Cin] or None"""
        B, H, W, C0 = x0.shape
        C1 = x1.shape[-1] if x1 is not None else 0
        Cout = dy.shape[-1]
        dw = torch.empty((Cout, C0 + C1, ksize, ksize), dtype=torch.float32, device=self.device)
        db = torch.empty((Cout,), dtype=torch.float32, device=self.device)
        self._ck(self.lib.cddpm_op_conv_wgrad(self._h, x0.data_ptr(), C0, x1.data_ptr() if x1 is not None else None, C1,
                                              coef.data_ptr() if coef is not None else None, int(bool(silu)), int(bool(upsample)),
                                              dy.data_ptr(), Cout, ksize, dw.data_ptr(), db.data_ptr(), dy.shape[0], dy.shape[1], dy.shape[2],
                                              _stream_ptr(self.device)), "cddpm_op_conv_wgrad")
        return dw, db

    def op_attention_backward(self, qkv, da):
        """dL/dqkv [B,N,3C] of the attention core for da = dL/d(output) [B,N,C]"""
        B, N, C3 = qkv.shape
        dqkv = torch.empty_like(qkv)
        self._ck(self.lib.cddpm_op_attention_backward(self._h, qkv.data_ptr(), da.data_ptr(), dqkv.data_ptr(), B, N, C3 // 3,
                                                      _stream_ptr(self.device)), "cddpm_op_attention_backward")
        return dqkv

    def op_linear_backward(self, x, w, dy, silu_in=False):
        """backward of y = [SiLU](x) W^T + b: x [M,K], w [N,K], dy [M,N] on the device -> (dW, db, dx)"""
        M, K = x.shape
        N = w.shape[0]
        dw, db, dx = torch.empty_like(w), torch.empty((N,), dtype=torch.float32, device=self.device), torch.empty_like(x)
        self._ck(self.lib.cddpm_op_linear_backward(self._h, x.data_ptr(), w.data_ptr(), dy.data_ptr(), M, N, K, int(bool(silu_in)),
                                                   dw.data_ptr(), db.data_ptr(), dx.data_ptr(), _stream_ptr(self.device)),
                 "cddpm_op_linear_backward")
        return dw, db, dx

    def op_gn_silu_backward(self, x, da, gamma, beta, film, silu=True):
        """backward of act(GroupNorm32(x) * (1 + scale) + shift): x, da NHWC [B,H,W,C] -> (dx, dgamma, dbeta, dfilm or None)"""
        B, H, W, C = x.shape
        f = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
        g, b = f(gamma), f(beta)
        dx = torch.empty_like(x)
        dg = torch.empty((C,), dtype=torch.float32, device=self.device)
        db = torch.empty((C,), dtype=torch.float32, device=self.device)
        dfilm = torch.empty((B, 2 * C), dtype=torch.float32, device=self.device) if film is not None else None
        self._ck(self.lib.cddpm_op_gn_silu_backward(self._h, x.data_ptr(), None, 0, da.data_ptr(), g.ctypes.data, b.ctypes.data,
                                                    film.data_ptr() if film is not None else None, int(bool(silu)), dx.data_ptr(), None,
                                                    dg.data_ptr(), db.data_ptr(), dfilm.data_ptr() if dfilm is not None else None, None, 0, None,
                                                    B, H * W, C, _stream_ptr(self.device)), "cddpm_op_gn_silu_backward")
        return dx, dg, db, dfilm

    def op_gn_coef(self, src0, src1, gamma, beta, film):
        B = src0.shape[0]
        HW = src0.shape[1] * src0.shape[2]
        C0 = src0.shape[-1]
        C1 = src1.shape[-1] if src1 is not None else 0
        g = np.ascontiguousarray(gamma.detach().cpu().numpy(), dtype=np.float32)
        b = np.ascontiguousarray(beta.detach().cpu().numpy(), dtype=np.float32)
        coef = torch.empty((3, B, C0 + C1), dtype=torch.float32, device=self.device)
        self._ck(self.lib.cddpm_op_gn_coef(self._h, src0.data_ptr(), C0, src1.data_ptr() if src1 is not None else None, C1,
                                           g.ctypes.data, b.ctypes.data, film.data_ptr() if film is not None else None,
                                           coef.data_ptr(), B, HW, _stream_ptr(self.device)), "cddpm_op_gn_coef")
        return coef

    def op_attention(self, qkv):
        B, N, C3 = qkv.shape
        out = torch.empty((B, N, C3 // 3), dtype=torch.float32, device=self.device)
        self._ck(self.lib.cddpm_op_attention(self._h, qkv.data_ptr(), out.data_ptr(), B, N, C3 // 3,
                                             _stream_ptr(self.device)), "cddpm_op_attention")
        return out
