"""Glue between the host-side mirrors (OpenAI_Unet.UNetModel, cond_DDPM.GaussianDiffusion) and the HIP engine.

A UNetModel owns one HipBackend. The backend (re)creates the CddpmEngine when the device, the capacity
(batch / image size), the weights (load_state_dict, .to()) or the schedule change, so user code keeps the
reference's workflow: build modules, load a checkpoint, move to the GPU, call forward / p_sample_loop.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import schedule as _schedule
from .engine import CddpmEngine


class HipBackend:
    def __init__(self, unet):
        self._unet_desc = dict(
            model_channels=unet.model_channels, channel_mult=tuple(unet.channel_mult),
            num_res_blocks=unet.num_res_blocks, attention_resolutions=tuple(unet.attention_resolutions),
            head_channels=unet.num_head_channels, cond_dim=unet.num_classes or 0,
            in_channels=unet.in_channels, out_channels=unet.out_channels)
        self.timesteps = 1000
        self.objective = "pred_x0"
        self.buffers: Optional[Dict[str, torch.Tensor]] = None
        self.engine: Optional[CddpmEngine] = None
        self._key = None
        self._sched_key = None

    # the diffusion wrapper registers its schedule here; a bare UNetModel uses the default cosine one
    def set_schedule(self, buffers: Dict[str, torch.Tensor], objective: str):
        self.buffers = {k: v.detach().cpu() for k, v in buffers.items()}
        self.timesteps = int(self.buffers["betas"].shape[0])
        self.objective = objective
        self._sched_key = None

    @staticmethod
    def _weights_key(unet):
        return tuple((p.data_ptr(), p._version) for p in unet.parameters())

    def get(self, unet, B: int, H: int, W: int, device: torch.device) -> CddpmEngine:
        if device.type != "cuda":
            raise RuntimeError(f"the cDDPM HIP path runs on an MI355X only; tensors are on {device}. "
                               "There is no CPU fallback (use the reference, or oracle/ in tests).")
        wkey = self._weights_key(unet)
        e = self.engine
        need = (e is None or e.device != device or e.timesteps != self.timesteps or B > e.max_batch
                or H * W > e.max_h * e.max_w or H > e.max_h or W > e.max_w or wkey != self._key)
        if need:
            if e is not None:
                e.close()
            max_b = max(B, e.max_batch if e is not None else 1)
            max_h = max(H, e.max_h if e is not None else 0)
            max_w = max(W, e.max_w if e is not None else 0)
            e = CddpmEngine(timesteps=self.timesteps, max_batch=max_b, max_h=max_h, max_w=max_w, device=device,
                            **self._unet_desc)
            e.load_weights(unet.state_dict())
            self.engine, self._key, self._sched_key = e, wkey, None
        skey = (id(self.buffers), self.objective)
        if self._sched_key != skey:
            bufs = self.buffers if self.buffers is not None else _schedule.schedule_buffers(self.timesteps)
            e.set_schedule(bufs, self.objective)
            self._sched_key = skey
        return e

    def invalidate(self):
        """the parameters were updated outside torch's version counters (the training step's Adam kernel writes them in place): the next
        `get` re-packs the inference engine's weights"""
        self._key = None

    def close(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None
