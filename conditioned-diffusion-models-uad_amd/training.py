"""Host side of the TRAINING step on the HIP kernels: forward with saved activations, loss, backward,
Adam -- the reference's `DDPM_2D.training_step` -> `GaussianDiffusion.forward` -> `p_losses` with gradients
(reference src/models/DDPM_2D.py:114-135, :305-306; src/models/modules/cond_DDPM.py:565-655; src/models/modules/OpenAI_Unet.py:823-1006).

The reference's training step is Python sequencing torch operators and autograd; this module is Python sequencing the HIP
operators of libcddpm_hip.so through their C-ABI entry points (include/cddpm.h, "training step" section): every FLOP of the forward
and of the backward runs in csrc/*.hip; torch is used for device memory, views, torch.cat of saved tensors and torch.distributed
(the gradient all-reduce). fp32 throughout (the gradients carry fp32 accuracy and are checked against autograd on the oracle).

State of this round: UNet forward / backward / Adam are complete for the reference's conditioned configuration and checked against
autograd (tests/test_gpu_training.py). Not done: the per-call host re-packing of the convolution weights (a step at 128x128 is
dominated by it: the packers of cddpm_op_conv* run on the CPU), bf16 autocast (BASELINE config 5), the encoder's backward.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import numpy as np
import torch

from . import schedule as _schedule
from .engine import CddpmEngine, _stream_ptr


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class UNetTrainer:
    """Forward + backward of the conditioned UNet on NHWC device tensors. `params`: reference state_dict names -> fp32 CUDA tensors
    (the master weights, updated in place by `adam_step`)."""

    def __init__(self, params: Dict[str, torch.Tensor], *, model_channels=128, channel_mult=(1, 2, 2), num_res_blocks=3,
                 cond_dim=128, engine: Optional[CddpmEngine] = None, device=None):
        self.dev = torch.device(device) if device is not None else next(iter(params.values())).device
        self.eng = engine
        self._cfg = dict(model_channels=model_channels, channel_mult=tuple(channel_mult), num_res_blocks=num_res_blocks, cond_dim=cond_dim)
        self._fit(1, 16, 16)
        self.C, self.mult, self.nres, self.cond_dim = model_channels, tuple(channel_mult), num_res_blocks, cond_dim
        self.p = {k: v.detach().to(self.dev, torch.float32).contiguous() for k, v in params.items()}
        self.program = self._build_program()
        self.state: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ program (mirrors UNetModel.__init__, OpenAI_Unet.py:604-797)
    def _build_program(self):
        C_, prog, chans = self.C, [], []
        ch, idx = C_, 1
        prog.append(("in", "input_blocks.0.0", None))
        chans.append(C_)
        for level, m in enumerate(self.mult):
            co = m * C_
            for _ in range(self.nres):
                prog.append(("res", f"input_blocks.{idx}.0", dict(cin=ch, cout=co, kind="plain", push=True)))
                ch = co
                chans.append(ch)
                idx += 1
            if level != len(self.mult) - 1:
                prog.append(("res", f"input_blocks.{idx}.0", dict(cin=ch, cout=ch, kind="down", push=True)))
                chans.append(ch)
                idx += 1
        prog.append(("res", "middle_block.0", dict(cin=ch, cout=ch, kind="plain")))
        prog.append(("attn", "middle_block.1", dict(c=ch)))
        prog.append(("res", "middle_block.2", dict(cin=ch, cout=ch, kind="plain")))
        idx = 0
        for level in reversed(range(len(self.mult))):
            co = self.mult[level] * C_
            for i in range(self.nres + 1):
                ich = chans.pop()
                prog.append(("res", f"output_blocks.{idx}.0", dict(cin=ch + ich, cout=co, kind="plain", concat=ich)))
                ch = co
                if level > 0 and i == self.nres:
                    prog.append(("res", f"output_blocks.{idx}.1", dict(cin=ch, cout=ch, kind="up")))
                idx += 1
        prog.append(("head", "out", dict(c=ch)))
        return prog

    def _fit(self, B, H, W):
        """the handle whose scratch (statistics records, split buffers) the operators use: grown when a larger batch arrives"""
        e = self.eng
        if e is None or e.max_batch < B or e.max_h < H or e.max_w < W:
            if e is not None and getattr(self, "_own", False):
                e.close()
            self.eng = CddpmEngine(timesteps=2, max_batch=B, max_h=H, max_w=W, device=self.dev, **self._cfg)
            self._own = True
        self.lib, self.h = self.eng.lib, self.eng._h

    # ------------------------------------------------------------------ thin operator wrappers (device pointers in, tensors out)
    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.cddpm_last_error(self.h).decode()}")

    def _s(self):
        return _stream_ptr(self.dev)

    def linear(self, x, w, b, silu_in=False):
        M, K = x.shape
        y = torch.empty((M, w.shape[0]), dtype=torch.float32, device=self.dev)
        self._ck(self.lib.cddpm_op_linear(self.h, _p(x), _p(w), _p(b), M, w.shape[0], K, int(silu_in), _p(y), self._s()), "op_linear")
        return y

    def linear_bwd(self, x, w, dy, silu_in=False, need_dx=True):
        dw, db, dx = self.eng.op_linear_backward(x, w, dy, silu_in=silu_in)
        return dw, db, (dx if need_dx else None)

    def gn_coef(self, x0, x1, name, film=None):
        return self.eng.op_gn_coef(x0, x1, self.p[name + ".weight"], self.p[name + ".bias"], film)

    def conv(self, x0, x1, coef, silu, up, wname, bias, res, res_up, k):
        return self.eng.op_conv(x0, x1, coef, silu, up, self.p[wname], bias, res, res_up, k)

    def unpool2(self, dyp, scale, into=None):
        B, h, w, Cc = dyp.shape
        out = into if into is not None else torch.empty((B, 2 * h, 2 * w, Cc), dtype=torch.float32, device=self.dev)
        self._ck(self.lib.cddpm_op_unpool2(self.h, _p(dyp), _p(out), B, 2 * h, 2 * w, Cc, C.c_float(scale), int(into is not None), self._s()),
                 "op_unpool2")
        return out

    def sumpool2(self, dy, into=None):
        B, H, W, Cc = dy.shape
        out = into if into is not None else torch.empty((B, H // 2, W // 2, Cc), dtype=torch.float32, device=self.dev)
        self._ck(self.lib.cddpm_op_sumpool2(self.h, _p(dy), _p(out), B, H, W, Cc, int(into is not None), self._s()), "op_sumpool2")
        return out

    def add_(self, a, b):
        self._ck(self.lib.cddpm_op_add_inplace(self.h, _p(a), _p(b), a.numel(), self._s()), "op_add_inplace")
        return a

    def bias_grad(self, dy):
        Cc = dy.shape[-1]
        db = torch.empty((Cc,), dtype=torch.float32, device=self.dev)
        self._ck(self.lib.cddpm_op_bias_grad(self.h, _p(dy), dy.numel() // Cc, Cc, _p(db), self._s()), "op_bias_grad")
        return db

    # ------------------------------------------------------------------ forward (OpenAI_Unet.py:823-1006), activations saved
    def forward(self, x: torch.Tensor, t: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
        """x [B,1,H,W], t [B] int, cond [B,cond_dim] on the device -> model output [B,1,H,W]; everything backward needs is kept"""
        p, sv = self.p, {}
        B, _c, H, W = x.shape
        self._fit(B, H, W)
        x = x.contiguous().float()
        # timestep embedding (util.py:151-171): cos first, float32 arithmetic as torch does
        half = self.C // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        args = t.detach().cpu().float()[:, None] * freqs[None]
        temb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1).to(self.dev)
        y1 = self.linear(temb, p["time_embed.0.weight"], p["time_embed.0.bias"])
        et = self.linear(y1, p["time_embed.2.weight"], p["time_embed.2.bias"], silu_in=True)
        l1 = self.linear(cond.float().contiguous(), p["label_emb.0.weight"], p["label_emb.0.bias"])
        ec = self.linear(l1, p["label_emb.2.weight"], p["label_emb.2.bias"], silu_in=True)
        emb = torch.cat([et, ec], dim=1).contiguous()
        sv.update(temb=temb, y1=y1, l1=l1, cond=cond.float().contiguous(), emb=emb, x=x)
        hs: List[torch.Tensor] = []
        cur = None
        for kind, name, a in self.program:
            if kind == "in":
                cur = torch.empty((B, H, W, self.C), dtype=torch.float32, device=self.dev)
                wv = p[name + ".weight"].reshape(self.C, 9).contiguous()
                self._ck(self.lib.cddpm_op_conv_in1(self.h, _p(x), _p(wv), _p(p[name + ".bias"]), _p(cur), B, H, W, self.C, self._s()), "op_conv_in1")
                hs.append(cur)
            elif kind == "res":
                x1 = hs.pop() if a.get("concat") else None
                film = self.linear(emb, p[name + ".emb_layers.1.weight"], p[name + ".emb_layers.1.bias"], silu_in=True)
                r = dict(x0=cur, x1=x1, film=film)
                coef1 = self.gn_coef(cur, x1, name + ".in_layers.0")
                r["coef1"] = coef1
                if a["kind"] == "down":
                    Bc, h_, w_, Cc = cur.shape
                    hp = torch.empty((Bc, h_ // 2, w_ // 2, Cc), dtype=torch.float32, device=self.dev)
                    xp = torch.empty_like(hp)
                    self._ck(self.lib.cddpm_op_pool_act(self.h, _p(cur), _p(coef1), _p(hp), _p(xp), Bc, h_, w_, Cc, self._s()), "op_pool_act")
                    h1 = self.conv(hp, None, None, False, 0, name + ".in_layers.2.weight", p[name + ".in_layers.2.bias"], None, False, 3)
                    r.update(hp=hp)
                    resid, res_up = xp, False
                elif a["kind"] == "up":
                    h1 = self.conv(cur, None, coef1, True, 2, name + ".in_layers.2.weight", p[name + ".in_layers.2.bias"], None, False, 3)
                    resid, res_up = cur, True
                else:
                    h1 = self.conv(cur, x1, coef1, True, 0, name + ".in_layers.2.weight", p[name + ".in_layers.2.bias"], None, False, 3)
                    resid, res_up = cur, False
                coef2 = self.gn_coef(h1, None, name + ".out_layers.0", film)
                r.update(h1=h1, coef2=coef2)
                if a["cin"] != a["cout"]:
                    xin = cur if x1 is None else torch.cat([cur, x1], dim=-1).contiguous()
                    r["xin"] = xin
                    out = self.eng.op_conv_skip(h1, coef2, True, p[name + ".out_layers.3.weight"],
                                                p[name + ".out_layers.3.bias"] + p[name + ".skip_connection.bias"], xin,
                                                p[name + ".skip_connection.weight"])
                else:
                    out = self.conv(h1, None, coef2, True, 0, name + ".out_layers.3.weight", p[name + ".out_layers.3.bias"], resid, res_up, 3)
                sv[name] = r
                cur = out
                if a.get("push"):
                    hs.append(cur)
            elif kind == "attn":
                Bc, h_, w_, Cc = cur.shape
                coefn = self.gn_coef(cur, None, name + ".norm")
                qkv = self.conv(cur, None, coefn, False, 0, name + ".qkv.weight", p[name + ".qkv.bias"], None, False, 1)
                att = self.eng.op_attention(qkv.reshape(Bc, h_ * w_, 3 * Cc))
                out = self.conv(att.reshape(Bc, h_, w_, Cc), None, None, False, 0, name + ".proj_out.weight", p[name + ".proj_out.bias"], cur, False, 1)
                sv[name] = dict(x=cur, coefn=coefn, qkv=qkv, att=att)
                cur = out
            else:   # head: GroupNorm -> SiLU -> Conv2d(C -> 1)
                coefo = self.gn_coef(cur, None, "out.0")
                w9 = p["out.2.weight"].reshape(a["c"], 9).t().contiguous()
                out = torch.empty((B, 1, H, W), dtype=torch.float32, device=self.dev)
                self._ck(self.lib.cddpm_op_head(self.h, _p(cur), _p(coefo), _p(w9), C.c_float(float(p["out.2.bias"][0])), _p(out), B, H, W, a["c"],
                                                self._s()), "op_head")
                sv["out"] = dict(x=cur, coefo=coefo, w9=w9)
                cur = out
        self.saved = sv
        return cur

    # ------------------------------------------------------------------ backward: dL/d(model output) -> gradients of every parameter
    def backward(self, dout: torch.Tensor) -> Dict[str, torch.Tensor]:
        p, sv, g = self.p, self.saved, {}
        B, _c, H, W = dout.shape
        dout = dout.contiguous().float()
        demb = torch.zeros_like(sv["emb"])
        skip_grads: List[torch.Tensor] = []      # gradients of popped skip tensors, consumed when the pushing op is reached (reverse order)
        d = None
        for kind, name, a in reversed(self.program):
            if kind == "head":
                r, Cc = sv["out"], a["c"]
                dact = torch.empty((B, H, W, Cc), dtype=torch.float32, device=self.dev)
                self._ck(self.lib.cddpm_op_head_dgrad(self.h, _p(dout), _p(r["w9"]), _p(dact), B, H, W, Cc, self._s()), "op_head_dgrad")
                dw = torch.empty((Cc, 9), dtype=torch.float32, device=self.dev)
                self._ck(self.lib.cddpm_op_chan_image_corr(self.h, _p(r["x"]), _p(r["coefo"]), 1, _p(dout), -1, _p(dw), B, H, W, Cc, self._s()),
                         "op_chan_image_corr")
                g["out.2.weight"] = dw.reshape(1, Cc, 3, 3)
                g["out.2.bias"] = dout.sum().reshape(1)
                d, g["out.0.weight"], g["out.0.bias"], _ = self.eng.op_gn_silu_backward(r["x"], dact, p["out.0.weight"], p["out.0.bias"], None, True)
            elif kind == "attn":
                r = sv[name]
                Bc, h_, w_, Cc = r["x"].shape
                da = self.eng.op_conv_dgrad(d, p[name + ".proj_out.weight"])
                g[name + ".proj_out.weight"], g[name + ".proj_out.bias"] = self.eng.op_conv_wgrad(r["att"].reshape(Bc, h_, w_, Cc), None, None, False, d, ksize=1)
                dqkv = self.eng.op_attention_backward(r["qkv"].reshape(Bc, h_ * w_, 3 * Cc), da.reshape(Bc, h_ * w_, Cc)).reshape(Bc, h_, w_, 3 * Cc)
                dn = self.eng.op_conv_dgrad(dqkv, p[name + ".qkv.weight"])
                g[name + ".qkv.weight"], g[name + ".qkv.bias"] = self.eng.op_conv_wgrad(r["x"], None, r["coefn"], False, dqkv, ksize=1)
                dx, g[name + ".norm.weight"], g[name + ".norm.bias"], _ = self.eng.op_gn_silu_backward(r["x"], dn, p[name + ".norm.weight"],
                                                                                                     p[name + ".norm.bias"], None, False)
                d = self.add_(dx, d)
            elif kind == "res":
                if a.get("push"):        # this op's output also fed a skip connection: add that gradient
                    d = self.add_(d, skip_grads.pop())
                r = sv[name]
                x0, x1, h1, film = r["x0"], r["x1"], r["h1"], r["film"]
                w1, w2 = p[name + ".in_layers.2.weight"], p[name + ".out_layers.3.weight"]
                # out = conv2(act2(h1)) + skip(x)
                da2 = self.eng.op_conv_dgrad(d, w2)
                g[name + ".out_layers.3.weight"], g[name + ".out_layers.3.bias"] = self.eng.op_conv_wgrad(h1, None, r["coef2"], True, d, ksize=3)
                if a["cin"] != a["cout"]:
                    ws = p[name + ".skip_connection.weight"]
                    dxs = self.eng.op_conv_dgrad(d, ws)                       # [B,H,W,Cin] over the concatenation
                    c0 = x0.shape[-1]
                    g[name + ".skip_connection.weight"], g[name + ".skip_connection.bias"] = self.eng.op_conv_wgrad(x0, x1, None, False, d, ksize=1)
                else:
                    dxs = None
                dh1, g[name + ".out_layers.0.weight"], g[name + ".out_layers.0.bias"], dfilm = self.eng.op_gn_silu_backward(
                    h1, da2, p[name + ".out_layers.0.weight"], p[name + ".out_layers.0.bias"], film, True)
                # film = Linear(SiLU(emb))
                g[name + ".emb_layers.1.weight"], g[name + ".emb_layers.1.bias"], de = self.linear_bwd(sv["emb"], p[name + ".emb_layers.1.weight"], dfilm, True)
                self.add_(demb, de)
                # h1 = conv1(...)
                xin = x0 if x1 is None else r.get("xin", None)
                if x1 is not None and xin is None:
                    xin = torch.cat([x0, x1], dim=-1).contiguous()
                if a["kind"] == "down":
                    dhp = self.eng.op_conv_dgrad(dh1, w1)
                    g[name + ".in_layers.2.weight"], g[name + ".in_layers.2.bias"] = self.eng.op_conv_wgrad(r["hp"], None, None, False, dh1, ksize=3)
                    da1 = self.unpool2(dhp, 0.25)
                    dx, g[name + ".in_layers.0.weight"], g[name + ".in_layers.0.bias"], _ = self.eng.op_gn_silu_backward(
                        x0, da1, p[name + ".in_layers.0.weight"], p[name + ".in_layers.0.bias"], None, True)
                    self.unpool2(d, 0.25, into=dx)                            # identity skip through avg_pool(x)
                elif a["kind"] == "up":
                    dau = self.eng.op_conv_dgrad(dh1, w1)                      # gradient of the upsampled activation
                    g[name + ".in_layers.2.weight"], g[name + ".in_layers.2.bias"] = self.eng.op_conv_wgrad(x0, None, r["coef1"], True, dh1, ksize=3, upsample=True)
                    da1 = self.sumpool2(dau)
                    dx, g[name + ".in_layers.0.weight"], g[name + ".in_layers.0.bias"], _ = self.eng.op_gn_silu_backward(
                        x0, da1, p[name + ".in_layers.0.weight"], p[name + ".in_layers.0.bias"], None, True)
                    self.sumpool2(d, into=dx)                                 # identity skip through the upsampled x
                else:
                    da1 = self.eng.op_conv_dgrad(dh1, w1)
                    g[name + ".in_layers.2.weight"], g[name + ".in_layers.2.bias"] = self.eng.op_conv_wgrad(x0, x1, r["coef1"], True, dh1, ksize=3)
                    dx, g[name + ".in_layers.0.weight"], g[name + ".in_layers.0.bias"], _ = self.eng.op_gn_silu_backward(
                        xin, da1, p[name + ".in_layers.0.weight"], p[name + ".in_layers.0.bias"], None, True)
                    if dxs is not None:
                        self.add_(dx, dxs)
                    else:
                        self.add_(dx, d)                                      # identity skip
                if x1 is not None:       # split the gradient of the concatenation: [h | popped skip tensor]
                    c0 = x0.shape[-1]
                    skip_grads.append(dx[..., c0:].contiguous())
                    d = dx[..., :c0].contiguous()
                else:
                    d = dx
            else:   # input conv
                d = self.add_(d, skip_grads.pop())
                dw = torch.empty((self.C, 9), dtype=torch.float32, device=self.dev)
                self._ck(self.lib.cddpm_op_chan_image_corr(self.h, _p(d), None, 0, _p(sv["x"]), 1, _p(dw), B, H, W, self.C, self._s()),
                         "op_chan_image_corr")
                g[name + ".weight"] = dw.reshape(self.C, 1, 3, 3)
                g[name + ".bias"] = self.bias_grad(d)
        assert not skip_grads
        # embedding MLPs (OpenAI_Unet.py:598-602, :583-590)
        hw = sv["emb"].shape[1] // 2
        det, dec = demb[:, :hw].contiguous(), demb[:, hw:].contiguous()
        g["time_embed.2.weight"], g["time_embed.2.bias"], dy1 = self.linear_bwd(sv["y1"], p["time_embed.2.weight"], det, True)
        g["time_embed.0.weight"], g["time_embed.0.bias"], _ = self.linear_bwd(sv["temb"], p["time_embed.0.weight"], dy1, False, need_dx=False)
        g["label_emb.2.weight"], g["label_emb.2.bias"], dl1 = self.linear_bwd(sv["l1"], p["label_emb.2.weight"], dec, True)
        g["label_emb.0.weight"], g["label_emb.0.bias"], dcond = self.linear_bwd(sv["cond"], p["label_emb.0.weight"], dl1, False)
        self.dcond = dcond           # gradient w.r.t. the context vector (what the encoder's backward would consume)
        return g

    # ------------------------------------------------------------------ loss of p_losses + one optimizer step
    def loss_and_grad(self, model_out, target, p2w=None, loss_type="l1", grad_scale=None):
        """-> (loss, grad_scale * dL/d(model_out)); grad_scale defaults to B*H*W rounded up to a power of two (see cddpm_op_loss)"""
        B, _c, H, W = model_out.shape
        dout = torch.empty_like(model_out)
        loss_b = torch.empty((B,), dtype=torch.float32, device=self.dev)
        self.grad_scale = float(grad_scale) if grad_scale is not None else float(2 ** math.ceil(math.log2(B * H * W)))
        self._ck(self.lib.cddpm_op_loss(self.h, _p(model_out), _p(target.contiguous().float()), _p(p2w), int(loss_type == "l2"), B, H * W,
                                        C.c_float(self.grad_scale), _p(dout), _p(loss_b), self._s()), "op_loss")
        return loss_b.mean(), dout

    def adam_step(self, grads: Dict[str, torch.Tensor], lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=None):
        """torch.optim.Adam(lr=1e-4) of DDPM_2D.configure_optimizers (DDPM_2D.py:305-306) on every UNet parameter"""
        st = self.state
        unscale = 1.0 / (grad_scale if grad_scale is not None else getattr(self, "grad_scale", 1.0))
        st["step"] = st.get("step", 0) + 1
        for k, gr in grads.items():
            w = self.p[k]
            if k not in st:
                st[k] = (torch.zeros_like(w), torch.zeros_like(w))
            m, v = st[k]
            gr = gr.reshape(w.shape).contiguous()
            self._ck(self.lib.cddpm_op_adam(self.h, _p(w), _p(gr), _p(m), _p(v), w.numel(), C.c_float(lr), C.c_float(betas[0]), C.c_float(betas[1]),
                                            C.c_float(eps), st["step"], C.c_float(unscale), self._s()), "op_adam")


def training_step(trainer: UNetTrainer, x01: torch.Tensor, cond: torch.Tensor, *, t: torch.Tensor, noise: torch.Tensor, timesteps=1000,
                  objective="pred_x0", loss_type="l1", all_reduce=False, lr=1e-4):
    """One optimisation step of the diffusion loss (cond_DDPM.py:647-655 -> :565-645; DDPM_2D.py:114-135): x01 [B,1,H,W] in [0,1], context
    cond [B,cond_dim], per-sample timesteps t and noise given by the caller. Returns the loss. `all_reduce`: average the gradients over
    the ranks of torch.distributed (RCCL) before the update -- the data-parallel training of the reference (Lightning DDP)."""
    buf = _schedule.schedule_buffers(timesteps)
    dev = trainer.dev
    x0 = x01.float() * 2 - 1
    sa = buf["sqrt_alphas_cumprod"].to(dev)[t].reshape(-1, 1, 1, 1)
    s1 = buf["sqrt_one_minus_alphas_cumprod"].to(dev)[t].reshape(-1, 1, 1, 1)
    xt = (sa * x0 + s1 * noise.float()).contiguous()      # q_sample (cond_DDPM.py:548-554); elementwise plumbing, not a hot operator
    out = trainer.forward(xt, t, cond)
    target = noise if objective == "pred_noise" else x0
    p2w = buf["p2_loss_weight"].to(dev)[t].contiguous()
    loss, dout = trainer.loss_and_grad(out, target, p2w, loss_type)
    grads = trainer.backward(dout)
    if all_reduce:
        import torch.distributed as dist
        flat = torch.cat([grads[k].reshape(-1) for k in sorted(grads)])
        dist.all_reduce(flat)
        flat /= dist.get_world_size()
        off = 0
        for k in sorted(grads):
            n = grads[k].numel()
            grads[k] = flat[off:off + n].reshape(grads[k].shape)
            off += n
    trainer.adam_step(grads, lr=lr)
    return loss
