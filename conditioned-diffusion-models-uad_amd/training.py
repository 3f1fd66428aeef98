"""Host side of the TRAINING step on the HIP kernels: forward with saved activations, loss, backward,
Adam -- the reference's `DDPM_2D.training_step` -> `GaussianDiffusion.forward` -> `p_losses` with gradients
(reference src/models/DDPM_2D.py:114-135, :305-306; src/models/modules/cond_DDPM.py:565-655; src/models/modules/OpenAI_Unet.py:823-1006).

The reference's training step is Python sequencing torch operators and autograd; this module is Python sequencing the HIP
operators of libcddpm_hip.so through their C-ABI entry points (include/cddpm.h, "training step" section): every FLOP of the forward
and of the backward runs in csrc/*.hip; torch is used for device memory, views, torch.cat of saved tensors and torch.distributed
(the gradient all-reduce). fp32 throughout (the gradients carry fp32 accuracy and are checked against autograd on the oracle).

State: device-resident (flat parameter / gradient / Adam buffers, weight images re-packed on the device after every update, operator
temporaries from one scratch arena, no synchronisation inside a step); loss and all gradients checked against float64 autograd through the
oracle, three complete steps against float64 autograd + torch Adam (tests/test_gpu_training.py); the context encoder is trained jointly by
encoder_training.EncoderTrainer. Convolutions keep the inference path's fp32-grade arithmetic (fp16 two-term splits) by default; the
reference trainer's `precision: 16` arithmetic (plain fp16 operands, fp32 accumulation) is selected per process (CDDPM_TRAIN_PRECISION=16,
set from the Trainer's precision by the DDPM_2D mirror). A step whose gradients hold inf / NaN is skipped on the device (`guard`), as
torch's GradScaler does for the reference. Measured: DESIGN.md section 4b.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import torch

from . import schedule as _schedule
from .engine import CddpmEngine, _stream_ptr


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def unet_program(model_channels=128, channel_mult=(1, 2, 2), num_res_blocks=3):
    """the UNet as a list of (kind, state_dict prefix, attributes) in forward order -- UNetModel.__init__ / forward of the reference
    (OpenAI_Unet.py:604-797, :823-1006) for the cDDPM configuration (resblock_updown, attention in the middle block only): the input
    convolution, ResBlocks ('plain' | 'down' | 'up'; `push`: the output also goes on the skip stack; `concat`: channels popped from it
    and concatenated to the input), the middle attention block, the output head."""
    C_, prog, chans = model_channels, [], []
    ch, idx = C_, 1
    prog.append(("in", "input_blocks.0.0", None))
    chans.append(C_)
    for level, m in enumerate(channel_mult):
        co = m * C_
        for _ in range(num_res_blocks):
            prog.append(("res", f"input_blocks.{idx}.0", dict(cin=ch, cout=co, kind="plain", push=True)))
            ch = co
            chans.append(ch)
            idx += 1
        if level != len(channel_mult) - 1:
            prog.append(("res", f"input_blocks.{idx}.0", dict(cin=ch, cout=ch, kind="down", push=True)))
            chans.append(ch)
            idx += 1
    prog.append(("res", "middle_block.0", dict(cin=ch, cout=ch, kind="plain")))
    prog.append(("attn", "middle_block.1", dict(c=ch)))
    prog.append(("res", "middle_block.2", dict(cin=ch, cout=ch, kind="plain")))
    idx = 0
    for level in reversed(range(len(channel_mult))):
        co = channel_mult[level] * C_
        for i in range(num_res_blocks + 1):
            ich = chans.pop()
            prog.append(("res", f"output_blocks.{idx}.0", dict(cin=ch + ich, cout=co, kind="plain", concat=ich)))
            ch = co
            if level > 0 and i == num_res_blocks:
                prog.append(("res", f"output_blocks.{idx}.1", dict(cin=ch, cout=ch, kind="up")))
            idx += 1
    prog.append(("head", "out", dict(c=ch)))
    return prog


def conv_table(program):
    """every convolution that runs on the fused kernel: name -> (Cout, Cin, k, folded, exponent group). A ResBlock's second convolution
    and its 1x1 skip_connection run as ONE launch and share the pre-scale exponent."""
    t = {}
    for kind, name, a in program:
        if kind == "res":
            t[name + ".in_layers.2"] = (a["cout"], a["cin"], 3, a["kind"] == "up", name + ".in_layers.2")
            t[name + ".out_layers.3"] = (a["cout"], a["cout"], 3, False, name + ".out_layers.3")
            if a["cin"] != a["cout"]:
                t[name + ".skip_connection"] = (a["cout"], a["cin"], 1, False, name + ".out_layers.3")
        elif kind == "attn":
            t[name + ".qkv"] = (3 * a["c"], a["c"], 1, False, name + ".qkv")
            t[name + ".proj_out"] = (a["c"], a["c"], 1, False, name + ".proj_out")
    return t


class UNetTrainer:
    """Forward + backward + Adam of the conditioned UNet on NHWC device tensors, device-resident: the parameters live in ONE flat fp32
    buffer (`flat`; `p[name]` are views with the reference's state_dict names and shapes), their gradients in a second one (`gflat`,
    `g[name]`): every backward operator writes its parameter gradients straight into those views, the data-parallel all-reduce is one
    collective over `gflat` and Adam one launch over `flat`. The convolution weights are re-packed on the device after every update
    (cddpm_op_pack_conv) into the images the fused convolution kernel reads: a forward image and a transposed / flipped image for the
    input gradient. No operator synchronises: temporaries come from the handle's scratch arena."""

    def __init__(self, params: Dict[str, torch.Tensor], *, model_channels=128, channel_mult=(1, 2, 2), num_res_blocks=3,
                 cond_dim=128, device=None, exp_refresh=50, overlap_wgrad=None):
        self.dev = torch.device(device) if device is not None else next(iter(params.values())).device
        self._cfg = dict(model_channels=model_channels, channel_mult=tuple(channel_mult), num_res_blocks=num_res_blocks, cond_dim=cond_dim)
        self.C, self.mult, self.nres, self.cond_dim = model_channels, tuple(channel_mult), num_res_blocks, cond_dim
        # flat layout: the 27 ResBlocks' emb_layers.1 weights first, in program order and without gaps, then their biases: together they are
        # ONE [sum 2 Cout, E] matrix (11776 x 1024) -- one Linear forward and one backward per step instead of 27 (the inference engine's
        # table does the same); then every other tensor, each 256-byte aligned
        self.program = self._build_program()
        emb_names = [n + ".emb_layers.1" for kind, n, _a in self.program if kind == "res" and n + ".emb_layers.1.weight" in params]
        lead = [n + ".weight" for n in emb_names] + [n + ".bias" for n in emb_names]
        names = lead + [k for k in params if k not in set(lead)]
        self._lead = set(lead)          # written by ONE linear backward at the end of the backward pass (when batched: emb_rows > 0)
        sizes = [(int(params[k].numel()) + 63) // 64 * 64 for k in names]       # 256-byte aligned views
        self.flat = torch.zeros(sum(sizes), dtype=torch.float32, device=self.dev)
        self.gflat = torch.zeros_like(self.flat)
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        off = 0
        for k, n in zip(names, sizes):
            v = params[k]
            self.p[k] = self.flat[off:off + v.numel()].view(v.shape)
            self.g[k] = self.gflat[off:off + v.numel()].view(v.shape)
            self.p[k].copy_(v.detach().to(self.dev, torch.float32))
            off += n
        self.emb_off, self.emb_rows = {}, 0
        if emb_names and all(params[n + ".weight"].numel() % 64 == 0 and params[n + ".bias"].numel() % 64 == 0 for n in emb_names):
            E = params[emb_names[0] + ".weight"].shape[1]
            for n in emb_names:
                self.emb_off[n[:-len(".emb_layers.1")]] = self.emb_rows
                self.emb_rows += params[n + ".weight"].shape[0]
            nw = self.emb_rows * E
            self.emb_w, self.emb_gw = self.flat[:nw].view(self.emb_rows, E), self.gflat[:nw].view(self.emb_rows, E)
            self.emb_b, self.emb_gb = self.flat[nw:nw + self.emb_rows], self.gflat[nw:nw + self.emb_rows]
        self.state: Dict[str, object] = {}
        self.eng: Optional[CddpmEngine] = None
        # weight gradients on a SIDE stream (and a second handle: an operator's temporaries come from its handle's arena, one call at a
        # time): nothing in the backward pass waits for a weight gradient, so the k-image passes (memory bound) and the wgrad GEMMs run
        # beside the dgrad / GroupNorm-backward chain of the main stream instead of between its links. CDDPM_TRAIN_OVERLAP=0 switches off.
        import os as _os
        self.overlap_wgrad = (_os.environ.get("CDDPM_TRAIN_OVERLAP", "1") != "0") if overlap_wgrad is None else bool(overlap_wgrad)
        self.eng_w: Optional[CddpmEngine] = None
        self.side = None
        self.grad_scale = 1.0
        self.exp_refresh = exp_refresh
        self._convs = self._conv_table()
        self._fit(1, 16, 16)
        if self._convs:
            self.refresh_exponents()
            self.repack()

    # ------------------------------------------------------------------ program (mirrors UNetModel.__init__, OpenAI_Unet.py:604-797)
    def _build_program(self):
        return unet_program(self.C, self.mult, self.nres)

    def _conv_table(self):
        return conv_table(self.program) if "input_blocks.0.0.weight" in self.p else {}

    # ------------------------------------------------------------------ handle, packed weights
    def _fit(self, B, H, W):
        """the handle the operators run on; its scratch arena is sized for the batch (attention backward keeps two B x heads x N x N
        planes, the weight-gradient kernel its partial tiles)"""
        e = self.eng
        if e is None or e.max_batch < B or e.max_h < H or e.max_w < W:
            if e is not None:
                torch.cuda.synchronize(self.dev)
                e.close()
            self.eng = CddpmEngine(timesteps=2, max_batch=B, max_h=H, max_w=W, device=self.dev, **self._cfg)
            nmid = (H >> (len(self.mult) - 1)) * (W >> (len(self.mult) - 1))
            cmid = self.mult[-1] * self.C
            arena = 2 * B * (cmid // 64) * nmid * nmid * 4 + (192 << 20)
            # the 3x3 weight gradient's two k-images (fp16 hi | mid, batch padded to groups of 8): the largest (Cin + Cout) x pixels of the net
            lv, worst = 0, 0
            for kind, _name, a_ in self.program:
                if kind == "res":
                    worst = max(worst, (a_["cin"] + a_["cout"]) * (H >> lv) * (W >> lv), 2 * a_["cout"] * (H >> lv) * (W >> lv))
                    lv += 1 if a_["kind"] == "down" else -1 if a_["kind"] == "up" else 0
            arena += 2 * ((B + 7) // 8) * worst * 16
            rc = self.eng.lib.cddpm_op_set_scratch(self.eng._h, arena)
            if rc != 0:
                raise RuntimeError("cddpm_op_set_scratch failed: " + self.eng.lib.cddpm_last_error(self.eng._h).decode())
            if self.overlap_wgrad:
                if self.eng_w is not None:
                    self.eng_w.close()
                # the weight-gradient handle: only its operator arena is used (k-images + partial tiles); smallest geometry
                self.eng_w = CddpmEngine(timesteps=2, max_batch=1, max_h=16, max_w=16, device=self.dev, **self._cfg)
                if self.eng_w.lib.cddpm_op_set_scratch(self.eng_w._h, 2 * ((B + 7) // 8) * worst * 16 + (192 << 20)) != 0:
                    raise RuntimeError("cddpm_op_set_scratch failed: " + self.eng_w.lib.cddpm_last_error(self.eng_w._h).decode())
                if self.side is None:
                    self.side = torch.cuda.Stream(device=self.dev)
        self.lib, self.h = self.eng.lib, self.eng._h

    def refresh_exponents(self):
        """power-of-two pre-scale of each convolution weight for the fp16 operand split: the largest e in [0, 24] with max|w| 2^(e+1) < 2^14
        (one bit of headroom: the weights move between refreshes; folded upsample classes sum up to four taps: two more bits)"""
        names = list(self._convs)
        mx = torch.zeros(len(names), dtype=torch.float32, device=self.dev)
        for i, k in enumerate(names):
            w = self.p[k + ".weight"]
            self._ck(self.lib.cddpm_op_absmax(self.h, _p(w), w.numel(), mx[i:].data_ptr(), self._s()), "op_absmax")
        mx = mx.cpu().tolist()                                   # the one host read-back, every `exp_refresh` steps
        raw = {}
        for k, m in zip(names, mx):
            bound = m * (8.0 if self._convs[k][3] else 2.0)
            e = 24
            while e > 0 and math.ldexp(bound, e) >= 16384.0:
                e -= 1
            raw[k] = e
        self.wexp = {k: min(raw[j] for j in names if self._convs[j][4] == self._convs[k][4]) for k in names}

    def repack(self):
        """device images of every convolution weight for the forward operator and for the input-gradient operator: ONE launch over a
        device table of jobs (cddpm_op_pack_conv_batch; 69 convolutions x 2 operators, a folded-upsample image = four jobs); the table is
        rebuilt when the pre-scale exponents change"""
        import numpy as np
        if not hasattr(self, "pk"):
            self.pk, self.pkT = {}, {}
            for k, (co, ci, ks, folded, _g) in self._convs.items():
                nb = 4 * self.lib.cddpm_packed_conv_bytes(co, ci, 4) if folded else self.lib.cddpm_packed_conv_bytes(co, ci, ks * ks)
                self.pk[k] = torch.empty(nb, dtype=torch.uint8, device=self.dev)
                self.pkT[k] = torch.empty(self.lib.cddpm_packed_conv_bytes(ci, co, ks * ks), dtype=torch.uint8, device=self.dev)
            self._jobs_key = None
        key = tuple(self.wexp[k] for k in self._convs)
        if self._jobs_key != key:
            job = np.dtype([("w", "<u8"), ("dst", "<u8"), ("O", "<i4"), ("I", "<i4"), ("taps", "<i4"), ("mode", "<i4"), ("wexp", "<i4"), ("cls", "<i4")])
            rows = []
            for k, (co, ci, ks, folded, _g) in self._convs.items():
                w, e = self.p[k + ".weight"].data_ptr(), self.wexp[k]
                if folded:
                    rows += [(w, self.pk[k].data_ptr(), co, ci, 4, 2, e, cls) for cls in range(4)]
                else:
                    rows.append((w, self.pk[k].data_ptr(), co, ci, ks * ks, 0, e, 0))
                rows.append((w, self.pkT[k].data_ptr(), ci, co, ks * ks, 1, e, 0))        # input gradient: the operator's O = Cin, I = Cout
            tab = np.array(rows, dtype=job)
            assert tab.itemsize == 40
            self._jobs = torch.from_numpy(tab.view(np.uint8).copy()).to(self.dev)
            self._njobs = len(rows)
            self._job_units = max((r[2] // 128) * (r[3] // 32) * r[4] * 512 for r in rows)
            self._jobs_key = key
        self._ck(self.lib.cddpm_op_pack_conv_batch(self.h, _p(self._jobs), self._njobs, self._job_units, self._s()), "op_pack_conv_batch")

    # ------------------------------------------------------------------ thin operator wrappers (device pointers in, tensors out)
    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.cddpm_last_error(self.h).decode()}")

    def _s(self):
        return _stream_ptr(self.dev)

    def _new(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def linear(self, x, name, silu_in=False):
        w, b = self.p[name + ".weight"], self.p[name + ".bias"]
        M, K = x.shape
        y = self._new(M, w.shape[0])
        self._ck(self.lib.cddpm_op_linear(self.h, _p(x), _p(w), _p(b), M, w.shape[0], K, int(silu_in), _p(y), self._s()), "op_linear")
        return y

    def linear_bwd(self, x, name, dy, silu_in=False):
        """writes dW, db of Linear `name` into the gradient buffer, returns dx"""
        w = self.p[name + ".weight"]
        M, K = x.shape
        dx = self._new(M, K)
        self._ck(self.lib.cddpm_op_linear_backward(self.h, _p(x), _p(w), _p(dy), M, w.shape[0], K, int(bool(silu_in)), _p(self.g[name + ".weight"]),
                                                   _p(self.g[name + ".bias"]), _p(dx), self._s()), "op_linear_backward")
        return dx

    def gn_coef(self, x0, x1, name, film=None):
        """GroupNorm (+ FiLM) coefficient planes of cat[x0, x1]; the statistics come from the records the producing convolutions wrote in
        their epilogues (self.rec) where both sources have them, else from a sweep over the tensors"""
        B, HW, C0 = x0.shape[0], x0.shape[1] * x0.shape[2], x0.shape[-1]
        C1 = x1.shape[-1] if x1 is not None else 0
        coef = self._new(3, B, C0 + C1)
        r0, r1 = self.rec.get(x0.data_ptr()), (self.rec.get(x1.data_ptr()) if x1 is not None else None)
        if r0 is not None and (x1 is None or r1 is not None):
            self._ck(self.lib.cddpm_op_gn_coef_rec(self.h, _p(r0), r0.shape[1], C0, _p(r1), r1.shape[1] if r1 is not None else 0, C1,
                                                   _p(self.p[name + ".weight"]), _p(self.p[name + ".bias"]), _p(film), _p(coef), B, HW, self._s()),
                     "op_gn_coef_rec")
        else:
            self._ck(self.lib.cddpm_op_gn_coef(self.h, _p(x0), C0, _p(x1), C1, _p(self.p[name + ".weight"]), _p(self.p[name + ".bias"]), _p(film),
                                               _p(coef), B, HW, self._s()), "op_gn_coef")
        return coef

    def gn_bwd(self, x, da, name, film=None, silu=True, rec=None, add=None, x1=None):
        """backward of act(GroupNorm32(cat[x, x1]) (1 + scale) + shift): dgamma / dbeta go to the gradient buffer; -> (dx, dfilm or None),
        dx = (dx, dx1) when x1 is given (the input gradient split like the input). rec: the input's statistics records kept from the
        forward pass (else they are swept again); add: a tensor over all channels added to dx (skip-path gradient)"""
        B, H, W, c0 = x.shape
        two = x1 is not None
        if two and rec is None:                   # no records for the pair: one concatenated tensor, swept
            x, x1 = torch.cat([x, x1], dim=-1), None
        C1 = x1.shape[-1] if x1 is not None else 0
        Cc = x.shape[-1] + C1
        dx = torch.empty_like(x)
        dx1 = torch.empty_like(x1) if x1 is not None else None
        dfilm = self._new(B, 2 * Cc) if film is not None else None
        self._ck(self.lib.cddpm_op_gn_silu_backward(self.h, _p(x), _p(x1), C1, _p(da), _p(self.p[name + ".weight"]), _p(self.p[name + ".bias"]), _p(film),
                                                    int(bool(silu)), _p(dx), _p(dx1), _p(self.g[name + ".weight"]), _p(self.g[name + ".bias"]),
                                                    _p(dfilm), _p(rec), rec.shape[1] if rec is not None else 0, _p(add), B, H * W, Cc, self._s()),
                 "op_gn_silu_backward")
        if two and dx1 is None:
            dx, dx1 = dx[..., :c0].contiguous(), dx[..., c0:].contiguous()
        return ((dx, dx1) if two else dx), dfilm

    def rec_of(self, x0, x1=None):
        """statistics records of cat[x0, x1] when both sources have records of the same count (record tensors concatenate along channels)"""
        r0 = self.rec.get(x0.data_ptr())
        if x1 is None:
            return r0
        r1 = self.rec.get(x1.data_ptr())
        if r0 is None or r1 is None or r0.shape[1] != r1.shape[1]:
            return None
        return torch.cat([r0, r1], dim=2).contiguous()

    def conv(self, name, x0, x1=None, coef=None, silu=False, res=None, res_up=False, skip=None, skip_name=None, bias=None, stats=True,
             skip1=None):
        """fused forward convolution `name` on its packed image: conv_k(act(cat[x0, x1])) [+ conv1x1(skip)] + bias [+ res]"""
        co, ci, ks, folded, _g = self._convs[name]
        B, h_, w_, C0 = x0.shape
        H, W = (2 * h_, 2 * w_) if folded else (h_, w_)
        out = self._new(B, H, W, co)
        b = bias if bias is not None else self.p[name + ".bias"]
        rec = self._new(B, self.lib.cddpm_stat_records(H, W, 1 if folded else 0), co, 2) if stats else None
        self._ck(self.lib.cddpm_op_conv_packed(
            self.h, _p(x0), C0, _p(x1), x1.shape[-1] if x1 is not None else 0, _p(coef), int(bool(silu)), int(folded), _p(self.pk[name]), self.wexp[name],
            _p(b), co, ks, _p(res), int(bool(res_up)), _p(skip), skip.shape[-1] if skip is not None else 0,
            _p(skip1), skip1.shape[-1] if skip1 is not None else 0,
            _p(self.pk[skip_name]) if skip_name else None, _p(out), _p(rec), B, H, W, self._s()), "op_conv_packed")
        if rec is not None:
            self.rec[out.data_ptr()] = rec          # GroupNorm statistics of the output, written by the convolution's epilogue
        return out

    def dgrad(self, name, dy):
        """dL/d(input) of convolution `name`: the same kernel on the transposed / flipped image"""
        co, ci, ks, _folded, _g = self._convs[name]
        B, H, W, _c = dy.shape
        dx = self._new(B, H, W, ci)
        self._ck(self.lib.cddpm_op_conv_packed(self.h, _p(dy), co, None, 0, None, 0, 0, _p(self.pkT[name]), self.wexp[name], None, ci, ks, None, 0,
                                               None, 0, None, 0, None, _p(dx), None, B, H, W, self._s()), "op_conv_packed (input gradient)")
        return dx

    def wgrad(self, name, x0, x1, coef, silu, dy, upsample=False, bias=True):
        """dL/dW (and dL/db) of convolution `name` into the gradient buffer"""
        co, ci, ks, _folded, _g = self._convs[name]
        B, H, W, _c = dy.shape
        if self.overlap_wgrad and self.side is not None:
            main = torch.cuda.current_stream(self.dev)
            self.side.wait_stream(main)                       # the operands are products of the main stream's work so far
            rc = self.lib.cddpm_op_conv_wgrad(self.eng_w._h, _p(x0), x0.shape[-1], _p(x1), x1.shape[-1] if x1 is not None else 0, _p(coef),
                                              int(bool(silu)), int(bool(upsample)), _p(dy), co, ks, _p(self.g[name + ".weight"]),
                                              _p(self.g[name + ".bias"]) if bias else None, B, H, W, self.side.cuda_stream)
            if rc != 0:
                raise RuntimeError("op_conv_wgrad failed: " + self.lib.cddpm_last_error(self.eng_w._h).decode())
            for t_ in (x0, x1, coef, dy):                      # their memory must outlive the side stream's reads
                if t_ is not None:
                    t_.record_stream(self.side)
            self._side_busy = True
            return
        self._ck(self.lib.cddpm_op_conv_wgrad(self.h, _p(x0), x0.shape[-1], _p(x1), x1.shape[-1] if x1 is not None else 0, _p(coef), int(bool(silu)),
                                              int(bool(upsample)), _p(dy), co, ks, _p(self.g[name + ".weight"]),
                                              _p(self.g[name + ".bias"]) if bias else None, B, H, W, self._s()), "op_conv_wgrad")

    def close(self):
        """releases the handles (operator arenas, workspaces)"""
        for e in (self.eng, self.eng_w):
            if e is not None:
                e.close()
        self.eng = self.eng_w = None

    def join_side(self):
        """the main stream waits for the weight gradients issued on the side stream so far (before the gradient buffer is read: an
        all-reduce bucket, the guard, Adam)"""
        if getattr(self, "_side_busy", False):
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
            self._side_busy = False

    def unpool2(self, dyp, scale, into=None):
        B, h, w, Cc = dyp.shape
        out = into if into is not None else self._new(B, 2 * h, 2 * w, Cc)
        self._ck(self.lib.cddpm_op_unpool2(self.h, _p(dyp), _p(out), B, 2 * h, 2 * w, Cc, C.c_float(scale), int(into is not None), self._s()),
                 "op_unpool2")
        return out

    def sumpool2(self, dy, into=None):
        B, H, W, Cc = dy.shape
        out = into if into is not None else self._new(B, H // 2, W // 2, Cc)
        self._ck(self.lib.cddpm_op_sumpool2(self.h, _p(dy), _p(out), B, H, W, Cc, int(into is not None), self._s()), "op_sumpool2")
        return out

    def add_(self, a, b):
        self._ck(self.lib.cddpm_op_add_inplace(self.h, _p(a), _p(b), a.numel(), self._s()), "op_add_inplace")
        return a

    # ------------------------------------------------------------------ forward (OpenAI_Unet.py:823-1006), activations saved
    def forward(self, x: torch.Tensor, t: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
        """x [B,1,H,W], t [B] int, cond [B,cond_dim] on the device -> model output [B,1,H,W]; everything backward needs is kept"""
        p, sv = self.p, {}
        B, _c, H, W = x.shape
        self._fit(B, H, W)
        self.rec: Dict[int, torch.Tensor] = {}             # data_ptr of a saved activation -> its GroupNorm statistics records
        x = x.contiguous().float()
        # timestep embedding (util.py:151-171): cos first, float32 arithmetic as torch does
        half = self.C // 2
        if not hasattr(self, "_freqs"):
            self._freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half).to(self.dev)
        args = t.to(self.dev).float()[:, None] * self._freqs[None]          # on the device: no read-back of t (B x 64 values of plumbing)
        temb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1).contiguous()
        y1 = self.linear(temb, "time_embed.0")
        et = self.linear(y1, "time_embed.2", silu_in=True)
        if "label_emb.0.weight" in p:
            if cond is None:
                raise ValueError("this UNet is conditioned (label_emb present): training needs the context vector")
            cond = cond.float().contiguous()
            l1 = self.linear(cond, "label_emb.0")
            ec = self.linear(l1, "label_emb.2", silu_in=True)
            emb = torch.cat([et, ec], dim=1).contiguous()
        else:                       # cfg.condition False (num_classes None, OpenAI_Unet.py:583-590, :849-852): emb = time_embed(t) alone
            cond = l1 = None
            emb = et
        sv.update(temb=temb, y1=y1, l1=l1, cond=cond, emb=emb, x=x)
        film_all = None
        if self.emb_rows:           # FiLM (scale | shift) of all ResBlocks: one [B, E] x [E, 11776] product
            film_all = self._new(B, self.emb_rows)
            self._ck(self.lib.cddpm_op_linear(self.h, _p(emb), _p(self.emb_w), _p(self.emb_b), B, self.emb_rows, emb.shape[1], 1, _p(film_all),
                                              self._s()), "op_linear")
        hs: List[torch.Tensor] = []
        cur = None
        for kind, name, a in self.program:
            if kind == "in":
                cur = self._new(B, H, W, self.C)
                self._ck(self.lib.cddpm_op_conv_in1(self.h, _p(x), _p(p[name + ".weight"]), _p(p[name + ".bias"]), _p(cur), B, H, W, self.C, self._s()),
                         "op_conv_in1")
                hs.append(cur)
            elif kind == "res":
                x1 = hs.pop() if a.get("concat") else None
                if film_all is not None:
                    o, n2 = self.emb_off[name], 2 * a["cout"]
                    film = film_all[:, o:o + n2].contiguous()
                else:
                    film = self.linear(emb, name + ".emb_layers.1", silu_in=True)
                r = dict(x0=cur, x1=x1, film=film, rec_in=self.rec_of(cur, x1))
                coef1 = self.gn_coef(cur, x1, name + ".in_layers.0")
                r["coef1"] = coef1
                c1, c2 = name + ".in_layers.2", name + ".out_layers.3"
                if a["kind"] == "down":
                    Bc, h_, w_, Cc = cur.shape
                    hp, xp = self._new(Bc, h_ // 2, w_ // 2, Cc), self._new(Bc, h_ // 2, w_ // 2, Cc)
                    self._ck(self.lib.cddpm_op_pool_act(self.h, _p(cur), _p(coef1), _p(hp), _p(xp), Bc, h_, w_, Cc, self._s()), "op_pool_act")
                    h1 = self.conv(c1, hp)
                    r.update(hp=hp)
                    resid, res_up = xp, False
                elif a["kind"] == "up":
                    h1 = self.conv(c1, cur, None, coef1, True)
                    resid, res_up = cur, True
                else:
                    h1 = self.conv(c1, cur, x1, coef1, True)
                    resid, res_up = cur, False
                coef2 = self.gn_coef(h1, None, name + ".out_layers.0", film)
                r.update(h1=h1, coef2=coef2, rec_h1=self.rec_of(h1))
                if a["cin"] != a["cout"]:        # the 1x1 skip_connection reads cat[cur, x1] as two tensors: nothing is concatenated in memory
                    out = self.conv(c2, h1, None, coef2, True, skip=cur, skip1=x1, skip_name=name + ".skip_connection",
                                    bias=p[c2 + ".bias"] + p[name + ".skip_connection.bias"])
                else:
                    out = self.conv(c2, h1, None, coef2, True, res=resid, res_up=res_up)
                sv[name] = r
                cur = out
                if a.get("push"):
                    hs.append(cur)
            elif kind == "attn":
                Bc, h_, w_, Cc = cur.shape
                coefn = self.gn_coef(cur, None, name + ".norm")
                qkv = self.conv(name + ".qkv", cur, None, coefn, False, stats=False)
                att = self._new(Bc, h_, w_, Cc)
                self._ck(self.lib.cddpm_op_attention(self.h, _p(qkv), _p(att), Bc, h_ * w_, Cc, self._s()), "op_attention")
                out = self.conv(name + ".proj_out", att, res=cur)
                sv[name] = dict(x=cur, coefn=coefn, qkv=qkv, att=att, rec=self.rec_of(cur))
                cur = out
            else:   # head: GroupNorm -> SiLU -> Conv2d(C -> 1)
                coefo = self.gn_coef(cur, None, "out.0")
                w9 = p["out.2.weight"].reshape(a["c"], 9).t().contiguous()
                out = self._new(B, 1, H, W)
                self._ck(self.lib.cddpm_op_head(self.h, _p(cur), _p(coefo), _p(w9), C.c_float(0.0), _p(p["out.2.bias"]), _p(out), B, H, W, a["c"],
                                                self._s()), "op_head")
                sv["out"] = dict(x=cur, coefo=coefo, w9=w9, rec=self.rec_of(cur))
                cur = out
        self.saved = sv
        return cur

    # ------------------------------------------------------------------ backward: dL/d(model output) -> gradients of every parameter
    def grad_offset(self, prefix: str) -> int:
        """offset in `gflat` of the first parameter whose name starts with `prefix` (the flat buffer is in forward order: when the backward
        pass is done with the operator of that name, [offset, end) is final)"""
        if not hasattr(self, "_goff"):
            base = self.gflat.data_ptr()
            self._goff = {k: (v.data_ptr() - base) // 4 for k, v in self.g.items()}
        batched = self._lead if self.emb_rows else ()
        return min((o for k, o in self._goff.items() if k.startswith(prefix) and k not in batched), default=self.gflat.numel())

    def backward(self, dout: torch.Tensor, buckets: Optional["GradBuckets"] = None, join: bool = True) -> Dict[str, torch.Tensor]:
        """fills `g` (views of `gflat`) from dout = grad_scale * dL/d(model output); returns `g`. buckets: the gradient exchange, told after
        every operator how much of the buffer's tail is final (GradBuckets: all-reduce overlapped with the rest of the backward pass)"""
        p, sv, g = self.p, self.saved, self.g
        B, _c, H, W = dout.shape
        dout = dout.contiguous().float()
        demb = torch.zeros_like(sv["emb"])
        dfilm_all = self._new(B, self.emb_rows) if self.emb_rows else None
        skip_grads: List[torch.Tensor] = []      # gradients of popped skip tensors, consumed when the pushing op is reached (reverse order)
        d = None
        for kind, name, a in reversed(self.program):
            if kind == "head":
                r, Cc = sv["out"], a["c"]
                dact = self._new(B, H, W, Cc)
                self._ck(self.lib.cddpm_op_head_dgrad(self.h, _p(dout), _p(r["w9"]), _p(dact), B, H, W, Cc, self._s()), "op_head_dgrad")
                self._ck(self.lib.cddpm_op_chan_image_corr(self.h, _p(r["x"]), _p(r["coefo"]), 1, _p(dout), -1, _p(g["out.2.weight"]), B, H, W, Cc,
                                                           self._s()), "op_chan_image_corr")
                g["out.2.bias"].copy_(dout.sum().reshape(1))          # one scalar
                d, _ = self.gn_bwd(r["x"], dact, "out.0", rec=r["rec"])
            elif kind == "attn":
                r = sv[name]
                self.wgrad(name + ".proj_out", r["att"], None, None, False, d)
                da = self.dgrad(name + ".proj_out", d)
                Bc, h_, w_, Cc = r["x"].shape
                dqkv = torch.empty_like(r["qkv"])
                self._ck(self.lib.cddpm_op_attention_backward(self.h, _p(r["qkv"]), _p(da), _p(dqkv), Bc, h_ * w_, Cc, self._s()),
                         "op_attention_backward")
                self.wgrad(name + ".qkv", r["x"], None, r["coefn"], False, dqkv)
                dn = self.dgrad(name + ".qkv", dqkv)
                d, _ = self.gn_bwd(r["x"], dn, name + ".norm", None, False, rec=r["rec"], add=d)       # + the residual path x + h
            elif kind == "res":
                if a.get("push"):        # this op's output also fed a skip connection: add that gradient
                    d = self.add_(d, skip_grads.pop())
                r = sv[name]
                x0, x1, h1, film = r["x0"], r["x1"], r["h1"], r["film"]
                c1, c2 = name + ".in_layers.2", name + ".out_layers.3"
                # out = conv2(act2(h1)) + skip(x)
                self.wgrad(c2, h1, None, r["coef2"], True, d)             # (weight gradients first: they start on the side stream while
                if a["cin"] != a["cout"]:                                  #  the main stream runs the input gradients)
                    self.wgrad(name + ".skip_connection", x0, x1, None, False, d)
                da2 = self.dgrad(c2, d)
                dxs = None
                if a["cin"] != a["cout"]:
                    dxs = self.dgrad(name + ".skip_connection", d)            # [B,H,W,Cin] over the concatenation
                dh1, dfilm = self.gn_bwd(h1, da2, name + ".out_layers.0", film, rec=r["rec_h1"])
                if dfilm_all is not None:
                    dfilm_all[:, self.emb_off[name]:self.emb_off[name] + dfilm.shape[1]].copy_(dfilm)     # one backward for all blocks below
                else:
                    self.add_(demb, self.linear_bwd(sv["emb"], name + ".emb_layers.1", dfilm, True))       # film = Linear(SiLU(emb))
                if a["kind"] == "down":
                    self.wgrad(c1, r["hp"], None, None, False, dh1)
                    dhp = self.dgrad(c1, dh1)
                    da1 = self.unpool2(dhp, 0.25)
                    dx, _ = self.gn_bwd(x0, da1, name + ".in_layers.0", rec=r["rec_in"])
                    self.unpool2(d, 0.25, into=dx)                            # identity skip through avg_pool(x)
                elif a["kind"] == "up":
                    self.wgrad(c1, x0, None, r["coef1"], True, dh1, upsample=True)
                    dau = self.dgrad(c1, dh1)                                 # gradient of the upsampled activation
                    da1 = self.sumpool2(dau)
                    dx, _ = self.gn_bwd(x0, da1, name + ".in_layers.0", rec=r["rec_in"])
                    self.sumpool2(d, into=dx)                                 # identity skip through the upsampled x
                else:
                    self.wgrad(c1, x0, x1, r["coef1"], True, dh1)
                    da1 = self.dgrad(c1, dh1)
                    dx, _ = self.gn_bwd(x0, da1, name + ".in_layers.0", rec=r["rec_in"], x1=x1,
                                        add=dxs if dxs is not None else d)        # + the 1x1 skip_connection's gradient, or the identity skip's
                if x1 is not None:       # the gradient of the concatenation arrives split: [h | popped skip tensor]
                    d, dskip = dx
                    skip_grads.append(dskip)
                else:
                    d = dx
            else:   # input conv
                d = self.add_(d, skip_grads.pop())
                self._ck(self.lib.cddpm_op_chan_image_corr(self.h, _p(d), None, 0, _p(sv["x"]), 1, _p(g[name + ".weight"]), B, H, W, self.C, self._s()),
                         "op_chan_image_corr")
                self._ck(self.lib.cddpm_op_bias_grad(self.h, _p(d), B * H * W, self.C, _p(g[name + ".bias"]), self._s()), "op_bias_grad")
            if buckets is not None:      # this operator's tensors are final (its emb_layers too unless they are batched at the buffer's head)
                if getattr(buckets, "on", True):
                    self.join_side()     # ... once the side stream's weight gradients have landed (a collective reads the buffer)
                buckets.mark_final(self.grad_offset(name + "."))
        assert not skip_grads
        if dfilm_all is not None:       # film_all = Linear(SiLU(emb)): dW / db of all 27 emb_layers (contiguous in the gradient buffer) and demb
            emb = sv["emb"]
            self._ck(self.lib.cddpm_op_linear_backward(self.h, _p(emb), _p(self.emb_w), _p(dfilm_all), B, self.emb_rows, emb.shape[1], 1,
                                                       _p(self.emb_gw), _p(self.emb_gb), _p(demb), self._s()), "op_linear_backward")
        # embedding MLPs (OpenAI_Unet.py:598-602, :583-590)
        if sv["cond"] is not None:
            hw = sv["emb"].shape[1] // 2
            det, dec = demb[:, :hw].contiguous(), demb[:, hw:].contiguous()
            dl1 = self.linear_bwd(sv["l1"], "label_emb.2", dec, True)
            self.dcond = self.linear_bwd(sv["cond"], "label_emb.0", dl1, False)  # gradient w.r.t. the context vector (the encoder's input gradient)
        else:
            det, self.dcond = demb, None
        dy1 = self.linear_bwd(sv["y1"], "time_embed.2", det, True)
        self.linear_bwd(sv["temb"], "time_embed.0", dy1, False)
        if join:                 # every gradient of the buffer is in place for whoever reads it next (join=False: the caller joins later --
            self.join_side()     # the step lets the last weight gradients run beside the encoder's backward chain)
        self.saved = None
        return g

    # ------------------------------------------------------------------ loss of p_losses + one optimizer step
    def loss_and_grad(self, model_out, target, p2w=None, loss_type="l1", grad_scale=None):
        """-> (loss, grad_scale * dL/d(model_out)); grad_scale defaults to B*H*W rounded up to a power of two (see cddpm_op_loss)"""
        B, _c, H, W = model_out.shape
        dout = torch.empty_like(model_out)
        loss_b = self._new(B)
        self.grad_scale = float(grad_scale) if grad_scale is not None else float(2 ** math.ceil(math.log2(B * H * W)))
        self._ck(self.lib.cddpm_op_loss(self.h, _p(model_out), _p(target.contiguous().float()), _p(p2w), int(loss_type == "l2"), B, H * W,
                                        C.c_float(self.grad_scale), _p(dout), _p(loss_b), self._s()), "op_loss")
        return loss_b.mean(), dout

    # ------------------------------------------------------------------ the guarded update (GradScaler's skip of a non-finite step)
    def _ctrl(self):
        """int32[8] on the device: {non-finite flag, optimizer step, skip, skipped so far, bias corrections} (include/cddpm.h,
        cddpm_op_guard_commit); shared with an EncoderTrainer running on this handle: one optimizer, one decision"""
        if getattr(self, "ctrl", None) is None:
            self.ctrl = torch.zeros(8, dtype=torch.int32, device=self.dev)
        return self.ctrl

    def guard(self, others=(), betas=(0.9, 0.999)):
        """checks this trainer's gradients (and those of `others`) for inf / NaN and commits the decision for this step on the device:
        a step with a non-finite gradient neither updates parameters / moments nor advances the step count"""
        ctrl = self._ctrl()
        for tr_ in (self, *others):
            self._ck(self.lib.cddpm_op_grad_check(self.h, _p(tr_.gflat), tr_.gflat.numel(), _p(ctrl), self._s()), "op_grad_check")
        self._ck(self.lib.cddpm_op_guard_commit(self.h, _p(ctrl), C.c_float(betas[0]), C.c_float(betas[1]), self._s()), "op_guard_commit")

    @property
    def step_count(self) -> int:
        """optimizer steps taken (skipped ones do not count); reads the device counter (synchronises)"""
        return int(self._ctrl()[1].item())

    @property
    def skipped_steps(self) -> int:
        return int(self._ctrl()[3].item())

    def adam_step(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=None, guarded=False):
        """torch.optim.Adam(lr=1e-4) of DDPM_2D.configure_optimizers (DDPM_2D.py:305-306) on every parameter: one launch over the flat buffers
        (gradient = gflat / grad_scale), then the convolution images are re-packed from the updated weights. guarded: `guard()` was already
        called for this step (jointly with the encoder's gradients); otherwise it is called here."""
        st = self.state
        if "m" not in st:
            st["m"], st["v"] = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        if not guarded:
            self.guard(betas=betas)
        st["calls"] = st.get("calls", 0) + 1
        unscale = 1.0 / (grad_scale if grad_scale is not None else self.grad_scale)
        self._ck(self.lib.cddpm_op_adam_guarded(self.h, _p(self.flat), _p(self.gflat), _p(st["m"]), _p(st["v"]), self.flat.numel(), C.c_float(lr),
                                                C.c_float(betas[0]), C.c_float(betas[1]), C.c_float(eps), C.c_float(unscale), _p(self._ctrl()),
                                                self._s()), "op_adam_guarded")
        if self._convs:
            if self.exp_refresh and st["calls"] % self.exp_refresh == 0:
                self.refresh_exponents()
            self.repack()

    # ------------------------------------------------------------------ checkpoint state (Adam moments, step count)
    def optimizer_state(self) -> Dict[str, torch.Tensor]:
        """what a checkpoint must carry besides the parameters to resume this optimizer: Adam's m, v (flat, this trainer's layout) and the
        device control block with the step count"""
        st = self.state
        if "m" not in st:
            st["m"], st["v"] = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        # wexp / calls: the convolutions' pre-scale exponents are refreshed every `exp_refresh` updates, not every update -- a resumed run
        # must multiply with the exponents (and refresh on the schedule) the uninterrupted run would
        return {"m": st["m"].detach().clone(), "v": st["v"].detach().clone(), "ctrl": self._ctrl().detach().clone(),
                "layout": [(k, int(v.numel())) for k, v in self.p.items()], "wexp": dict(getattr(self, "wexp", {})),
                "calls": int(st.get("calls", 0))}

    def load_optimizer_state(self, state) -> None:
        layout = [(k, int(v.numel())) for k, v in self.p.items()]
        if [tuple(x) for x in state["layout"]] != layout:
            raise ValueError("optimizer state was saved for another parameter layout")
        self.state["m"] = state["m"].to(self.dev, torch.float32).clone()
        self.state["v"] = state["v"].to(self.dev, torch.float32).clone()
        self._ctrl().copy_(state["ctrl"].to(self.dev, torch.int32))
        self.state["calls"] = int(state.get("calls", 0))
        if state.get("wexp") and self._convs:
            self.wexp = {k: int(state["wexp"][k]) for k in self._convs}
            self.repack()

    def parameters_changed(self) -> None:
        """the flat parameter buffer was written from outside (load_state_dict into the aliased module parameters): refresh the
        pre-scale exponents and the packed convolution images the operators read"""
        if self._convs:
            self.refresh_exponents()
            self.repack()


def set_precision(precision) -> int:
    """selects the arithmetic of the training operators from a Lightning-style precision value: 32 / "32" / "32-true" / None -> fp32-grade
    (two-term fp16 operand splits); 16 / "16" / "16-mixed" -> plain fp16 operands with fp32 accumulation, the arithmetic of the reference
    trainer (`precision: 16`, configs/trainer/default.yaml:7). "bf16" / "bf16-mixed" (BASELINE config 5's wording) is served by the same
    fp16-operand kernels: 11 significand bits instead of bf16's 8, the exponent range covered by the per-tensor power-of-two pre-scales, the
    loss scale and the non-finite-step guard. Process-wide (include/cddpm.h: cddpm_set_train_precision). Returns the bits in effect."""
    from . import _lib
    key = str(precision).lower() if precision is not None else "32"
    if key in ("32", "32-true", "64", "64-true", "none"):
        bits = 32
    elif key in ("16", "16-mixed", "16-true", "bf16", "bf16-mixed", "bf16-true"):
        bits = 16
    else:
        raise ValueError(f"unknown precision {precision!r}")
    lib = _lib.load_library()
    if lib.cddpm_set_train_precision(bits) < 0:
        raise RuntimeError("cddpm_set_train_precision failed")
    return bits


def get_precision() -> int:
    from . import _lib
    return int(_lib.load_library().cddpm_get_train_precision())


class GradBuckets:
    """The data-parallel gradient exchange overlapped with the backward pass (the reference trains under Lightning DDP, src/train.py:62-65:
    bucketed all-reduce behind autograd hooks). The flat gradient buffer is laid out in FORWARD order and the backward pass fills it from
    its tail: once the operator that owns offset `lo` has written its gradients, everything in [lo, end) is final. `mark_final(lo)` is
    called after every backward operator; whenever at least `bucket_floats` finished floats are waiting, their slice is handed to an
    asynchronous all-reduce (torch.distributed runs it on its own stream behind the kernels already enqueued on the current one, so it
    overlaps the operators that follow). xGMI rings are per-link bound, so buckets are large (default 8 M floats = 32 MB: the UNet's 176 MB
    go out in 6 collectives, the last one small). `finish()` flushes the head of the buffer and waits; returns the number of ranks summed
    over (the caller folds the mean into Adam's unscale factor). Without an initialised process group every call is a no-op."""

    def __init__(self, flat: torch.Tensor, bucket_floats: int = 8 << 20, enabled: bool = True):
        import torch.distributed as dist
        self.flat, self.bucket = flat, int(bucket_floats)
        self.on = bool(enabled) and dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.on else 1
        self.hi = flat.numel()            # [hi, end) has been handed to a collective
        self.lo = flat.numel()            # [lo, end) is final
        self.work, self.issued = [], []

    def mark_final(self, lo: int):
        lo = max(0, min(int(lo), self.lo))
        self.lo = lo
        if self.on and self.hi - self.lo >= self.bucket:
            self._issue(self.lo, self.hi)

    def _issue(self, lo: int, hi: int):
        import torch.distributed as dist
        if hi > lo:
            self.work.append(dist.all_reduce(self.flat[lo:hi], async_op=True))
            self.issued.append((lo, hi))
            self.hi = lo

    def finish(self) -> int:
        if self.on:
            self._issue(0, self.hi)       # whatever is left, head of the buffer included
            for w in self.work:
                w.wait()
        self.work = []
        return self.world


def all_reduce_sum_(flat: torch.Tensor) -> int:
    """the data-parallel gradient exchange (the reference trains under Lightning DDP, src/train.py:62-65): ONE all-reduce over the flat
    gradient buffer (43.9 M floats = 176 MB: a single large ring collective, what xGMI's per-link bandwidth wants). Returns the number of
    ranks summed over; the caller divides (training_step folds it into Adam's unscale factor)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat)
        return dist.get_world_size()
    return 1


def training_step(trainer: UNetTrainer, x01: torch.Tensor, cond: Optional[torch.Tensor], *, t: torch.Tensor, noise: torch.Tensor, timesteps=1000,
                  objective="pred_x0", loss_type="l1", all_reduce=False, lr=1e-4, buffers=None, encoder=None):
    """One optimisation step of the diffusion loss (cond_DDPM.py:647-655 -> :565-645; DDPM_2D.py:114-135): x01 [B,1,H,W] in [0,1], context
    cond [B,cond_dim], per-sample timesteps t and noise given by the caller; `buffers`: the diffusion's schedule tables (default: the
    cosine schedule of `timesteps`). Returns the loss. `all_reduce`: sum the gradients over the ranks of torch.distributed (RCCL) before
    the update -- the data-parallel training of the reference (Lightning DDP, src/train.py:62-65).
    `encoder` (an encoder_training.EncoderTrainer): the context is computed by it in training mode (cond is ignored) and it is trained
    jointly -- dL/d(context) of the UNet's backward flows into its backward, its gradients join the all-reduce and its own Adam step runs
    with the same learning rate: `features = self(input)` + `optim.Adam(self.parameters())` of the reference."""
    buf = buffers if buffers is not None else _schedule.schedule_buffers(timesteps)
    dev = trainer.dev
    trainer._fit(*[x01.shape[i] for i in (0, 2, 3)])       # the handle and its scratch arena before the first operator runs
    if encoder is not None:
        cond = encoder.forward(x01)
    x0 = x01.float() * 2 - 1
    sa = buf["sqrt_alphas_cumprod"].to(dev)[t].reshape(-1, 1, 1, 1)
    s1 = buf["sqrt_one_minus_alphas_cumprod"].to(dev)[t].reshape(-1, 1, 1, 1)
    xt = (sa * x0 + s1 * noise.float()).contiguous()      # q_sample (cond_DDPM.py:548-554); elementwise plumbing, not a hot operator
    out = trainer.forward(xt, t, cond)
    target = noise if objective == "pred_noise" else x0
    p2w = buf["p2_loss_weight"].to(dev)[t].contiguous()
    loss, dout = trainer.loss_and_grad(out, target, p2w, loss_type)
    # the gradient exchange runs bucket by bucket BEHIND the backward pass: a bucket's all-reduce is issued as soon as its slice of the
    # flat buffer is final and overlaps the backward operators that follow (and, for the UNet's last buckets, the encoder's backward)
    buckets = GradBuckets(trainer.gflat, enabled=all_reduce)
    trainer.backward(dout, buckets, join=(encoder is None))
    enc_buckets = None
    if encoder is not None:
        enc_buckets = GradBuckets(encoder.gflat, enabled=all_reduce)
        encoder.backward(trainer.dcond, enc_buckets)       # carries the same loss scale; joins the side stream at its end
    trainer.join_side()
    world = buckets.finish()
    if enc_buckets is not None:
        enc_buckets.finish()
    # one decision for the whole optimizer (after the all-reduce: an inf / NaN on any rank reaches every rank through the sum)
    trainer.guard(others=(encoder,) if encoder is not None else ())
    trainer.adam_step(lr=lr, grad_scale=trainer.grad_scale * world, guarded=True)     # the mean over ranks folds into Adam's unscale factor
    if encoder is not None:
        encoder.adam_step(lr=lr, grad_scale=trainer.grad_scale * world, guarded=True)
    return loss
