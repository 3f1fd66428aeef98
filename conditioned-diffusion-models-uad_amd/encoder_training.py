"""Training of the context encoder on the HIP operators (SURVEY.md section 8 rows f2 + f4): timm's ResNet-50 (in_chans = 1,
num_classes = cond_dim; reference src/models/modules/DDPM_encoder.py:21-23, wrapped by SparK_2D_encoder with drop_path_rate 0.05,
spark/models.py:89-109) in TRAINING mode -- BatchNorm on batch statistics with running-statistics updates, optional stochastic depth --
forward with saved activations and the backward of every operator, what `loss.backward()` does to `self.encoder` in the reference's
training_step (src/models/DDPM_2D.py:114-135; Adam over self.parameters(), :305-306).

Same construction as training.UNetTrainer: Python sequencing C-ABI operators (include/cddpm.h, "training-mode operators of the context
encoder"; csrc/encoder_train.hip), one flat fp32 parameter buffer / gradient buffer / Adam state, weight images re-packed on the device
after every update. Parity: unpinned like the encoder's forward (timm is not in the image): checked against float64 autograd through
oracle/encoder_oracle.py's restatement (tests/test_gpu_encoder_training.py)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from .engine import _stream_ptr

STAGES = ((64, 3), (128, 4), (256, 6), (512, 3))
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class EncoderTrainer:
    """params: timm state_dict of the ResNet-50 (names conv1.weight, bn1.*, layerS.I.convJ.weight, layerS.I.bnJ.*, layerS.0.downsample.*,
    fc.*; running_mean / running_var included). `ops`: an object with `.lib`, `.h`, `.dev` (a training.UNetTrainer: the operators run on
    its handle and scratch arena)."""

    def __init__(self, params: Dict[str, torch.Tensor], ops, drop_path_rate: float = 0.0):
        self.ops, self.dev = ops, ops.dev
        train_names = [k for k in params if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]
        sizes = [(int(params[k].numel()) + 63) // 64 * 64 for k in train_names]
        self.flat = torch.zeros(sum(sizes), dtype=torch.float32, device=self.dev)
        self.gflat = torch.zeros_like(self.flat)
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        off = 0
        for k, n in zip(train_names, sizes):
            v = params[k]
            self.p[k] = self.flat[off:off + v.numel()].view(v.shape)
            self.g[k] = self.gflat[off:off + v.numel()].view(v.shape)
            self.p[k].copy_(v.detach().to(self.dev, torch.float32))
            off += n
        self.buf = {k: params[k].detach().to(self.dev, torch.float32).clone() for k in params
                    if k.endswith("running_mean") or k.endswith("running_var")}
        self.blocks = []
        cin, nb, idx = 64, sum(n for _p2, n in STAGES), 0
        for s, (planes, nblocks) in enumerate(STAGES):
            for i in range(nblocks):
                stride = 2 if (i == 0 and s > 0) else 1
                self.blocks.append(dict(name=f"layer{s + 1}.{i}", cin=cin, planes=planes, stride=stride, down=(i == 0),
                                        drop=drop_path_rate * idx / (nb - 1)))        # timm: linearly increasing per block
                cin, idx = 4 * planes, idx + 1
        self.convs = {}
        for bk in self.blocks:
            n, pl = bk["name"], bk["planes"]
            self.convs[n + ".conv1"] = (pl, bk["cin"], 1, 1)
            self.convs[n + ".conv2"] = (pl, pl, 3, bk["stride"])
            self.convs[n + ".conv3"] = (4 * pl, pl, 1, 1)
            if bk["down"]:
                self.convs[n + ".downsample.0"] = (4 * pl, bk["cin"], 1, bk["stride"])
        self.wf = {k: torch.empty(co * ci * kk * kk, dtype=torch.float32, device=self.dev) for k, (co, ci, kk, _s) in self.convs.items()}
        self.wd = {k: torch.empty(co * ci * kk * kk, dtype=torch.float32, device=self.dev) for k, (co, ci, kk, _s) in self.convs.items()}
        self.state: Dict[str, object] = {}
        self.repack()

    # ------------------------------------------------------------------ plumbing
    @property
    def lib(self):
        return self.ops.lib

    @property
    def h(self):
        return self.ops.h

    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.cddpm_last_error(self.h).decode()}")

    def _s(self):
        return _stream_ptr(self.dev)

    def _new(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def repack(self):
        for k, (co, ci, kk, _s) in self.convs.items():
            self._ck(self.lib.cddpm_op_enc_pack_w(self.h, _p(self.p[k + ".weight"]), co, ci, kk, _p(self.wf[k]), _p(self.wd[k]), self._s()),
                     "op_enc_pack_w")

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """timm names -> current tensors (parameters are views of the flat buffer; running statistics as updated by the forward passes)"""
        out = dict(self.p)
        out.update(self.buf)
        return out

    # ------------------------------------------------------------------ operators
    def conv(self, name, x, transposed=False, in_hw=None):
        co, ci, kk, st = self.convs[name]
        B = x.shape[0]
        if not transposed:
            H, W = x.shape[1], x.shape[2]
            out = self._new(B, (H + st - 1) // st, (W + st - 1) // st, co)
            img = self.wf[name]
        else:
            H, W = in_hw
            out = self._new(B, H, W, ci)
            img = self.wd[name]
        self._ck(self.lib.cddpm_op_enc_conv(self.h, _p(x), _p(img), _p(out), B, H, W, ci, co, kk, st, int(transposed), self._s()), "op_enc_conv")
        return out

    def wgrad(self, name, x, dz):
        co, ci, kk, st = self.convs[name]
        B, H, W, _c = x.shape
        ops = self.ops
        if getattr(ops, "overlap_wgrad", False) and getattr(ops, "side", None) is not None:
            # as the UNet's weight gradients (training.UNetTrainer.wgrad): on the side stream and its handle, beside the main stream's chain
            # of small BatchNorm / input-gradient kernels
            ops.side.wait_stream(torch.cuda.current_stream(self.dev))
            rc = self.lib.cddpm_op_enc_conv_wgrad(ops.eng_w._h, _p(x), _p(dz), _p(self.g[name + ".weight"]), B, H, W, ci, co, kk, st, ops.side.cuda_stream)
            if rc != 0:
                raise RuntimeError("op_enc_conv_wgrad failed: " + self.lib.cddpm_last_error(ops.eng_w._h).decode())
            x.record_stream(ops.side)
            dz.record_stream(ops.side)
            ops._side_busy = True
            return
        self._ck(self.lib.cddpm_op_enc_conv_wgrad(self.h, _p(x), _p(dz), _p(self.g[name + ".weight"]), B, H, W, ci, co, kk, st, self._s()),
                 "op_enc_conv_wgrad")

    def bn(self, name, z, relu, res=None, sscale=None):
        B, H, W, Cc = z.shape
        mr, y = self._new(2, Cc), torch.empty_like(z)
        self._ck(self.lib.cddpm_op_enc_bn_forward(self.h, _p(z), _p(self.p[name + ".weight"]), _p(self.p[name + ".bias"]), _p(sscale), _p(res),
                                                  int(relu), C.c_float(BN_EPS), C.c_float(BN_MOMENTUM), _p(self.buf[name + ".running_mean"]),
                                                  _p(self.buf[name + ".running_var"]), _p(mr), _p(y), B * H * W, H * W, Cc, self._s()),
                 "op_enc_bn_forward")
        return y, mr

    def bn_bwd(self, name, z, y, dy, mr, relu, sscale=None, want_dres=False):
        B, H, W, Cc = z.shape
        dz = torch.empty_like(z)
        dres = torch.empty_like(z) if want_dres else None
        self._ck(self.lib.cddpm_op_enc_bn_backward(self.h, _p(z), _p(y), _p(dy), _p(mr), _p(self.p[name + ".weight"]), _p(sscale), int(relu), _p(dz),
                                                   _p(dres), _p(self.g[name + ".weight"]), _p(self.g[name + ".bias"]), B * H * W, H * W, Cc,
                                                   self._s()), "op_enc_bn_backward")
        return dz, dres

    # ------------------------------------------------------------------ forward (training mode), activations saved
    def forward(self, x: torch.Tensor, drop_scales: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        """x [B,1,H,W] on the device -> context [B, cond_dim]. drop_scales (tests): per-block [B] scales instead of fresh random draws."""
        x = x.float().contiguous()
        B, _c, H, W = x.shape
        sv: Dict[str, object] = dict(x=x)
        H1, W1 = (H + 1) // 2, (W + 1) // 2
        z0 = self._new(B, H1, W1, 64)
        self._ck(self.lib.cddpm_op_enc_stem(self.h, _p(x), _p(self.p["conv1.weight"]), _p(z0), B, H, W, self._s()), "op_enc_stem")
        a0, mr0 = self.bn("bn1", z0, True)
        H2, W2 = (H1 + 1) // 2, (W1 + 1) // 2
        cur = self._new(B, H2, W2, 64)
        self._ck(self.lib.cddpm_op_enc_maxpool(self.h, _p(a0), _p(cur), B, H1, W1, 64, 0, None, None, self._s()), "op_enc_maxpool")
        sv.update(z0=z0, a0=a0, mr0=mr0)
        for bk in self.blocks:
            n = bk["name"]
            ss = None
            if drop_scales is not None:
                ss = drop_scales.get(n)
                ss = None if ss is None else ss.to(self.dev, torch.float32).contiguous()
            elif bk["drop"] > 0:
                keep = 1.0 - bk["drop"]
                ss = (torch.floor(keep + torch.rand(B, device=self.dev)) / keep).contiguous()     # timm.layers.drop_path
            z1 = self.conv(n + ".conv1", cur)
            a1, mr1 = self.bn(n + ".bn1", z1, True)
            z2 = self.conv(n + ".conv2", a1)
            a2, mr2 = self.bn(n + ".bn2", z2, True)
            z3 = self.conv(n + ".conv3", a2)
            r = dict(inp=cur, z1=z1, a1=a1, mr1=mr1, z2=z2, a2=a2, mr2=mr2, z3=z3, ss=ss)
            sc = cur
            if bk["down"]:
                zd = self.conv(n + ".downsample.0", cur)
                sc, mrd = self.bn(n + ".downsample.1", zd, False)
                r.update(zd=zd, mrd=mrd)
            out, mr3 = self.bn(n + ".bn3", z3, True, res=sc, sscale=ss)
            r.update(mr3=mr3, out=out)
            sv[n] = r
            cur = out
        Bc, Hc, Wc, Cc = cur.shape
        gap = self._new(B, Cc)
        self._ck(self.lib.cddpm_op_enc_avgpool(self.h, _p(cur), _p(gap), B, Hc * Wc, Cc, 0, self._s()), "op_enc_avgpool")
        wfc, bfc = self.p["fc.weight"], self.p["fc.bias"]
        out = self._new(B, wfc.shape[0])
        self._ck(self.lib.cddpm_op_linear(self.h, _p(gap), _p(wfc), _p(bfc), B, wfc.shape[0], Cc, 0, _p(out), self._s()), "op_linear")
        sv.update(gap=gap, last=cur)
        self.saved = sv
        return out

    # ------------------------------------------------------------------ backward: dL/d(context) -> gradients of every parameter
    def grad_offset(self, prefix: str) -> int:
        if not hasattr(self, "_goff"):
            base = self.gflat.data_ptr()
            self._goff = {k: (v.data_ptr() - base) // 4 for k, v in self.g.items()}
        return min((o for k, o in self._goff.items() if k.startswith(prefix)), default=self.gflat.numel())

    def backward(self, dcond: torch.Tensor, buckets=None) -> Dict[str, torch.Tensor]:
        """buckets (training.GradBuckets): told after every block how much of the flat gradient buffer's tail is final"""
        sv, g = self.saved, self.g
        gap, last = sv["gap"], sv["last"]
        B, Hc, Wc, Cc = last.shape
        wfc = self.p["fc.weight"]
        dgap = self._new(B, Cc)
        self._ck(self.lib.cddpm_op_linear_backward(self.h, _p(gap), _p(wfc), _p(dcond.float().contiguous()), B, wfc.shape[0], Cc, 0, _p(g["fc.weight"]),
                                                   _p(g["fc.bias"]), _p(dgap), self._s()), "op_linear_backward")
        d = torch.empty_like(last)
        self._ck(self.lib.cddpm_op_enc_avgpool(self.h, _p(dgap), _p(d), B, Hc * Wc, Cc, 1, self._s()), "op_enc_avgpool (backward)")
        for bk in reversed(self.blocks):
            n, r = bk["name"], sv[bk["name"]]
            inp = r["inp"]
            hw = (inp.shape[1], inp.shape[2])
            dz3, dsc = self.bn_bwd(n + ".bn3", r["z3"], r["out"], d, r["mr3"], True, r["ss"], want_dres=True)
            da2 = self.conv(n + ".conv3", dz3, transposed=True, in_hw=(r["a2"].shape[1], r["a2"].shape[2]))
            self.wgrad(n + ".conv3", r["a2"], dz3)
            dz2, _ = self.bn_bwd(n + ".bn2", r["z2"], r["a2"], da2, r["mr2"], True)
            da1 = self.conv(n + ".conv2", dz2, transposed=True, in_hw=hw)
            self.wgrad(n + ".conv2", r["a1"], dz2)
            dz1, _ = self.bn_bwd(n + ".bn1", r["z1"], r["a1"], da1, r["mr1"], True)
            dinp = self.conv(n + ".conv1", dz1, transposed=True, in_hw=hw)
            self.wgrad(n + ".conv1", inp, dz1)
            if bk["down"]:
                dzd, _ = self.bn_bwd(n + ".downsample.1", r["zd"], None, dsc, r["mrd"], False)
                self.ops.add_(dinp, self.conv(n + ".downsample.0", dzd, transposed=True, in_hw=hw))
                self.wgrad(n + ".downsample.0", inp, dzd)
            else:
                self.ops.add_(dinp, dsc)
            d = dinp
            if buckets is not None:
                if getattr(buckets, "on", True) and hasattr(self.ops, "join_side"):
                    self.ops.join_side()          # the side stream's weight gradients have landed before a collective reads the buffer
                buckets.mark_final(self.grad_offset(n + "."))
        a0, z0 = sv["a0"], sv["z0"]
        B, H1, W1, _c = a0.shape
        da0 = torch.empty_like(a0)
        self._ck(self.lib.cddpm_op_enc_maxpool(self.h, _p(a0), None, B, H1, W1, 64, 1, _p(d), _p(da0), self._s()), "op_enc_maxpool (backward)")
        dz0, _ = self.bn_bwd("bn1", z0, a0, da0, sv["mr0"], True)
        x = sv["x"]
        self._ck(self.lib.cddpm_op_enc_stem_wgrad(self.h, _p(x), _p(dz0), _p(g["conv1.weight"]), B, x.shape[2], x.shape[3], self._s()),
                 "op_enc_stem_wgrad")
        if hasattr(self.ops, "join_side"):
            self.ops.join_side()
        self.saved = None
        return g

    def adam_step(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, guarded=False):
        """Adam over the flat buffer under the guard of the handle's trainer (`ops`: training.UNetTrainer): the encoder and the UNet are ONE
        optimizer in the reference (`optim.Adam(self.parameters())`, DDPM_2D.py:305-306), so the step count and the skip decision are the
        UNet trainer's control block. guarded: `ops.guard(others=(self,))` was called for this step; else the encoder is stepped alone."""
        st = self.state
        if "m" not in st:
            st["m"], st["v"] = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        if not guarded:
            if getattr(self, "ctrl", None) is None:
                self.ctrl = torch.zeros(8, dtype=torch.int32, device=self.dev)
            self._ck(self.lib.cddpm_op_grad_check(self.h, _p(self.gflat), self.gflat.numel(), _p(self.ctrl), self._s()), "op_grad_check")
            self._ck(self.lib.cddpm_op_guard_commit(self.h, _p(self.ctrl), C.c_float(betas[0]), C.c_float(betas[1]), self._s()), "op_guard_commit")
            ctrl = self.ctrl
        else:
            ctrl = self.ops._ctrl()
        self._ck(self.lib.cddpm_op_adam_guarded(self.h, _p(self.flat), _p(self.gflat), _p(st["m"]), _p(st["v"]), self.flat.numel(), C.c_float(lr),
                                                C.c_float(betas[0]), C.c_float(betas[1]), C.c_float(eps), C.c_float(1.0 / grad_scale), _p(ctrl),
                                                self._s()), "op_adam_guarded")
        self.repack()

    def optimizer_state(self) -> Dict[str, torch.Tensor]:
        st = self.state
        if "m" not in st:
            st["m"], st["v"] = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        return {"m": st["m"].detach().clone(), "v": st["v"].detach().clone(), "layout": [(k, int(v.numel())) for k, v in self.p.items()]}

    def load_optimizer_state(self, state) -> None:
        if [tuple(x) for x in state["layout"]] != [(k, int(v.numel())) for k, v in self.p.items()]:
            raise ValueError("encoder optimizer state was saved for another parameter layout")
        self.state["m"] = state["m"].to(self.dev, torch.float32).clone()
        self.state["v"] = state["v"].to(self.dev, torch.float32).clone()

    def parameters_changed(self) -> None:
        """parameters were written from outside (load_state_dict into the aliased module): rebuild the operators' weight images"""
        self.repack()
