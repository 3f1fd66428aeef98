"""Host-side mirror of the reference LightningModule `DDPM_2D` (reference src/models/DDPM_2D.py:17-308):
the Hydra target `src.models.DDPM_2D.DDPM_2D` of `experiment=cDDPM/DDPM_cond_spark_2D`.

Kept: constructor `DDPM_2D(cfg, prefix=None)` reading the same cfg keys with the same defaults (:37-77),
attributes `.encoder`, `.diffusion`, `.test_timesteps`, `forward(x) -> c`, the state_dict prefixes
`encoder.*` / `diffusion.model.*` / `diffusion.<buffers>`, `configure_optimizers`.
Added, behind a cfg switch that is absent (= reference behaviour) by default:
    cfg.reverse_sampling: true   ->  test-time reconstruction = the iterative reverse loop
                                     (GaussianDiffusion.p_sample_loop) instead of the single-step x0 estimate
    cfg.reverse_start_t: int     ->  start_t of that loop (0 = all `timesteps` steps)

`test_step` follows the reference's evaluation call (:171-286): 4 centre slices, `noise_ensemble` / `step_ensemble`
averaging, a fresh `gen_noise` (device simplex) field per reconstruction.
The context encoder is this package's native ResNet-50 (DDPM_encoder.py) unless an
`encoder=` module is supplied; `cfg.pretrained_encoder` loads a SparK pre-training checkpoint into it with the reference's key rewrite
(:79-96). `training_step` runs the optimisation step of the UNet AND of the native context encoder (trained jointly, as
`optim.Adam(self.parameters())` does in the reference) on the HIP operators (training.py, encoder_training.py); Adam's state travels in
checkpoints (`on_save_checkpoint` / `on_load_checkpoint`). The scipy post-processing of utils_eval lives in utils_eval.py. pytorch_lightning / omegaconf are used when installed and replaced by
nn.Module / a plain attribute dict when not.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .OpenAI_Unet import UNetModel as OpenAI_UNet
from .cond_DDPM import GaussianDiffusion
from .generate_noise import gen_noise

try:  # Lightning 1.5 path first (what the reference pins), then 2.x, then a plain Module
    from pytorch_lightning.core.lightning import LightningModule as _Base  # type: ignore
except Exception:  # pragma: no cover - depends on the environment
    try:
        from pytorch_lightning import LightningModule as _Base  # type: ignore
    except Exception:
        _Base = nn.Module


class AttrDict(dict):
    """cfg stand-in when omegaconf is absent: cfg.key, cfg['key'], cfg.get(key, default)"""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def _cfg_get(cfg, key, default=None):
    try:
        v = cfg.get(key, default)
    except AttributeError:
        v = getattr(cfg, key, default)
    return default if v is None else v


def build_encoder(cfg):
    """get_encoder (reference src/models/modules/DDPM_encoder.py:6-29, called at DDPM_2D.py:32): the native MI355X
    encoder of this package (DDPM_encoder.py / csrc/encoder.hip: timm's resnet50 layout with in_chans=1,
    num_classes=cond_dim, plain or wrapped as SparK_2D_encoder as the cfg's `backbone` says). Its parity is unpinned
    offline: timm 0.6.7 is not in the image (oracle/encoder_oracle.py restates the published architecture)."""
    from .DDPM_encoder import get_encoder
    return get_encoder(cfg)


def rewrite_pretrained_encoder_keys(state_dict_pretrained):
    """the key rewrite of the reference's pretrained-encoder ingest (src/models/DDPM_2D.py:84-95), for the state_dict of a SparK
    pre-training checkpoint (Spark_2D LightningModule): `...slice_encoder...X` -> `slice_encoder` + what follows the LAST 'encoder';
    `sparse_encoder.sp_cnn.X` -> `encoder.X` with the pre-training head `fc.weight` / `fc.bias` dropped; every other key kept as it is
    (load_state_dict(strict=False) ignores what the encoder does not have)."""
    from collections import OrderedDict
    new_statedict = OrderedDict()
    for key, value in state_dict_pretrained.items():
        if "slice_encoder" in key:
            new_statedict["slice_encoder" + key.split("encoder")[-1]] = value
        elif "sparse_encoder" in key:
            if "fc.weight" not in key and "fc.bias" not in key:
                new_statedict["encoder" + key.split("sp_cnn")[-1]] = value
        else:
            new_statedict[key] = value
    return new_statedict


def load_pretrained_encoder(encoder, encoder_path):
    """reference :79-96: `torch.load(encoder_path)['state_dict']` -> key rewrite -> `encoder.load_state_dict(new, strict=False)`.
    The file is read with `weights_only=True` (nothing in it is executed); a checkpoint that only unpickles with arbitrary classes must be
    re-saved as {'state_dict': ...} first. Returns load_state_dict's (missing_keys, unexpected_keys)."""
    ckpt = torch.load(encoder_path, map_location="cpu", weights_only=True)
    if not isinstance(ckpt, dict) or "state_dict" not in ckpt:
        raise KeyError(f"{encoder_path}: no 'state_dict' entry (expected a Lightning checkpoint of the SparK pre-training)")
    return encoder.load_state_dict(rewrite_pretrained_encoder_keys(ckpt["state_dict"]), strict=False)


class DDPM_2D(_Base):
    def __init__(self, cfg, prefix=None, encoder=None):
        super().__init__()
        if isinstance(cfg, dict) and not isinstance(cfg, AttrDict):
            cfg = AttrDict(cfg)
        self.cfg = cfg
        if _cfg_get(cfg, "condition", True):
            cfg["cond_dim"] = _cfg_get(cfg, "unet_dim", 128)
            if encoder is not None:
                self.encoder, out_features = encoder, int(cfg["cond_dim"])
            else:
                self.encoder, out_features = build_encoder(cfg)
        else:
            out_features = None
        size = (int(cfg["imageDim"][0] / cfg["rescaleFactor"]), int(cfg["imageDim"][1] / cfg["rescaleFactor"]))
        model = OpenAI_UNet(
            image_size=size, in_channels=1, model_channels=_cfg_get(cfg, "unet_dim", 64), out_channels=1,
            num_res_blocks=_cfg_get(cfg, "num_res_blocks", 3), attention_resolutions=tuple(_cfg_get(cfg, "att_res", [3, 6, 12])),
            dropout=_cfg_get(cfg, "dropout_unet", 0), channel_mult=_cfg_get(cfg, "dim_mults", [1, 2, 4, 8]),
            conv_resample=True, dims=2, num_classes=out_features, use_checkpoint=False, use_fp16=True, num_heads=1,
            num_head_channels=64, num_heads_upsample=-1, use_scale_shift_norm=True, resblock_updown=True,
            use_new_attention_order=True, use_spatial_transformer=_cfg_get(cfg, "spatial_transformer", False),
            transformer_depth=1)
        model.convert_to_fp16()
        timesteps = _cfg_get(cfg, "timesteps", 1000)
        self.test_timesteps = _cfg_get(cfg, "test_timesteps", 150)
        self.diffusion = GaussianDiffusion(
            model, image_size=size, timesteps=timesteps, sampling_timesteps=_cfg_get(cfg, "sampling_timesteps", timesteps),
            objective=_cfg_get(cfg, "objective", "pred_x0"), channels=1, loss_type=_cfg_get(cfg, "loss", "l1"),
            p2_loss_weight_gamma=_cfg_get(cfg, "p2_gamma", 0), cfg=cfg)
        if _cfg_get(cfg, "pretrained_encoder", False):        # reference :79-96
            if not _cfg_get(cfg, "condition", True):
                raise ValueError("pretrained_encoder is set but condition is False: there is no encoder to load into")
            path = _cfg_get(cfg, "encoder_path", None)
            assert path is not None, "pretrained_encoder is set but cfg.encoder_path is missing"      # the reference's assert (:81)
            self.pretrained_encoder_keys = load_pretrained_encoder(self.encoder, path)
        self.prefix = prefix
        if hasattr(self, "save_hyperparameters") and _Base is not nn.Module:
            try:
                self.save_hyperparameters()
            except Exception:
                pass

    def forward(self, x):
        """encode slices [D,1,H,W] -> context c [D, cond_dim] (reference :102-111)"""
        if _cfg_get(self.cfg, "condition", True):
            return self.encoder(x)
        return None

    @torch.no_grad()
    def reconstruct(self, input, features=None, noise=None, t=None):
        """The reconstruction call of test_step (reference :225-247), [D,1,H,W] in [0,1] -> (loss, reco).
        reverse_sampling off: `self.diffusion(input, cond=features, t=t-1, noise=noise)` (single step);
        reverse_sampling on : the reverse loop from pure noise / from start_t (p_sample_loop)."""
        if features is None:
            features = self(input)
        with torch.autocast("cuda", enabled=False):          # the path is fp32 whatever the Trainer's precision
            if _cfg_get(self.cfg, "reverse_sampling", False):
                start_t = int(_cfg_get(self.cfg, "reverse_start_t", 0))
                reco = self.diffusion.p_sample_loop(tuple(input.shape), cond=features, start_t=start_t)
                loss = (reco - input).abs().mean()
                return loss, reco
            t = self.test_timesteps if t is None else t
            return self.diffusion(input, cond=features, t=t - 1, noise=noise)

    def _gen_noise(self, shape, device, engine=None):
        """`gen_noise(self.cfg, input.shape).to(self.device)` of the reference (:231, :241): a simplex field drawn on the
        device (generate_noise.py mirror, bit-exact for a given numpy seed) or None when cfg.noisetype is unset (the
        diffusion then draws Gaussian noise itself, cond_DDPM.py:577). The generator needs no model state: any handle on the
        device serves (`engine`: the training step passes its trainer's handle, so that drawing noise never touches -- or
        rebuilds -- the inference engine whose packed weights the step is about to invalidate)."""
        if _cfg_get(self.cfg, "noisetype", None) is None:
            return None
        B, _c, H, W = shape
        return gen_noise(self.cfg, shape, engine=engine if engine is not None else self.diffusion._engine(B, H, W, device))

    @torch.no_grad()
    def test_step(self, batch, batch_idx: int):
        """The evaluation call of the reference (src/models/DDPM_2D.py:171-286), same order of operations:
        the 4 centre slices of the volume (`num_eval_slices` is hard-wired to 4 at :193; :194-203), depth to the batch
        axis (:210), context c = encoder(slices) (:214), then either the noise ensemble (`noise_ensemble: True` in the
        experiment yaml :22: reconstructions at t in `step_ensemble` = [250, 500, 750], each from a FRESH `gen_noise`
        field, averaged, :225-236) or one reconstruction at `test_timesteps` (:240-247); volume re-assembled as
        [1,1,H,W,D] (:256-275). cfg.reverse_sampling (absent in the reference) swaps the single-step estimate for the
        reverse loop at the same call site. The scipy/sklearn post-processing `_test_step` (:277) runs when the reference's
        `src.utils.utils_eval` is importable (this class dropped into the reference tree) and the batch carries its
        inputs; the tensors are returned either way."""
        def data_of(key):
            v = batch.get(key) if hasattr(batch, "get") else None
            if v is None:
                return None
            return v["data"] if isinstance(v, dict) else v

        def field(key, default=None):
            return batch.get(key, default) if hasattr(batch, "get") else default

        self.dataset = field("Dataset")                                     # (:176) read by utils_eval._test_step
        self.stage = field("stage")                                         # (:184)
        input = data_of("vol")                                              # [1,1,H,W,D]
        data_orig, data_seg, data_mask = data_of("vol_orig"), data_of("seg_orig"), data_of("mask_orig")
        if data_orig is not None and (data_seg is None or not field("seg_available", data_seg is not None)):
            data_seg = torch.zeros_like(data_orig)                          # (:179)
        self.cfg["num_eval_slices"] = 4                                      # (:193)
        D = input.size(4)
        num_slices = _cfg_get(self.cfg, "num_eval_slices", D)
        ind_offset = 0
        if num_slices != D:
            start_slice = int((D - num_slices) / 2)                          # (:196)
            sl = slice(start_slice, start_slice + num_slices)
            input = input[..., sl]
            data_orig = data_orig[..., sl] if data_orig is not None else None
            data_seg = data_seg[..., sl] if data_seg is not None else None
            data_mask = data_mask[..., sl] if data_mask is not None else None
            ind_offset = start_slice
        assert input.shape[0] == 1, "Batch size must be 1"
        input = input.squeeze(0).permute(3, 0, 1, 2).contiguous()           # [D,1,H,W]   (:210)
        features = self(input)
        if _cfg_get(self.cfg, "noise_ensemble", False):
            timesteps = list(_cfg_get(self.cfg, "step_ensemble", [250, 500, 750]))
            reco_ensemble = torch.zeros_like(input)
            for t in timesteps:
                noise = self._gen_noise(input.shape, input.device)
                loss_diff, reco = self.reconstruct(input, features, noise, t=t)
                reco_ensemble += reco
            reco = reco_ensemble / len(timesteps)
        else:
            timesteps = [self.test_timesteps]
            noise = self._gen_noise(input.shape, input.device)
            loss_diff, reco = self.reconstruct(input, features, noise, t=self.test_timesteps)
        final_volume = reco.clone().squeeze().permute(1, 2, 0).unsqueeze(0).unsqueeze(0)   # (:256-275)
        out = {"loss": loss_diff, "final_volume": final_volume, "input": input, "features": features,
               "timesteps": timesteps, "ind_offset": ind_offset}
        if hasattr(self, "eval_dict"):
            self._record_volume_scores(features, input, loss_diff)
        if data_orig is not None and data_mask is not None and hasattr(self, "eval_dict"):
            try:
                from src.utils.utils_eval import _test_step  # type: ignore  (reference tree on sys.path)
            except ImportError:                 # standalone (no reference tree): the reconstruction is returned without the metric pass
                _test_step = None
            if _test_step is not None:
                _test_step(self, final_volume, data_orig, data_seg, data_mask, batch_idx, field("ID"), field("label"))
        return out

    def _record_volume_scores(self, features, input, loss_diff):
        """the per-volume bookkeeping the reference's test_step does before `_test_step` (:216-220, :249-254, :258-271): the mean context
        vector of the volume, the L1 reconstruction loss as the three anomaly scores; `_test_end` / `calc_thresh` read these lists"""
        import numpy as np
        ed = self.eval_dict
        if _cfg_get(self.cfg, "condition", True) and features is not None:
            latent = [features.mean(0).squeeze().detach().cpu()]
        else:
            latent = [torch.tensor([0], dtype=float).repeat(input.shape[0])]
        self.latentSpace_slice.extend(latent)
        ed.setdefault("latentSpace", []).append(torch.mean(torch.stack(latent), 0))
        score = float(np.mean([loss_diff.detach().cpu()]))          # AnomalyScoreReg = AnomalyScoreReco = AnomalyScoreComb = loss_diff
        ed.setdefault("AnomalyScoreRegPerVol", []).append(score)
        if not _cfg_get(self.cfg, "use_postprocessed_score", True):
            ed.setdefault("AnomalyScoreRecoPerVol", []).append(score)
            ed.setdefault("AnomalyScoreCombPerVol", []).append(score)
            ed.setdefault("AnomalyScoreCombiPerVol", []).append(score * score)
            ed.setdefault("AnomalyScoreCombPriorPerVol", []).append(score + _cfg_get(self.cfg, "beta", 0) * 0)
            ed.setdefault("AnomalyScoreCombiPriorPerVol", []).append(score * 0)

    def on_test_start(self):
        """reference :156-170: the bookkeeping `_test_step` / `_test_end` of the reference's evaluation write into. The metric code itself
        (src/utils/utils_eval.py: sklearn / monai / skimage) is outside the hot path and is used from the reference tree when this class
        runs inside it; standalone, the lists are created and test_step returns its reconstruction without the metric pass."""
        try:
            from src.utils.utils_eval import get_eval_dictionary  # type: ignore  (reference tree on sys.path)
            self.eval_dict = get_eval_dictionary()
        except ImportError:       # standalone: no eval_dict, test_step returns its tensors; a BROKEN reference install still raises
            pass
        self.inds, self.latentSpace_slice, self.diffs_list, self.seg_list = [], [], [], []
        self.new_size = [160, 190, 160]
        if not hasattr(self, "threshold"):
            self.threshold = {}

    def on_test_end(self):
        """reference :288-291: `_test_end(self)` of the reference's utils_eval when it is importable; nothing to aggregate otherwise"""
        try:
            from src.utils.utils_eval import _test_end  # type: ignore
        except ImportError:
            return
        _test_end(self)

    # ------------------------------------------------------------------ training (reference :114-135, :305-306)
    def hip_trainer(self, device):
        """the UNet's training state on the HIP operators (training.UNetTrainer). From here on the UNet module's parameters ARE views of
        the trainer's flat buffer: state_dict() / checkpoints see the trained values, and the evaluation path re-packs them on its next call."""
        if getattr(self, "_hip_unet_trainer", None) is None:
            from .training import UNetTrainer
            unet = self.diffusion.model
            ds, levels = 1, len(unet.channel_mult)
            for _lvl in range(levels):
                if ds in tuple(unet.attention_resolutions):
                    raise NotImplementedError("training: attention inside the resolution levels is not built (the cDDPM experiment has none: "
                                              "att_res [3, 6, 12] never matches ds in {1, 2, 4})")
                ds *= 2
            self._hip_unet_trainer = UNetTrainer({k: v for k, v in unet.state_dict().items()}, model_channels=unet.model_channels,
                                        channel_mult=tuple(unet.channel_mult), num_res_blocks=unet.num_res_blocks,
                                        cond_dim=unet.num_classes, device=device)
            self._alias_unet()
            self._load_pending_optimizer_state()
        return self._hip_unet_trainer

    def _alias_unet(self):
        """the UNet module's parameters become (again) views of the trainer's flat buffer. `module.to()` / `.cpu()` / `.float()` /
        `.half()` silently replace `param.data` (Lightning's teardown calls `.cpu()`): before every step the alias is verified, and a broken
        one is repaired in the direction that loses nothing -- the module's current values are copied into the flat buffer first."""
        tr_, changed = self._hip_unet_trainer, False
        for k, prm in self.diffusion.model.named_parameters():
            view = tr_.p[k]
            if prm.data_ptr() != view.data_ptr() or prm.device != view.device or prm.dtype != view.dtype:
                if getattr(self, "_aliased", False):            # was aliased before: the module holds the values the user sees
                    view.copy_(prm.data.detach().to(view.device, view.dtype))
                    changed = True
                prm.data = view
        versions = tuple(prm._version for prm in self.diffusion.model.parameters())
        if getattr(self, "_aliased", False) and (changed or versions != self._param_versions):
            tr_.parameters_changed()          # load_state_dict / a repaired alias wrote the flat buffer: exponents + packed images follow
        self._aliased, self._param_versions = True, versions

    def hip_encoder_trainer(self, device):
        """training state of the native context encoder (encoder_training.EncoderTrainer) when `self.encoder` is this package's ResNet-50
        (plain or inside SparK_2D_encoder, whose timm model carries drop_path_rate 0.05: spark/models.py:89-109); None for any other
        encoder module (it is then used as a frozen feature extractor). The module's parameters and BatchNorm buffers become views of the
        trainer's tensors, as for the UNet."""
        if getattr(self, "_hip_enc_trainer", None) is None:
            from .DDPM_encoder import ResNet50Encoder, SparK_2D_encoder
            from .encoder_training import EncoderTrainer
            enc = getattr(self, "encoder", None)
            spark = isinstance(enc, SparK_2D_encoder)
            core = enc.encoder if spark else enc
            if not isinstance(core, ResNet50Encoder):
                return None
            sd = {k: v for k, v in core.state_dict().items() if not k.endswith("num_batches_tracked")}
            # SparK_2D_encoder: build_encoder(..., drop_path_rate=cfg.get('dp', 0)) overrides timm's kwarg only when non-zero, the
            # resnet50 default of pre_train_d is 0.05 (spark/Spark_2D.py:277-282, spark/models.py:50, :91-93); plain timm resnet50: 0
            dp = float(_cfg_get(self.cfg, "dp", 0) or 0)
            self._hip_enc_trainer = EncoderTrainer(sd, self.hip_trainer(device), drop_path_rate=(dp if dp != 0 else 0.05) if spark else 0.0)
            self.__dict__["_enc_core"] = core          # NOT a registered submodule: state_dict() must keep the reference's key set
            self._alias_encoder()
            self._load_pending_optimizer_state()
        return self._hip_enc_trainer

    def _alias_encoder(self):
        """as _alias_unet, for the native encoder's parameters and BatchNorm running statistics"""
        et, core, changed = self._hip_enc_trainer, self._enc_core, False
        full = et.state_dict()
        tensors = [(k, t) for k, t in list(core.named_parameters()) + list(core.named_buffers()) if k in full]
        for k, prm in tensors:
            view = full[k]
            if prm.data_ptr() != view.data_ptr() or prm.device != view.device or prm.dtype != view.dtype:
                if getattr(self, "_enc_aliased", False):
                    view.copy_(prm.data.detach().to(view.device, view.dtype))
                    changed = True
                prm.data = view
        versions = tuple(t._version for _k, t in tensors)
        if getattr(self, "_enc_aliased", False) and (changed or versions != self._enc_versions):
            et.parameters_changed()
        self._enc_aliased, self._enc_versions = True, versions

    # ------------------------------------------------------------------ checkpoints: Adam's state lives in the trainers, not in torch.optim
    def hip_optimizer_state(self):
        """Adam moments + step count of the HIP trainers (None before the first training step)"""
        out = {}
        if getattr(self, "_hip_unet_trainer", None) is not None:
            out["unet"] = self._hip_unet_trainer.optimizer_state()
        if getattr(self, "_hip_enc_trainer", None) is not None:
            out["encoder"] = self._hip_enc_trainer.optimizer_state()
        return out or None

    def load_hip_optimizer_state(self, state):
        """restores what hip_optimizer_state returned; before the trainers exist it is kept and applied when they are created"""
        self._pending_opt_state = state
        self._load_pending_optimizer_state()

    def _load_pending_optimizer_state(self):
        st = getattr(self, "_pending_opt_state", None)
        if not st:
            return
        if "unet" in st and getattr(self, "_hip_unet_trainer", None) is not None:
            self._hip_unet_trainer.load_optimizer_state(st.pop("unet"))
        if "encoder" in st and getattr(self, "_hip_enc_trainer", None) is not None:
            self._hip_enc_trainer.load_optimizer_state(st.pop("encoder"))

    def on_save_checkpoint(self, checkpoint):
        """Lightning hook: `configure_optimizers` returns a torch Adam that is never stepped (manual optimisation on the HIP operators), so
        the checkpoint's `optimizer_states` is empty; the real Adam state (m, v, step count -- what the reference's checkpoints carry in
        `optimizer_states`) goes under its own key"""
        st = self.hip_optimizer_state()
        if st is not None:
            checkpoint["hip_optimizer_state"] = {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in st.items()}

    def on_load_checkpoint(self, checkpoint):
        st = checkpoint.get("hip_optimizer_state")
        if st is not None:
            self.load_hip_optimizer_state({k: dict(v) for k, v in st.items()})

    def _train_precision(self):
        """the Trainer's `precision` (the reference trains with 16: configs/trainer/default.yaml:7) or cfg.precision; None = leave the
        process default (CDDPM_TRAIN_PRECISION or fp32-grade)"""
        prec = _cfg_get(self.cfg, "precision", None)
        if prec is None:
            try:
                tr_ = getattr(self, "trainer", None)        # Lightning attaches it; a bare nn.Module base has none
            except Exception:
                tr_ = None
            prec = getattr(tr_, "precision", None) if tr_ is not None else None
        return prec

    def training_step(self, batch, batch_idx: int):
        """One optimisation step of the reference's training_step (:114-135): input = batch['vol'][DATA].squeeze(-1), context = encoder(input),
        noise = gen_noise(cfg) or Gaussian, loss = diffusion(input, cond, noise) at random t (cond_DDPM.py:647-655) -- with the gradient,
        the data-parallel all-reduce and Adam(lr = cfg.lr) (:305-306) done HERE on the HIP operators (training.training_step): this module
        runs under Lightning's manual optimisation (`automatic_optimization = False`), there is no autograd graph to hand back.
        The native context encoder (this package's ResNet-50, plain or SparK-wrapped) is trained jointly, in training mode (BatchNorm batch
        statistics, stochastic depth), as `optim.Adam(self.parameters())` does in the reference; any other `encoder=` module is used as a
        frozen feature extractor (`hip_trainer(...).dcond` holds dL/d(context) for whoever trains it)."""
        from . import training as _training
        prec = self._train_precision()
        if prec is not None:
            _training.set_precision(prec)
        vol = batch["vol"]
        input = vol["data"].squeeze(-1).float()              # torchio's DATA key is the string "data"
        dev = input.device
        trainer = self.hip_trainer(dev)
        self._alias_unet()                 # the module's parameters must still BE the trainer's (a .cpu() / load_state_dict in between?)
        enc_trainer = self.hip_encoder_trainer(dev) if _cfg_get(self.cfg, "condition", True) else None
        if enc_trainer is not None:
            self._alias_encoder()
        features = None if enc_trainer is not None else self(input)        # the native encoder is run (in training mode) by the step itself
        noise = self._gen_noise(input.shape, dev, engine=trainer.eng)      # simplex (the experiment's noisetype) on the trainer's own handle
        if noise is None:
            noise = torch.randn_like(input)
        d = self.diffusion
        t = torch.randint(0, d.num_timesteps, (input.shape[0],), device=dev).long()
        ddp = torch.distributed.is_available() and torch.distributed.is_initialized()
        loss = _training.training_step(trainer, input, None if features is None else features.float(), t=t, noise=noise.float(),
                                       timesteps=d.num_timesteps, encoder=enc_trainer,
                                       objective=d.objective, loss_type=d.loss_type, all_reduce=ddp, lr=_cfg_get(self.cfg, "lr", 1e-4),
                                       buffers={k: getattr(d, k) for k in ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                                                                           "p2_loss_weight")})
        d.model._hip.invalidate()
        if enc_trainer is not None:
            self._enc_core._key = None                    # the inference encoder re-reads the updated weights and running statistics
            for name, buf in self._enc_core.named_buffers():
                if name.endswith("num_batches_tracked"):  # BatchNorm's forward counter in training mode (torch: += 1 per forward)
                    buf += 1
        if hasattr(self, "log") and _Base is not nn.Module:
            try:
                self.log(f"{self.prefix}train/Loss", loss, prog_bar=False, on_step=False, on_epoch=True, batch_size=input.shape[0], sync_dist=True)
            except Exception:
                pass
        return {"loss": loss.detach()}

    @torch.no_grad()
    def validation_step(self, batch, batch_idx: int):
        """reference :137-155: the training loss on a validation batch -- context = encoder(input), `gen_noise` or Gaussian noise, a random
        timestep per slice, `loss, reco = self.diffusion(input, cond=features, noise=noise)` -- forward only, on the HIP path"""
        input = batch["vol"]["data"].squeeze(-1).float()
        features = self(input)
        noise = self._gen_noise(input.shape, input.device)
        with torch.autocast("cuda", enabled=False):
            loss, _reco = self.diffusion(input, cond=features, noise=noise)
        if hasattr(self, "log") and _Base is not nn.Module:
            try:
                self.log(f"{self.prefix}val/Loss_comb", loss, prog_bar=False, on_step=False, on_epoch=True, batch_size=input.shape[0], sync_dist=True)
            except Exception:
                pass
        return {"loss": loss}

    def update_prefix(self, prefix):
        """reference :308"""
        self.prefix = prefix

    @property
    def automatic_optimization(self):        # Lightning: training_step above steps the optimizer itself
        return False

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=_cfg_get(self.cfg, "lr", 1e-4))
