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
`encoder=` module is supplied. `training_step` runs the UNet's optimisation step on the HIP operators (training.py; the encoder is
not updated). The scipy post-processing of utils_eval lives in utils_eval.py. pytorch_lightning / omegaconf are used when installed and replaced by
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

    def _gen_noise(self, shape, device):
        """`gen_noise(self.cfg, input.shape).to(self.device)` of the reference (:231, :241): a simplex field drawn on the
        device (generate_noise.py mirror, bit-exact for a given numpy seed) or None when cfg.noisetype is unset (the
        diffusion then draws Gaussian noise itself, cond_DDPM.py:577)."""
        if _cfg_get(self.cfg, "noisetype", None) is None:
            return None
        B, _c, H, W = shape
        return gen_noise(self.cfg, shape, engine=self.diffusion._engine(B, H, W, device))

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

        input = data_of("vol")                                              # [1,1,H,W,D]
        data_orig, data_seg, data_mask = data_of("vol_orig"), data_of("seg_orig"), data_of("mask_orig")
        if data_seg is None and data_orig is not None:
            data_seg = torch.zeros_like(data_orig)
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
        if data_orig is not None and data_mask is not None and hasattr(self, "eval_dict"):
            try:
                from src.utils.utils_eval import _test_step  # type: ignore  (reference tree on sys.path)
            except Exception:
                _test_step = None
            if _test_step is not None:
                _test_step(self, final_volume, data_orig, data_seg, data_mask, batch_idx, batch.get("ID"), batch.get("label"))
        return out

    def on_test_start(self):
        """reference :156-170: the bookkeeping `_test_step` / `_test_end` of the reference's evaluation write into. The metric code itself
        (src/utils/utils_eval.py: sklearn / monai / skimage) is outside the hot path and is used from the reference tree when this class
        runs inside it; standalone, the lists are created and test_step returns its reconstruction without the metric pass."""
        try:
            from src.utils.utils_eval import get_eval_dictionary  # type: ignore  (reference tree on sys.path)
            self.eval_dict = get_eval_dictionary()
        except Exception:
            pass
        self.inds, self.latentSpace_slice, self.diffs_list, self.seg_list = [], [], [], []
        self.new_size = [160, 190, 160]
        if not hasattr(self, "threshold"):
            self.threshold = {}

    def on_test_end(self):
        """reference :288-291: `_test_end(self)` of the reference's utils_eval when it is importable; nothing to aggregate otherwise"""
        try:
            from src.utils.utils_eval import _test_end  # type: ignore
        except Exception:
            return
        _test_end(self)

    # ------------------------------------------------------------------ training (reference :114-135, :305-306)
    def hip_trainer(self, device):
        """the UNet's training state on the HIP operators (training.UNetTrainer). From here on the UNet module's parameters ARE views of
        the trainer's flat buffer: state_dict() / checkpoints see the trained values, and the evaluation path re-packs them on its next call."""
        if getattr(self, "_trainer", None) is None:
            from .training import UNetTrainer
            unet = self.diffusion.model
            ds, levels = 1, len(unet.channel_mult)
            for _lvl in range(levels):
                if ds in tuple(unet.attention_resolutions):
                    raise NotImplementedError("training: attention inside the resolution levels is not built (the cDDPM experiment has none: "
                                              "att_res [3, 6, 12] never matches ds in {1, 2, 4})")
                ds *= 2
            self._trainer = UNetTrainer({k: v for k, v in unet.state_dict().items()}, model_channels=unet.model_channels,
                                        channel_mult=tuple(unet.channel_mult), num_res_blocks=unet.num_res_blocks,
                                        cond_dim=unet.num_classes, device=device)
            for k, prm in unet.named_parameters():
                prm.data = self._trainer.p[k]
        return self._trainer

    def hip_encoder_trainer(self, device):
        """training state of the native context encoder (encoder_training.EncoderTrainer) when `self.encoder` is this package's ResNet-50
        (plain or inside SparK_2D_encoder, whose timm model carries drop_path_rate 0.05: spark/models.py:89-109); None for any other
        encoder module (it is then used as a frozen feature extractor). The module's parameters and BatchNorm buffers become views of the
        trainer's tensors, as for the UNet."""
        if getattr(self, "_enc_trainer", None) is None:
            from .DDPM_encoder import ResNet50Encoder, SparK_2D_encoder
            from .encoder_training import EncoderTrainer
            enc = getattr(self, "encoder", None)
            spark = isinstance(enc, SparK_2D_encoder)
            core = enc.encoder if spark else enc
            if not isinstance(core, ResNet50Encoder):
                return None
            sd = {k: v for k, v in core.state_dict().items() if not k.endswith("num_batches_tracked")}
            self._enc_trainer = EncoderTrainer(sd, self.hip_trainer(device), drop_path_rate=0.05 if spark else 0.0)
            full = self._enc_trainer.state_dict()
            for k, prm in list(core.named_parameters()) + list(core.named_buffers()):
                if k in full:
                    prm.data = full[k]
            self._enc_core = core
        return self._enc_trainer

    def training_step(self, batch, batch_idx: int):
        """One optimisation step of the reference's training_step (:114-135): input = batch['vol'][DATA].squeeze(-1), context = encoder(input),
        noise = gen_noise(cfg) or Gaussian, loss = diffusion(input, cond, noise) at random t (cond_DDPM.py:647-655) -- with the gradient,
        the data-parallel all-reduce and Adam(lr = cfg.lr) (:305-306) done HERE on the HIP operators (training.training_step): this module
        runs under Lightning's manual optimisation (`automatic_optimization = False`), there is no autograd graph to hand back.
        The native context encoder (this package's ResNet-50, plain or SparK-wrapped) is trained jointly, in training mode (BatchNorm batch
        statistics, stochastic depth), as `optim.Adam(self.parameters())` does in the reference; any other `encoder=` module is used as a
        frozen feature extractor (`hip_trainer(...).dcond` holds dL/d(context) for whoever trains it)."""
        from . import training as _training
        vol = batch["vol"]
        input = vol["data"].squeeze(-1).float()              # torchio's DATA key is the string "data"
        dev = input.device
        trainer = self.hip_trainer(dev)
        enc_trainer = self.hip_encoder_trainer(dev) if _cfg_get(self.cfg, "condition", True) else None
        features = None if enc_trainer is not None else self(input)        # the native encoder is run (in training mode) by the step itself
        noise = self._gen_noise(input.shape, dev)
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
