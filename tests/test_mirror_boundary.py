"""Host logic of the DDPM_2D mirror at the boundary of the path (CPU): the pretrained-encoder ingest (reference
src/models/DDPM_2D.py:79-96), the state test_step hands to the reference's `_test_step` / `_test_end` (:171-286), the checkpoint hooks
that carry the HIP trainers' Adam state, the precision mapping."""
import sys
import types

import pytest
import torch

from conftest import load_pkg

CFG = dict(imageDim=[64, 64, 100], rescaleFactor=2, unet_dim=128, dim_mults=[1, 2, 2], condition=True, test_timesteps=500,
           noise_ensemble=True, spatial_transformer=False, backbone="Spark_Encoder_2D", version="resnet50", cond_dim=128)


def test_pretrained_encoder_key_rewrite():
    M = load_pkg("DDPM_2D")
    sd = {"sparse_encoder.sp_cnn.conv1.weight": 1, "sparse_encoder.sp_cnn.layer1.0.bn1.running_mean": 2,
          "sparse_encoder.sp_cnn.fc.weight": 3, "sparse_encoder.sp_cnn.fc.bias": 4,          # the pre-training head: dropped (:89)
          "dense_decoder.dec.0.weight": 5, "mask_token": 6,                                  # kept as they are (:92-93)
          "model.slice_encoder.encoder.layer2.0.conv1.weight": 7}                            # 'slice_encoder' + after the LAST 'encoder'
    out = M.rewrite_pretrained_encoder_keys(sd)
    assert out == {"encoder.conv1.weight": 1, "encoder.layer1.0.bn1.running_mean": 2, "dense_decoder.dec.0.weight": 5, "mask_token": 6,
                   "slice_encoder.layer2.0.conv1.weight": 7}
    assert list(out) == ["encoder.conv1.weight", "encoder.layer1.0.bn1.running_mean", "dense_decoder.dec.0.weight", "mask_token",
                         "slice_encoder.layer2.0.conv1.weight"]


def test_pretrained_encoder_is_loaded_into_the_spark_wrapper(tmp_path, synth):
    """a synthetic SparK pre-training checkpoint (keys `sparse_encoder.sp_cnn.*`, a 2048-wide pre-training fc, decoder debris): the
    backbone tensors land under encoder.encoder.*, the mirror's own fc (cond_dim outputs) is left alone, strict=False swallows the rest"""
    M = load_pkg("DDPM_2D")
    enc_sd = synth.synth_encoder_state_dict(5)
    ck = {"sparse_encoder.sp_cnn." + k: torch.from_numpy(v) for k, v in enc_sd.items() if not k.startswith("fc.")}
    ck["sparse_encoder.sp_cnn.fc.weight"] = torch.full((1000, 2048), 7.0)
    ck["sparse_encoder.sp_cnn.fc.bias"] = torch.full((1000,), 7.0)
    ck["dense_decoder.proj.weight"] = torch.zeros(3, 3)
    path = tmp_path / "spark.ckpt"
    torch.save({"state_dict": ck, "epoch": 3}, path)
    mod = M.DDPM_2D(dict(CFG, pretrained_encoder=True, encoder_path=str(path)))
    got = mod.encoder.state_dict()
    for k, v in enc_sd.items():
        if k.startswith("fc."):
            assert not torch.equal(got["encoder." + k], torch.full_like(got["encoder." + k], 7.0))       # head not overwritten
        else:
            assert torch.equal(got["encoder." + k], torch.from_numpy(v)), k
    missing, unexpected = mod.pretrained_encoder_keys
    assert set(missing) >= {"encoder.fc.weight", "encoder.fc.bias"} and "dense_decoder.proj.weight" in unexpected
    # the switch without a path is the reference's assert (:81); without the switch nothing is read
    with pytest.raises(AssertionError):
        M.DDPM_2D(dict(CFG, pretrained_encoder=True))
    plain = M.DDPM_2D(dict(CFG, encoder_path=str(path)))
    assert not hasattr(plain, "pretrained_encoder_keys")
    with pytest.raises(KeyError):
        torch.save({"weights": ck}, tmp_path / "bad.ckpt")
        M.DDPM_2D(dict(CFG, pretrained_encoder=True, encoder_path=str(tmp_path / "bad.ckpt")))


def _stub_utils_eval(monkeypatch, record):
    src, utils, ue = types.ModuleType("src"), types.ModuleType("src.utils"), types.ModuleType("src.utils.utils_eval")

    def get_eval_dictionary():
        return {"latentSpace": [], "AnomalyScoreRegPerVol": [], "AnomalyScoreRecoPerVol": [], "AnomalyScoreCombPerVol": [],
                "AnomalyScoreCombiPerVol": [], "AnomalyScoreCombPriorPerVol": [], "AnomalyScoreCombiPriorPerVol": []}

    def _test_step(self, final_volume, data_orig, data_seg, data_mask, batch_idx, ID, label):
        # what the reference's _test_step reads from `self` before anything else (utils_eval.py:77, :92, :101, :146, :176)
        record.append(dict(dataset=self.dataset[0], stage=self.stage, vol=tuple(final_volume.shape), seg_sum=float(data_seg.sum()),
                           ID=ID, label=label, n_latent=len(self.eval_dict["latentSpace"])))

    def _test_end(self):
        record.append(dict(end=True, n=len(self.eval_dict["AnomalyScoreRegPerVol"])))

    ue.get_eval_dictionary, ue._test_step, ue._test_end = get_eval_dictionary, _test_step, _test_end
    src.utils, utils.utils_eval = utils, ue
    for name, m in (("src", src), ("src.utils", utils), ("src.utils.utils_eval", ue)):
        monkeypatch.setitem(sys.modules, name, m)


def test_test_step_sets_what_the_reference_metric_code_reads(monkeypatch):
    M = load_pkg("DDPM_2D")
    record = []
    _stub_utils_eval(monkeypatch, record)
    enc = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.LazyLinear(128))
    mod = M.DDPM_2D(dict(CFG, noise_ensemble=False, use_postprocessed_score=False, beta=0.5), encoder=enc)

    class FakeDiffusion(torch.nn.Module):
        def forward(self, img, cond=None, t=None, noise=None):
            return torch.tensor(0.25), torch.full_like(img, 0.5)

    mod.diffusion = FakeDiffusion()
    mod.on_test_start()
    assert set(mod.eval_dict) >= {"latentSpace", "AnomalyScoreRegPerVol"} and mod.latentSpace_slice == [] and mod.new_size == [160, 190, 160]
    D = 6
    vol = torch.rand(1, 1, 32, 32, D)
    batch = {"vol": {"data": vol}, "vol_orig": {"data": vol.clone()}, "mask_orig": {"data": torch.ones_like(vol)},
             "seg_orig": {"data": torch.ones_like(vol)}, "seg_available": False, "Dataset": ["IXI"], "stage": "test", "ID": ["s1"],
             "label": ["healthy"], "age": [50]}
    mod.test_step(batch, 0)
    assert record[0] == dict(dataset="IXI", stage="test", vol=(1, 1, 32, 32, 4), seg_sum=0.0, ID=["s1"], label=["healthy"], n_latent=1)
    ed = mod.eval_dict
    assert ed["latentSpace"][0].shape == (128,) and len(mod.latentSpace_slice) == 1
    assert ed["AnomalyScoreRegPerVol"] == [0.25] and ed["AnomalyScoreRecoPerVol"] == [0.25] and ed["AnomalyScoreCombPerVol"] == [0.25]
    assert ed["AnomalyScoreCombiPerVol"] == [0.0625] and ed["AnomalyScoreCombPriorPerVol"] == [0.25] and ed["AnomalyScoreCombiPriorPerVol"] == [0.0]
    # seg_available: the segmentation is passed on (:179)
    mod.test_step(dict(batch, seg_available=True), 1)
    assert record[1]["seg_sum"] == 32 * 32 * 4 and record[1]["n_latent"] == 2
    mod.on_test_end()
    assert record[-1] == dict(end=True, n=2)


def test_a_broken_reference_install_is_not_swallowed(monkeypatch):
    """only ImportError (no reference tree on sys.path) means 'standalone'; any other failure of the reference's metric module surfaces"""
    M = load_pkg("DDPM_2D")

    class Boom(types.ModuleType):
        def __getattr__(self, name):
            raise RuntimeError("broken install")

    for name in ("src", "src.utils"):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    monkeypatch.setitem(sys.modules, "src.utils.utils_eval", Boom("src.utils.utils_eval"))
    mod = M.DDPM_2D(CFG, encoder=torch.nn.Identity())
    with pytest.raises(RuntimeError, match="broken install"):
        mod.on_test_start()


def test_checkpoint_hooks_carry_the_hip_optimizer_state():
    """no GPU here: the hooks' bookkeeping with stand-in trainers (the device round trip is tests/test_gpu_mirror.py)"""
    M = load_pkg("DDPM_2D")
    mod = M.DDPM_2D(CFG, encoder=torch.nn.Identity())
    ck = {}
    mod.on_save_checkpoint(ck)
    assert "hip_optimizer_state" not in ck                      # nothing trained yet: nothing to save

    class T:
        def __init__(self):
            self.loaded = None

        def optimizer_state(self):
            return {"m": torch.ones(3), "v": torch.zeros(3), "ctrl": torch.tensor([0, 7, 0, 1, 0, 0, 0, 0], dtype=torch.int32), "layout": [("a", 3)]}

        def load_optimizer_state(self, st):
            self.loaded = st

    mod._hip_unet_trainer = T()
    mod.on_save_checkpoint(ck)
    assert set(ck["hip_optimizer_state"]) == {"unet"} and int(ck["hip_optimizer_state"]["unet"]["ctrl"][1]) == 7
    fresh = M.DDPM_2D(CFG, encoder=torch.nn.Identity())
    fresh.on_load_checkpoint(ck)                               # before the trainer exists: kept ...
    assert fresh._pending_opt_state is not None and "unet" in fresh._pending_opt_state
    fresh._hip_unet_trainer = T()
    fresh._load_pending_optimizer_state()                      # ... and applied when it is created
    assert int(fresh._hip_unet_trainer.loaded["ctrl"][1]) == 7 and "unet" not in fresh._pending_opt_state


def test_precision_mapping(monkeypatch):
    tr = load_pkg("training")
    seen = []

    class Lib:
        def cddpm_set_train_precision(self, bits):
            seen.append(bits)
            return 32

    monkeypatch.setattr(load_pkg("_lib"), "load_library", lambda *a, **k: Lib())
    assert [tr.set_precision(p) for p in (32, "32-true", None, 16, "16-mixed", "bf16", "bf16-mixed")] == [32, 32, 32, 16, 16, 16, 16]
    assert seen == [32, 32, 32, 16, 16, 16, 16]
    with pytest.raises(ValueError):
        tr.set_precision("fp8")
    M = load_pkg("DDPM_2D")
    mod = M.DDPM_2D(dict(CFG, precision=16), encoder=torch.nn.Identity())
    assert mod._train_precision() == 16
    assert M.DDPM_2D(CFG, encoder=torch.nn.Identity())._train_precision() is None


def test_state_dict_keys_are_the_reference_prefixes_only():
    """encoder.* / diffusion.model.* / diffusion.<buffers> and nothing else -- also after the training plumbing attached itself"""
    M = load_pkg("DDPM_2D")
    mod = M.DDPM_2D(dict(CFG, backbone="resnet50"))
    mod.__dict__["_enc_core"] = mod.encoder          # what hip_encoder_trainer keeps: must not register a second copy
    mod._hip_unet_trainer, mod._hip_enc_trainer = None, None
    keys = list(mod.state_dict())
    assert all(k.startswith(("encoder.", "diffusion.")) for k in keys), [k for k in keys if not k.startswith(("encoder.", "diffusion."))][:3]
