"""CPU: the Hydra-less composer on the reference's own YAML files (read in place when the reference tree is
mounted; skipped on machines without it -- no config text is copied into this repository)."""
import os

import pytest
import torch

from conftest import load_pkg

REF_CFG = "/root/reference/configs"


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference tree not mounted")
def test_compose_cddpm_experiment():
    config = load_pkg("config")
    cfg = config.compose(REF_CFG, "cDDPM/DDPM_cond_spark_2D", overrides=["model.cfg.pretrained_encoder=False"])
    m = cfg["model"]["cfg"]
    assert cfg["model"]["_target_"] == "src.models.DDPM_2D.DDPM_2D"
    assert m["unet_dim"] == 128 and m["dim_mults"] == [1, 2, 2] and m["test_timesteps"] == 500
    assert m["imageDim"] == [192, 192, 100] and m["rescaleFactor"] == 2          # ${datamodule.cfg.*} resolved
    assert m["pretrained_encoder"] is False and m["noisetype"] == "simplex"
    enc = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.LazyLinear(128))
    mod = config.instantiate_model(cfg, encoder=enc)
    assert mod.diffusion.model.image_size == (96, 96) and mod.diffusion.num_timesteps == 1000
    assert mod.diffusion.model.channel_mult == (1, 2, 2) and mod.test_timesteps == 500
