"""GPU: the native context encoder (csrc/encoder.hip behind DDPM_encoder.ResNet50Encoder) against the
torch restatement of timm's ResNet-50 in oracle/encoder_oracle.py. PARITY UNPINNED with respect to the reference: timm is
not installed in the build image, so the restatement itself cannot be checked against `timm.create_model('resnet50',
in_chans=1, num_classes=128)`; what is tested is that the HIP kernels compute the restated network."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_pkg

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "oracle"))


@pytest.fixture(scope="module")
def enc_and_weights(synth):
    E = load_pkg("DDPM_encoder")
    w = synth.synth_encoder_state_dict(0, 128)
    enc = E.ResNet50Encoder(num_classes=128)
    missing, unexpected = enc.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    yield enc, {k: torch.from_numpy(v) for k, v in w.items()}
    enc.close()


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (3, 96, 96), (1, 128, 128), (2, 80, 96)])
def test_encoder_forward_vs_oracle(enc_and_weights, synth, B, H, W):
    import encoder_oracle as EO
    enc, sdt = enc_and_weights
    x = torch.from_numpy(synth.synth_slices(4, 0, B, H, W))
    ref = EO.resnet50_forward(x, sdt)
    out = enc(x.cuda()).cpu()
    scale = float(ref.abs().max())
    err = float((out - ref).abs().max())
    print(f"encoder {B}x{H}x{W}: max|delta| {err:.3e} (|ref| max {scale:.3e})")
    assert out.shape == (B, 128)
    assert err < 1e-4 * max(1.0, scale)
    # float64 yardstick: the fp32 oracle's own distance from a float64 run of the same network
    ref64 = EO.resnet50_forward(x.double(), {k: v.double() for k, v in sdt.items()})
    e_ref, e_hip = float((ref.double() - ref64).abs().max()), float((out.double() - ref64).abs().max())
    print(f"   vs float64: torch fp32 {e_ref:.3e}, HIP {e_hip:.3e}")
    assert e_hip < 10 * e_ref + 1e-6


def test_encoder_in_ddpm2d(enc_and_weights, synth):
    """DDPM_2D.forward(x) -> c with the SparK-wrapped native encoder (the experiment's configuration)"""
    M = load_pkg("DDPM_2D")
    cfg = dict(imageDim=[192, 192, 100], rescaleFactor=2, unet_dim=128, dim_mults=[1, 2, 2], condition=True,
               test_timesteps=500, backbone="Spark_Encoder_2D", version="resnet50")
    mod = M.DDPM_2D(cfg)
    _enc, sdt = enc_and_weights
    mod.encoder.encoder.load_state_dict(sdt, strict=False)
    x = torch.from_numpy(synth.synth_slices(4, 0, 2, 96, 96)).cuda()
    c = mod(x)
    assert c.shape == (2, 128) and bool(torch.isfinite(c).all())
    import encoder_oracle as EO
    assert float((c.cpu() - EO.resnet50_forward(x.cpu(), sdt)).abs().max()) < 1e-4 * max(1.0, float(c.abs().max()))
    with pytest.raises(RuntimeError):
        mod.encoder(x.cpu())
    mod.encoder.encoder.close()
