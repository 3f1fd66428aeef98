"""GPU: training-mode forward and backward of the context encoder (SURVEY.md section 8 rows f2 + f4; encoder_training.py over
csrc/encoder_train.hip) against float64 autograd through oracle/encoder_oracle.py's restatement of timm's ResNet-50 in training mode
(BatchNorm on batch statistics, stochastic depth as a given per-sample scale). PARITY UNPINNED like the encoder's forward: timm is not in
the image, so neither the reference nor a fixture pins these numbers; what is checked is this implementation against the restated
architecture (reference: src/models/modules/DDPM_encoder.py:21-23, trained by src/models/DDPM_2D.py:114-135)."""
import numpy as np
import pytest
import torch

from conftest import load_pkg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,drop", [(4, 64, 64, False), (3, 64, 96, True)])
def test_encoder_training_forward_and_all_gradients_vs_autograd(synth, B, H, W, drop):
    import encoder_oracle as eo
    tr, et = load_pkg("training"), load_pkg("encoder_training")
    dev = torch.device("cuda", 0)
    sd_np = synth.synth_encoder_state_dict(0)
    torch.manual_seed(B + H)
    x = torch.rand(B, 1, H, W)
    dcond = torch.randn(B, 128)
    scales = None
    if drop:          # two blocks with a dropped / rescaled residual branch for some samples
        scales = {"layer2.1": torch.tensor([1 / 0.9, 0.0, 1 / 0.9][:B]), "layer4.2": torch.tensor([0.0, 1 / 0.95, 1 / 0.95][:B])}
    sd = {k: torch.from_numpy(v).double() for k, v in sd_np.items()}
    for k, v in sd.items():
        if not ("running" in k):
            v.requires_grad_(True)
    ops = tr.UNetTrainer({"w": torch.zeros(64)}, device=dev)            # the handle and scratch arena the operators run on
    ops._fit(B, H, W)
    enc = et.EncoderTrainer({k: torch.from_numpy(v) for k, v in sd_np.items()}, ops)
    out = enc.forward(x.to(dev), drop_scales=scales)
    # the yardstick differentiates the branch of the network the implementation is on: its ReLU masks (see the oracle's docstring)
    nchw = lambda t: t.permute(0, 3, 1, 2).cpu() > 0
    masks = {"bn1": nchw(enc.saved["a0"])}
    for bk in enc.blocks:
        r = enc.saved[bk["name"]]
        masks[bk["name"] + ".bn1"], masks[bk["name"] + ".bn2"], masks[bk["name"] + ".out"] = nchw(r["a1"]), nchw(r["a2"]), nchw(r["out"])
    stats = {}
    ref = eo.resnet50_forward(x.double(), sd, training=True, drop_scales=scales, stats=stats, relu_masks=masks)
    ref.backward(dcond.double())
    e_out = float((out.double().cpu() - ref.detach()).abs().max() / ref.detach().abs().max())
    grads = enc.backward(dcond.to(dev))
    torch.cuda.synchronize()
    worst = []
    for k, v in sd.items():
        if v.grad is None:
            continue
        g = grads[k].double().cpu().reshape(v.grad.shape)
        assert torch.isfinite(g).all(), k
        worst.append((float((g - v.grad).abs().max() / (v.grad.abs().max() + 1e-30)), k))
    assert len(worst) == len(enc.p)
    import os
    if os.environ.get("CDDPM_GRAD_REPORT"):
        with open(os.environ["CDDPM_GRAD_REPORT"] + f".enc{B}", "w") as f:
            for e, k in worst:
                f.write(f"{e:.3e} {k}\n")
    worst.sort(reverse=True)
    e_run = max(float((enc.buf[k].double().cpu() - stats[k]).abs().max() / (stats[k].abs().max() + 1e-30)) for k in stats)
    print(f"context rel err {e_out:.2e}; running statistics rel err {e_run:.2e}; worst gradient rel errs",
          [(f"{e:.2e}", k) for e, k in worst[:4]], "median", float(np.median([e for e, _ in worst])))
    # scale of these numbers: torch's own fp32 forward of the same network (training-mode BatchNorm over 16 ... 4096 samples per channel,
    # 53 convolutions) is 5.6e-5 away from the float64 run on these inputs
    assert e_out < 2e-4 and e_run < 5e-5
    assert worst[0][0] < 1e-3 and float(np.median([e for e, _ in worst])) < 3e-4


def test_encoder_adam_steps_change_the_context(synth):
    tr, et = load_pkg("training"), load_pkg("encoder_training")
    dev = torch.device("cuda", 0)
    sd_np = synth.synth_encoder_state_dict(0)
    ops = tr.UNetTrainer({"w": torch.zeros(64)}, device=dev)
    ops._fit(4, 64, 64)
    enc = et.EncoderTrainer({k: torch.from_numpy(v) for k, v in sd_np.items()}, ops, drop_path_rate=0.05)
    x = torch.rand(4, 1, 64, 64, device=dev)
    target = torch.zeros(4, 128, device=dev)
    losses = []
    for _ in range(5):
        out = enc.forward(x)
        losses.append(float(((out - target) ** 2).mean()))
        enc.backward(2 * (out - target) / out.numel())
        enc.adam_step(lr=1e-3)
    print("encoder toy losses", losses)
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
