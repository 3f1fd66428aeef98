"""GPU: the training step (host sequencing in training.py, kernels in csrc/train_kernels.hip + the forward
kernels) against torch autograd through the oracle's UNet in float64 on the same seeded inputs: the loss of p_losses and the gradient of
every one of the UNet's parameters (reference src/models/modules/cond_DDPM.py:565-655 with autograd; src/models/DDPM_2D.py:114-135),
then Adam (DDPM_2D.py:305-306)."""
import numpy as np
import pytest
import torch

from conftest import load_pkg

pytestmark = pytest.mark.gpu


def _inputs(synth, B, H, W, T, seed):
    x01 = torch.from_numpy(synth.synth_slices(seed, 0, B, H, W)).reshape(B, 1, H, W)
    cond = torch.from_numpy(synth.synth_cond(seed, 0, B))
    noise = torch.from_numpy(synth.noise_xT(seed, 0, B, H, W)).reshape(B, 1, H, W)
    t = torch.tensor([(137 * (i + 1) + seed) % T for i in range(B)], dtype=torch.long)
    return x01, cond, noise, t


def _loss_of(out, target, p2w, loss_type):
    d = out - target
    per = (d.abs() if loss_type == "l1" else d ** 2).reshape(d.shape[0], -1).mean(dim=1) * p2w
    return per.mean()


@pytest.mark.parametrize("B,H,W,objective,loss_type", [(2, 32, 32, "pred_x0", "l1"), (2, 16, 48, "pred_noise", "l2"),
                                                        (1, 96, 96, "pred_x0", "l1")])       # 96 x 96: the experiment's own slice size (24 x 24 = 576 tokens)
def test_loss_and_all_gradients_vs_autograd(oracle, synth, sd_np, B, H, W, objective, loss_type):
    """Yardstick: float64 autograd through the oracle's UNet. The L1 loss's derivative sign(out - target) is discontinuous, so the chain
    is checked in two links that share no ambiguity: (1) loss and dL/d(out) from the HIP loss kernel against autograd of the loss formula
    AT the HIP forward's own output; (2) the HIP backward against the oracle's vector-Jacobian product for that same dL/d(out)."""
    tr = load_pkg("training")
    T = 1000
    x01, cond, noise, t = _inputs(synth, B, H, W, T, 3)
    sd = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in sd_np.items()}
    buf64 = oracle.to_float64(oracle.schedule_buffers(T))
    x0 = x01 * 2 - 1
    ref_out = oracle.unet_forward(oracle.q_sample(x0.double(), t, noise.double(), buf64), t, cond.double(), sd)
    target = noise if objective == "pred_noise" else x0
    ref_loss = float(_loss_of(ref_out.detach(), target.double(), buf64["p2_loss_weight"][t], loss_type))

    dev = torch.device("cuda", 0)
    trainer = tr.UNetTrainer({k: torch.from_numpy(v).to(dev) for k, v in sd_np.items()}, device=dev)
    buf = load_pkg("schedule").schedule_buffers(T)
    xt = (buf["sqrt_alphas_cumprod"][t].reshape(-1, 1, 1, 1) * x0 + buf["sqrt_one_minus_alphas_cumprod"][t].reshape(-1, 1, 1, 1) * noise)
    out = trainer.forward(xt.to(dev), t.to(dev), cond.to(dev))
    assert float((out.double().cpu() - ref_out.detach()).abs().max()) < 2e-5
    # link 1: the loss kernel
    loss, dout = trainer.loss_and_grad(out, target.to(dev), buf["p2_loss_weight"][t].to(dev).contiguous(), loss_type)
    assert abs(float(loss) - ref_loss) < 2e-6 * max(1.0, abs(ref_loss))
    o64 = out.double().cpu().requires_grad_(True)
    _loss_of(o64, target.double(), buf64["p2_loss_weight"][t], loss_type).backward()
    S = trainer.grad_scale            # the loss scale of the backward pass (a power of two: exact)
    assert S == 2 ** round(np.log2(S)) and S >= B * H * W
    assert float((dout.double().cpu() / S - o64.grad).abs().max()) <= 1e-6 * float(o64.grad.abs().max())
    # link 2: the backward pass
    grads = trainer.backward(dout)
    torch.cuda.synchronize()
    ref_out.backward(dout.double().cpu() / S)
    ref_g = {k: v.grad for k, v in sd.items()}
    assert set(grads) == set(ref_g), (set(ref_g) - set(grads), set(grads) - set(ref_g))
    worst = []
    for k in sorted(ref_g):
        r = ref_g[k]
        g = grads[k].double().cpu().reshape(r.shape) / S
        assert torch.isfinite(g).all(), k
        worst.append((float((g - r).abs().max() / (r.abs().max() + 1e-30)), k))   # relative to the parameter's largest gradient entry
    import os
    if os.environ.get("CDDPM_GRAD_REPORT"):
        with open(os.environ["CDDPM_GRAD_REPORT"] + f".{loss_type}", "w") as f:
            for _kind, name, _a in trainer.program:
                for e, k in worst:
                    if k.startswith(name + "."):
                        f.write(f"{e:.3e} {k}\n")
    worst.sort(reverse=True)
    print("worst relative gradient errors:", [(f"{e:.2e}", k) for e, k in worst[:5]], "median", float(np.median([e for e, _ in worst])))
    assert worst[0][0] < 1e-4, worst[:5]
    assert float(np.median([e for e, _ in worst])) < 1e-5


def test_adam_update_vs_torch():
    """cddpm_op_adam against torch.optim.Adam (lr 1e-4, default betas / eps: DDPM_2D.py:305-306) over three steps"""
    tr = load_pkg("training")
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    w0 = torch.randn(4099)
    ref = torch.nn.Parameter(w0.clone().double())
    opt = torch.optim.Adam([ref], lr=1e-4)
    trainer = tr.UNetTrainer({"w": w0.clone()}, device=dev)
    for step in range(3):
        gr = torch.randn(4099) * (10.0 ** (step - 1))
        ref.grad = gr.double()
        opt.step()
        trainer.g["w"].copy_(gr.to(dev))
        trainer.adam_step(lr=1e-4, grad_scale=1.0)
    got = trainer.p["w"].double().cpu()
    assert float((got - ref.detach()).abs().max()) < 1e-6      # fp32 master weights of magnitude ~1 against a float64 optimizer


def test_training_steps_reduce_the_loss(synth, sd_np):
    """a few optimisation steps on one fixed batch (same t and noise) lower its loss: forward, loss, backward and Adam compose"""
    tr = load_pkg("training")
    dev = torch.device("cuda", 0)
    trainer = tr.UNetTrainer({k: torch.from_numpy(v).to(dev) for k, v in sd_np.items()}, device=dev)
    x01, cond, noise, t = _inputs(synth, 2, 32, 32, 1000, 5)
    losses = [float(tr.training_step(trainer, x01.to(dev), cond.to(dev), t=t.to(dev), noise=noise.to(dev), lr=1e-4)) for _ in range(4)]
    print("losses", losses)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_three_optimisation_steps_track_float64_training(oracle, synth, sd_np):
    """BASELINE config 5 end to end at small size: three complete optimisation steps (q_sample, UNet forward, noise-prediction MSE, backward,
    Adam lr 1e-4, device re-packing of the updated weights) against the same three steps done by float64 autograd through the oracle +
    torch.optim.Adam: the loss of every step and the parameters after the third. Adam's first steps move every parameter by ~lr * sign(g),
    so a parameter whose gradient is within rounding of zero may legitimately go the other way (2 lr per step): the maximum deviation is
    bounded by that, the mean deviation has to be far below it."""
    tr = load_pkg("training")
    T, B, H, W, lr, steps = 1000, 2, 32, 32, 1e-4, 3
    dev = torch.device("cuda", 0)
    buf64 = oracle.to_float64(oracle.schedule_buffers(T))
    sd = {k: torch.nn.Parameter(torch.from_numpy(v).double()) for k, v in sd_np.items()}
    opt = torch.optim.Adam(list(sd.values()), lr=lr)
    trainer = tr.UNetTrainer({k: torch.from_numpy(v) for k, v in sd_np.items()}, device=dev)
    ref_losses, got_losses = [], []
    for s in range(steps):
        x01, cond, noise, t = _inputs(synth, B, H, W, T, 11 + s)
        x0 = x01.double() * 2 - 1
        out = oracle.unet_forward(oracle.q_sample(x0, t, noise.double(), buf64), t, cond.double(), sd)
        loss = _loss_of(out, noise.double(), buf64["p2_loss_weight"][t], "l2")
        opt.zero_grad()
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
        got_losses.append(float(tr.training_step(trainer, x01.to(dev), cond.to(dev), t=t.to(dev), noise=noise.to(dev), timesteps=T,
                                                 objective="pred_noise", loss_type="l2", lr=lr)))
    print("losses float64:", ref_losses, "HIP:", got_losses)
    for a, b in zip(ref_losses, got_losses):
        assert abs(a - b) < 2e-5 * max(1.0, abs(a)), (ref_losses, got_losses)
    worst, mean_dev, moved = 0.0, [], []
    for k, v in sd.items():
        d = (trainer.p[k].double().cpu() - v.detach()).abs()
        worst = max(worst, float(d.max()))
        mean_dev.append(float(d.mean()))
        moved.append(float((v.detach() - torch.from_numpy(sd_np[k]).double()).abs().mean()))
    print(f"after {steps} steps: max parameter deviation {worst:.2e} (bound {2 * lr * steps:.1e}), mean deviation {np.mean(mean_dev):.2e}, "
          f"mean parameter movement {np.mean(moved):.2e}")
    assert worst <= 2.05 * lr * steps
    assert np.mean(mean_dev) < 0.02 * np.mean(moved)


def test_precision16_mode_gradients_are_fp16_grade():
    """CDDPM_TRAIN_PRECISION=16: the training operators multiply plain fp16 operands with fp32 accumulation (the arithmetic of the reference
    trainer's `precision: 16`, configs/trainer/default.yaml:7) -- a third of the MFMAs of the default fp32-grade split. Own process (the
    arithmetic is chosen once per process). Gradients then carry fp16 operand rounding (2^-11 per element, averaged over the contraction),
    not fp32 accuracy: bounded here at 2 % of a parameter's largest entry, median below 0.3 %."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "train_grad_check.py"), "2", "32", "32"],
                       env=dict(os.environ, CDDPM_TRAIN_PRECISION="16"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    print("precision 16:", res)
    assert res["finite"] and res["n"] == 316
    assert res["forward_max_abs_err"] < 5e-3 and res["worst"] < 2e-2 and res["median"] < 3e-3
    assert res["median"] > 1e-5          # (the mode is really on: fp32-grade arithmetic gives 2.5e-6 here)
