"""Soak run of the training step: N optimisation steps with fresh random t and noise per step on one fixed batch of synthetic slices, UNet +
context encoder, printing the loss every 10 steps (exercises the exponent refresh every 50 steps, the re-packing and the running statistics).
usage: python tools/train_soak.py [steps] [B] [S]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "conditioned-diffusion-models-uad_amd"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = int(sys.argv[3]) if len(sys.argv) > 3 else 96
tr, et, synth = (importlib.import_module(PKG + "." + m) for m in ("training", "encoder_training", "synth"))
dev = torch.device("cuda", 0)
trainer = tr.UNetTrainer({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(0).items()}, device=dev)
enc = et.EncoderTrainer({k: torch.from_numpy(v) for k, v in synth.synth_encoder_state_dict(0).items()}, trainer, drop_path_rate=0.05)
x01 = torch.from_numpy(synth.synth_slices(1, 0, B, S, S)).reshape(B, 1, S, S).to(dev)
torch.manual_seed(0)
acc = []
for step in range(N):
    t = torch.randint(0, 1000, (B,), device=dev)
    noise = torch.randn(B, 1, S, S, device=dev)
    loss = tr.training_step(trainer, x01, None, t=t, noise=noise, objective="pred_x0", loss_type="l1", lr=1e-4, encoder=enc)
    acc.append(float(loss))
    if (step + 1) % 10 == 0:
        print(f"step {step + 1:4d}  mean loss of the last 10: {sum(acc[-10:]) / 10:.4f}", flush=True)
assert all(l == l and abs(l) < 1e6 for l in acc), "non-finite loss"
print("first 10:", sum(acc[:10]) / 10, "last 10:", sum(acc[-10:]) / 10, "exponents", sorted(set(trainer.wexp.values())),
      "precision", tr.get_precision(), "optimizer steps", trainer.step_count, "skipped (non-finite gradients)", trainer.skipped_steps)
