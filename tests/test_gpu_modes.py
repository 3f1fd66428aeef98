"""GPU: the two alternative convolution families stay parity-green. The family is chosen once per process
(CDDPM_CONV, csrc/conv_x6.hip::conv_mode), so each one runs the kernel parity tests in a child process:
x6 = exact three-term bf16 split (six MFMAs per product group), f32 = fp32 MFMA (conv_mfma.hip).
The default family (two-term fp16 split) is what every other GPU test exercises."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["x6", "f32", "nb2"])
def test_alternative_conv_family(mode):
    # nb2: the opt-in 256-cout-workgroup plan of the default family (conv_x6.hip, NB = 2: the chunk's patch shared by two cout blocks,
    # two-level accumulation), forced onto the tests' small shapes
    env = dict(os.environ, CDDPM_NB2="force") if mode == "nb2" else dict(os.environ, CDDPM_CONV=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_kernels.py"), "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


def test_reverse_chain_with_alternative_families():
    """50-step loop at 32x32 against the reference golden in the x6 and f32 families (north-star bound 1e-4)."""
    code = r"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, %r)
PKG = "conditioned-diffusion-models-uad_amd"
synth = importlib.import_module(PKG + ".synth"); eng_mod = importlib.import_module(PKG + ".engine"); sched = importlib.import_module(PKG + ".schedule")
e = eng_mod.CddpmEngine(timesteps=50, max_batch=2, max_h=32, max_w=32)
e.load_weights(synth.synth_state_dict(0)); e.set_schedule(sched.schedule_buffers(50), "pred_x0")
B, H, W, T = 2, 32, 32, 50
x = torch.from_numpy(synth.noise_xT(2, 0, B, H, W)); cond = torch.from_numpy(synth.synth_cond(1, 0, B))
noise = np.zeros((T, B, 1, H, W), np.float32)
for t in range(1, T): noise[t] = synth.noise_z(3, t, 0, B, H, W)
out = e.reverse(x.cuda(), cond.cuda(), T, noise=torch.from_numpy(noise).cuda()).cpu().numpy()
ref = np.load(os.path.join(%r, "tests", "golden", "loop_B2_32x32_T50_start0.npz"))["out"]
err = float(np.abs(out - ref).max()); print("ERR", err); assert err < 1e-4, err
""" % (ROOT, ROOT)
    for mode in ("x6", "f32", "nb2"):
        env = dict(os.environ, CDDPM_NB2="force") if mode == "nb2" else dict(os.environ, CDDPM_CONV=mode)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ERR" in r.stdout, mode + ": " + r.stdout[-2000:] + r.stderr[-2000:]


def test_training_gradients_through_the_256_cout_workgroups():
    """the training operators use the 256-cout-workgroup convolution (two-level accumulation) wherever a call fills the chip with it; at
    the gradient test's small shapes only when forced -- loss and all 316 gradients against float64 autograd through it"""
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_training.py"), "-m", "gpu", "-x", "-q", "-k",
                        "(test_loss_and_all_gradients_vs_autograd and 32-32) or test_precision16_mode_gradients_are_fp16_grade",
                        "-p", "no:cacheprovider"], env=dict(os.environ, CDDPM_NB2="force"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]      # (the precision-16 child inherits the switch)
