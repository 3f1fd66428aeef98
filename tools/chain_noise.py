#!/usr/bin/env python3
"""Rounding noise of the reverse loop against the float64 goldens (tests/golden/*_fp64.npz) and the reference's fp32
goldens, for the library selected by CDDPM_LIB / CDDPM_CONV. The max-|delta| of a chaotic 50-step chain is a noisy
statistic; the rms against float64 is the stable one to compare kernel variants on.
    python tools/chain_noise.py            # on the GPU box
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "conditioned-diffusion-models-uad_amd"
GOLD = os.path.join(ROOT, "tests", "golden")

CASES = [("loop_cfg1_B4_128x128_T50_start0", 50, 0, 4, 128, 128, 0),
         ("loop_B2_32x32_T50_start0", 50, 0, 2, 32, 32, 0),
         ("loop_B1_128x128_T1000_start50", 1000, 50, 1, 128, 128, 0),
         ("loop_B2_32x32_T1000_start0", 1000, 0, 2, 32, 32, 0)]


def main():
    synth = importlib.import_module(PKG + ".synth")
    eng_mod = importlib.import_module(PKG + ".engine")
    sched = importlib.import_module(PKG + ".schedule")
    sd = synth.synth_state_dict(0)
    engines = {}
    for name, T, start_t, B, H, W, slice0 in CASES:
        if T not in engines:
            e = eng_mod.CddpmEngine(timesteps=T, max_batch=4, max_h=128, max_w=128)
            e.load_weights(sd)
            e.set_schedule(sched.schedule_buffers(T), "pred_x0")
            engines[T] = e
        eng = engines[T]
        steps = T if start_t == 0 else start_t
        x = torch.from_numpy(synth.noise_xT(2, slice0, B, H, W))
        cond = torch.from_numpy(synth.synth_cond(1, slice0, B))
        noise = np.zeros((steps, B, 1, H, W), np.float32)
        for t in range(1, steps):
            noise[t] = synth.noise_z(3, t, slice0, B, H, W)
        out = eng.reverse(x.cuda(), cond.cuda(), steps, noise=torch.from_numpy(noise).cuda()).cpu().numpy()
        ref = np.load(os.path.join(GOLD, name + ".npz"))["out"]
        truth = np.load(os.path.join(GOLD, name + "_fp64.npz"))["out"]
        r = lambda a, b: (float(np.abs(a - b).max()), float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2))))
        print(f"{name:36s} HIP-ref max {r(out, ref)[0]:.3e} rms {r(out, ref)[1]:.3e} | HIP-fp64 max {r(out, truth)[0]:.3e} "
              f"rms {r(out, truth)[1]:.3e} | ref-fp64 max {r(ref, truth)[0]:.3e} rms {r(ref, truth)[1]:.3e}")


if __name__ == "__main__":
    main()
