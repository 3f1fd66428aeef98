"""ORACLE tooling: float64 runs of the oracle on the golden loop cases (no reference involved).

The float64 result is the rounding-free yardstick: |reference_fp32 - fp64| is the reference's own rounding
noise, |HIP - fp64| is ours; two correct fp32 implementations differ by about the root-sum-square of the two.
    python oracle/make_fp64_yardstick.py        # ~15 min on 8 cores (config 1 dominates)
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
synth = importlib.import_module("conditioned-diffusion-models-uad_amd.synth")
import cddpm_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = [("loop_B2_32x32_T50_start0", 50, 0, 2, 32, 32, 0),
         ("loop_B1_128x128_T1000_start50", 1000, 50, 1, 128, 128, 0),
         ("loop_cfg1_B4_128x128_T50_start0", 50, 0, 4, 128, 128, 0)]

if __name__ == "__main__":
    sd64 = O.to_float64(O.to_torch_sd(synth.synth_state_dict(0)))
    for name, T, start_t, B, H, W, slice0 in CASES:
        t0 = time.time()
        buf64 = O.to_float64(O.schedule_buffers(T))
        x = torch.from_numpy(synth.noise_xT(2, slice0, B, H, W)).double()
        cond = torch.from_numpy(synth.synth_cond(1, slice0, B)).double()
        out = O.p_sample_loop(x, cond, sd64, buf64,
                              lambda t: torch.from_numpy(synth.noise_z(3, t, slice0, B, H, W)).double(), start_t=start_t)
        ref = np.load(os.path.join(GOLD, name + ".npz"))["out"]
        err = float(np.abs(ref - out.numpy()).max())
        np.savez_compressed(os.path.join(GOLD, name + "_fp64.npz"), out=out.numpy().astype(np.float64),
                            reference_fp32_vs_fp64_maxabs=np.float64(err))
        print(name, "reference fp32 vs fp64:", err, f"{time.time() - t0:.0f}s", flush=True)
