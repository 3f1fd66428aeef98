#!/usr/bin/env python3
"""Where may the reverse loop switch from the two-level-accumulation convolution plan (256-cout workgroups, conv_x6.hip NB = 2) to the
three-level kernel? Runs the headline parity chain (B = 2 x 128 x 128 x T = 1000, explicit noise, the B = 64 handle's plan) for several
switch steps and prints the final image's deviation from the reference golden and from the float64 chain, and the state entering t = 50.
    python tools/accum_switch_sweep.py [t_switch ...]          (GPU box; reads tests/golden only)"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "conditioned-diffusion-models-uad_amd"
synth, eng_mod, sched = (importlib.import_module(PKG + m) for m in (".synth", ".engine", ".schedule"))
NAME, B, H, W, T = "loop_cfg2_B2_128x128_T1000_start0", 2, 128, 128, 1000
gold = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(gold, NAME + ".npz"))
ref, x50 = g["out"], g["x_t50"]
truth = np.load(os.path.join(gold, NAME + "_fp64.npz"))["out"]
e = eng_mod.CddpmEngine(timesteps=T, max_batch=64, max_h=H, max_w=W)
e.load_weights(synth.synth_state_dict(0)); e.set_schedule(sched.schedule_buffers(T), "pred_x0")
x = torch.from_numpy(synth.noise_xT(2, 0, B, H, W)).cuda(); cond = torch.from_numpy(synth.synth_cond(1, 0, B)).cuda()
noise = torch.empty((T, B, 1, H, W), dtype=torch.float32); noise[0] = 0
for t in range(1, T):
    noise[t] = torch.from_numpy(synth.noise_z(3, t, 0, B, H, W))
nz = noise.cuda()
rows = []
for ts in [int(v) for v in sys.argv[1:]] or [1000, 0, 100, 150, 200, 300, 500]:
    e.set_accumulation_switch(ts)
    e.prepare_cond(cond, B)
    img = x.clone()
    for t in range(T - 1, 50, -1):
        img = e.p_sample(img, t, None, z=nz[t])
    d50 = float(np.abs(img.cpu().numpy() - x50).max())
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    out = e.reverse(x, cond, T, noise=nz).cpu().numpy()
    dt = time.perf_counter() - t0
    d, d64 = np.abs(out.astype(np.float64) - ref), np.abs(out - truth)
    row = dict(t_switch=ts, x_t50_max=d50, vs_ref_max=float(d.max()), vs_ref_rms=float(np.sqrt((d ** 2).mean())), vs_ref_n_over=int((d > 1e-4).sum()),
               vs_fp64_max=float(d64.max()), vs_fp64_rms=float(np.sqrt((d64 ** 2).mean())), seconds=round(dt, 2))
    rows.append(row)
    print(json.dumps(row), flush=True)
out_dir = os.path.join(ROOT, "gpurun_out")
if os.path.isdir(out_dir):
    json.dump(rows, open(os.path.join(out_dir, "accum_switch_sweep.json"), "w"), indent=1)
