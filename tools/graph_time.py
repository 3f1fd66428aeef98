#!/usr/bin/env python3
"""cddpm_reverse with (CDDPM_GRAPH=1) and without HIP-graph replay of the step, at B = 64, 4, 1 (128x128). On the GPU box:
    python tools/graph_time.py"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.getcwd())
PKG = "conditioned-diffusion-models-uad_amd"
synth = importlib.import_module(PKG + ".synth"); sched = importlib.import_module(PKG + ".schedule"); eng_mod = importlib.import_module(PKG + ".engine")
for B in (64, 4, 1):
    eng = eng_mod.CddpmEngine(timesteps=1000, max_batch=B, max_h=128, max_w=128)
    eng.load_weights(synth.synth_state_dict(0)); eng.set_schedule(sched.schedule_buffers(1000))
    x = torch.from_numpy(synth.noise_xT(2, 0, B, 128, 128)).cuda(); cond = torch.from_numpy(synth.synth_cond(1, 0, B)).cuda()
    for mode in ("0", "1", "0", "1"):
        os.environ["CDDPM_GRAPH"] = mode
        eng.reverse(x, cond, 6)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.reverse(x, cond, 30)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"B={B} graph={mode} {dt / 30 * 1e3:.3f} ms/step", flush=True)
    del eng
