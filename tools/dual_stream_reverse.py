#!/usr/bin/env python3
"""Experiment: the B = 64 reverse steps as TWO half-batches on two streams / two handles, enqueued step by step in alternation, against the
single-handle run. Slices are independent, so the results are the same bits; the question is whether one half's kernel tails and small
(GroupNorm finalize, attention, posterior) kernels hide behind the other half's convolutions.
    python tools/dual_stream_reverse.py [steps] [B]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "conditioned-diffusion-models-uad_amd"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S, T = 128, 1000
eng_mod, synth, sched = (importlib.import_module(PKG + "." + m) for m in ("engine", "synth", "schedule"))
dev = torch.device("cuda", 0)
sd, buf = synth.synth_state_dict(0), sched.schedule_buffers(T)


def make(b):
    e = eng_mod.CddpmEngine(timesteps=T, max_batch=B, max_h=S, max_w=S, device=dev)       # the B = 64 plan for every handle: same bits
    e.load_weights(sd); e.set_schedule(buf)
    return e


cond = torch.from_numpy(synth.synth_cond(1, 0, B)).to(dev)
one = make(B)
x0 = one.noise_fill(B, S, S, seed=2, stream_id=synth.STREAM_XT, slice0=0)
one.prepare_cond(cond, B)
xa = x0.clone()
one.reverse_range_(xa, T - 1, T - 5, seed=3)
torch.cuda.synchronize()
xa = x0.clone()
t0 = time.perf_counter()
one.reverse_range_(xa, T - 1, T - N, seed=3)
torch.cuda.synchronize()
t_one = (time.perf_counter() - t0) / N * 1e3

h = B // 2
ea, eb = make(h), make(h)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
xb = x0.clone()
halves = [(ea, sa, xb[:h], cond[:h], 0), (eb, sb, xb[h:], cond[h:], h)]
for e, s, x, c, s0 in halves:
    with torch.cuda.stream(s):
        e.prepare_cond(c.contiguous(), h)
torch.cuda.synchronize()


def run(n_hi, n):
    for t in range(n_hi, n_hi - n, -1):
        for e, s, x, c, s0 in halves:
            with torch.cuda.stream(s):
                e.reverse_range_(x, t, t, seed=3, slice0=s0)


xw = xb.clone()
run(T - 1, 5)
torch.cuda.synchronize()
xb.copy_(x0)
halves = [(ea, sa, xb[:h], cond[:h], 0), (eb, sb, xb[h:], cond[h:], h)]
torch.cuda.synchronize()
t0 = time.perf_counter()
run(T - 1, N)
torch.cuda.synchronize()
t_two = (time.perf_counter() - t0) / N * 1e3
print(f"B={B} {S}x{S}: one handle {t_one:.3f} ms per reverse step; two half-batches on two streams {t_two:.3f} ms "
      f"({(t_one / t_two - 1) * 100:+.1f} %); identical bits: {bool(torch.equal(xa, xb))}")
