#!/usr/bin/env python3
"""bench.py -- reconstructed 128x128 slices/sec @ T=1000 of the cDDPM reverse-diffusion path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--size S] [--no-cpu] [--no-alt] [--no-profile]
        (N > 1 without RANK in the environment: starts the N ranks itself -- the torchrun line below as a child process --
         and relays rank 0's JSON line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): the reference UNet configuration (43.87 M parameters, fp32), a batch of 64 synthetic
single-channel 128x128 slices per GPU, T = 1000: the reference's p_sample_loop (cond_DDPM.py:446-464) executed by
cddpm_reverse of libcddpm_hip.so, z_t drawn on the device (Philox), x_T / weights / context synthetic (synth.py) and
resident in HBM before timing.

What a bench "step" is: a SEGMENT of 50 consecutive reverse steps (1/20 of a reconstruction, one cddpm_reverse_range call's
worth of chain). The timed region executes exactly K of them back to back -- K = 20 (the default, and what the round
driver passes) is ONE COMPLETE reconstruction x_T -> x_0 in [0,1] of the whole batch: a single cddpm_reverse(t_start =
1000) call, ~45 s of sustained load, followed (under torchrun) by the path's one collective, the all_gather of the
reconstructions, inside the timed region. Other K: K*50 reverse steps = floor(K/20) complete reconstructions
plus the head of the next chain.
    value [slices/s] = n_gpus * B * (K * 50 / 1000) / seconds        (max over ranks, barrier + synchronize both sides)
Untimed before it: W segments of warm-up. Untimed after it: a short pass with per-kernel HIP events (roofline), the
small-batch line, the alternative arithmetic paths and the CPU baseline (rank 0, one GPU only).

Extra objects on the JSON line:
  roofline      dominant kernel = the fused 3x3 convolution (conv_x6.hip: conv_split_kernel<9,8,2,true> and the
                folded-upsample form <4,8,2,true>). Default arithmetic: every fp32 operand split into two fp16 terms,
                three v_mfma_f32_16x16x32_f16 per product group (hi*hi + hi*mid + mid*hi), fp32 accumulation.
                achieved = EXECUTED 16-bit-pipe FLOPs (3 x the algorithmic fp32 FLOPs of the launches) / their HIP-event
                durations, events recorded on the launch stream, against the 2.5 PFLOP/s dense fp16 MFMA peak of
                MI355X_MICROARCH.md; the fp32-equivalent rate is reported beside it.
  cpu_baseline  oracle/cddpm_oracle.py (torch CPU restatement of the reference path, "port": the reference cannot travel)
                timed on this host over thread counts {8, 16, 32, all} x B {1, 4} (1 warm-up + 2 timed p_sample steps per point,
                median); the fastest point is re-sampled (5 timed steps): value = median, `range` = slowest .. fastest step.
  config.alt_paths   the same workload (10 reverse steps, B = 64) under CDDPM_CONV=f32 (strict fp32 MFMA,
                v_mfma_f32_32x32x2_f32 -- the arithmetic `north_star` names) and CDDPM_CONV=x6 (exact 3-term bf16 split),
                each in a child process (the family is chosen once per process); `h3_nb2`: the default family with the opt-in
                256-cout-workgroup plan on every step (cddpm_set_accumulation_switch(0): faster, two-level accumulation).
  config.small_batch the reference's real call shape (DDPM_2D.py:193: 4 slices per volume): B = 4, 50 reverse steps;
                config.small_batch_two_streams: the same as two half-batches on two streams (engine.reverse_two_streams, identical bits).
  config.training_step  BASELINE config 5's per-GPU share (16 x 1 x 128 x 128, noise-prediction MSE, Adam; the context encoder trained jointly,
                as the reference does): ms per optimisation step on the HIP operators (training.py), 1 warm-up + 3 timed steps.
                Carries `precision`, a `training_roofline` object per MFMA operator class (weight-gradient GEMMs; forward + input-gradient
                convolutions: HIP events on the launch stream over one extra step) and the per-class times of that step.
  config.training_step_precision16  the same step with precision 16 (plain fp16 operands, fp32 accumulation: the reference trainer's
                `precision: 16`; selected at run time as the DDPM_2D mirror does from the Trainer's precision).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "conditioned-diffusion-models-uad_amd"
T_TOTAL = 1000
SEG = 50                           # reverse steps per bench step
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 MFMA peak (= fp32 vector peak)
PEAK_16BIT_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak
CONV_MODE = {"f32": "f32", "x6": "x6"}.get(os.environ.get("CDDPM_CONV", ""), "h3")     # mirrors conv_mode() in csrc/conv_x6.hip
PEAK_HBM_TBPS = 8.0
FLOP_PER_SLICE_STEP = {128: 265.6e9, 96: 149.1e9, 256: 1075.1e9}     # the reference's operation count
# executed by this implementation: the FIRST conv of the two up ResBlocks ("nearest x2 upsample -> conv3x3", 2.416 + 9.664
# GMAC @128^2, SURVEY 8a block table: half of output_blocks.3.1 / .7.1) runs as four 2x2-tap convolutions of the
# low-resolution input = 4/9 of its multiplies; their second conv is a plain 3x3
EXECUTED_FRACTION = 1.0 - (2.416 + 9.664) * (5.0 / 9.0) / 132.79
BYTES_PER_SLICE_STEP_128 = 1.043e9                                    # SURVEY.md 8(d), fused-kernel model
DTYPE = {"h3": "f32_emulated_f16x3", "x6": "f32_emulated_bf16x6", "f32": "f32"}[CONV_MODE]
ARITH = {"h3": "fp32 in, fp32 out, fp32 accumulation; convolution products formed from two-term fp16 splits of both operands "
               "(|x - hi - mid| <= 2^-23 |x| for |x| >= 2^-2, absolute <= 2^-25 below; weights pre-scaled by a power of two) "
               "on the fp16 MFMA, 3 of 4 partial products (the dropped mid*mid term <= 2^-22 |ab|); domain |activation| < 65504 "
               "(beyond it the result is NaN and the engine raises); parity bar 1e-4 vs the fp32 reference holds at full length",
         "x6": "fp32 in, fp32 out, fp32 accumulation; convolution products formed from exact 3-way bf16 splits of both operands "
               "on the bf16 MFMA (6 of 9 partial products, the rest < 2^-24 relative); no range limit",
         "f32": "fp32 MFMA (v_mfma_f32_32x32x2_f32) throughout: exact fp32 products"}[CONV_MODE]


def cpu_baseline(synth, size: int):
    """time the oracle (CPU restatement) on this host: bounded sample, NOT the thing shipped or measured. The host is shared (the
    figure moves by +-20 % with its other tenants): every point is the MEDIAN of its timed steps, the winning point is re-sampled, and
    the line carries the range next to the value -- quote the range, not one ratio."""
    import statistics
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cddpm_oracle as O
    sd = O.to_torch_sd(synth.synth_state_dict(0))
    buf = O.schedule_buffers(T_TOTAL)
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    points, best = [], None
    t_begin = time.perf_counter()

    def sample(threads, batch, nsteps):
        torch.set_num_threads(threads)
        x = torch.from_numpy(synth.noise_xT(2, 0, batch, size, size))
        cond = torch.from_numpy(synth.synth_cond(1, 0, batch))
        times = []
        with torch.no_grad():
            for i in range(1 + nsteps):
                t = T_TOTAL - 1 - i
                z = torch.from_numpy(synth.noise_z(3, t, 0, batch, size, size))
                t0 = time.perf_counter()
                x = O.p_sample(x, t, cond, sd, buf, z)
                times.append(time.perf_counter() - t0)
        return times[1:]                                          # the first step is the warm-up

    try:
        for threads in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu), default_threads}):
            for batch in (1, 4):
                if time.perf_counter() - t_begin > 45.0:          # bounded: the sweep never takes more than about a minute
                    break
                per_step = statistics.median(sample(threads, batch, 2))
                pt = {"threads": threads, "batch": batch, "s_per_slice_step": per_step / batch, "slices_per_s": batch / (T_TOTAL * per_step)}
                points.append(pt)
                if best is None or pt["slices_per_s"] > best["slices_per_s"]:
                    best = pt
        resampled = sample(best["threads"], best["batch"], 5)      # the winner again: five more steps
    finally:
        torch.set_num_threads(default_threads)
    rates = sorted(best["batch"] / (T_TOTAL * s_) for s_ in resampled)
    value = statistics.median(rates)
    return {"value": value, "unit": "slices/s", "cores": best["threads"], "kind": "port",
            "range": [rates[0], rates[-1]],
            "sample": f"oracle p_sample at {size}x{size}; sweep: 1 warm-up + 2 timed steps (median) per point over threads x batch = "
                      f"{[(p['threads'], p['batch']) for p in points]}; the fastest point (threads={best['threads']}, B={best['batch']}) "
                      f"re-sampled with 1 warm-up + 5 timed steps: value = their median, range = slowest .. fastest step; extrapolated "
                      f"x{T_TOTAL} steps; host has {ncpu} logical CPUs",
            "s_per_slice_step": 1.0 / (T_TOTAL * value), "points": points}


def make_engine(torch, dev, B, S):
    synth = importlib.import_module(PKG + ".synth")
    sched = importlib.import_module(PKG + ".schedule")
    eng_mod = importlib.import_module(PKG + ".engine")
    eng = eng_mod.CddpmEngine(timesteps=T_TOTAL, max_batch=B, max_h=S, max_w=S, device=dev)
    eng.load_weights(synth.synth_state_dict(0))
    eng.set_schedule(sched.schedule_buffers(T_TOTAL))
    if os.environ.get("CDDPM_BENCH_ACCUM_SWITCH"):            # the opt-in two-level accumulation plan (alt path h3_nb2; profiling runs)
        eng.set_accumulation_switch(int(os.environ["CDDPM_BENCH_ACCUM_SWITCH"]))
    return eng, synth


def run_chain(eng, synth, x, n_rev, slice0):
    """n_rev reverse steps on x in place: complete reconstructions from fresh x_T, then the head of the next chain"""
    done = 0
    while done < n_rev:
        n = min(T_TOTAL, n_rev - done)
        if done:
            x.copy_(eng.noise_fill(x.shape[0], x.shape[2], x.shape[3], seed=2 + done, stream_id=synth.STREAM_XT, slice0=slice0))
        eng.reverse_range_(x, T_TOTAL - 1, T_TOTAL - n, seed=3, slice0=slice0)
        done += n
    return x


def training_rate(torch, dev, synth, B=16, S=128, steps=3, precision=32):
    """one optimisation step (training.training_step: q_sample, UNet forward / backward, encoder forward / backward, guarded Adam) on
    B x 1 x S x S, in the arithmetic `precision` selects (32: fp32-grade two-term fp16 splits; 16: plain fp16 operands, fp32 accumulation --
    the reference trainer's `precision: 16`, configs/trainer/default.yaml:7, which also serves BASELINE config 5's "bf16": see
    training.set_precision). Then one more step with per-operator HIP events: the training_roofline object."""
    tr, et = importlib.import_module(PKG + ".training"), importlib.import_module(PKG + ".encoder_training")
    bits = tr.set_precision(precision)
    try:
        trainer = tr.UNetTrainer({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(0).items()}, device=dev)
        enc = et.EncoderTrainer({k: torch.from_numpy(v) for k, v in synth.synth_encoder_state_dict(0).items()}, trainer, drop_path_rate=0.05)
        x01 = torch.from_numpy(synth.synth_slices(1, 0, B, S, S)).reshape(B, 1, S, S).to(dev)
        noise = torch.from_numpy(synth.noise_xT(1, 0, B, S, S)).reshape(B, 1, S, S).to(dev)
        t = torch.tensor([(137 * (i + 1)) % 1000 for i in range(B)], dtype=torch.long, device=dev)
        kw = dict(t=t, noise=noise, objective="pred_noise", loss_type="l2", encoder=enc)
        losses = [float(tr.training_step(trainer, x01, None, **kw))]
        tr.training_step(trainer, x01, None, **kw)             # second warm-up: allocator and scratch arena at their final size
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.training_step(trainer, x01, None, **kw)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        losses.append(float(loss))
        # per-operator-class HIP events over one more step (outside the timed steps)
        # the timed steps above run the weight gradients on a side stream beside the main stream's chain; for per-class kernel times the
        # profiled step runs everything on ONE stream (concurrent streams would charge each class the other's time)
        overlap = getattr(trainer, "overlap_wgrad", False)
        trainer.overlap_wgrad = False
        trainer.eng.set_profiling(True)
        tr.training_step(trainer, x01, None, **kw)
        torch.cuda.synchronize(dev)
        trainer.eng.set_profiling(False)
        trainer.overlap_wgrad = overlap
        prof = trainer.eng.get_profile()
        skipped = trainer.skipped_steps
        trainer.close()
    finally:
        tr.set_precision(32)
    nprod = 3.0 if bits == 32 else 1.0                # MFMAs executed per product group: hi*hi + hi*mid + mid*hi, or hi*hi alone
    roof = {}
    for cls, what in (("wgrad", "conv weight gradients: k-image passes + fp16 GEMM (conv_wgrad_img_kernel) + partial-tile folds"),
                      ("conv3x3_mfma", "forward + input-gradient 3x3 convolutions (conv_split_kernel on the forward / transposed image)")):
        c = prof[cls]
        if c["ms"] > 0:
            ach = c["flops"] / (c["ms"] * 1e-3) / 1e12
            roof[cls] = {"bound": "mfma", "kernel": what, "achieved": nprod * ach, "peak": PEAK_16BIT_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": nprod * ach / PEAK_16BIT_MFMA_TFLOPS, "fp32_equivalent_tflops": ach, "ms_per_step": c["ms"], "launches": c["launches"],
                         "products_per_multiply": nprod}
    # (one-stream times: their sum exceeds the timed step by what the side stream hides)
    classes = {k: {"ms_per_step": v["ms"], "launches": v["launches"]} for k, v in prof.items() if v["launches"]}
    return {"workload": f"{B}x1x{S}x{S}, noise-pred MSE, Adam, UNet + context encoder", "ms_per_step": dt * 1e3, "slices_per_s": B / dt,
            "precision": bits, "dtype": ("f32_emulated_f16x3 (convolutions), f32 elsewhere" if bits == 32 else
                                         "f16 operands with f32 accumulation (convolutions: the reference trainer's precision 16), f32 elsewhere"),
            "losses_first_last": losses, "skipped_steps": skipped, "training_roofline": roof, "kernel_classes": classes,
            "weight_gradients_on_side_stream": bool(overlap),
            "profiled_step_ms_one_stream": sum(v["ms"] for v in prof.values())}


def short_rate(torch, dev, B, S, n_rev, warm):
    """ms per reverse step of a fresh engine at batch B (alt-path children and the small-batch line)"""
    eng, synth = make_engine(torch, dev, B, S)
    cond = torch.from_numpy(synth.synth_cond(1, 0, B)).to(dev)
    x = eng.noise_fill(B, S, S, seed=2, stream_id=synth.STREAM_XT, slice0=0)
    eng.prepare_cond(cond, B)
    eng.reverse_range_(x, T_TOTAL - 1, T_TOTAL - warm, seed=3)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    eng.reverse_range_(x, T_TOTAL - 1 - warm, T_TOTAL - warm - n_rev, seed=3)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) * 1e3 / n_rev
    finite = bool(torch.isfinite(x).all().item())
    eng.close()
    return {"batch": B, "reverse_steps_timed": n_rev, "ms_per_reverse_step": ms, "slices_per_s": B / (T_TOTAL * ms * 1e-3),
            "finite": finite}


def launch_plan(gpus: int, argv, env, script=None):
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment is a request to run N ranks, not a rank: the
    command line of the launcher child (one process per GPU over RCCL, rendezvous on 127.0.0.1), or None when this process
    is itself a rank (under torchrun) or N == 1. Decided BEFORE anything touches the GPU: the launcher process never does."""
    if gpus <= 1 or "RANK" in env:
        return None
    import socket
    port = env.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def self_launch(cmd, env):
    """run the launcher child, relay rank 0's JSON line (the only thing the ranks write to stdout), exit with its code"""
    env = dict(env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    if r.returncode != 0 or not lines:
        raise SystemExit(r.returncode or 1)
    raise SystemExit(0)


def residual_workload(args, torch, dev, rank, world, dist, result_fd):
    """BASELINE.json configs[3] ("full IXI-shaped volume eval: 8192 synthetic 128x128 slices sharded over 8 MI355X, RCCL gather of residual
    maps"): N slices in contiguous blocks per rank (sharding.shard_range), each rank reconstructs its block in chunks of `--chunk` slices
    (x_T and z_t drawn on the device, keyed by the GLOBAL slice index) and keeps |x - reconstruction| on the device; ONE all_gather of the
    residual maps at the end, inside the timed region. value = N / seconds (max over ranks).
        python bench.py --gpus 8 --workload residual --slices 8192              # the configuration as named (about 12 minutes)
        python bench.py --workload residual --slices 128 --t-start 50           # a one-GPU smoke of the same code path"""
    sharding = importlib.import_module(PKG + ".sharding")
    N, S, chunk = args.slices, args.size, args.chunk
    eng, synth = make_engine(torch, dev, chunk, S)
    kw = dict(seed_inputs=4, seed_cond=1, seed_noise=3, t_start=args.t_start, chunk=chunk, gather="all")
    # warm-up: one small chunk through the same code (kernel first launches, RCCL communicator)
    sharding.residual_maps_sharded(eng, min(N, 2 * world), S, S, **dict(kw, t_start=min(args.t_start, 5)))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()

    def progress(s0, cnt):        # a chunk of 64 full-length reconstructions takes ~45 s: one stderr line per chunk from rank 0
        if rank == 0:
            print(f"[residual] rank 0: slices {s0} .. {s0 + cnt - 1} of its block, {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)

    res = sharding.residual_maps_sharded(eng, N, S, S, progress=progress, **kw)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ok = tuple(res.shape) == (N, 1, S, S) and bool(torch.isfinite(res).all().item()) and bool(((res >= 0) & (res <= 1)).all().item())
    lo, hi = sharding.shard_range(N, rank, world)
    out = {"metric": f"residual maps of {S}x{S} slices/sec @ {args.t_start} reverse steps", "value": N / elapsed, "unit": "slices/s",
           "n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": elapsed * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
           "config": {"workload": f"configs[3]: {N} synthetic {S}x{S} slices sharded over {world} GPU(s) in contiguous blocks, chunks of {chunk}, "
                                  f"{args.t_start} reverse steps each, one all_gather of |x - reconstruction| at the end (inside the timed region)",
                      "slices": N, "slices_this_rank": hi - lo, "chunk": chunk, "t_start": args.t_start, "size": S, "timed_region_s": elapsed,
                      "conv_family": CONV_MODE, "arithmetic": ARITH, "result_shape": list(res.shape), "finite_and_in_unit_range": ok,
                      "residual_mean": float(res.mean().item()),
                      "checksum_first_last": [float(res[0].double().sum().item()), float(res[-1].double().sum().item())]}}
    eng.close()
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("bench.py --workload residual: the residual maps are not finite / not in [0,1]")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="reverse", choices=("reverse", "residual"),
                    help="reverse: configs[1] (the headline); residual: configs[3], sharded residual maps + one gather")
    ap.add_argument("--slices", type=int, default=8192, help="--workload residual: total slices over all ranks")
    ap.add_argument("--t-start", type=int, default=1000, help="--workload residual: reverse steps per reconstruction")
    ap.add_argument("--chunk", type=int, default=64, help="--workload residual: slices per cddpm_reverse call")
    ap.add_argument("--steps", type=int, default=20, help="bench steps = segments of 50 reverse steps; 20 = one reconstruction")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="slices per GPU (configs[1]: 64)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-alt", action="store_true", help="skip the alternative-arithmetic children and the small-batch line")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP event pass (no roofline object)")
    ap.add_argument("--alt-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    plan = launch_plan(args.gpus, sys.argv[1:], os.environ)
    if plan is not None:
        self_launch(plan, os.environ)

    # stdout carries exactly ONE line (the JSON result): anything libraries print there (RCCL's version banner,
    # MIOpen/HIP notices) is sent to stderr for the duration of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B, S = args.batch, args.size

    if args.alt_child:      # one alternative arithmetic family, chosen by CDDPM_CONV in this child's environment
        r = short_rate(torch, dev, B, S, n_rev=10, warm=3)
        r["conv_family"] = CONV_MODE + ("_nb2" if os.environ.get("CDDPM_BENCH_ACCUM_SWITCH") == "0" else "")
        os.write(result_fd, (json.dumps(r) + "\n").encode())
        return

    dist = None
    if world > 1 or "RANK" in os.environ:     # under torchrun the RCCL path is exercised even with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL

    if args.workload == "residual":
        return residual_workload(args, torch, dev, rank, world, dist, result_fd)

    slice0 = rank * B                                   # weak scaling: every rank owns B distinct slices
    eng, synth = make_engine(torch, dev, B, S)
    cond = torch.from_numpy(synth.synth_cond(1, slice0, B)).to(dev)
    eng.prepare_cond(cond, B)
    gathered = torch.empty((world * B, 1, S, S), dtype=torch.float32, device=dev) if dist is not None else None

    # ---- warm-up: W segments on a scratch chain (also the first, eager, launch of every kernel)
    scratch = eng.noise_fill(B, S, S, seed=7, stream_id=synth.STREAM_XT, slice0=slice0)
    if args.warmup > 0:
        run_chain(eng, synth, scratch, args.warmup * SEG, slice0)
    if dist is not None:
        dist.all_gather_into_tensor(gathered, scratch)          # RCCL warm-up (communicator setup is not part of the metric)
    x = eng.noise_fill(B, S, S, seed=2, stream_id=synth.STREAM_XT, slice0=slice0)   # x_T on device
    n_rev = args.steps * SEG

    # ---- timed region: exactly K segments (K = 20: one complete reconstruction) + the path's one collective
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run_chain(eng, synth, x, n_rev, slice0)
    t_gather0 = None
    if dist is not None:
        torch.cuda.synchronize(dev)
        t_gather0 = time.perf_counter()
        dist.all_gather_into_tensor(gathered, x)                # gather of the per-rank results over xGMI
    torch.cuda.synchronize(dev)
    t_end_local = time.perf_counter()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gather_ms = (t_end_local - t_gather0) * 1e3 if t_gather0 is not None else 0.0

    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = bool(torch.isfinite(x).all().item())
    complete = (n_rev % T_TOTAL == 0)
    in_range = bool(((x >= 0) & (x <= 1)).all().item()) if complete else None

    value = world * B * (n_rev / T_TOTAL) / elapsed
    s_per_rev = elapsed / n_rev
    flop_step = FLOP_PER_SLICE_STEP.get(S, 265.6e9 * (S / 128.0) ** 2) * B
    workload = ("configs[1]" if (S == 128 and B == 64) else ("configs[2] geometry (256x256)" if S == 256 else "non-headline geometry"))
    out = {
        "metric": "reconstructed 128x128 slices/sec @ T=1000" if S == 128 else f"reconstructed {S}x{S} slices/sec @ T=1000",
        "value": value, "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": workload + f": reference UNet (43.87M params, fp32 weights), {B}x1x{S}x{S} slices per GPU, T={T_TOTAL}; "
                               f"one bench step = {SEG} consecutive reverse steps; timed region = {n_rev} reverse steps = "
                               f"{n_rev / T_TOTAL:g} complete reconstruction(s) x_T -> x_0 by cddpm_reverse"
                               + (" + all_gather of the results" if dist is not None else ""),
                   "reverse_steps_per_bench_step": SEG, "reverse_steps_timed": n_rev,
                   "complete_reconstructions_timed": n_rev / T_TOTAL, "timed_region_s": elapsed,
                   "ms_per_reverse_step": s_per_rev * 1e3,
                   "arithmetic": ARITH, "conv_family": CONV_MODE,
                   "batch_per_gpu": B, "global_batch": world * B, "size": S, "T": T_TOTAL,
                   "parallelism": f"slice-sharded x{world}, no collective in the loop, one all_gather at the end (inside the timed region)",
                   "gather_ms": gather_ms, "finite": finite, "reconstruction_in_unit_range": in_range,
                   "whole_step_tflops_fp32_equivalent": flop_step / s_per_rev / 1e12,
                   "whole_step_flops_note": "reference operation count (SURVEY 8d); executed count is x%.4f (folded upsample convs)" % EXECUTED_FRACTION,
                   "executed_tflops_fp32_equivalent": flop_step * EXECUTED_FRACTION / s_per_rev / 1e12,
                   "hbm_frac_fused_model": (BYTES_PER_SLICE_STEP_128 * (S / 128.0) ** 2 * B + 0.1755e9) / s_per_rev / (PEAK_HBM_TBPS * 1e12)},
    }

    # ---- per-kernel-class HIP events: a second, short pass outside the timed region
    if not args.no_profile:
        nprof = 6
        eng.set_profiling(True)
        eng.reverse_range_(scratch, T_TOTAL - 1, T_TOTAL - nprof, seed=3, slice0=slice0)
        torch.cuda.synchronize(dev)
        eng.set_profiling(False)
        prof = eng.get_profile()
        c3 = prof["conv3x3_mfma"]
        total_ms = sum(v["ms"] for v in prof.values())
        ach = c3["flops"] / (c3["ms"] * 1e-3) / 1e12 if c3["ms"] > 0 else 0.0
        traffic, traffic_src = None, None
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_conv3x3_hbm_traffic.json")))
        if tfiles:
            try:
                traffic = json.load(open(tfiles[-1])).get("bytes_per_launch")
                traffic_src = (f"profiles/{os.path.basename(tfiles[-1])} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, builder run of the same "
                               "command; not measured in this run)")
            except Exception:
                traffic = None
        if CONV_MODE != "f32":
            nprod = 3.0 if CONV_MODE == "h3" else 6.0
            kern = ("conv_split_kernel<9,8,%d> and <4,8,%d> (fused GN/FiLM/SiLU + 3x3 conv [+ folded upsample] + 1x1 skip + GN statistics; "
                    "fp32 operands as %s, %d %s per product group, fp32 accumulate); achieved = executed 16-bit-pipe FLOPs = %d x algorithmic"
                    % ((2, 2, "2 fp16 terms", 3, "v_mfma_f32_16x16x32_f16", 3) if CONV_MODE == "h3"
                       else (3, 3, "3 bf16 terms", 6, "v_mfma_f32_32x32x16_bf16", 6)))
            roof = {"achieved": nprod * ach, "peak": PEAK_16BIT_MFMA_TFLOPS, "frac": nprod * ach / PEAK_16BIT_MFMA_TFLOPS,
                    "fp32_equivalent_tflops": ach, "fp32_mfma_peak": PEAK_FP32_MFMA_TFLOPS,
                    "fp32_equivalent_over_fp32_mfma_peak": ach / PEAK_FP32_MFMA_TFLOPS}
        else:
            kern = "conv_mfma_kernel<9,4> and <4,4> (fused GN/FiLM/SiLU + 3x3 conv [+ folded upsample] + skip + GN statistics, fp32 MFMA); FLOPs = executed"
            roof = {"achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "frac": ach / PEAK_FP32_MFMA_TFLOPS}
        out["roofline"] = {"bound": "mfma", "kernel": kern, **roof, "unit": "TFLOP/s", "traffic": traffic, "traffic_source": traffic_src,
                           "launches": c3["launches"], "avg_launch_ms": c3["ms"] / max(1, c3["launches"]),
                           "flops_per_launch": c3["flops"] / max(1, c3["launches"]),
                           "algorithmic_bytes_per_launch": c3["bytes"] / max(1, c3["launches"]),
                           "share_of_profiled_step_time": c3["ms"] / total_ms if total_ms > 0 else None,
                           "measured_in": f"{nprof} reverse steps with HIP events on the launch stream, after the timed region"}
        out["kernel_classes"] = {k: {"ms_per_step": v["ms"] / nprof, "launches_per_step": v["launches"] / nprof,
                                     "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else 0.0,
                                     "algorithmic_GBps": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else 0.0}
                                 for k, v in prof.items()}
    eng.close()
    del eng, x, scratch
    torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_alt:
        # the reference's real call shape: 4 slices per volume (DDPM_2D.py:193)
        out["config"]["small_batch"] = short_rate(torch, dev, 4, S, n_rev=50, warm=10)
        try:        # the same call as two half-batches on two streams / two handles (engine.reverse_two_streams: identical bits)
            ea, synth_ = make_engine(torch, dev, 4, S)
            eb, _ = make_engine(torch, dev, 4, S)
            c4 = torch.from_numpy(synth_.synth_cond(1, 0, 4)).to(dev)
            x4 = ea.noise_fill(4, S, S, seed=2, stream_id=synth_.STREAM_XT, slice0=0)
            ea.reverse_two_streams(eb, x4, c4, 10, seed=3)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ea.reverse_two_streams(eb, x4, c4, 50, seed=3)
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) * 1e3 / 50
            out["config"]["small_batch_two_streams"] = {"batch": 4, "reverse_steps_timed": 50, "ms_per_reverse_step": ms,
                                                        "slices_per_s": 4 / (T_TOTAL * ms * 1e-3)}
            ea.close(); eb.close()
        except Exception as e:
            out["config"]["small_batch_two_streams"] = {"error": repr(e)}
        # the strict-fp32 and exact-bf16-split families on the same workload, one child process each
        alts = {}
        for fam in ("f32", "x6", "h3_nb2"):
            if fam == CONV_MODE or (fam == "h3_nb2" and CONV_MODE != "h3"):
                continue
            try:
                # h3_nb2: the default family with 256-cout workgroups on every step (cddpm_set_accumulation_switch(0); conv_x6.hip NB = 2:
                # faster, two-level accumulation -- opt-in because it costs accuracy at full length)
                env = dict(os.environ, CDDPM_BENCH_ACCUM_SWITCH="0") if fam == "h3_nb2" else dict(os.environ, CDDPM_CONV=fam)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--alt-child", "--batch", str(B), "--size", str(S)],
                                   env=env, capture_output=True, text=True, timeout=300)
                alts[fam] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-300:]}
            except Exception as e:      # the headline does not depend on the side measurements
                alts[fam] = {"error": repr(e)}
        out["config"]["alt_paths"] = alts
        try:
            out["config"]["training_step"] = training_rate(torch, dev, synth)
        except Exception as e:
            out["config"]["training_step"] = {"error": repr(e)}
        try:        # the same step in the reference trainer's precision-16 arithmetic (a runtime setting: cddpm_set_train_precision)
            out["config"]["training_step_precision16"] = training_rate(torch, dev, synth, precision=16)
        except Exception as e:
            out["config"]["training_step_precision16"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(synth, S)
        out["config"]["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        out["config"]["speedup_vs_cpu_baseline_range"] = [value / out["cpu_baseline"]["range"][1], value / out["cpu_baseline"]["range"][0]]
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()
    if not finite or in_range is False:
        raise SystemExit("bench.py: the reconstruction is not finite / not in [0,1] -- the number above is INVALID")


if __name__ == "__main__":
    main()
