"""Times the training step (training.py: forward + loss + backward + Adam, all on the HIP operators) on one GPU.
    python tools/train_step_bench.py [--batch 16] [--size 128] [--steps 5] [--warmup 2]
    python tools/train_step_bench.py --gpus N            (starts the N ranks itself: the line below as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_step_bench.py   (data-parallel, RCCL)
BASELINE config 5 is 128x1x128x128 over 8 GPUs data-parallel = 16 slices per GPU and step (weak scaling: --batch is per rank)."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "conditioned-diffusion-models-uad_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--encoder", action="store_true", help="train the context encoder (ResNet-50) jointly, as the reference does")
    ap.add_argument("--phases", action="store_true", help="time forward / backward / adam separately (synchronises between them)")
    ap.add_argument("--gpus", type=int, default=1, help="N > 1 without RANK in the environment: start the N ranks (torchrun child) and relay rank 0's line")
    a = ap.parse_args()
    from bench import launch_plan, self_launch           # decided before anything touches the GPU
    plan = launch_plan(a.gpus, sys.argv[1:], os.environ, script=os.path.abspath(__file__))
    if plan is not None:
        self_launch(plan, os.environ)
    import torch
    tr = importlib.import_module(PKG + ".training")
    synth = importlib.import_module(PKG + ".synth")
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    ddp = "RANK" in os.environ                   # under torchrun (also with one rank: the RCCL path runs)
    if ddp:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    sd = synth.synth_state_dict(0)
    trainer = tr.UNetTrainer({k: torch.from_numpy(v) for k, v in sd.items()}, device=dev)
    enc = None
    if a.encoder:
        et = importlib.import_module(PKG + ".encoder_training")
        enc = et.EncoderTrainer({k: torch.from_numpy(v) for k, v in synth.synth_encoder_state_dict(0).items()}, trainer, drop_path_rate=0.05)
    B, S, T = a.batch, a.size, 1000
    x01 = torch.from_numpy(synth.synth_slices(1, rank * B, B, S, S)).reshape(B, 1, S, S).to(dev)      # each rank its own slices
    cond = torch.from_numpy(synth.synth_cond(1, rank * B, B)).to(dev)
    noise = torch.from_numpy(synth.noise_xT(1, rank * B, B, S, S)).reshape(B, 1, S, S).to(dev)
    t = torch.tensor([(137 * (i + 1)) % T for i in range(B)], dtype=torch.long, device=dev)
    losses = []
    for _ in range(a.warmup):
        losses.append(float(tr.training_step(trainer, x01, cond, t=t, noise=noise, objective="pred_noise", loss_type="l2", all_reduce=ddp, encoder=enc)))
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.training_step(trainer, x01, cond, t=t, noise=noise, objective="pred_noise", loss_type="l2", all_reduce=ddp, encoder=enc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    if ddp:
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    losses.append(float(loss))
    bits = tr.get_precision()
    arith = "fp32-emulated convolutions" if bits == 32 else "fp16-operand convolutions (precision 16)"
    res = {"workload": f"training step {B}x1x{S}x{S} (noise-pred MSE, {arith}, Adam" + (", context encoder trained jointly)" if enc else ")"), "ms_per_step": dt * 1e3,
           "precision": bits,
           "slices_per_s": world * B / dt, "n_gpus": world, "losses": losses, "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}
    if a.phases:
        buf = importlib.import_module(PKG + ".schedule").schedule_buffers(T)
        x0 = x01 * 2 - 1
        xt = (buf["sqrt_alphas_cumprod"].to(dev)[t].reshape(-1, 1, 1, 1) * x0 + buf["sqrt_one_minus_alphas_cumprod"].to(dev)[t].reshape(-1, 1, 1, 1) * noise)
        ph = {}
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = trainer.forward(xt, t, cond)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            _l, dout = trainer.loss_and_grad(out, noise, None, "l2")
            trainer.backward(dout)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            trainer.adam_step()
            torch.cuda.synchronize(); t3 = time.perf_counter()
            ph = {"forward_ms": (t1 - t0) * 1e3, "loss_backward_ms": (t2 - t1) * 1e3, "adam_repack_ms": (t3 - t2) * 1e3}
        res["phases"] = ph
    if rank == 0:
        print(json.dumps(res))
    if ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
