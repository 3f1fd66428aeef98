"""rocprofv3 target: N reverse steps at a small batch (default 4 x 128 x 128, the reference's real call shape) on a handle created for that batch
(split-K plan active).  usage: rocprofv3 --kernel-trace --stats -d out -o sb -- python3 tools/small_batch_profile.py [B] [S] [steps]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "conditioned-diffusion-models-uad_amd"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 50
eng_mod, synth, sched = (importlib.import_module(PKG + "." + m) for m in ("engine", "synth", "schedule"))
e = eng_mod.CddpmEngine(timesteps=1000, max_batch=B, max_h=S, max_w=S)
e.load_weights(synth.synth_state_dict(0))
e.set_schedule(sched.schedule_buffers(1000), "pred_x0")
x = torch.from_numpy(synth.noise_xT(2, 0, B, S, S)).reshape(B, 1, S, S).cuda()
cond = torch.from_numpy(synth.synth_cond(1, 0, B)).cuda()
e.reverse(x, cond, 3, noise=None, seed=3, slice0=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
e.reverse(x, cond, N, noise=None, seed=3, slice0=0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"B={B} {S}x{S}: {dt / N * 1e3:.3f} ms per reverse step = {B / (dt / N * 1000):.3f} slices/s at T=1000")
