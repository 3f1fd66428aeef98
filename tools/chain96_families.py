import importlib, os, sys, json, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
PKG = "conditioned-diffusion-models-uad_amd"
synth, eng_mod, sched = (importlib.import_module(PKG + m) for m in (".synth", ".engine", ".schedule"))
NAME, B, H, W, T = "loop_full_B4_96x96_T1000_start0", 4, 96, 96, 1000
g = np.load(os.path.join(ROOT, "tests", "golden", NAME + ".npz"))
x = torch.from_numpy(synth.noise_xT(2, 0, B, H, W)).cuda(); cond = torch.from_numpy(synth.synth_cond(1, 0, B)).cuda()
noise = torch.empty((T, B, 1, H, W), dtype=torch.float32); noise[0] = 0
for t in range(1, T): noise[t] = torch.from_numpy(synth.noise_z(3, t, 0, B, H, W))
nz = noise.cuda()
for mb in (int(v) for v in sys.argv[1:]):
    e = eng_mod.CddpmEngine(timesteps=T, max_batch=mb, max_h=H, max_w=W)
    e.load_weights(synth.synth_state_dict(0)); e.set_schedule(sched.schedule_buffers(T), "pred_x0")
    out = e.reverse(x, cond, T, noise=nz).cpu().numpy()
    d = np.abs(out.astype(np.float64) - g["out"])
    print(json.dumps(dict(family=os.environ.get("CDDPM_CONV", "h3"), max_batch=mb, max=float(d.max()), rms=float(np.sqrt((d**2).mean())), n_over=int((d > 1e-4).sum()))), flush=True)
    e.close()
