import hashlib, importlib, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
eng_mod = importlib.import_module("conditioned-diffusion-models-uad_amd.engine")
e = eng_mod.CddpmEngine(timesteps=10, max_batch=4, max_h=64, max_w=64)
torch.manual_seed(0)
for (C0, C1, Cout, H, W, coef, res, k) in [(128, 0, 128, 64, 64, True, False, 3), (256, 128, 256, 32, 40, True, True, 3), (256, 0, 256, 24, 24, False, False, 3), (256,0,768,32,32,True,False,1)]:
    B = 3
    x0 = torch.randn(B, H, W, C0, device="cuda")
    x1 = torch.randn(B, H, W, C1, device="cuda") if C1 else None
    cf = torch.stack([torch.randn(B, C0 + C1) * 0.2, 1 + 0.2 * torch.randn(B, C0 + C1), torch.randn(B, C0 + C1) * 0.2]).cuda().contiguous() if coef else None
    w = torch.randn(Cout, C0 + C1, k, k) / ((C0 + C1) * k * k) ** 0.5
    bias = torch.randn(Cout) * 0.1
    r = torch.randn(B, H, W, Cout, device="cuda") if res else None
    out = e.op_conv(x0, x1, cf, coef, 0, w, bias, r, False, k)
    torch.cuda.synchronize()
    # float64 reference
    xx = torch.cat([x0] + ([x1] if x1 is not None else []), -1).double().cpu()
    if coef:
        c = cf.double().cpu(); xx = (xx - c[0][:, None, None, :]) * c[1][:, None, None, :] + c[2][:, None, None, :]; xx = xx * torch.sigmoid(xx)
    ref = torch.nn.functional.conv2d(xx.permute(0, 3, 1, 2), w.double(), bias.double(), padding=k // 2).permute(0, 2, 3, 1)
    if r is not None: ref = ref + r.double().cpu()
    err = (out.double().cpu() - ref)
    print((C0, C1, Cout, H, W, k), os.environ.get("CDDPM_NB2"), hashlib.md5(out.cpu().numpy().tobytes()).hexdigest()[:8], "rms err / rms %.3e max %.3e" % (float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), float(err.abs().max())))
