"""GPU: the `pred_noise` objective (cond_DDPM.py:411-414, :379-383, :612-644) and the linear beta schedule
(cond_DDPM.py:271-275) end to end through the C ABI, against the reference's own outputs
(oracle/make_golden_objectives.py -> tests/golden/pn_*.npz, lin_*.npz)."""
import numpy as np
import pytest
import torch

from conftest import golden, load_pkg

pytestmark = pytest.mark.gpu
TOL = 1e-4


def inputs(synth, B, H, W):
    return torch.from_numpy(synth.noise_xT(2, 0, B, H, W)), torch.from_numpy(synth.synth_cond(1, 0, B))


def explicit_noise(synth, steps, B, H, W):
    noise = np.zeros((steps, B, 1, H, W), np.float32)
    for t in range(1, steps):
        noise[t] = synth.noise_z(3, t, 0, B, H, W)
    return torch.from_numpy(noise).cuda()


@pytest.mark.parametrize("name,T,start_t,objective,kind", [
    ("pn_loop_B2_32x32_T1000_start8", 1000, 8, "pred_noise", "cosine"),
    ("pn_loop_B2_32x32_T50_start0", 50, 0, "pred_noise", "cosine"),
    ("lin_loop_B2_32x32_T1000_start8", 1000, 8, "pred_x0", "linear")])
def test_reverse_loop_objective_and_schedule(engine_factory, synth, name, T, start_t, objective, kind):
    """cddpm_reverse with step_kernel's pred_noise branch (x0 = sqrt_recip x - sqrt_recipm1 eps, clamped) / the linear
    schedule tables, explicit z_t, vs the reference's output"""
    B, H, W = 2, 32, 32
    eng = engine_factory(timesteps=T, max_batch=B, max_h=H, max_w=W, objective=objective, beta_schedule=kind)
    steps = T if start_t == 0 else start_t
    x, cond = inputs(synth, B, H, W)
    out = eng.reverse(x.cuda(), cond.cuda(), steps, noise=explicit_noise(synth, steps, B, H, W)).cpu().numpy()
    ref = golden(name)["out"]
    err = np.abs(out - ref).max()
    print(name, f"max|delta| vs reference golden: {err:.3e}")
    assert err < TOL and out.min() >= 0 and out.max() <= 1
    # the same chain replayed as a HIP graph (captures the pred_noise / linear tables too): bit-identical
    if start_t == 0:
        import os
        old = os.environ.get("CDDPM_GRAPH")
        os.environ["CDDPM_GRAPH"] = "1"
        try:
            again = eng.reverse(x.cuda(), cond.cuda(), steps, noise=explicit_noise(synth, steps, B, H, W)).cpu().numpy()
        finally:
            if old is None:
                os.environ.pop("CDDPM_GRAPH")
            else:
                os.environ["CDDPM_GRAPH"] = old
        assert np.array_equal(out, again)


@pytest.fixture(scope="module")
def diffusion_pn(sd_np):
    U, D = load_pkg("OpenAI_Unet"), load_pkg("cond_DDPM")
    m = U.UNetModel(image_size=(32, 32), in_channels=1, model_channels=128, out_channels=1, num_res_blocks=3,
                    attention_resolutions=(3, 6, 12), dropout=0, channel_mult=[1, 2, 2], conv_resample=True, dims=2,
                    num_classes=128, use_checkpoint=False, use_fp16=True, num_heads=1, num_head_channels=64,
                    num_heads_upsample=-1, use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True,
                    use_spatial_transformer=False, transformer_depth=1)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    d = D.GaussianDiffusion(m, image_size=(32, 32), timesteps=1000, sampling_timesteps=1000, objective="pred_noise",
                            channels=1, loss_type="l2", p2_loss_weight_gamma=0, cfg=None).cuda()
    yield d
    m._hip.close()


def test_single_step_forward_pred_noise_mirror(diffusion_pn, synth):
    """GaussianDiffusion.forward -> p_losses under pred_noise / l2: reco = unnormalize(x_t - sqrt(1 - abar_t) * eps_hat)"""
    g = golden("pn_p_losses_B2_32x32_t499")
    B, H, W = 2, 32, 32
    x01 = torch.from_numpy(synth.synth_slices(2, 0, B, H, W)).cuda()
    cond = torch.from_numpy(synth.synth_cond(1, 0, B)).cuda()
    noise = torch.from_numpy(synth.noise_z(3, 0, 0, B, H, W)).cuda()
    loss, reco = diffusion_pn(x01, t=499, cond=cond, noise=noise)
    assert np.abs(reco.cpu().numpy() - g["reco"]).max() < TOL
    assert abs(float(loss) - float(g["loss"])) < 1e-5


def test_p_sample_loop_pred_noise_mirror(diffusion_pn, synth):
    B, H, W, steps = 2, 32, 32, 8
    x, cond = inputs(synth, B, H, W)
    out = diffusion_pn.p_sample_loop((B, 1, H, W), cond=cond.cuda(), start_t=steps, x_T=x.cuda(),
                                     z_noise=explicit_noise(synth, steps, B, H, W))
    assert np.abs(out.cpu().numpy() - golden("pn_loop_B2_32x32_T1000_start8")["out"]).max() < TOL


def test_ddim_sample_pred_noise_mirror(diffusion_pn, synth, oracle):
    """ddim_step_kernel's pred_noise branch (eps = model output, x0 from eps, clamped) vs the reference's ddim_sample"""
    B, H, W, S = 2, 32, 32, 10
    d = diffusion_pn
    old = (d.sampling_timesteps, d.is_ddim_sampling, d.ddim_sampling_eta)
    d.sampling_timesteps, d.is_ddim_sampling, d.ddim_sampling_eta = S, True, 1.0
    try:
        x, cond = inputs(synth, B, H, W)
        pairs = d.ddim_time_pairs(0)
        assert pairs == oracle.ddim_time_pairs(1000, S, 0)
        zs = {t: torch.from_numpy(synth.noise_z(3, t, 0, B, H, W)).cuda() for t, nxt in pairs if nxt > 0}
        out = d.ddim_sample((B, 1, H, W), cond=cond.cuda(), x_T=x.cuda(), z_noise=zs)
        err = np.abs(out.cpu().numpy() - golden("pn_ddim_B2_32x32_T1000_S10_eta1")["out"]).max()
        print(f"pred_noise DDIM max|delta| vs reference golden: {err:.3e}")
        assert err < TOL
    finally:
        d.sampling_timesteps, d.is_ddim_sampling, d.ddim_sampling_eta = old


def test_timestep_indices_are_validated(engine_factory, synth):
    """per-sample t index device tables: the engine refuses out-of-range values before any upload"""
    eng = engine_factory(timesteps=50, max_batch=2, max_h=32, max_w=32)
    x, cond = inputs(synth, 2, 32, 32)
    with pytest.raises(IndexError):
        eng.unet_forward(x.cuda(), torch.tensor([3, 50]), cond.cuda())
    with pytest.raises(IndexError):
        eng.unet_forward(x.cuda(), torch.tensor([-1, 4]), cond.cuda())
    with pytest.raises(IndexError):
        eng.q_sample(x.cuda() * 0 + 0.5, torch.tensor([0, 77]), x.cuda())
    with pytest.raises(RuntimeError, match="outside"):
        eng.q_sample(x.cuda() * 0 + 0.5, 50, x.cuda())
    out = eng.unet_forward(x.cuda(), torch.tensor([0, 49]), cond.cuda())
    assert bool(torch.isfinite(out).all())


def test_clip_denoised_false_mirror(engine_factory, sd_np, synth):
    """p_sample(clip_denoised=False) and ddim_sample(clip_denoised=False) (cond_DDPM.py:433, :467) through the mirror vs the
    reference's outputs; the engine flag is restored to the default (clipping) afterwards"""
    U, D = load_pkg("OpenAI_Unet"), load_pkg("cond_DDPM")
    m = U.UNetModel(image_size=(32, 32), in_channels=1, model_channels=128, out_channels=1, num_res_blocks=3,
                    attention_resolutions=(3, 6, 12), dropout=0, channel_mult=[1, 2, 2], conv_resample=True, dims=2,
                    num_classes=128, use_checkpoint=False, use_fp16=True, num_heads=1, num_head_channels=64,
                    num_heads_upsample=-1, use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True,
                    use_spatial_transformer=False, transformer_depth=1)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    d = D.GaussianDiffusion(m, image_size=(32, 32), timesteps=1000, sampling_timesteps=1000, objective="pred_x0",
                            channels=1, loss_type="l1", p2_loss_weight_gamma=0, cfg=None).cuda()
    B, H, W = 2, 32, 32
    x, cond = inputs(synth, B, H, W)
    z = torch.from_numpy(synth.noise_z(3, 5, 0, B, H, W)).cuda()
    out = d.p_sample(x.cuda(), 5, clip_denoised=False, cond=cond.cuda(), z=z).cpu().numpy()
    ref = golden("noclip_p_sample_B2_32x32_t5")["out"]
    assert np.abs(out - ref).max() < TOL
    clipped = d.p_sample(x.cuda(), 5, cond=cond.cuda(), z=z).cpu().numpy()
    assert np.abs(clipped - ref).max() > 1e-3
    d.sampling_timesteps, d.is_ddim_sampling, d.ddim_sampling_eta = 10, True, 1.0
    zs = {t: torch.from_numpy(synth.noise_z(3, t, 0, B, H, W)).cuda() for t, nxt in d.ddim_time_pairs(0) if nxt > 0}
    out = d.ddim_sample((B, 1, H, W), clip_denoised=False, cond=cond.cuda(), x_T=x.cuda(), z_noise=zs).cpu().numpy()
    assert np.abs(out - golden("noclip_ddim_B2_32x32_T1000_S10_eta1")["out"]).max() < TOL
    out = d.ddim_sample((B, 1, H, W), cond=cond.cuda(), x_T=x.cuda(), z_noise=zs).cpu().numpy()      # default: clipping again
    assert np.abs(out - golden("ddim_B2_32x32_T1000_S10_eta1")["out"]).max() < TOL
    m._hip.close()


def test_ddim_sample_with_simplex_noise_vs_oracle_composition(sd_np, synth, oracle, sd_torch):
    """ddim_sample under cfg.noisetype == 'simplex' (cond_DDPM.py:501-503): every pair with time_next > 0 draws a fresh simplex
    field (the device generator, bit-exact with the reference's for a given numpy seed). Oracle composition: the golden-pinned
    ddim_sample restatement fed the simplex oracle's fields for the seeds numpy yields in the same order."""
    import simplex_oracle as SO
    GN = load_pkg("generate_noise")
    U, D = load_pkg("OpenAI_Unet"), load_pkg("cond_DDPM")
    M = load_pkg("DDPM_2D")
    m = U.UNetModel(image_size=(32, 32), in_channels=1, model_channels=128, out_channels=1, num_res_blocks=3,
                    attention_resolutions=(3, 6, 12), dropout=0, channel_mult=[1, 2, 2], conv_resample=True, dims=2,
                    num_classes=128, use_checkpoint=False, use_fp16=True, num_heads=1, num_head_channels=64,
                    num_heads_upsample=-1, use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True,
                    use_spatial_transformer=False, transformer_depth=1)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    d = D.GaussianDiffusion(m, image_size=(32, 32), timesteps=1000, sampling_timesteps=6, objective="pred_x0", channels=1,
                            loss_type="l1", p2_loss_weight_gamma=0, cfg=M.AttrDict(noisetype="simplex")).cuda()
    B, H, W, S = 2, 32, 32, 6
    x, cond = inputs(synth, B, H, W)
    np.random.seed(77)
    out = d.ddim_sample((B, 1, H, W), cond=cond.cuda(), x_T=x.cuda()).cpu().numpy()
    np.random.seed(77)
    fields = {}
    for time, nxt in oracle.ddim_time_pairs(1000, S, 0):
        if nxt > 0:
            GN.draw_seed()
            fields[time] = torch.from_numpy(SO.gen_noise(GN.draw_seed(), (B, 1, H, W))).float()
    ref = oracle.ddim_sample(x, cond, sd_torch, oracle.schedule_buffers(1000), lambda t: fields[t], S, 1.0, 0, None).numpy()
    err = float(np.abs(out - ref).max())
    print(f"DDIM with simplex noise vs oracle composition: max|delta| {err:.3e}")
    assert err < TOL and ref.std() > 0.01
    m._hip.close()
