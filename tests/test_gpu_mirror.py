"""GPU: the host-side mirrors of the reference classes (UNetModel, GaussianDiffusion, DDPM_2D) driving the
HIP engine -- same call sites as the reference, results against the reference's golden vectors."""
import numpy as np
import pytest
import torch

from conftest import golden, load_pkg

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def diffusion(sd_np):
    U, D = load_pkg("OpenAI_Unet"), load_pkg("cond_DDPM")
    m = U.UNetModel(image_size=(32, 32), in_channels=1, model_channels=128, out_channels=1, num_res_blocks=3,
                    attention_resolutions=(3, 6, 12), dropout=0, channel_mult=[1, 2, 2], conv_resample=True, dims=2,
                    num_classes=128, use_checkpoint=False, use_fp16=True, num_heads=1, num_head_channels=64,
                    num_heads_upsample=-1, use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True,
                    use_spatial_transformer=False, transformer_depth=1)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    d = D.GaussianDiffusion(m, image_size=(32, 32), timesteps=1000, sampling_timesteps=1000, objective="pred_x0",
                            channels=1, loss_type="l1", p2_loss_weight_gamma=0, cfg=None)
    d = d.cuda()
    yield d
    m._hip.close()


def test_unet_mirror_forward(diffusion, synth):
    g = golden("unet_fwd_B2_32x32")
    x = torch.from_numpy(synth.noise_xT(2, 0, 2, 32, 32)).cuda()
    cond = torch.from_numpy(synth.synth_cond(1, 0, 2)).cuda()
    m = diffusion.model
    out = m(x, torch.full((2,), 500, device="cuda", dtype=torch.long), cond=cond)
    assert np.abs(out.cpu().numpy() - g["t500"]).max() < TOL
    out = m.forward_with_cond_scale(x, torch.tensor([123, 877], device="cuda"), cond=cond, cond_scale=3.0)
    assert np.abs(out.cpu().numpy() - g["tmixed"]).max() < TOL


def test_p_sample_loop_mirror(diffusion, synth):
    B, H, W, steps = 2, 32, 32, 8
    x = torch.from_numpy(synth.noise_xT(2, 0, B, H, W)).cuda()
    cond = torch.from_numpy(synth.synth_cond(1, 0, B)).cuda()
    noise = np.zeros((steps, B, 1, H, W), np.float32)
    for t in range(1, steps):
        noise[t] = synth.noise_z(3, t, 0, B, H, W)
    out = diffusion.p_sample_loop((B, 1, H, W), cond=cond, start_t=steps, x_T=x, z_noise=torch.from_numpy(noise).cuda())
    ref = golden("loop_B2_32x32_T1000_start8")["out"]
    assert np.abs(out.cpu().numpy() - ref).max() < TOL
    # reference call shape (everything drawn internally); reproducible through torch.manual_seed
    torch.manual_seed(5)
    a = diffusion.p_sample_loop((B, 1, H, W), cond=cond, start_t=4)
    torch.manual_seed(5)
    b = diffusion.p_sample_loop((B, 1, H, W), cond=cond, start_t=4)
    assert torch.equal(a, b) and float(a.min()) >= 0 and float(a.max()) <= 1
    c = diffusion.sample(batch_size=B, cond=cond, start_t=2)
    assert c.shape == (B, 1, H, W)


@pytest.mark.parametrize("name,S,eta,start_t", [("ddim_B2_32x32_T1000_S10_eta1", 10, 1.0, 0),
                                                ("ddim_B2_32x32_T1000_S10_eta0", 10, 0.0, 0),
                                                ("ddim_B2_32x32_T1000_S6_eta1_start300", 6, 1.0, 300)])
def test_ddim_sample_mirror(diffusion, synth, oracle, name, S, eta, start_t):
    """GaussianDiffusion.ddim_sample (cond_DDPM.py:466-515) on the device vs the reference's golden output"""
    B, H, W = 2, 32, 32
    d = diffusion
    old = (d.sampling_timesteps, d.is_ddim_sampling, d.ddim_sampling_eta)
    d.sampling_timesteps, d.is_ddim_sampling, d.ddim_sampling_eta = S, True, eta
    try:
        assert d.ddim_time_pairs(start_t) == oracle.ddim_time_pairs(1000, S, start_t)
        x = torch.from_numpy(synth.noise_xT(2, 0, B, H, W)).cuda()
        cond = torch.from_numpy(synth.synth_cond(1, 0, B)).cuda()
        x_start = (torch.from_numpy(synth.synth_slices(4, 0, B, H, W)) * 2 - 1).cuda() if start_t else None
        zs = {t: torch.from_numpy(synth.noise_z(3, t, 0, B, H, W)).cuda() for t, nxt in d.ddim_time_pairs(start_t) if nxt > 0}
        out = d.ddim_sample((B, 1, H, W), cond=cond, x_start=x_start, start_t=start_t, x_T=x, z_noise=zs)
        ref = golden(name)["out"]
        err = np.abs(out.cpu().numpy() - ref).max()
        print(name, f"max|delta| vs reference golden: {err:.3e}")
        assert err < TOL, err
        if start_t == 0:
            # reference call shape through sample(): draws come from the counter RNG, reproducible through torch.manual_seed
            torch.manual_seed(7)
            a = d.sample(batch_size=B, cond=cond)
            torch.manual_seed(7)
            b = d.sample(batch_size=B, cond=cond)
            assert torch.equal(a, b) and a.shape == (B, 1, H, W) and float(a.min()) >= 0 and float(a.max()) <= 1
    finally:
        d.sampling_timesteps, d.is_ddim_sampling, d.ddim_sampling_eta = old


def test_single_step_forward_mirror(diffusion, synth):
    g = golden("p_losses_B2_32x32_t499")
    B, H, W = 2, 32, 32
    x01 = torch.from_numpy(synth.synth_slices(2, 0, B, H, W)).cuda()
    cond = torch.from_numpy(synth.synth_cond(1, 0, B)).cuda()
    noise = torch.from_numpy(synth.noise_z(3, 0, 0, B, H, W)).cuda()
    loss, reco = diffusion(x01, t=499, cond=cond, noise=noise)
    assert np.abs(reco.cpu().numpy() - g["reco"]).max() < TOL
    assert abs(float(loss) - float(g["loss"])) < 1e-5


def test_weights_reload_repacks(diffusion, synth, sd_np):
    """load_state_dict after the first forward must be picked up (the engine re-packs)"""
    m = diffusion.model
    x = torch.from_numpy(synth.noise_xT(2, 0, 1, 32, 32)).cuda()
    cond = torch.from_numpy(synth.synth_cond(1, 0, 1)).cuda()
    a = m(x, torch.tensor([10], device="cuda"), cond=cond)
    sd2 = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    sd2["out.2.bias"] = sd2["out.2.bias"] + 1.0
    m.load_state_dict(sd2)
    b = m(x, torch.tensor([10], device="cuda"), cond=cond)
    assert torch.allclose(b, a + 1.0, atol=1e-5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})


def test_ddpm2d_reconstruct_switch(sd_np, synth):
    M = load_pkg("DDPM_2D")
    cfg = dict(imageDim=[64, 64, 100], rescaleFactor=2, unet_dim=128, dim_mults=[1, 2, 2], condition=True,
               test_timesteps=500, timesteps=1000)

    class Enc(torch.nn.Module):          # stand-in for the timm ResNet-50 (out of scope)
        def forward(self, x):
            return x.flatten(1)[:, :128].contiguous()

    mod = M.DDPM_2D(cfg, encoder=Enc())
    mod.diffusion.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    mod = mod.cuda()
    inp = torch.from_numpy(synth.synth_slices(2, 0, 3, 32, 32)).cuda()
    loss, reco = mod.reconstruct(inp, noise=torch.randn_like(inp))               # reference behaviour: single step
    assert reco.shape == inp.shape and torch.isfinite(reco).all()
    mod.cfg["reverse_sampling"], mod.cfg["reverse_start_t"] = True, 3            # the reverse loop at the same call site
    loss2, reco2 = mod.reconstruct(inp)
    assert reco2.shape == inp.shape and float(reco2.min()) >= 0 and float(reco2.max()) <= 1
    out = mod.test_step({"vol": {"data": inp.permute(1, 2, 3, 0).unsqueeze(0)}}, 0)
    assert out["final_volume"].shape == (1, 1, 32, 32, 3)
    mod.diffusion.model._hip.close()


def test_ddpm2d_test_step_noise_ensemble_vs_oracle_composition(sd_np, synth, oracle, sd_torch):
    """The package's own LightningModule with the reference experiment's cfg (noise_ensemble: True, noisetype: simplex,
    configs/experiment/cDDPM/DDPM_cond_spark_2D.yaml:22,33) against a composition of the already-golden pieces: the 4 centre
    slices, three single-step reconstructions (oracle p_losses_recon, golden-pinned) at t = 249, 499, 749, each from the
    simplex field of the seed numpy's RNG yields at that point (simplex oracle, bit-exact golden), averaged."""
    import simplex_oracle as SO
    M = load_pkg("DDPM_2D")
    GN = load_pkg("generate_noise")
    cfg = dict(imageDim=[64, 64, 100], rescaleFactor=2, unet_dim=128, dim_mults=[1, 2, 2], condition=True,
               test_timesteps=500, timesteps=1000, noise_ensemble=True, noisetype="simplex")

    class Enc(torch.nn.Module):          # stand-in for the context encoder (row f2 has its own tests)
        def forward(self, x):
            return x.flatten(1)[:, :128].contiguous() * 2 - 1

    mod = M.DDPM_2D(cfg, encoder=Enc())
    mod.diffusion.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    mod = mod.cuda()
    D, H, W = 9, 32, 32
    vol = torch.from_numpy(synth.synth_slices(2, 0, D, H, W)).permute(1, 2, 3, 0).unsqueeze(0).contiguous()   # [1,1,H,W,D]
    np.random.seed(2024)
    out = mod.test_step({"vol": {"data": vol.cuda()}}, 0)
    got = out["final_volume"].cpu()[0, 0].permute(2, 0, 1).unsqueeze(1)                  # [4,1,H,W]
    # oracle composition
    start = int((D - 4) / 2)
    x01 = vol[0, 0].permute(2, 0, 1).unsqueeze(1)[start:start + 4].contiguous()
    assert torch.equal(out["input"].cpu(), x01) and out["ind_offset"] == start
    cond = Enc()(x01)
    buf = oracle.schedule_buffers(1000)
    np.random.seed(2024)
    acc = torch.zeros_like(x01)
    for t in (250, 500, 750):
        GN.draw_seed()
        seed = GN.draw_seed()
        noise = torch.from_numpy(SO.gen_noise(seed, (4, 1, H, W))).float()
        _loss, reco = oracle.p_losses_recon(x01, torch.full((4,), t - 1), cond, noise, sd_torch, buf)
        acc += reco
    ref = acc / 3
    err = float((got - ref).abs().max())
    print(f"test_step (noise ensemble, simplex) vs oracle composition: max|delta| {err:.3e}")
    assert err < TOL and float(ref.std()) > 0.01
    mod.diffusion.model._hip.close()


def test_ddpm2d_training_step_updates_the_unet(sd_np, synth):
    """the LightningModule mirror's training_step (reference src/models/DDPM_2D.py:114-135 + :305-306): a few steps on one batch run on the
    HIP operators, lower the loss of a fixed probe, move the UNet's parameters (state_dict sees them) and the evaluation path picks the
    updated weights up (the inference engine re-packs)"""
    M = load_pkg("DDPM_2D")
    cfg = dict(imageDim=[64, 64, 100], rescaleFactor=2, unet_dim=128, dim_mults=[1, 2, 2], condition=True, test_timesteps=500, timesteps=1000,
               lr=1e-4)

    class Enc(torch.nn.Module):          # stand-in for the context encoder (not updated by training_step)
        def forward(self, x):
            return x.flatten(1)[:, :128].contiguous() * 2 - 1

    mod = M.DDPM_2D(cfg, encoder=Enc())
    mod.diffusion.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    mod = mod.cuda()
    vol = torch.from_numpy(synth.synth_slices(4, 0, 2, 32, 32)).reshape(2, 1, 32, 32, 1).cuda()       # what batch['vol'][DATA] holds for 2D slices
    inp = vol.squeeze(-1)
    probe_noise = torch.randn_like(inp)
    loss0, reco0 = mod.reconstruct(inp, noise=probe_noise)
    w_before = mod.diffusion.model.state_dict()["middle_block.0.in_layers.2.weight"].clone()
    torch.manual_seed(0)
    losses = [float(mod.training_step({"vol": {"data": vol}}, i)["loss"]) for i in range(6)]
    assert all(np.isfinite(losses))
    w_after = mod.diffusion.model.state_dict()["middle_block.0.in_layers.2.weight"]
    assert float((w_after - w_before).abs().max()) > 1e-5                      # Adam moved the weights (lr 1e-4 per step)
    loss1, reco1 = mod.reconstruct(inp, noise=probe_noise)
    assert float((reco1 - reco0).abs().max()) > 1e-4                           # ... and evaluation runs on the updated weights
    print("training losses", losses, "probe", float(loss0), "->", float(loss1))


def test_ddpm2d_training_step_trains_the_native_encoder_jointly(sd_np, synth):
    """with this package's own context encoder (the ResNet-50 behind get_encoder) the training step runs it in training mode and updates
    it together with the UNet (`features = self(input)` with a gradient + `Adam(self.parameters())`, reference src/models/DDPM_2D.py:114-135,
    :305-306): encoder weights and BatchNorm running statistics move, state_dict() sees them, evaluation re-reads them"""
    M, E = load_pkg("DDPM_2D"), load_pkg("DDPM_encoder")
    cfg = dict(imageDim=[128, 128, 100], rescaleFactor=2, unet_dim=128, dim_mults=[1, 2, 2], condition=True, test_timesteps=500, timesteps=1000,
               lr=1e-4, backbone="resnet50", cond_dim=128)
    enc = E.ResNet50Encoder(num_classes=128, in_chans=1)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_encoder_state_dict(0).items()}, strict=False)
    mod = M.DDPM_2D(cfg, encoder=enc)
    mod.diffusion.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    mod = mod.cuda()
    vol = torch.from_numpy(synth.synth_slices(4, 0, 4, 64, 64)).reshape(4, 1, 64, 64, 1).cuda()
    inp = vol.squeeze(-1)
    c0 = mod(inp).clone()
    w0 = mod.encoder.state_dict()["layer3.2.conv2.weight"].clone()
    rm0 = mod.encoder.state_dict()["layer1.0.bn1.running_mean"].clone()
    torch.manual_seed(1)
    losses = [float(mod.training_step({"vol": {"data": vol}}, i)["loss"]) for i in range(3)]
    assert all(np.isfinite(losses))
    sd = mod.encoder.state_dict()
    assert float((sd["layer3.2.conv2.weight"] - w0).abs().max()) > 1e-5
    assert float((sd["layer1.0.bn1.running_mean"] - rm0).abs().max()) > 1e-6
    c1 = mod(inp)
    assert torch.isfinite(c1).all() and float((c1 - c0).abs().max()) > 1e-4
    print("joint training losses", losses)


def test_ddpm2d_validation_step(sd_np, synth):
    """reference src/models/DDPM_2D.py:137-155: forward-only loss of a validation batch (random t per slice, Gaussian noise when cfg.noisetype
    is unset) -- finite, in the range of the training loss, and it leaves the weights alone"""
    M = load_pkg("DDPM_2D")
    cfg = dict(imageDim=[64, 64, 100], rescaleFactor=2, unet_dim=128, dim_mults=[1, 2, 2], condition=True, test_timesteps=500, timesteps=1000)

    class Enc(torch.nn.Module):
        def forward(self, x):
            return x.flatten(1)[:, :128].contiguous() * 2 - 1

    mod = M.DDPM_2D(cfg, encoder=Enc())
    mod.diffusion.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    mod = mod.cuda()
    vol = torch.from_numpy(synth.synth_slices(4, 0, 3, 32, 32)).reshape(3, 1, 32, 32, 1).cuda()
    w0 = mod.diffusion.model.state_dict()["out.2.weight"].clone()
    torch.manual_seed(3)
    out = mod.validation_step({"vol": {"data": vol}}, 0)
    assert torch.isfinite(out["loss"]) and 0.05 < float(out["loss"]) < 2.0
    assert torch.equal(mod.diffusion.model.state_dict()["out.2.weight"], w0)
    mod.update_prefix("fold1/")
    assert mod.prefix == "fold1/"


def _experiment_module(sd_np, synth, **over):
    M, E = load_pkg("DDPM_2D"), load_pkg("DDPM_encoder")
    cfg = dict(imageDim=[128, 128, 100], rescaleFactor=2, unet_dim=128, dim_mults=[1, 2, 2], condition=True, test_timesteps=500, timesteps=1000,
               lr=1e-4, backbone="resnet50", cond_dim=128, objective="pred_x0", loss="l1", noisetype="simplex")
    cfg.update(over)
    enc = E.ResNet50Encoder(num_classes=128, in_chans=1)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_encoder_state_dict(0).items()}, strict=False)
    mod = M.DDPM_2D(cfg, encoder=enc)
    mod.diffusion.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    return mod.cuda()


def test_training_step_with_simplex_noise_does_not_touch_the_inference_engine(sd_np, synth, monkeypatch):
    """the experiment's own training configuration (pred_x0, l1, noisetype simplex: DDPM_cond_spark_2D.yaml): the simplex field of a
    step is drawn on the TRAINER's handle -- the inference engine (whose packed weights every step invalidates) is neither asked for
    nor rebuilt while training"""
    B = load_pkg("backend")
    mod = _experiment_module(sd_np, synth)
    gets = []
    orig = B.HipBackend.get
    monkeypatch.setattr(B.HipBackend, "get", lambda self, *a, **k: (gets.append(1), orig(self, *a, **k))[1])
    vol = torch.from_numpy(synth.synth_slices(4, 0, 4, 64, 64)).reshape(4, 1, 64, 64, 1).cuda()
    np.random.seed(0); torch.manual_seed(0)
    losses = [float(mod.training_step({"vol": {"data": vol}}, i)["loss"]) for i in range(3)]
    assert all(np.isfinite(losses)) and gets == [], (losses, len(gets))
    for name, buf in mod.encoder.named_buffers():
        if name.endswith("num_batches_tracked"):
            assert int(buf) == 3
    mod.reconstruct(vol.squeeze(-1))             # evaluation afterwards: ONE engine (re)build, on the updated weights
    assert len(gets) >= 1


def test_checkpoint_round_trip_resumes_adam(sd_np, synth):
    """save after two steps (state_dict + the HIP trainers' Adam state through on_save_checkpoint), load into a FRESH module, take the
    third step there and in the uninterrupted run with the same draws: identical parameters. Without the moments the resumed step would be
    Adam's first step again (bias correction 1, zero moments) and differ in the fourth digit."""
    vol = torch.from_numpy(synth.synth_slices(4, 0, 2, 64, 64)).reshape(2, 1, 64, 64, 1).cuda()
    a = _experiment_module(sd_np, synth)
    for i in range(2):
        np.random.seed(10 + i); torch.manual_seed(10 + i)
        a.training_step({"vol": {"data": vol}}, i)
    ck = {"state_dict": {k: v.detach().cpu().clone() for k, v in a.state_dict().items()}}
    a.on_save_checkpoint(ck)
    assert int(ck["hip_optimizer_state"]["unet"]["ctrl"][1]) == 2 and "encoder" in ck["hip_optimizer_state"]
    np.random.seed(99); torch.manual_seed(99)
    a.training_step({"vol": {"data": vol}}, 2)
    want = {k: v.detach().cpu().clone() for k, v in a.state_dict().items()}

    b = _experiment_module(sd_np, synth)
    b.load_state_dict(ck["state_dict"])
    b.on_load_checkpoint(ck)
    np.random.seed(99); torch.manual_seed(99)
    b.training_step({"vol": {"data": vol}}, 2)
    assert b.hip_trainer(vol.device).step_count == 3
    got = b.state_dict()
    worst = max(float((got[k].cpu().float() - want[k].float()).abs().max()) for k in want if want[k].is_floating_point())
    print("resumed vs uninterrupted, third step: max|delta|", worst)
    assert worst < 1e-6
    # and WITHOUT the optimizer state the same step lands elsewhere (the test can tell the difference)
    c = _experiment_module(sd_np, synth)
    c.load_state_dict(ck["state_dict"])
    np.random.seed(99); torch.manual_seed(99)
    c.training_step({"vol": {"data": vol}}, 2)
    k = "diffusion.model.middle_block.0.in_layers.2.weight"
    assert float((c.state_dict()[k].cpu() - want[k]).abs().max()) > 2e-5


def test_parameters_loaded_or_moved_after_the_trainer_exists(sd_np, synth):
    """load_state_dict into the aliased parameters re-packs the operators' weight images; module.cpu()/.cuda() (Lightning's teardown)
    breaks the alias and the next step repairs it, keeping the values the user sees"""
    vol = torch.from_numpy(synth.synth_slices(4, 0, 2, 64, 64)).reshape(2, 1, 64, 64, 1).cuda()
    other = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(3).items()}
    fresh = _experiment_module(sd_np, synth, noisetype=None)
    fresh.diffusion.model.load_state_dict(other)
    torch.manual_seed(5)
    want = float(fresh.training_step({"vol": {"data": vol}}, 0)["loss"])
    mod = _experiment_module(sd_np, synth, noisetype=None)
    torch.manual_seed(4)
    mod.training_step({"vol": {"data": vol}}, 0)               # trainer exists, packed images built from seed-0 weights
    mod2 = _experiment_module(sd_np, synth, noisetype=None)    # same start as `fresh`, but the weights arrive AFTER the trainer exists
    mod2.hip_trainer(vol.device); mod2.hip_encoder_trainer(vol.device)
    mod2.diffusion.model.load_state_dict(other)
    torch.manual_seed(5)
    got = float(mod2.training_step({"vol": {"data": vol}}, 0)["loss"])
    assert abs(got - want) < 1e-6 * abs(want), (got, want)
    # broken alias: .cpu() / .cuda() replace param.data
    w_before = mod.diffusion.model.state_dict()["out.2.weight"].clone()
    mod = mod.cpu().cuda()
    assert torch.equal(mod.diffusion.model.state_dict()["out.2.weight"], w_before)
    torch.manual_seed(6)
    mod.training_step({"vol": {"data": vol}}, 1)
    tr_ = mod.hip_trainer(vol.device)
    p = dict(mod.diffusion.model.named_parameters())["out.2.weight"]
    assert p.data_ptr() == tr_.p["out.2.weight"].data_ptr()
    assert float((mod.diffusion.model.state_dict()["out.2.weight"] - w_before).abs().max()) > 1e-6     # the step is visible in state_dict


def test_training_step_without_conditioning(synth):
    """cfg.condition False (the reference's pDDPM-style configuration: no encoder, num_classes None, emb = time_embed(t) alone,
    OpenAI_Unet.py:583-590, :849-852): the mirror's training step runs on the HIP operators, the loss falls, the weights move"""
    M = load_pkg("DDPM_2D")
    cfg = dict(imageDim=[64, 64, 100], rescaleFactor=2, unet_dim=128, dim_mults=[1, 2, 2], condition=False, test_timesteps=500, timesteps=1000,
               lr=1e-4)
    mod = M.DDPM_2D(cfg)
    assert not hasattr(mod, "encoder") and mod.diffusion.model.num_classes is None
    sd = synth.synth_state_dict(0, num_classes=None)
    mod.diffusion.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    mod = mod.cuda()
    vol = torch.from_numpy(synth.synth_slices(4, 0, 2, 32, 32)).reshape(2, 1, 32, 32, 1).cuda()
    w0 = mod.diffusion.model.state_dict()["time_embed.2.weight"].clone()
    torch.manual_seed(0)
    losses = [float(mod.training_step({"vol": {"data": vol}}, i)["loss"]) for i in range(5)]
    print("unconditioned training losses", losses)
    assert all(np.isfinite(losses))
    assert float((mod.diffusion.model.state_dict()["time_embed.2.weight"] - w0).abs().max()) > 1e-5
    assert mod.hip_trainer(vol.device).dcond is None


def test_p_sample_loop_box_branch_vs_reference_golden(diffusion, synth):
    """p_sample_loop(box=...) of the mirror (reference cond_DDPM.py:455-459, the x_T masking as written) on the HIP path against the
    reference's own output"""
    g = golden("box_loop_B3_32x32_T1000_start6")
    B, H, W, start_t = 3, 32, 32, 6
    cond = torch.from_numpy(synth.synth_cond(1, 0, B)).cuda()
    xT = torch.from_numpy(synth.noise_xT(2, 0, B, H, W)).cuda()
    z = torch.zeros((start_t, B, 1, H, W), dtype=torch.float32)
    for t in range(1, start_t):
        z[t] = torch.from_numpy(synth.noise_z(3, t, 0, B, H, W))
    out = diffusion.p_sample_loop((B, 1, H, W), cond=cond, start_t=start_t, box=torch.from_numpy(g["box"]), x_T=xT, z_noise=z.cuda())
    err = float(np.abs(out.cpu().numpy() - g["out"]).max())
    print("box branch vs reference golden:", err)
    assert err < TOL


@pytest.mark.parametrize("objective,loss_type,inpaint", [("pred_noise", "l2", False), ("pred_x0", "l1", False), ("pred_x0", "l1", True),
                                                         ("pred_noise", "l2", True)])
def test_p_losses_box_and_inpaint_variants_vs_reference_golden(sd_np, synth, objective, loss_type, inpaint):
    """GaussianDiffusion.forward -> p_losses with `box` (only the box is noised; pred_noise: the target is the noise inside the box) and
    with inpaint=True (the model's box pasted back into x_start before the loss): reference cond_DDPM.py:592-598, :611-615, :626-633;
    (loss, reco) against the reference's own outputs (oracle/make_golden_box.py)"""
    U, D = load_pkg("OpenAI_Unet"), load_pkg("cond_DDPM")
    g = golden("box_p_losses_B3_32x32_t350")
    m = U.UNetModel(image_size=(32, 32), in_channels=1, model_channels=128, out_channels=1, num_res_blocks=3,
                    attention_resolutions=(3, 6, 12), dropout=0, channel_mult=[1, 2, 2], conv_resample=True, dims=2,
                    num_classes=128, use_checkpoint=False, use_fp16=True, num_heads=1, num_head_channels=64,
                    num_heads_upsample=-1, use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True,
                    use_spatial_transformer=False, transformer_depth=1)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    d = D.GaussianDiffusion(m, image_size=(32, 32), timesteps=1000, sampling_timesteps=1000, objective=objective, channels=1,
                            loss_type=loss_type, p2_loss_weight_gamma=0, inpaint=inpaint, cfg=None).cuda()
    B, H, W = 3, 32, 32
    x01 = torch.from_numpy(synth.synth_slices(2, 0, B, H, W)).cuda()
    cond = torch.from_numpy(synth.synth_cond(1, 0, B)).cuda()
    noise = torch.from_numpy(synth.noise_z(3, 0, 0, B, H, W)).cuda()
    loss, reco = d(x01, t=350, cond=cond, noise=noise, box=torch.from_numpy(g["box"]))
    key = f"{objective}_{loss_type}_{'inpaint' if inpaint else 'box'}"
    e_reco = float(np.abs(reco.cpu().numpy() - g[key + "_reco"]).max())
    e_loss = abs(float(loss) - float(g[key + "_loss"]))
    print(key, "reco", e_reco, "loss", e_loss)
    assert e_reco < TOL and e_loss < 1e-5
    m._hip.close()


def test_interpolate_mirror(diffusion, synth):
    """GaussianDiffusion.interpolate (reference cond_DDPM.py:532-546, intended semantics): q_sample both images at t, mix, t reverse steps
    -- the composition of the mirror's own (golden-checked) q_sample and p_sample, and lam = 0 / 1 reduce to the single-image chains"""
    B, H, W, t = 2, 32, 32, 5
    x1 = torch.from_numpy(synth.synth_slices(2, 0, B, H, W)).cuda() * 2 - 1
    x2 = torch.from_numpy(synth.synth_slices(2, 7, B, H, W)).cuda() * 2 - 1
    cond = torch.from_numpy(synth.synth_cond(1, 0, B)).cuda()
    tb = torch.full((B,), t, device="cuda", dtype=torch.long)
    torch.manual_seed(0)
    got = diffusion.interpolate(x1, x2, t=t, lam=0.25, cond=cond, seed=7)
    torch.manual_seed(0)
    img = (0.75 * diffusion.q_sample(x1, tb) + 0.25 * diffusion.q_sample(x2, tb)).contiguous()
    for i in reversed(range(t)):
        img = diffusion.p_sample(img, i, cond=cond, seed=7)
    assert got.shape == x1.shape and torch.equal(got, img) and bool(torch.isfinite(got).all())
    torch.manual_seed(0)
    a = diffusion.interpolate(x1, x2, t=t, lam=0.0, cond=cond, seed=7)
    torch.manual_seed(0)
    b = diffusion.q_sample(x1, tb)
    for i in reversed(range(t)):
        b = diffusion.p_sample(b, i, cond=cond, seed=7)
    assert torch.equal(a, b)
