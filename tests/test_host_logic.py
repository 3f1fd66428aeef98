"""CPU: host-side logic of the package -- counter RNG, synthetic weights, schedule, the class mirrors'
surfaces, the C ABI library (loads, exports every declared symbol, fails loudly without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden, load_pkg


def test_philox_known_answers(synth):
    """Random123 known-answer vectors for Philox4x32-10"""
    def run(c, k):
        return [int(v) for v in synth.philox4x32(np.uint32(c[0]), np.uint32(c[1]), np.uint32(c[2]), np.uint32(c[3]), k[0], k[1])]
    assert run((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert run((f, f, f, f), (f, f)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_noise_is_keyed_by_global_slice(synth):
    a = synth.noise_z(3, 17, 0, 4, 8, 8)
    b = synth.noise_z(3, 17, 2, 2, 8, 8)
    np.testing.assert_array_equal(a[2:], b)                      # shard [2,4) regenerates its own draws
    assert not np.array_equal(synth.noise_z(3, 18, 0, 1, 8, 8), a[:1])
    z = synth.noise_xT(5, 0, 8, 64, 64)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02 and np.isfinite(z).all()
    u = synth.synth_slices(1, 0, 2, 16, 16)
    assert u.min() > 0 and u.max() < 1


def test_synthetic_state_dict_matches_reference_inventory(sd_np, synth):
    assert len(sd_np) == 316 and sum(v.size for v in sd_np.values()) == 43871873
    assert sd_np["middle_block.1.qkv.weight"].shape == (768, 256, 1)
    assert sd_np["output_blocks.7.1.in_layers.2.weight"].shape == (256, 256, 3, 3)      # the up ResBlock
    assert sd_np["output_blocks.8.0.skip_connection.weight"].shape == (128, 384, 1, 1)
    assert sd_np["input_blocks.1.0.emb_layers.1.weight"].shape == (256, 1024)
    assert all(np.abs(v).max() > 0 for v in sd_np.values())       # nothing zero-initialised


def test_package_schedule_equals_oracle(oracle):
    sched = load_pkg("schedule")
    for T, kind in ((1000, "cosine"), (50, "cosine"), (200, "linear")):
        a, b = sched.schedule_buffers(T, kind), oracle.schedule_buffers(T, kind)
        assert list(a.keys()) == list(sched.BUFFER_NAMES)
        for k in a:
            assert torch.equal(a[k], b[k]), k
    with pytest.raises(ValueError):
        sched.schedule_buffers(10, "quadratic")


def test_unet_mirror_state_dict_and_surface(sd_np, synth):
    U = load_pkg("OpenAI_Unet")
    m = U.UNetModel(image_size=(128, 128), in_channels=1, model_channels=128, out_channels=1, num_res_blocks=3,
                    attention_resolutions=(3, 6, 12), dropout=0, channel_mult=[1, 2, 2], conv_resample=True, dims=2,
                    num_classes=128, use_checkpoint=False, use_fp16=True, num_heads=1, num_head_channels=64,
                    num_heads_upsample=-1, use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True,
                    use_spatial_transformer=False, transformer_depth=1)
    sd = m.state_dict()
    assert list(sd.keys()) == list(sd_np.keys())                  # same names, same registration order
    assert all(tuple(sd[k].shape) == sd_np[k].shape for k in sd)
    # zero-initialised modules as in the reference (zero_module): out.2, out_layers.3, proj_out
    assert float(sd["out.2.weight"].abs().max()) == 0 and float(sd["middle_block.1.proj_out.weight"].abs().max()) == 0
    assert float(sd["input_blocks.1.0.out_layers.3.weight"].abs().max()) == 0
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    m.convert_to_fp16()
    assert all(p.dtype == torch.float32 for p in m.parameters())   # no-op, as in the reference
    with pytest.raises(RuntimeError, match="no CPU fallback|MI355X"):
        m(torch.zeros(1, 1, 32, 32), torch.zeros(1, dtype=torch.long), cond=torch.zeros(1, 128))
    with pytest.raises(NotImplementedError):
        U.UNetModel(image_size=32, in_channels=1, model_channels=128, out_channels=1, num_res_blocks=3,
                    attention_resolutions=(), use_scale_shift_norm=False, resblock_updown=True,
                    use_new_attention_order=True, num_head_channels=64)


def test_diffusion_mirror_buffers_and_errors(oracle):
    U, D = load_pkg("OpenAI_Unet"), load_pkg("cond_DDPM")
    m = U.UNetModel(image_size=(32, 32), in_channels=1, model_channels=128, out_channels=1, num_res_blocks=3,
                    attention_resolutions=(3, 6, 12), channel_mult=[1, 2, 2], num_classes=128, num_head_channels=64,
                    use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True)
    d = D.GaussianDiffusion(m, image_size=(32, 32), timesteps=1000, sampling_timesteps=1000, objective="pred_x0",
                            channels=1, loss_type="l1", p2_loss_weight_gamma=0, cfg=None)
    g = golden("schedule_T1000")
    names = [n for n, _ in d.named_buffers()]
    assert names == list(g.files) or set(names) == set(g.files)
    for n in g.files:
        np.testing.assert_array_equal(getattr(d, n).numpy(), g[n], err_msg=n)
    assert d.num_timesteps == 1000 and not d.is_ddim_sampling and d.use_spatial_transformer is False
    keys = d.state_dict().keys()
    assert "model.input_blocks.0.0.weight" in keys and "betas" in keys          # -> diffusion.model.* in DDPM_2D
    with pytest.raises(RuntimeError):
        d.p_sample_loop((1, 1, 32, 32), cond=torch.zeros(1, 128))                # CPU tensors: loud, no fallback
    with pytest.raises(NotImplementedError):
        d.sample(batch_size=1, cond=torch.zeros(1, 128), box=torch.zeros(1, 4), x_start=torch.zeros(1, 1, 32, 32))   # ddim_sample_box: undefined in the reference
    # the box lines of p_sample_loop (cond_DDPM.py:455-459) as written: sample 0 keeps x_T inside its box, the others start from zeros
    x = torch.arange(3 * 1 * 4 * 5, dtype=torch.float32).reshape(3, 1, 4, 5) + 1
    m = D.mask_x_T_to_box(x, torch.tensor([[1, 0, 4, 2], [0, 0, 5, 4], [2, 1, 3, 3]]))
    want = torch.zeros_like(x)
    want[0, :, 0:2, 1:4] = x[0, :, 0:2, 1:4]
    assert torch.equal(m, want)
    with pytest.raises(AssertionError):
        D.GaussianDiffusion(m, image_size=32, objective="pred_v")
    with pytest.raises(ValueError):
        D.GaussianDiffusion(m, image_size=32, beta_schedule="sigmoid", objective="pred_x0")


def test_ddpm2d_mirror_builds_from_experiment_cfg():
    M = load_pkg("DDPM_2D")
    cfg = dict(imageDim=[192, 192, 100], rescaleFactor=2, unet_dim=128, dim_mults=[1, 2, 2], condition=True,
               test_timesteps=500, noise_ensemble=True, spatial_transformer=False, noisetype="simplex")
    enc = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.LazyLinear(128))
    mod = M.DDPM_2D(cfg, encoder=enc)
    assert mod.test_timesteps == 500 and mod.diffusion.num_timesteps == 1000 and mod.diffusion.objective == "pred_x0"
    assert mod.diffusion.model.image_size == (96, 96) and mod.cfg["cond_dim"] == 128
    keys = list(mod.state_dict().keys())
    assert any(k.startswith("diffusion.model.output_blocks.11.0.") for k in keys) and "diffusion.betas" in keys
    assert mod(torch.zeros(3, 1, 96, 96)).shape == (3, 128)
    # without `encoder=` the package's native encoder is built: the experiment's backbone is the SparK wrapper around a
    # resnet50 whose weights sit under encoder.encoder.* in the reference's checkpoints (Spark_2D.py:268-290)
    full = M.DDPM_2D(dict(cfg, backbone="Spark_Encoder_2D", version="resnet50"))
    ek = [k for k in full.state_dict() if k.startswith("encoder.")]
    assert "encoder.encoder.conv1.weight" in ek and "encoder.encoder.layer4.2.bn3.running_var" in ek and "encoder.encoder.fc.bias" in ek
    assert full.state_dict()["encoder.encoder.fc.weight"].shape == (128, 2048)
    assert full.state_dict()["encoder.encoder.conv1.weight"].shape == (64, 1, 7, 7)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="MI355X|CUDA"):
            full(torch.zeros(1, 1, 96, 96))           # no CPU fallback
    with pytest.raises(NotImplementedError):
        M.DDPM_2D(dict(cfg, backbone="resnet101"))


def test_ddpm2d_test_step_follows_the_reference_call_sequence(monkeypatch):
    """src/models/DDPM_2D.py:171-286 with the experiment's cfg (noise_ensemble: True, noisetype: simplex): the 4 CENTRE
    slices (:193-200), one fresh gen_noise field per ensemble member (:231), self.diffusion(input, cond=features, t=t-1,
    noise=noise) at t in [250, 500, 750] (:235), the mean of the three reconstructions (:238), volume as [1,1,H,W,D].
    Host logic only: the diffusion and the noise generator are recorders."""
    M = load_pkg("DDPM_2D")
    cfg = dict(imageDim=[64, 64, 100], rescaleFactor=2, unet_dim=128, dim_mults=[1, 2, 2], condition=True,
               test_timesteps=500, noise_ensemble=True, spatial_transformer=False, noisetype="simplex")
    enc = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.LazyLinear(128))
    mod = M.DDPM_2D(cfg, encoder=enc)
    calls, draws = [], []

    class FakeDiffusion(torch.nn.Module):
        def forward(self, img, cond=None, t=None, noise=None):
            calls.append((tuple(img.shape), tuple(cond.shape), int(t), None if noise is None else float(noise.flatten()[0])))
            return torch.tensor(float(t)), torch.full_like(img, float(t))

        def _engine(self, B, H, W, device):
            return "engine"

    def fake_gen_noise(cfg_, shape, engine=None, **kw):
        assert engine == "engine" and cfg_ is mod.cfg
        draws.append(tuple(shape))
        return torch.full(tuple(shape), float(len(draws)), dtype=torch.float16)

    mod.diffusion = FakeDiffusion()
    monkeypatch.setattr(M, "gen_noise", fake_gen_noise)
    D = 10
    vol = torch.arange(D, dtype=torch.float32).reshape(1, 1, 1, 1, D).expand(1, 1, 32, 32, D).contiguous()
    out = mod.test_step({"vol": {"data": vol}}, 0)
    assert mod.cfg["num_eval_slices"] == 4 and out["ind_offset"] == 3                 # int((10 - 4) / 2)
    assert out["input"].shape == (4, 1, 32, 32) and out["input"][:, 0, 0, 0].tolist() == [3.0, 4.0, 5.0, 6.0]
    assert [c[2] for c in calls] == [249, 499, 749] and out["timesteps"] == [250, 500, 750]
    assert draws == [(4, 1, 32, 32)] * 3 and [c[3] for c in calls] == [1.0, 2.0, 3.0]      # a FRESH field per member
    assert all(c[0] == (4, 1, 32, 32) and c[1] == (4, 128) for c in calls)
    assert out["final_volume"].shape == (1, 1, 32, 32, 4)
    assert torch.allclose(out["final_volume"], torch.full((1, 1, 32, 32, 4), (249 + 499 + 749) / 3.0))
    assert float(out["loss"]) == 749.0                                               # the last member's loss (:235, :249)
    # step_ensemble override, and the single reconstruction at test_timesteps when noise_ensemble is off
    calls.clear(); draws.clear()
    mod.cfg["step_ensemble"] = [100, 900]
    mod.test_step({"vol": {"data": vol}}, 0)
    assert [c[2] for c in calls] == [99, 899] and len(draws) == 2
    calls.clear(); draws.clear()
    mod.cfg["noise_ensemble"] = False
    mod.cfg["noisetype"] = None
    out = mod.test_step({"vol": {"data": vol[..., :4]}}, 0)                          # exactly 4 slices: no selection
    assert [c[2] for c in calls] == [499] and draws == [] and calls[0][3] is None and out["ind_offset"] == 0


def test_encoder_mirror_inventory_and_oracle_shapes(synth):
    """timm resnet50 (in_chans=1) inventory: 161 weight tensors + 53 x (running_mean, running_var, num_batches_tracked);
    the oracle restatement runs on the synthetic weights and is sensitive to every stage."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import encoder_oracle as EO
    E = load_pkg("DDPM_encoder")
    shapes = synth.encoder_param_shapes(128)
    enc = E.ResNet50Encoder(num_classes=128)
    sd = enc.state_dict()
    assert set(k for k in sd if not k.endswith("num_batches_tracked")) == set(shapes)
    assert sum(k.endswith("num_batches_tracked") for k in sd) == 53
    # torchvision/timm resnet50: 25,557,032 parameters; 1-channel stem (-6,272) and a 128-way fc (-1,786,728)
    assert sum(int(np.prod(s)) for n, s in shapes.items() if "running" not in n) == 25_557_032 - 6_272 - 1_786_728
    w = synth.synth_encoder_state_dict(0, 128)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    sdt = {k: torch.from_numpy(v) for k, v in w.items()}
    x = torch.from_numpy(synth.synth_slices(4, 0, 2, 64, 64))
    y = EO.resnet50_forward(x, sdt)
    assert y.shape == (2, 128) and torch.isfinite(y).all() and 0.05 < float(y.std()) < 50
    sdt2 = dict(sdt)
    sdt2["layer3.4.conv2.weight"] = sdt["layer3.4.conv2.weight"] * 1.01
    assert float((EO.resnet50_forward(x, sdt2) - y).abs().max()) > 1e-6


def test_library_exports_every_declared_symbol():
    lib_mod = load_pkg("_lib")
    header = open(os.path.join(ROOT, "include", "cddpm.h")).read()
    declared = set(re.findall(r"\b(cddpm_[a-z0-9_]+)\s*\(", header)) - {"cddpm_ctx"}
    assert declared == set(lib_mod.SYMBOLS), declared ^ set(lib_mod.SYMBOLS)
    assert os.path.exists(lib_mod.LIB_PATH), "run `python __graft_entry__.py` to build the HIP library"
    lib = lib_mod.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    # pure host entry point: callable without a GPU
    d = lib_mod.UnetDesc()
    d.in_channels = d.out_channels = 1
    d.model_channels, d.num_levels, d.num_res_blocks, d.head_channels, d.cond_dim = 128, 3, 3, 64, 128
    d.channel_mult[0], d.channel_mult[1], d.channel_mult[2] = 1, 2, 2
    d.num_attention_resolutions = 0
    d.timesteps, d.max_batch, d.max_h, d.max_w = 1000, 64, 128, 128
    nbytes = lib.cddpm_workspace_bytes(ctypes.byref(d))
    assert 4e9 < nbytes < 20e9, nbytes               # ~3.4 GB skip stack + 3.2 GB ping-pong + tables + weights
    d.model_channels = 64
    assert lib.cddpm_workspace_bytes(ctypes.byref(d)) == 0      # unsupported shape -> refused
    assert b"model_channels" in lib.cddpm_last_error(None)


def _desc(lib_mod, model_channels=128, mults=(1, 2, 2), max_batch=4, max_h=128, max_w=128):
    d = lib_mod.UnetDesc()
    d.in_channels = d.out_channels = 1
    d.model_channels, d.num_levels, d.num_res_blocks, d.head_channels, d.cond_dim = model_channels, len(mults), 3, 64, 128
    for i, m in enumerate(mults):
        d.channel_mult[i] = m
    d.num_attention_resolutions = 0
    d.timesteps, d.max_batch, d.max_h, d.max_w = 1000, max_batch, max_h, max_w
    return d


def test_create_refuses_a_concatenation_wider_than_the_kernels_hold():
    """(128, (1,2,4,8)) -- the mirror's own default dim_mults -- concatenates 1024 + 1024 channels on the output path:
    more than gn_finalize_kernel's per-channel LDS arrays and the conv's coefficient cache hold. The
    descriptor must be refused by name, before any device is touched, instead of faulting later."""
    lib_mod = load_pkg("_lib")
    lib = lib_mod.load_library()
    d = _desc(lib_mod, 128, (1, 2, 4, 8), max_h=128, max_w=128)
    assert lib.cddpm_workspace_bytes(ctypes.byref(d)) == 0
    assert b"concatenated channels" in lib.cddpm_last_error(None)
    h = ctypes.c_void_p()
    assert lib.cddpm_create(ctypes.byref(h), ctypes.byref(d), 0) != 0 and not h.value
    assert b"concatenated channels" in lib.cddpm_last_error(None)
    # the widest supported: 384 x (1, 2) concatenates 768 + 768 = 1536 at the deepest level; 512 x (1, 2) would be 2048
    assert lib.cddpm_workspace_bytes(ctypes.byref(_desc(lib_mod, 384, (1, 2), max_h=64, max_w=64))) > 0
    assert lib.cddpm_workspace_bytes(ctypes.byref(_desc(lib_mod, 512, (1, 2), max_h=64, max_w=64))) == 0
    assert lib.cddpm_workspace_bytes(ctypes.byref(_desc(lib_mod, 128, (1, 2, 4), max_h=64, max_w=64))) > 0


def test_groupnorm_record_counts_fit_buffers_sized_at_the_maximum_geometry():
    """The statistics buffers are sized once, at max_h x max_w (cddpm_api.hip::plan_workspace); a call at any smaller
    H x W (each <= its maximum, multiples of 4) must never need more records -- the class of bug behind the round-1
    memory fault (a buffer sized for 2 record classes, a kernel writing 8). Every producer launch also checks its count
    against the capacity at run time (conv_launch / stats_of)."""
    lib = load_pkg("_lib").load_library()
    rec = lambda h, w, k: lib.cddpm_stat_records(h, w, k)
    for Hm, Wm in ((128, 128), (256, 256), (96, 96), (64, 96), (32, 48), (160, 192)):
        for ds in (1, 2, 4):
            cap = max(rec(Hm // ds, Wm // ds, k) for k in (0, 1, 2))
            for h in range(4, Hm + 1, 4):
                for w in range(4, Wm + 1, 4):
                    if h % ds or w % ds or h // ds < 1 or w // ds < 1:
                        continue
                    hh, ww = h // ds, w // ds
                    need = max(rec(hh, ww, 0), rec(hh, ww, 2), rec(hh, ww, 1) if (hh % 2 == 0 and ww % 2 == 0) else 0)
                    assert need <= cap, (Hm, Wm, ds, h, w, need, cap)
    assert rec(128, 128, 0) == 2 * 4 * 32 and rec(128, 128, 1) == 8 * 2 * 16 and rec(0, 4, 0) == -1


def _bf16_rne(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 (round to nearest even), returned as float32"""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32)


def test_split_weight_image_reproduces_the_weights():
    """The convolution forms fp32 products from a few 16-bit terms per operand (conv_x6.hip); the whole parity claim
    rests on the terms adding up to the operand: exactly for the bf16 three-term split, to within 2^-23 relative (one
    fp32 ulp, most values exactly) for the pre-scaled fp16 two-term split. Checked on the host packer through the C ABI (no GPU),
    together with the documented image layouts (include/cddpm.h)."""
    lib = load_pkg("_lib").load_library()
    Cout, Cin, taps = 128, 64, 9
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * 0.05).astype(np.float32)
    w.flat[:10] = [0.0, -0.0, 0.25, -0.25, 3.0e-20, 0.2, 0.1 + 2.0 ** -23, -(2.0 ** -100), 0.1, 0.25 - 2.0 ** -26]
    n = lib.cddpm_packed_conv_bytes(Cout, Cin, taps)
    assert lib.cddpm_packed_conv_bytes(100, Cin, taps) == 0 and lib.cddpm_packed_conv_bytes(Cout, Cin, 5) == 0
    buf = (ctypes.c_uint8 * n)()
    wexp = ctypes.c_int(-1)
    fmt = lib.cddpm_pack_conv_weights(w.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), Cout, Cin, taps, buf, ctypes.byref(wexp))
    assert fmt in (0, 1, 2)
    wt = w.reshape(Cout, Cin, taps)
    if fmt == 0:                                    # fp32-MFMA image (CDDPM_CONV=f32)
        assert n == w.size * 4 and wexp.value == 0
        img = np.frombuffer(buf, np.float32).reshape(Cin // 32, taps, 128, 8, 4)
        for j in (0, 1, 2, 77, 127):
            for sl in range(8):
                got = img[:, :, j, sl ^ ((j >> 1) & 7), :]                      # [chunk][tap][4 channels]
                want = wt[j].reshape(Cin // 32, 8, 4, taps)[:, sl].transpose(0, 2, 1)
                assert np.array_equal(got, want)
        return
    ns = 3 if fmt == 1 else 2
    assert n == w.size * 2 * ns
    img = np.frombuffer(buf, np.uint16).reshape(Cin // 32, taps, 128, 4 * ns, 8)
    parts = np.zeros((ns, Cout, Cin, taps), np.float32)
    for j in range(128):
        for s in range(ns):
            for u in range(4):
                slot = (4 * s + (u ^ ((j >> 2) & 3))) if ns == 3 else ((4 * s + u) ^ ((j >> 1) & 7))
                raw = img[:, :, j, slot, :]                                                     # [chunk][tap][8]
                v = (raw.astype(np.uint32) << 16).view(np.float32) if ns == 3 else raw.view(np.float16).astype(np.float32)
                parts[s, j].reshape(Cin // 32, 4, 8, taps)[:, u] = v.transpose(0, 2, 1)
    total = parts.astype(np.float64).sum(axis=0)
    if ns == 3:
        assert wexp.value == 0
        assert np.array_equal(total, wt.astype(np.float64)), "three-term bf16 split must reproduce every weight exactly"
        assert np.array_equal(parts[0], _bf16_rne(wt))                              # hi = bf16(w), round to nearest even
        assert np.array_equal(parts[1], _bf16_rne(wt - parts[0]))                   # mid = bf16(w - hi)
    else:
        e = wexp.value
        mx = float(np.abs(w).max())
        assert 0 <= e <= 24 and 8192.0 <= mx * 2.0 ** e < 16384.0                   # max |w| 2^e in [2^13, 2^14)
        ws = wt.astype(np.float64) * 2.0 ** e
        err = np.abs(total - ws)
        # two 11-bit terms: one fp32 ulp at worst while mid is a normal fp16 (|w 2^e| >= 2^-2), an absolute 2^-25 below
        assert (err <= np.maximum(np.abs(ws) * 2.0 ** -23, 2.0 ** -25)).all()
        big = np.abs(ws) >= 0.25
        assert np.sqrt(np.mean((err[big] / np.abs(ws[big])) ** 2)) < 2.0 ** -24 and (err[big] == 0).mean() > 0.5
        assert np.array_equal(parts[0], (wt * np.float32(2.0 ** e)).astype(np.float16).astype(np.float32))   # hi = fp16(w 2^e)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_engine_fails_loudly_without_gpu():
    eng = load_pkg("engine")
    with pytest.raises(RuntimeError, match="no HIP device|MI355X"):
        eng.CddpmEngine(timesteps=10, max_batch=1, max_h=32, max_w=32)


def test_product_never_imports_oracle():
    pkg_dir = os.path.join(ROOT, "conditioned-diffusion-models-uad_amd")
    for dirpath, _d, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "cddpm_oracle" not in src and "import oracle" not in src and "ref_harness" not in src, f


# ---- LDS images of the fused convolution (conv_x6.hip): the swizzles are chosen against the bank / lane-group rules of
#      MI355X_MICROARCH.md (LDS): a ds_read_b128 is served in four groups of 16 lanes over 64 banks of 4 bytes, a
#      ds_write_b64 in four groups of 16 consecutive lanes over 32 banks. This restates the kernel's slot functions and
#      checks that every fragment read of every tap, and every patch store, is conflict-free.
_B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
_B128_GROUPS = _B128_GROUPS + [[l + 32 for l in g] for g in _B128_GROUPS]


def _read_b128_cycles(addr):          # byte address per lane -> LDS cycles (4 = conflict-free)
    total = 0
    for grp in _B128_GROUPS:
        banks = {}
        for l in grp:
            for d in range(4):
                banks.setdefault((addr[l] // 4 + d) % 64, set()).add(addr[l] // 4 + d)
        total += max(len(v) for v in banks.values())
    return total


def _write_b64_cycles(addr):          # None = lane masked off
    total = 0
    for g in range(4):
        banks = {}
        for l in range(16 * g, 16 * g + 16):
            if addr[l] is None:
                continue
            for d in range(2):
                banks.setdefault((addr[l] // 4 + d) % 32, set()).add(addr[l] // 4 + d)
        total += max([len(v) for v in banks.values()] + [1])
    return total


def _swz16(pc):
    return (2 * ((pc >> 1) & 3)) ^ (4 * (pc & 1))


def test_conv_lds_images_are_bank_conflict_free():
    PW = 34
    # 16x16x32 form, A operand: lane (r = l & 15, g = l >> 4) reads pixel group t16 of wave row pair wm, shifted by (ky, kx)
    for wm in range(4):
        for ky in range(3):
            for kx in range(3):
                for t16 in range(4):
                    for sp in range(2):
                        addr = []
                        for l in range(64):
                            r16, g = l & 15, l >> 4
                            pc = (t16 & 1) * 16 + r16 + kx
                            pix = (2 * wm + (t16 >> 1) + ky) * PW + pc
                            addr.append(16 * (pix * 8 + ((4 * sp + g) ^ _swz16(pc))))
                        assert _read_b128_cycles(addr) == 4, (wm, ky, kx, t16, sp)
    # ... and the slots of the four pixel groups are constant offsets from the first one (instruction offsets in the kernel)
    for kx in range(3):
        for r16 in range(16):
            for g in range(4):
                pc0 = r16 + kx
                base = pc0 * 8 + (g ^ _swz16(pc0))
                for t16 in range(4):
                    pc = (t16 & 1) * 16 + r16 + kx
                    assert ((t16 >> 1) * PW + pc) * 8 + (g ^ _swz16(pc)) == base + ((t16 >> 1) * PW + (t16 & 1) * 16) * 8
    # weights (host-packed image, both MFMA shapes): slot (4 s + u) ^ ((row >> 1) & 7) of cout row `row`
    wslot = lambda row, sp, u: row * 8 + ((4 * sp + u) ^ ((row >> 1) & 7))
    for wn in range(2):
        for n in range(4):
            for sp in range(2):
                addr = [16 * wslot(64 * wn + 16 * n + (l & 15), sp, l >> 4) for l in range(64)]
                assert _read_b128_cycles(addr) == 4
                assert all(wslot(64 * wn + 16 * n + r, sp, g) == wslot(64 * wn + r, 0, g) + 128 * n if sp == 0 else True
                           for r in range(16) for g in range(4))
        for nt in range(2):
            for sp in range(2):
                for jk in range(2):
                    addr = [16 * wslot(64 * wn + 32 * nt + (l & 31), sp, 2 * jk + (l >> 5)) for l in range(64)]
                    assert _read_b128_cycles(addr) == 4
    # 32x32x16 form, A operand: pixel-index swizzle with the parity flip
    aslot = lambda row, sp, u: row * 8 + ((4 * sp + u) ^ ((row >> 1) & 7) ^ (4 * (row & 1)))
    for base in range(0, 3 * PW + 3):
        for sp in range(2):
            for jk in range(2):
                addr = [16 * aslot(base + (l & 31), sp, 2 * jk + (l >> 5)) for l in range(64)]
                assert _read_b128_cycles(addr) == 4
    # patch stores (8 bytes per lane: thread = (pixel q, channel quad c4)), both swizzles
    for wave in range(8):
        for k in range(6):
            for sp in range(2):
                a16, a32 = [], []
                for l in range(64):
                    tid = wave * 64 + l
                    c4, q = tid & 7, (tid >> 3) + 64 * k
                    if q >= 10 * PW:
                        a16.append(None); a32.append(None)
                        continue
                    pc = q % PW
                    a16.append(16 * (q * 8 + ((4 * sp + (c4 >> 1)) ^ _swz16(pc))) + 8 * (c4 & 1))
                    a32.append(16 * aslot(q, sp, c4 >> 1) + 8 * (c4 & 1))
                assert _write_b64_cycles(a16) == 4 and _write_b64_cycles(a32) == 4



def test_training_program_covers_the_state_dict(synth):
    """host logic of the training step (training.unet_program / conv_table): the operator sequence mirrors UNetModel's construction
    (reference OpenAI_Unet.py:604-797) -- every convolution, GroupNorm and embedding Linear of the reference's state_dict is visited once,
    with the channel counts of its weight tensor, and the skip stack balances (each pushed tensor is popped by exactly one output block)"""
    tr = load_pkg("training")
    for mult, nres in (((1, 2, 2), 3), ((1, 2, 4), 2)):
        prog = tr.unet_program(128, mult, nres)
        tab = tr.conv_table(prog)
        shapes = synth.unet_param_shapes(model_channels=128, channel_mult=mult, num_res_blocks=nres)
        for k, (co, ci, ks, _folded, _grp) in tab.items():
            assert tuple(shapes[k + ".weight"][:2]) == (co, ci) and shapes[k + ".weight"][2] == ks, k
        convs = {k[:-7] for k, sh in shapes.items() if k.endswith(".weight") and len(sh) >= 3}
        assert convs - set(tab) == {"input_blocks.0.0", "out.2"}          # the two one-channel convolutions have their own kernels
        pushed = 1 + sum(1 for kind, _n, a in prog if kind == "res" and a.get("push"))
        popped = [a["concat"] for kind, _n, a in prog if kind == "res" and a.get("concat")]
        assert pushed == len(popped)
        visited = {n for _k, n, _a in prog}
        for k in shapes:
            if k.startswith(("time_embed", "label_emb")):
                continue
            assert any(k.startswith(v + ".") for v in visited), k
