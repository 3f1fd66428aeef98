"""GPU parity of the whole path through the C ABI: UNet forward (per block, vs the oracle run on the same
host), the committed reference golden vectors, and the reverse loops (config 1 and friends).
North-star tolerance: |delta| <= 1e-4 per pixel on the reconstruction; UNet outputs are O(1), same bound."""
import numpy as np
import pytest
import torch

from conftest import golden, load_pkg

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def eng1000(engine_factory):
    return engine_factory(timesteps=1000, max_batch=4, max_h=128, max_w=128)


@pytest.fixture(scope="module")
def eng50(engine_factory):
    return engine_factory(timesteps=50, max_batch=4, max_h=128, max_w=128)


def inputs(synth, B, H, W, slice0=0):
    x = torch.from_numpy(synth.noise_xT(2, slice0, B, H, W))
    cond = torch.from_numpy(synth.synth_cond(1, slice0, B))
    return x, cond


def test_unet_blocks_vs_oracle(eng1000, synth, oracle, sd_torch):
    B, H, W = 2, 32, 32
    x, cond = inputs(synth, B, H, W)
    taps = {}
    with torch.no_grad():
        ref = oracle.unet_forward(x, torch.full((B,), 500), cond, sd_torch, taps=taps)
    got = eng1000.forward_with_taps(x.cuda(), 500, cond.cuda())
    report, worst = [], 0.0
    for name in eng1000.block_names():
        r = ref if name == "out" else taps[name]
        g = got[name].cpu()
        assert g.shape == r.shape, (name, g.shape, r.shape)
        err = float((g - r).abs().max())
        rel = err / (1e-6 + float(r.abs().max()))
        report.append(f"{name:18s} max|d|={err:.3e} rel={rel:.3e}")
        worst = max(worst, rel)
    print("\n".join(report))
    assert worst < 2e-5, "\n".join(report)
    assert float((got["out"].cpu() - ref).abs().max()) < TOL


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 64, 96), (1, 96, 96), (1, 128, 128)])
def test_unet_forward_golden(eng1000, synth, B, H, W):
    g = golden(f"unet_fwd_B{B}_{H}x{W}")
    x, cond = inputs(synth, B, H, W)
    xd, cd = x.cuda(), cond.cuda()
    for key in g.files:
        if key == "tmixed":
            t = torch.tensor([123, 877][:B], dtype=torch.int32)
        else:
            t = int(key[1:])
        out = eng1000.unet_forward(xd, t, cd).cpu().numpy()
        err = np.abs(out - g[key]).max()
        assert err < TOL, (key, err)


LOOPS = [("loop_B2_32x32_T1000_start8", 1000, 8, 2, 32, 32, 0),
         ("loop_B2_32x32_T50_start0", 50, 0, 2, 32, 32, 0),
         ("loop_B3_32x48_T1000_start5_slice7", 1000, 5, 3, 32, 48, 7),
         ("loop_cfg1_B4_128x128_T50_start0", 50, 0, 4, 128, 128, 0),
         ("loop_B1_128x128_T1000_start50", 1000, 50, 1, 128, 128, 0),
         ("loop_B2_32x32_T1000_start0", 1000, 0, 2, 32, 32, 0)]       # the FULL T = 1000 chain from pure noise


@pytest.mark.parametrize("name,T,start_t,B,H,W,slice0", LOOPS, ids=[l[0] for l in LOOPS])
def test_reverse_loop_golden(eng1000, eng50, synth, name, T, start_t, B, H, W, slice0):
    """p_sample_loop vs the reference's own output (tests/golden), explicit z_t injected."""
    eng = eng1000 if T == 1000 else eng50
    steps = T if start_t == 0 else start_t
    x, cond = inputs(synth, B, H, W, slice0)
    noise = np.zeros((steps, B, 1, H, W), np.float32)
    for t in range(1, steps):
        noise[t] = synth.noise_z(3, t, slice0, B, H, W)
    out = eng.reverse(x.cuda(), cond.cuda(), steps, noise=torch.from_numpy(noise).cuda()).cpu().numpy()
    ref = golden(name)["out"]
    err = np.abs(out - ref).max()
    print(name, f"max|delta| vs reference golden: {err:.3e}  rms {np.sqrt(np.mean((out - ref) ** 2)):.3e}")
    assert out.min() >= 0.0 and out.max() <= 1.0
    assert err < TOL, err
    # where a float64 run of the oracle exists, show both implementations against it: the reference's own fp32
    # rounding noise is the floor of any |HIP - reference| comparison
    import os
    from conftest import GOLD
    if os.path.exists(os.path.join(GOLD, name + "_fp64.npz")):
        truth = golden(name + "_fp64")["out"]
        e_ref, e_hip = np.abs(ref - truth).max(), np.abs(out - truth).max()
        print(name, f"vs fp64: reference {e_ref:.3e} (rms {np.sqrt(np.mean((ref - truth) ** 2)):.3e}), "
                    f"HIP {e_hip:.3e} (rms {np.sqrt(np.mean((out - truth) ** 2)):.3e})")
        assert e_hip < TOL
    # The device-Philox path (what bench.py times): same kernels, z drawn on the device. Its integer stream is
    # synth.py's, but logf / sincosf differ from numpy's by ulps, i.e. its INPUTS differ slightly from the golden's.
    # Pinned exactly instead: (a) the Philox run equals, bit for bit, the explicit-noise run fed the downloaded Philox
    # draws (so the strict bound above covers its kernels), (b) the draws are synth.noise_z to a few ulp.
    zdev = torch.zeros((steps, B, 1, H, W), dtype=torch.float32, device="cuda")
    for t in range(1, steps):
        zdev[t] = eng.noise_fill(B, H, W, seed=3, stream_id=synth.STREAM_Z, t=t, slice0=slice0)
    out2 = eng.reverse(x.cuda(), cond.cuda(), steps, noise=None, seed=3, slice0=slice0)
    out3 = eng.reverse(x.cuda(), cond.cuda(), steps, noise=zdev, seed=0, slice0=0)
    assert torch.equal(out2, out3)
    assert float((zdev.cpu() - torch.from_numpy(noise)).abs().max()) < 4e-6
    err2 = np.abs(out2.cpu().numpy() - ref).max()
    print(name, f"device-RNG max|delta| (inputs differ by ulps): {err2:.3e}")
    assert err2 < TOL, err2       # north_star's bound holds for the device-RNG path too (its inputs differ from the golden's by ulps)


def test_rounding_yardstick_fp64(eng50, synth, oracle, sd_torch):
    """How far may two correct fp32 implementations differ? Run the oracle in float64 (no fp32 rounding) and
    compare both the reference's fp32 output (golden) and the HIP output with it on the 50-step loop."""
    B, H, W, T = 2, 32, 32, 50
    x, cond = inputs(synth, B, H, W)
    zs = {t: torch.from_numpy(synth.noise_z(3, t, 0, B, H, W)) for t in range(1, T)}
    sd64, buf64 = oracle.to_float64(sd_torch), oracle.to_float64(oracle.schedule_buffers(T))
    truth = oracle.p_sample_loop(x.double(), cond.double(), sd64, buf64, lambda t: zs[t].double(), start_t=0).numpy()
    noise = torch.zeros(T, B, 1, H, W)
    for t, z in zs.items():
        noise[t] = z
    hip = eng50.reverse(x.cuda(), cond.cuda(), T, noise=noise.cuda()).cpu().numpy()
    ref = golden("loop_B2_32x32_T50_start0")["out"]
    e_ref, e_hip, e_pair = np.abs(ref - truth).max(), np.abs(hip - truth).max(), np.abs(hip - ref).max()
    print(f"fp64 yardstick (T=50, 32x32): reference-fp32 vs fp64 {e_ref:.3e}; HIP vs fp64 {e_hip:.3e}; HIP vs reference {e_pair:.3e}")
    assert e_hip < TOL
    assert e_hip < 4 * e_ref + 1e-6, "HIP rounding noise should be of the order of the reference's own"


def test_p_sample_single_step(eng1000, synth, oracle, sd_torch):
    B, H, W = 2, 32, 32
    x, cond = inputs(synth, B, H, W)
    z = torch.from_numpy(synth.noise_z(3, 700, 0, B, H, W))
    buf = oracle.schedule_buffers(1000)
    with torch.no_grad():
        ref = oracle.p_sample(x, 700, cond, sd_torch, buf, z)
    got = eng1000.p_sample(x.cuda(), 700, cond.cuda(), z=z.cuda()).cpu()
    assert float((got - ref).abs().max()) < TOL


def test_reverse_graph_replay_equals_eager(eng50, synth, monkeypatch):
    """CDDPM_GRAPH=1: cddpm_reverse replays one captured step as a HIP graph (device-resident t) instead of launching
    every step. Same kernels, same order: the results are bit-identical, with explicit noise and with the device Philox."""
    B, H, W, steps = 2, 32, 32, 12
    x, cond = inputs(synth, B, H, W)
    noise = np.zeros((steps, B, 1, H, W), np.float32)
    for t in range(1, steps):
        noise[t] = synth.noise_z(3, t, 0, B, H, W)
    nz = torch.from_numpy(noise).cuda()
    for kw in (dict(noise=nz), dict(noise=None, seed=11, slice0=5)):
        monkeypatch.setenv("CDDPM_GRAPH", "0")
        eager = eng50.reverse(x.cuda(), cond.cuda(), steps, **kw)
        monkeypatch.setenv("CDDPM_GRAPH", "1")
        replay = eng50.reverse(x.cuda(), cond.cuda(), steps, **kw)
        again = eng50.reverse(x.cuda(), cond.cuda(), steps, **kw)
        assert torch.equal(eager, replay) and torch.equal(eager, again)
        assert float(eager.min()) >= 0.0 and float(eager.max()) <= 1.0


def test_sharding_invariance(eng1000, synth):
    """a slice's result depends only on its global index: batch [0..3] == batches [0,1] + [2,3] bit for bit"""
    H = W = 32
    x, cond = inputs(synth, 4, H, W)
    full = eng1000.reverse(x.cuda(), cond.cuda(), 6, seed=11, slice0=0)
    lo = eng1000.reverse(x[:2].cuda(), cond[:2].cuda(), 6, seed=11, slice0=0)
    hi = eng1000.reverse(x[2:].cuda(), cond[2:].cuda(), 6, seed=11, slice0=2)
    assert torch.equal(full[:2], lo) and torch.equal(full[2:], hi)


def test_full_batch_is_the_concatenation_of_small_batches(engine_factory, synth):
    """BASELINE config 2's size (64 slices of 128x128 on one GPU) through a size-independent property: every slice of
    the B = 64 run equals, bit for bit, the same slice reconstructed in a batch of 4 on the same handle (the kernel plan --
    tile order, split-K factors -- is a property of the handle's maximum geometry, never of the call's batch). The
    full-length version of this check is tests/test_gpu_headline.py."""
    eng = engine_factory(timesteps=1000, max_batch=64, max_h=128, max_w=128)
    H = W = 128
    x, cond = inputs(synth, 64, H, W)
    full = eng.reverse(x.cuda(), cond.cuda(), 3, seed=11, slice0=0)
    assert bool(torch.isfinite(full).all()) and float(full.std()) > 0.01
    for s0 in (0, 28, 60):
        part = eng.reverse(x[s0:s0 + 4].cuda(), cond[s0:s0 + 4].cuda(), 3, seed=11, slice0=s0)
        assert torch.equal(full[s0:s0 + 4], part), s0
    eng.close()


def test_small_batch_plan_and_large_batch_plan_agree_to_rounding(engine_factory, eng1000, synth):
    """A handle sized for small batches (max_batch 4) cuts the K loop of the layers that cannot fill the chip into split-K
    ranges (cddpm_api.hip::plan_ksplit) -- a different summation order from a max_batch 64 handle, so the two agree to fp32
    rounding, not bit for bit; each is checked against the reference goldens on its own (eng1000 here and in every golden
    test above: split; the 64-handle: unsplit, also tests/test_gpu_headline.py). Within one handle the bits never depend on B."""
    big = engine_factory(timesteps=1000, max_batch=64, max_h=128, max_w=128)
    for (B, H, W) in ((2, 32, 32), (1, 128, 128)):
        g = golden(f"unet_fwd_B{B}_{H}x{W}")
        x, cond = inputs(synth, B, H, W)
        a = eng1000.unet_forward(x.cuda(), 500, cond.cuda()).cpu().numpy()
        b = big.unet_forward(x.cuda(), 500, cond.cuda()).cpu().numpy()
        ea, eb, eab = np.abs(a - g["t500"]).max(), np.abs(b - g["t500"]).max(), np.abs(a - b).max()
        print(f"{B}x{H}x{W}: split-K plan vs golden {ea:.2e}, unsplit plan vs golden {eb:.2e}, between them {eab:.2e}")
        assert ea < TOL and eb < TOL and eab < 2e-5
    x, cond = inputs(synth, 4, 64, 96)
    one = torch.cat([eng1000.unet_forward(x[i:i + 1].cuda(), 321, cond[i:i + 1].cuda()) for i in range(4)])
    assert torch.equal(one, eng1000.unet_forward(x.cuda(), 321, cond.cuda()))
    big.close()


def test_errors_are_loud(eng1000, synth):
    x, cond = inputs(synth, 2, 32, 32)
    with pytest.raises(RuntimeError):
        eng1000.unet_forward(x, 3, cond)            # CPU tensors: no fallback
    with pytest.raises(RuntimeError):
        eng1000.unet_forward(x.cuda()[:, :, :30], 3, cond.cuda())   # H not a multiple of 4
    with pytest.raises(RuntimeError):
        eng1000.unet_forward(x.cuda(), 1000, cond.cuda())           # t out of range
    # a non-finite value anywhere upstream must surface (torch.clamp semantics in the posterior step), not be clamped away
    bad = x.clone()
    bad[1, 0, 5, 7] = float("nan")
    with pytest.raises(FloatingPointError, match="CDDPM_CONV"):
        eng1000.reverse(bad.cuda(), cond.cuda(), 2)
    ok = eng1000.reverse(x.cuda(), cond.cuda(), 2)
    assert bool(torch.isfinite(ok).all())


def test_unet_forward_256_config3(engine_factory, synth, oracle, sd_torch):
    """BASELINE config 3 geometry: 256x256 slices (attention over N = 4096 tokens), same weights."""
    eng = engine_factory(timesteps=1000, max_batch=1, max_h=256, max_w=256)
    x, cond = inputs(synth, 1, 256, 256)
    with torch.no_grad():
        ref = oracle.unet_forward(x, torch.full((1,), 500), cond, sd_torch)
    got = eng.unet_forward(x.cuda(), 500, cond.cuda()).cpu()
    err = float((got - ref).abs().max())
    print("256x256 forward max|delta| vs oracle:", err)
    assert err < TOL
    eng.close()


def test_reverse_loop_256_config3_golden(engine_factory, synth):
    """BASELINE config 3 geometry against the reference's own output: 256x256, the last 12 steps of a T = 1000 chain."""
    eng = engine_factory(timesteps=1000, max_batch=1, max_h=256, max_w=256)
    B, H, W, steps = 1, 256, 256, 12
    x, cond = inputs(synth, B, H, W)
    noise = np.zeros((steps, B, 1, H, W), np.float32)
    for t in range(1, steps):
        noise[t] = synth.noise_z(3, t, 0, B, H, W)
    out = eng.reverse(x.cuda(), cond.cuda(), steps, noise=torch.from_numpy(noise).cuda()).cpu().numpy()
    ref = golden("loop_cfg3_B1_256x256_T1000_start12")["out"]
    err = float(np.abs(out - ref).max())
    print("256x256 reverse loop max|delta| vs reference:", err)
    assert err < TOL
    eng.close()


def test_config4_sharded_residual_maps_single_rank(eng1000, synth):
    """BASELINE config 4 in miniature on one rank: slices walked in chunks, residual maps |x - reco| gathered;
    identical to reconstructing every slice in one batch (noise keyed by global slice index)."""
    sh = load_pkg("sharding")
    n, H, W, steps = 6, 32, 32, 3
    res = sh.residual_maps_sharded(eng1000, n, H, W, seed_inputs=5, seed_cond=1, seed_noise=9, t_start=steps, chunk=4)
    assert res.shape == (n, 1, H, W)
    x = torch.from_numpy(synth.synth_slices(5, 0, n, H, W)).cuda()
    cond = torch.from_numpy(synth.synth_cond(1, 0, n)).cuda()
    # engine capacity is 4 slices: do the full batch in two halves keyed by slice0 -- must equal the chunked walk
    parts = []
    for s0, cnt in ((0, 3), (3, 3)):
        xT = eng1000.noise_fill(cnt, H, W, seed=9, stream_id=synth.STREAM_XT, slice0=s0)
        parts.append((x[s0:s0 + cnt] - eng1000.reverse(xT, cond[s0:s0 + cnt], steps, seed=9, slice0=s0)).abs())
    assert torch.equal(res, torch.cat(parts, 0))


@pytest.mark.parametrize("B,H,W", [(1, 4, 4), (3, 12, 20), (2, 8, 44), (5, 36, 28), (2, 4, 132)])
def test_ragged_geometries_vs_oracle(engine_factory, synth, oracle, sd_torch, B, H, W):
    """Smallest and ragged image sizes (H, W only need to be multiples of 4: partial 8x32 tiles in every layer, 1x1 images at
    the deepest level for 4x4, widths beyond one tile), on handles created for exactly that geometry (small-batch split-K plan)
    and on a larger handle (different plan, partial use of its buffers): UNet forward per block and a 3-step reverse loop."""
    x, cond = inputs(synth, B, H, W)
    t = torch.tensor([(37 * i + 5) % 1000 for i in range(B)])
    taps = {}
    with torch.no_grad():
        ref = oracle.unet_forward(x, t, cond, sd_torch, taps=taps)
    zs = {s: torch.from_numpy(synth.noise_z(3, s, 0, B, H, W)) for s in range(1, 3)}
    loop_ref = oracle.p_sample_loop(x, cond, sd_torch, oracle.schedule_buffers(1000), lambda s: zs[s], start_t=3).numpy()
    noise = torch.zeros(3, B, 1, H, W)
    for s, z in zs.items():
        noise[s] = z
    q8 = lambda v: (v + 7) // 8 * 8
    for (mb, mh, mw) in ((B, H, W), (8, q8(H) + 8, q8(W) + 24)):
        eng = engine_factory(timesteps=1000, max_batch=mb, max_h=mh, max_w=mw)
        got = eng.forward_with_taps(x.cuda(), t, cond.cuda())
        worst = 0.0
        for name in eng.block_names():
            r = ref if name == "out" else taps[name]
            g = got[name].cpu()
            assert g.shape == r.shape, (name, g.shape, r.shape)
            worst = max(worst, float((g - r).abs().max()) / (1e-6 + float(r.abs().max())))
        assert worst < 2e-5, (mb, mh, mw, worst)
        out = eng.reverse(x.cuda(), cond.cuda(), 3, noise=noise.cuda()).cpu().numpy()
        assert np.abs(out - loop_ref).max() < TOL, (mb, mh, mw)
        eng.close()


def test_reverse_on_two_streams_is_bit_identical(engine_factory, synth):
    """engine.reverse_two_streams: a small batch as two half-batches on two handles / two streams, step by step in alternation (one
    half's small launches hide behind the other half's convolutions) -- the same bits as the one-stream reverse, odd batch included"""
    T, steps, H, W = 1000, 12, 32, 32
    a = engine_factory(timesteps=T, max_batch=4, max_h=H, max_w=W)
    b = engine_factory(timesteps=T, max_batch=4, max_h=H, max_w=W)
    for B in (4, 3):
        x = torch.from_numpy(synth.noise_xT(2, 5, B, H, W)).cuda()
        cond = torch.from_numpy(synth.synth_cond(1, 5, B)).cuda()
        one = a.reverse(x, cond, steps, seed=3, slice0=5)
        two = a.reverse_two_streams(b, x, cond, steps, seed=3, slice0=5)
        assert torch.equal(one, two), B
