"""Simplex noise: the oracle against the reference's own generator (golden, CPU), and the HIP
kernel against both (GPU) -- integer permutation and float64 field arithmetic, so everything is compared bit for bit."""
import numpy as np
import pytest
import torch

from conftest import golden, load_pkg

CASES = [(3, 32), (-9876543210, 32), (1234567, 96), (9999999999, 128), (-6534728787, 16)]


@pytest.fixture(scope="module")
def sx():
    import simplex_oracle
    return simplex_oracle


@pytest.mark.parametrize("seed,n", CASES)
def test_oracle_matches_reference_generator(sx, seed, n):
    g = golden("simplex")
    key = f"seed{seed}_n{n}"
    assert np.array_equal(sx.init_perm(seed).astype(np.int16), g[key + "_perm"])
    field = sx.rand_2d_octaves(sx.init_perm(seed), n, n)
    assert np.array_equal(field, g[key + "_f64"])                      # float64, bit for bit
    assert np.array_equal(sx.gen_noise(seed, (1, 1, n, n))[0, 0].view(np.uint16), g[key + "_f16"])
    assert float(g["oracle_vs_reference_f64_maxabs"]) == 0.0


def test_oracle_gen_noise_end_to_end(sx):
    g = golden("simplex")
    out = sx.gen_noise(int(g["gen_noise_seed"]), (3, 1, 32, 32))
    assert out.dtype == np.float16 and np.array_equal(out.view(np.uint16), g["gen_noise_f16"])
    assert np.array_equal(out[0], out[2])                               # the same field for every batch item
    with pytest.raises(ValueError):
        sx.rand_2d_octaves(sx.init_perm(1), 16, 24)                     # square only, as the reference


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n", CASES)
def test_hip_simplex_bit_exact(engine_factory, seed, n):
    eng = engine_factory(timesteps=50, max_batch=2, max_h=32, max_w=32)
    g = golden("simplex")
    out = eng.simplex_noise(3, n, n, seed=seed)
    assert out.dtype == torch.float16 and tuple(out.shape) == (3, 1, n, n)
    bits = out.cpu().numpy().view(np.uint16)
    ref = g[f"seed{seed}_n{n}_f16"]
    assert np.array_equal(bits[0, 0], ref), f"{int((bits[0, 0] != ref).sum())} of {ref.size} half values differ"
    assert np.array_equal(bits[1], bits[0]) and np.array_equal(bits[2], bits[0])


@pytest.mark.gpu
def test_gen_noise_mirror_and_simplex_loop(engine_factory, sd_np, synth, oracle, sd_torch):
    """gen_noise(cfg, shape) mirror (seed drawn from numpy's RNG like the reference) and the simplex branch of
    p_sample_loop against the oracle's restatement of that branch."""
    U, D, GN = load_pkg("OpenAI_Unet"), load_pkg("cond_DDPM"), load_pkg("generate_noise")
    g = golden("simplex")
    cfg = {"noisetype": "simplex"}
    m = U.UNetModel(image_size=(32, 32), in_channels=1, model_channels=128, out_channels=1, num_res_blocks=3,
                    attention_resolutions=(3, 6, 12), channel_mult=[1, 2, 2], num_classes=128, num_head_channels=64,
                    use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    d = D.GaussianDiffusion(m, image_size=(32, 32), timesteps=1000, objective="pred_x0", channels=1, cfg=cfg).cuda()
    eng = d._engine(3, 32, 32, torch.device("cuda", 0))
    np.random.seed(5)
    ns = GN.gen_noise(cfg, (3, 1, 32, 32), engine=eng)
    assert ns.dtype == torch.float16 and np.array_equal(ns.cpu().numpy().view(np.uint16), g["gen_noise_f16"])
    with pytest.raises(ValueError):
        GN.gen_noise({"noisetype": "gauss"}, (1, 1, 32, 32), engine=eng)
    # simplex branch of the reverse loop, T = 4: seeds = what numpy's RNG hands to newSeed, call by call
    B, H, W, T = 2, 32, 32, 4
    x0 = torch.from_numpy(synth.synth_slices(2, 0, B, H, W)) * 2 - 1
    cond = torch.from_numpy(synth.synth_cond(1, 0, B))
    np.random.seed(11)
    draws = [int(np.random.randint(-10000000000, 10000000000)) for _ in range(2 * T)]
    seeds = draws[1::2]                       # every gen_noise call draws twice, the second draw seeds the field
    ref = oracle.p_sample_loop_simplex(x0, cond, sd_torch, oracle.schedule_buffers(1000), seeds, start_t=T)
    np.random.seed(11)
    out = d.p_sample_loop((B, 1, H, W), cond=cond.cuda(), start_t=T, noise=torch.zeros(1), x_start=x0.cuda())
    err = float((out.cpu() - ref).abs().max())
    print("simplex-branch loop max|delta| vs oracle:", err)
    assert err < 1e-4
    with pytest.raises(ValueError):
        d.p_sample_loop((B, 1, H, W), cond=cond.cuda(), start_t=0, noise=torch.zeros(1), x_start=x0.cuda())
    m._hip.close()
