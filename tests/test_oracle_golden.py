"""CPU: the oracle (oracle/cddpm_oracle.py) against the committed golden vectors, which are outputs of the
REFERENCE itself (made by oracle/make_golden.py in the build container). Same torch ops, same inputs: the
expected difference is exactly zero on the same torch build; 2e-6 is allowed for other CPU/torch builds."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, golden

TOL = 2e-6


def test_manifest_pins_oracle_to_reference():
    m = json.load(open(os.path.join(GOLD, "MANIFEST.json")))
    for name, case in m["cases"].items():
        v = case.get("oracle_vs_reference_maxabs")
        if isinstance(v, dict):
            assert all(x <= 1e-6 for x in v.values()), (name, v)
        elif v is not None:
            assert v <= 1e-6, (name, v)


@pytest.mark.parametrize("T", [1000, 50])
def test_schedule_buffers(oracle, T):
    g = golden(f"schedule_T{T}")
    buf = oracle.schedule_buffers(T)
    assert set(g.files) == set(buf.keys()) and len(buf) == 13
    for k in g.files:
        np.testing.assert_array_equal(buf[k].numpy(), g[k], err_msg=k)


def test_schedule_known_answers(oracle):
    """values measured on the reference"""
    b = oracle.schedule_buffers(1000)
    kat = {"betas": {0: 4.12842237e-05, 1: 4.61417512e-05, 500: 0.00315569155, 998: 0.749999404, 999: 0.999000013},
           "alphas_cumprod": {0: 0.999958694, 500: 0.492285162, 999: 2.42876697e-09},
           "posterior_mean_coef1": {0: 1.0, 1: 0.527781427, 999: 0.00155689172},
           "posterior_mean_coef2": {0: 0.0, 1: 0.472218603, 999: 0.0316227004},
           "posterior_log_variance_clipped": {0: -46.0517006, 1: -10.7340822, 999: -0.00100292673}}
    for name, d in kat.items():
        for i, v in d.items():
            assert abs(float(b[name][i]) - v) <= 2e-7 * max(1.0, abs(v)), (name, i)
    assert abs(float(oracle.schedule_buffers(50)["betas"][25]) - 0.0630497783) < 1e-8


def test_timestep_embedding(oracle):
    g = golden("timestep_embedding")
    e = oracle.timestep_embedding(torch.from_numpy(g["t"]), 128).numpy()
    np.testing.assert_allclose(e, g["emb"], rtol=0, atol=TOL)
    e1 = oracle.timestep_embedding(torch.tensor([1, 999]), 128)
    assert abs(float(e1[0, 0]) - 0.540302336) < 1e-7 and abs(float(e1[0, 64]) - 0.841470957) < 1e-7   # cos first
    assert abs(float(e1[1, 63]) - 0.993353069) < 1e-6 and abs(float(e1[1, 127]) - 0.115106970) < 1e-6


def _inputs(synth, B, H, W, slice0=0):
    return (torch.from_numpy(synth.noise_xT(2, slice0, B, H, W)), torch.from_numpy(synth.synth_cond(1, slice0, B)))


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 64, 96), (1, 96, 96), (1, 128, 128)])
def test_unet_forward(oracle, synth, sd_torch, B, H, W):
    g = golden(f"unet_fwd_B{B}_{H}x{W}")
    x, cond = _inputs(synth, B, H, W)
    for key in g.files:
        t = torch.tensor([123, 877][:B]) if key == "tmixed" else torch.full((B,), int(key[1:]))
        with torch.no_grad():
            out = oracle.unet_forward(x, t, cond, sd_torch).numpy()
        assert np.abs(out - g[key]).max() <= TOL, key
        assert np.abs(g[key]).max() > 0.1      # non-vacuous: the synthetic weights are nowhere zero


LOOPS = [("loop_B2_32x32_T1000_start8", 1000, 8, 2, 32, 32, 0),
         ("loop_B2_32x32_T50_start0", 50, 0, 2, 32, 32, 0),
         ("loop_B3_32x48_T1000_start5_slice7", 1000, 5, 3, 32, 48, 7),
         ("loop_B1_128x128_T1000_start50", 1000, 50, 1, 128, 128, 0),
         ("loop_cfg3_B1_256x256_T1000_start12", 1000, 12, 1, 256, 256, 0)]     # BASELINE config 3 geometry, ~40 s
if os.environ.get("CDDPM_SLOW"):
    LOOPS.append(("loop_cfg1_B4_128x128_T50_start0", 50, 0, 4, 128, 128, 0))   # ~4 min on 8 cores
    LOOPS.append(("loop_B2_32x32_T1000_start0", 1000, 0, 2, 32, 32, 0))        # ~3 min: the full T = 1000 chain


@pytest.mark.parametrize("name,T,start_t,B,H,W,slice0", LOOPS, ids=[l[0] for l in LOOPS])
def test_reverse_loop(oracle, synth, sd_torch, name, T, start_t, B, H, W, slice0):
    x, cond = _inputs(synth, B, H, W, slice0)
    buf = oracle.schedule_buffers(T)
    out = oracle.p_sample_loop(x, cond, sd_torch, buf, lambda t: torch.from_numpy(synth.noise_z(3, t, slice0, B, H, W)),
                               start_t=start_t).numpy()
    ref = golden(name)["out"]
    assert np.abs(out - ref).max() <= TOL
    assert ref.min() >= 0 and ref.max() <= 1 and ref.std() > 0.01


DDIM = {"ddim_B2_32x32_T1000_S10_eta1": (32, 32, 2, 1000, 10, 1.0, 0),
        "ddim_B2_32x32_T1000_S10_eta0": (32, 32, 2, 1000, 10, 0.0, 0),
        "ddim_B2_32x32_T1000_S6_eta1_start300": (32, 32, 2, 1000, 6, 1.0, 300),
        "ddim_B1_64x64_T50_S7_eta05": (64, 64, 1, 50, 7, 0.5, 0)}


@pytest.mark.parametrize("name", list(DDIM))
def test_ddim_sample(oracle, synth, sd_torch, name):
    """ddim_sample restatement vs the reference's own output (oracle/make_golden_ddim.py); the manifest records
    max|oracle - reference| = 0.0 for these cases at generation time."""
    H, W, B, T, S, eta, start_t = DDIM[name]
    x, cond = _inputs(synth, B, H, W, 0)
    x_start = torch.from_numpy(synth.synth_slices(4, 0, B, H, W)) * 2 - 1 if start_t else None
    out = oracle.ddim_sample(x, cond, sd_torch, oracle.schedule_buffers(T),
                             lambda t: torch.from_numpy(synth.noise_z(3, t, 0, B, H, W)), S, eta, start_t, x_start).numpy()
    ref = golden(name)["out"]
    assert np.abs(out - ref).max() <= TOL
    assert ref.min() >= 0 and ref.max() <= 1 and ref.std() > 0.01
    pairs = oracle.ddim_time_pairs(T, S, start_t)
    assert len(pairs) == S and pairs[-1][1] == 0 and all(a > b for a, b in pairs)
    import json, os
    from conftest import GOLD
    man = json.load(open(os.path.join(GOLD, "MANIFEST.json")))["cases"][name]
    assert man["oracle_vs_reference_maxabs"] == 0.0 and [tuple(p) for p in man["time_pairs"]] == pairs


def test_single_step_reconstruction(oracle, synth, sd_torch):
    g = golden("p_losses_B2_32x32_t499")
    B, H, W = 2, 32, 32
    x01 = torch.from_numpy(synth.synth_slices(2, 0, B, H, W))
    cond = torch.from_numpy(synth.synth_cond(1, 0, B))
    noise = torch.from_numpy(synth.noise_z(3, 0, 0, B, H, W))
    loss, reco = oracle.p_losses_recon(x01, torch.full((B,), 499), cond, noise, sd_torch, oracle.schedule_buffers(1000))
    assert np.abs(reco.numpy() - g["reco"]).max() <= TOL and abs(float(loss) - float(g["loss"])) <= TOL


# ---- pred_noise objective and the linear schedule (oracle/make_golden_objectives.py; reference cond_DDPM.py:411-414,
#      :379-383, :612-644, :271-275) --------------------------------------------------------------------------------
OBJ_LOOPS = [("pn_loop_B2_32x32_T1000_start8", 1000, 8, "pred_noise", "cosine"),
             ("pn_loop_B2_32x32_T50_start0", 50, 0, "pred_noise", "cosine"),
             ("lin_loop_B2_32x32_T1000_start8", 1000, 8, "pred_x0", "linear")]


@pytest.mark.parametrize("name,T,start_t,objective,kind", OBJ_LOOPS, ids=[l[0] for l in OBJ_LOOPS])
def test_reverse_loop_objectives(oracle, synth, sd_torch, name, T, start_t, objective, kind):
    B, H, W = 2, 32, 32
    x, cond = _inputs(synth, B, H, W)
    out = oracle.p_sample_loop(x, cond, sd_torch, oracle.schedule_buffers(T, kind),
                               lambda t: torch.from_numpy(synth.noise_z(3, t, 0, B, H, W)), start_t=start_t,
                               objective=objective).numpy()
    ref = golden(name)["out"]
    assert np.abs(out - ref).max() <= TOL
    assert ref.min() >= 0 and ref.max() <= 1 and ref.std() > 0.01
    # the objective / schedule really matter: the pred_x0-cosine golden of the same inputs is a different image
    if start_t == 8:
        assert np.abs(ref - golden("loop_B2_32x32_T1000_start8")["out"]).max() > 1e-3


def test_linear_schedule_buffers(oracle):
    g = golden("lin_schedule_T1000")
    buf = oracle.schedule_buffers(1000, "linear")
    for k in g.files:
        assert np.array_equal(buf[k].numpy(), g[k]), k
    assert abs(float(g["betas"][0]) - 1e-4) < 1e-9 and abs(float(g["betas"][-1]) - 0.02) < 1e-8


def test_single_step_reconstruction_pred_noise(oracle, synth, sd_torch):
    g = golden("pn_p_losses_B2_32x32_t499")
    B, H, W = 2, 32, 32
    x01 = torch.from_numpy(synth.synth_slices(2, 0, B, H, W))
    cond = torch.from_numpy(synth.synth_cond(1, 0, B))
    noise = torch.from_numpy(synth.noise_z(3, 0, 0, B, H, W))
    loss, reco = oracle.p_losses_recon(x01, torch.full((B,), 499), cond, noise, sd_torch, oracle.schedule_buffers(1000),
                                       objective="pred_noise", loss_type="l2")
    assert np.abs(reco.numpy() - g["reco"]).max() <= TOL and abs(float(loss) - float(g["loss"])) <= TOL


def test_ddim_sample_pred_noise(oracle, synth, sd_torch):
    B, H, W, T, S = 2, 32, 32, 1000, 10
    x, cond = _inputs(synth, B, H, W)
    out = oracle.ddim_sample(x, cond, sd_torch, oracle.schedule_buffers(T),
                             lambda t: torch.from_numpy(synth.noise_z(3, t, 0, B, H, W)), S, 1.0, 0, None,
                             objective="pred_noise").numpy()
    ref = golden("pn_ddim_B2_32x32_T1000_S10_eta1")["out"]
    assert np.abs(out - ref).max() <= TOL and ref.std() > 0.01


def test_manifest_records_zero_oracle_error_for_every_loop():
    """every loop / ddim / single-step fixture was written together with max|oracle - reference| = 0.0 (bit-exact
    restatement at generation time), including the full-length headline chain that is too slow to re-run here"""
    import json
    from conftest import GOLD
    cases = json.load(open(os.path.join(GOLD, "MANIFEST.json")))["cases"]
    checked = 0
    for name, c in cases.items():
        if "oracle_vs_reference_maxabs" in c and isinstance(c["oracle_vs_reference_maxabs"], float):
            assert c["oracle_vs_reference_maxabs"] <= 1e-6, (name, c["oracle_vs_reference_maxabs"])
            checked += 1
    assert checked >= 15


@pytest.mark.skipif(not os.environ.get("CDDPM_SLOW"), reason="~30 min on 8 cores: the full T = 1000 chain at 128x128")
def test_reverse_loop_cfg2_full_length(oracle, synth, sd_torch):
    name, B, H, W, T = "loop_cfg2_B2_128x128_T1000_start0", 2, 128, 128, 1000
    x, cond = _inputs(synth, B, H, W)
    out = oracle.p_sample_loop(x, cond, sd_torch, oracle.schedule_buffers(T),
                               lambda t: torch.from_numpy(synth.noise_z(3, t, 0, B, H, W)), start_t=0).numpy()
    assert np.abs(out - golden(name)["out"]).max() <= TOL


def test_clip_denoised_false(oracle, synth, sd_torch):
    """clip_denoised=False (cond_DDPM.py:433, :467): p_sample step and a DDIM chain vs the reference's outputs"""
    B, H, W = 2, 32, 32
    x, cond = _inputs(synth, B, H, W)
    buf = oracle.schedule_buffers(1000)
    z = torch.from_numpy(synth.noise_z(3, 5, 0, B, H, W))
    with torch.no_grad():
        out = oracle.p_sample(x, 5, cond, sd_torch, buf, z, clip_denoised=False).numpy()
        clipped = oracle.p_sample(x, 5, cond, sd_torch, buf, z).numpy()
    ref = golden("noclip_p_sample_B2_32x32_t5")["out"]
    assert np.abs(out - ref).max() <= TOL and np.abs(clipped - ref).max() > 1e-3      # the flag matters on these inputs
    out = oracle.ddim_sample(x, cond, sd_torch, buf, lambda t: torch.from_numpy(synth.noise_z(3, t, 0, B, H, W)), 10, 1.0, 0, None,
                             clip_denoised=False).numpy()
    assert np.abs(out - golden("noclip_ddim_B2_32x32_T1000_S10_eta1")["out"]).max() <= TOL


def test_box_branch_of_p_sample_loop(oracle, synth, sd_torch):
    """the reference's `box` lines (cond_DDPM.py:455-459) as written -- sample 0 keeps x_T inside its box, every other sample starts from
    zeros (`img = img_patch` inside the loop) -- restated by cond_DDPM.mask_x_T_to_box: the oracle on that masked x_T reproduces the
    reference's own output (oracle/make_golden_box.py)"""
    from conftest import load_pkg
    D = load_pkg("cond_DDPM")
    g = golden("box_loop_B3_32x32_T1000_start6")
    B, H, W, T, start_t = 3, 32, 32, 1000, 6
    cond = torch.from_numpy(synth.synth_cond(1, 0, B))
    xT = D.mask_x_T_to_box(torch.from_numpy(synth.noise_xT(2, 0, B, H, W)), g["box"])
    assert float(xT[1:].abs().max()) == 0.0 and float(xT[0].abs().max()) > 0
    out = oracle.p_sample_loop(xT, cond, sd_torch, oracle.schedule_buffers(T),
                               lambda t: torch.from_numpy(synth.noise_z(3, t, 0, B, H, W)), start_t=start_t)
    assert float(np.abs(out.numpy() - g["out"]).max()) < 2e-6
