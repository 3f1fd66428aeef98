"""ORACLE tooling -- build-container only: run the REFERENCE's ddim_sample (cond_DDPM.py:466-515, imported from
/root/reference by oracle/ref_harness.py) on seeded synthetic inputs and store its outputs under tests/golden/.

    python oracle/make_golden_ddim.py

As in make_golden.py, inputs are regenerated from seeds (synth.py); the fixtures hold reference OUTPUTS only, and the
manifest records max|oracle - reference| for every case (0.0 = the restatement reproduces the reference bit for bit).
Draw order of the reference (Gaussian branch): one randn(shape) that is overwritten, then the draw that is used --
randn(shape) (:484) for start_t == 0, q_sample's randn_like (:482, :549) otherwise -- then one randn_like per pair
with time_next > 0. The z of a pair is keyed by its `time` (synth.noise_z(3, time, ...))."""
from __future__ import annotations

import importlib
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
synth = importlib.import_module("conditioned-diffusion-models-uad_amd.synth")
import cddpm_oracle as O  # noqa: E402
import ref_harness as R  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEED_W, SEED_COND, SEED_XT, SEED_Z, SEED_X0 = 0, 1, 2, 3, 4

CASES = {
    "ddim_B2_32x32_T1000_S10_eta1": dict(H=32, W=32, B=2, timesteps=1000, S=10, eta=1.0, start_t=0),
    "ddim_B2_32x32_T1000_S10_eta0": dict(H=32, W=32, B=2, timesteps=1000, S=10, eta=0.0, start_t=0),
    "ddim_B2_32x32_T1000_S6_eta1_start300": dict(H=32, W=32, B=2, timesteps=1000, S=6, eta=1.0, start_t=300),
    "ddim_B1_64x64_T50_S7_eta05": dict(H=64, W=64, B=1, timesteps=50, S=7, eta=0.5, start_t=0),
}


def run_case(sd, H, W, B, timesteps, S, eta, start_t):
    _model, diff = R.build_reference(sd, image_size=(H, W), timesteps=timesteps)
    diff.sampling_timesteps = S
    diff.is_ddim_sampling = S < timesteps
    diff.ddim_sampling_eta = eta
    diff.cfg = types.SimpleNamespace(noisetype="gauss")          # read at cond_DDPM.py:502
    cond = torch.from_numpy(synth.synth_cond(SEED_COND, 0, B))
    xT = torch.from_numpy(synth.noise_xT(SEED_XT, 0, B, H, W))
    x_start = torch.from_numpy(synth.synth_slices(SEED_X0, 0, B, H, W)) * 2 - 1 if start_t else None
    pairs = O.ddim_time_pairs(timesteps, S, start_t)
    zs = {time: torch.from_numpy(synth.noise_z(SEED_Z, time, 0, B, H, W)) for time, nxt in pairs if nxt > 0}
    draws = [torch.zeros(B, 1, H, W), xT] + [zs[time] for time, nxt in pairs if nxt > 0]
    with R.injected_randn(draws):
        ref = diff.ddim_sample((B, 1, H, W), cond=cond, x_start=x_start, start_t=start_t)
    buf = O.schedule_buffers(timesteps)
    ora = O.ddim_sample(xT, cond, sd, buf, lambda t: zs[t], S, eta, start_t, x_start)
    return ref.numpy(), float((ref - ora).abs().max()), pairs


def main():
    torch.manual_seed(0)
    sd = O.to_torch_sd(synth.synth_state_dict(SEED_W))
    mpath = os.path.join(GOLD, "MANIFEST.json")
    manifest = json.load(open(mpath))
    for name, kw in CASES.items():
        ref, err, pairs = run_case(sd, **kw)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), out=ref)
        manifest["cases"][name] = dict(kw, oracle_vs_reference_maxabs=err, time_pairs=[list(p) for p in pairs],
                                       seeds=dict(weights=SEED_W, cond=SEED_COND, xT=SEED_XT, z=SEED_Z, x_start=SEED_X0))
        print(name, "oracle-vs-ref", err, "range", float(ref.min()), float(ref.max()), flush=True)
    json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
