"""GPU: bench.py keeps its contract -- one JSON line on stdout with the driver's keys, the roofline and (at N = 1) the
CPU baseline objects; a short run at a reduced batch so that the test stays under a minute."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"}


def test_bench_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu", "--no-alt"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly one line on stdout; everything else goes to stderr
    out = json.loads(lines[0])
    assert REQUIRED <= set(out), REQUIRED - set(out)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["unit"] == "slices/s" and out["value"] > 0 and out["ms_per_step"] > 0 and out["vs_baseline"] is None
    # a bench step = 50 reverse steps: value = B * (K * 50 / 1000) / seconds, ms_per_step = seconds / K
    assert abs(out["value"] - 8 * 0.05 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    cfg = out["config"]
    assert "workload" in cfg and cfg["finite"] is True
    assert cfg["reverse_steps_per_bench_step"] == 50 and cfg["reverse_steps_timed"] == 100 and cfg["complete_reconstructions_timed"] == 0.1
    assert abs(cfg["timed_region_s"] - 2 * out["ms_per_step"] * 1e-3) < 1e-9
    roof = out["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert "cpu_baseline" not in out                    # --no-cpu
    assert "alt_paths" not in cfg and "small_batch" not in cfg      # --no-alt


def test_bench_complete_reconstruction_small():
    """K = 20 segments = ONE complete reconstruction (a single cddpm_reverse(t_start = 1000)); here at B = 2, 32x32 so
    that the test stays short; also exercises the small-batch line and one alternative-arithmetic child"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "1", "--batch", "2", "--size", "32",
                        "--no-cpu"], capture_output=True, text=True, timeout=900, env=dict(os.environ))
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    cfg = out["config"]
    assert out["steps"] == 20 and cfg["reverse_steps_timed"] == 1000 and cfg["complete_reconstructions_timed"] == 1.0
    assert cfg["finite"] is True and cfg["reconstruction_in_unit_range"] is True
    assert cfg["small_batch"]["batch"] == 4 and cfg["small_batch"]["slices_per_s"] > 0
    assert cfg["small_batch_two_streams"].get("slices_per_s", 0) > 0, cfg["small_batch_two_streams"]
    assert set(cfg["alt_paths"]) == {"f32", "x6", "h3_nb2"}
    for fam, v in cfg["alt_paths"].items():
        assert v.get("conv_family") == fam and v["slices_per_s"] > 0 and v["finite"] is True, (fam, v)


def test_bench_residual_workload_smoke():
    """BASELINE configs[3] as one command: `bench.py --workload residual` -- here 6 slices of 32x32, 20 reverse steps, in chunks of 4 and
    of 2: the gathered residual maps do not depend on how the block is chunked (checksums equal), one JSON line, strong scaling"""
    outs = []
    for chunk in ("4", "2"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "residual", "--slices", "6", "--size", "32", "--t-start", "20",
                            "--chunk", chunk], capture_output=True, text=True, timeout=600, env=dict(os.environ))
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1
        outs.append(json.loads(lines[0]))
    a, b = outs
    assert REQUIRED <= set(a) and a["scaling"] == "strong" and a["n_gpus"] == 1 and a["unit"] == "slices/s" and a["value"] > 0
    assert a["config"]["result_shape"] == [6, 1, 32, 32] and a["config"]["finite_and_in_unit_range"] is True
    assert a["config"]["chunk"] == 4 and b["config"]["chunk"] == 2


def test_residual_maps_do_not_depend_on_the_chunking(engine_factory, synth):
    """sharding.residual_maps_sharded on one rank: chunks of 4 == chunks of 3 == one call per slice pair, bit for bit (the handle's plan, not
    the call's batch, decides a slice's bits), and the residual is |x - reverse(x_T)| of the same engine calls made by hand"""
    sh = load_pkg_sharding()
    eng = engine_factory(timesteps=1000, max_batch=4, max_h=32, max_w=32)
    kw = dict(seed_inputs=4, seed_cond=1, seed_noise=3, t_start=12, gather="all")
    a = sh.residual_maps_sharded(eng, 7, 32, 32, chunk=4, **kw)
    b = sh.residual_maps_sharded(eng, 7, 32, 32, chunk=3, **kw)
    assert a.shape == (7, 1, 32, 32) and torch.equal(a, b)
    x = torch.from_numpy(synth.synth_slices(4, 5, 2, 32, 32)).cuda()
    cond = torch.from_numpy(synth.synth_cond(1, 5, 2)).cuda()
    xT = eng.noise_fill(2, 32, 32, seed=3, stream_id=synth.STREAM_XT, slice0=5)
    want = (x - eng.reverse(xT, cond, 12, seed=3, slice0=5)).abs()
    assert torch.equal(a[5:7], want)


def load_pkg_sharding():
    import importlib
    from conftest import PKG_NAME
    return importlib.import_module(PKG_NAME + ".sharding")
