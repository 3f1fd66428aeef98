"""GPU: bench.py keeps its contract -- one JSON line on stdout with the driver's keys, the roofline and (at N = 1) the
CPU baseline objects; a short run at a reduced batch so that the test stays under a minute."""
import json
import os
import subprocess
import sys

import pytest

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
    assert set(cfg["alt_paths"]) == {"f32", "x6"}
    for fam, v in cfg["alt_paths"].items():
        assert v.get("conv_family") == fam and v["slices_per_s"] > 0 and v["finite"] is True, (fam, v)
