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
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly one line on stdout; everything else goes to stderr
    out = json.loads(lines[0])
    assert REQUIRED <= set(out), REQUIRED - set(out)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["unit"] == "slices/s" and out["value"] > 0 and out["ms_per_step"] > 0 and out["vs_baseline"] is None
    assert abs(out["value"] - 8 / (1000 * out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]      # value = B / (T s_per_step)
    assert "workload" in out["config"] and out["config"]["finite"] is True
    roof = out["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert "cpu_baseline" not in out                    # --no-cpu
