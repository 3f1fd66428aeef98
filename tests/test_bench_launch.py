"""bench.py --gpus N starts its own ranks: the launch plan is decided on the host before anything touches a GPU (CPU tests),
and the launcher form runs the RCCL path on the one GPU of the box (gpu test)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402  (importing bench.py imports neither torch nor the HIP library)


def test_importing_bench_does_not_import_torch():
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; print('torch' in sys.modules)" % ROOT],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "False", r.stdout + r.stderr


def test_launch_plan_single_gpu_or_already_a_rank():
    assert bench.launch_plan(1, ["--gpus", "1"], {}) is None
    assert bench.launch_plan(8, ["--gpus", "8"], {"RANK": "3", "WORLD_SIZE": "8"}) is None       # under torchrun: this IS a rank


def test_launch_plan_starts_n_ranks_on_loopback():
    argv = ["--gpus", "4", "--steps", "20", "--warmup", "2"]
    cmd = bench.launch_plan(4, argv, {})
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    port = int(cmd[cmd.index("--master-port") + 1])
    assert 1024 < port < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                                   # the ranks see the caller's flags unchanged
    # an explicit MASTER_PORT is honoured; another script (tools/train_step_bench.py) can use the same plan
    cmd = bench.launch_plan(2, [], {"MASTER_PORT": "29511"}, script="/x/y.py")
    assert cmd[cmd.index("--master-port") + 1] == "29511" and cmd[-1] == "/x/y.py"


def test_self_launch_relays_one_json_line_and_the_exit_code():
    code = ("import sys; sys.path.insert(0, %r); import bench, os\n"
            "bench.self_launch([sys.executable, '-c', 'print(\"noise\"); print(\"{\\\\\"n_gpus\\\\\": 2}\")'], os.environ)") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert [json.loads(l) for l in r.stdout.splitlines()] == [{"n_gpus": 2}]
    code = ("import sys; sys.path.insert(0, %r); import bench, os\n"
            "bench.self_launch([sys.executable, '-c', 'import sys; sys.exit(3)'], os.environ)") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and r.stdout.strip() == ""


def test_train_step_bench_shares_the_plan():
    src = open(os.path.join(ROOT, "tools", "train_step_bench.py")).read()
    assert "launch_plan(a.gpus" in src and src.index("launch_plan(a.gpus") < src.index("import torch\n")


@pytest.mark.gpu
def test_bench_under_the_launcher_runs_the_rccl_path():
    """the driver's N > 1 command line with N = 1 (one GPU on the box): init_process_group('nccl'), the all_gather inside the
    timed region, MAX over ranks, one JSON line"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--batch", "4",
           "--size", "32", "--no-cpu", "--no-alt"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["global_batch"] == 4 and out["config"]["gather_ms"] > 0
    assert "all_gather" in out["config"]["workload"]
