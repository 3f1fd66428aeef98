"""CPU, world_size 2 over gloo: the slice sharding + single final gather of sharding.py. The per-chunk
reconstruction is replaced by a deterministic function of the GLOBAL slice index (what the real path
guarantees through the counter RNG), so the test checks partitioning, chunking, padding and the collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG_NAME, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_reco(slice0, count, H=8, W=12):
    idx = torch.arange(slice0, slice0 + count, dtype=torch.float32).reshape(count, 1, 1, 1)
    return (idx * 0.5 + torch.arange(H * W, dtype=torch.float32).reshape(1, 1, H, W) * 1e-3).contiguous()


def _worker(rank, world, port, n_slices, chunk, mode, q):
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module(PKG_NAME + ".sharding")
    calls = []

    def run(s0, cnt):
        calls.append((s0, cnt))
        return _fake_reco(s0, cnt)

    out = sh.reconstruct_sharded(n_slices, (8, 12), run, chunk=chunk, gather=mode)
    # by value (numpy -> pickle), not as a shared-memory tensor: the worker may exit before the parent has mapped it
    q.put((rank, calls, None if out is None else out.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_slices,chunk,mode", [(8, 3, "all"), (7, 2, "all"), (5, 64, "root"), (1, 4, "all")])
def test_two_rank_shard_and_gather(n_slices, chunk, mode):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_slices, chunk, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, calls, out = q.get(timeout=300)
        res[r] = (calls, None if out is None else torch.from_numpy(out))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = _fake_reco(0, n_slices)
    covered = sorted(s for r in res for (s0, c) in res[r][0] for s in range(s0, s0 + c))
    assert covered == list(range(n_slices))                       # every slice exactly once
    assert all(c <= chunk for r in res for (_s, c) in res[r][0])
    assert torch.equal(res[0][1], expect)                         # identical to the 1-rank result, bitwise
    if mode == "all":
        assert torch.equal(res[1][1], expect)
    else:
        assert res[1][1] is None


def test_shard_range_partition():
    import importlib
    sh = importlib.import_module(PKG_NAME + ".sharding")
    for n in (0, 1, 7, 8, 8192):
        for w in (1, 2, 3, 8):
            rs = [sh.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1


def _grad_worker(rank, world, port, q):
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = importlib.import_module(PKG_NAME + ".training")
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)          # rank r holds (r + 1) * g
    n = tr.all_reduce_sum_(flat)
    q.put((rank, n, flat.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_all_reduce():
    """the training step's data-parallel exchange (training.all_reduce_sum_): one collective over the flat gradient buffer; the mean is
    the sum divided by the returned rank count (folded into Adam's unscale factor by training_step)"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = (torch.arange(1000, dtype=torch.float32) * 3).numpy()
    for _rank, n, flat in got:
        assert n == 2 and (flat == expect).all()


def _bucket_worker(rank, world, port, q):
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = importlib.import_module(PKG_NAME + ".training")
    n = 10_000
    flat = torch.zeros(n)
    b = tr.GradBuckets(flat, bucket_floats=3000)
    # a "backward pass" that finalises the buffer from its tail in uneven pieces; nothing in [lo, end) changes after mark_final(lo)
    for lo in (9500, 9000, 6100, 6000, 2500, 2400, 100):
        flat[lo:b.lo] = torch.arange(lo, b.lo, dtype=torch.float32) * (rank + 1)
        b.mark_final(lo)
    flat[0:100] = torch.arange(0, 100, dtype=torch.float32) * (rank + 1)       # the head is flushed by finish()
    world_n = b.finish()
    q.put((rank, world_n, flat.numpy().copy(), list(b.issued)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucketed_all_reduce_behind_the_backward_pass():
    """training.GradBuckets: slices of the flat gradient buffer are all-reduced (async) as soon as they are final -- the overlapped form
    of the exchange. Every float is reduced exactly once, buckets hold at least `bucket_floats` (the last one what is left), the result is
    the plain sum over ranks."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = (torch.arange(10_000, dtype=torch.float32) * 3).numpy()
    for _rank, n, flat, issued in got:
        assert n == 2 and (flat == expect).all()
        assert issued == [(6100, 10000), (2500, 6100), (0, 2500)]            # contiguous, tail first, each >= 3000 floats but the last
