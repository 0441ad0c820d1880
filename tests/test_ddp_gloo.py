"""N>1 path on CPU: the bucketed gradient exchange with 2 gloo ranks (one process per rank)."""

import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maestro_amd.train.ddp import GradSync
    n = 1000
    g = torch.Generator().manual_seed(rank)
    grad = torch.randn(n, generator=g)
    mine = grad.clone()
    sync = GradSync(grad, bucket_bytes=4 * 200)   # 200-element buckets
    sync.begin()
    # readiness arrives tail-first but out of order and with a gap that only closes late (like mask-token slices)
    for lo, hi in [(900, 1000), (600, 700), (700, 900), (300, 500), (0, 100), (500, 600)]:
        sync.ready(lo, hi)
    launched_before_finish = list(sync.launched)
    scale = sync.finish()
    other = torch.randn(n, generator=torch.Generator().manual_seed(1 - rank))
    ok = torch.allclose(grad, mine + other) and scale == 0.5
    covered = sorted(sync.launched)
    contiguous = covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    out.put((rank, ok, contiguous, launched_before_finish))
    dist.destroy_process_group()


def test_gradsync_two_ranks():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, contiguous, early in res:
        assert ok, f"rank {rank}: all-reduced gradient is wrong"
        assert contiguous, f"rank {rank}: buckets do not tile the flat buffer exactly once"
        assert early == [(600, 1000), (300, 600)], early   # launched during "backward", tail first, gaps respected


def _split_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maestro_amd.train.ddp import GradSync
    n = 1000
    grad = torch.randn(n, generator=torch.Generator().manual_seed(rank))
    mine = grad.clone()
    sync = GradSync(grad, bucket_bytes=4 * 200)
    sync.begin()
    for lo, hi in [(700, 1000), (400, 700)]:
        sync.ready(lo, hi)
    scale, split, wait_tail = sync.finish_split()
    other = torch.randn(n, generator=torch.Generator().manual_seed(1 - rank))
    upper_ok = torch.allclose(grad[split:], (mine + other)[split:])     # reduced before the tail has been waited for
    wait_tail()
    out.put((rank, scale, split, upper_ok, torch.allclose(grad, mine + other), sorted(sync.launched)))
    dist.destroy_process_group()


def test_gradsync_split_finish_two_ranks():
    """The optimizer may start on [split, n) while the head bucket [0, split) is still in flight."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, scale, split, upper_ok, all_ok, launched in res:
        assert scale == 0.5 and split == 400 and upper_ok and all_ok, (rank, scale, split, upper_ok, all_ok)
        assert launched == [(0, 400), (400, 1000)] or launched == [(0, 400), (400, 700), (700, 1000)], launched


def test_gradsync_single_process_is_identity():
    from maestro_amd.train.ddp import GradSync
    grad = torch.arange(10.0)
    sync = GradSync(grad, bucket_bytes=16)
    sync.ready(4, 10)
    assert sync.finish() == 1.0 and torch.equal(grad, torch.arange(10.0))
