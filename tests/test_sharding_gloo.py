"""Multi-rank path on CPU: two gloo ranks shard a stream range, pack their decoded-frame
records and gather them on rank 0 in stream order (what bench.py does over RCCL)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wavenet_amd import sharding as SH


def test_shard_range_covers_everything():
    for n, world in ((100000, 8), (10, 3), (7, 8), (0, 2)):
        spans = [SH.shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert SH.shard_range(100000, 3, 8) == (37500, 50000)
    with pytest.raises(ValueError):
        SH.shard_range(10, 2, 2)


def test_pack_unpack_roundtrip():
    g = torch.Generator().manual_seed(1)
    fb = torch.randint(0, 256, (33, 128), dtype=torch.uint8, generator=g)
    rec = torch.randint(0, 256, (33, 8), dtype=torch.uint8, generator=g)
    lens = torch.randint(0, 127, (33,), dtype=torch.int32, generator=g)
    a, b, c = SH.unpack_records(SH.pack_records(fb, lens, rec))
    assert torch.equal(a, fb) and torch.equal(b, lens) and torch.equal(c, rec)


def _worker(rank, world, n_total, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = SH.shard_range(n_total, rank, world)
    idx = torch.arange(lo, hi, dtype=torch.int64)
    fb = ((idx[:, None] * 131 + torch.arange(128)[None, :]) % 256).to(torch.uint8)
    rec = ((idx[:, None] * 7 + torch.arange(8)[None, :]) % 256).to(torch.uint8)
    lens = (idx % 127).to(torch.int32)
    out = SH.gather_records(SH.pack_records(fb, lens, rec), n_total, dst=0)
    if rank == 0:
        a, b, c = SH.unpack_records(out)
        all_idx = torch.arange(n_total, dtype=torch.int64)
        ok = (out.shape == (n_total, SH.RECORD_BYTES)
              and torch.equal(a, ((all_idx[:, None] * 131 + torch.arange(128)[None, :]) % 256).to(torch.uint8))
              and torch.equal(b, (all_idx % 127).to(torch.int32))
              and torch.equal(c, ((all_idx[:, None] * 7 + torch.arange(8)[None, :]) % 256).to(torch.uint8)))
        q.put(bool(ok))
    else:
        q.put(out is None)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [1001, 64])
def test_gather_records_two_ranks(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_total % 7
    procs = [ctx.Process(target=_worker, args=(r, 2, n_total, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [q.get(timeout=10) for _ in procs] == [True, True]


def _worker_windows(rank, world, n_windows, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    whole = ((torch.arange(n_windows, dtype=torch.int64) * 2654435761) % 1021).to(torch.int16)
    lo, hi = SH.shard_range(n_windows, rank, world)
    got = SH.gather_window_classes(whole[lo:hi].clone(), n_windows)
    q.put(bool(torch.equal(got, whole)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_windows", [22503, 7])
def test_gather_window_classes_two_ranks(n_windows):
    """One capture cut into two window ranges: both ranks end with the whole class sequence."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + n_windows % 7
    procs = [ctx.Process(target=_worker_windows, args=(r, 2, n_windows, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [q.get(timeout=10) for _ in procs] == [True, True]
