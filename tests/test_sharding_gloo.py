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


class _HostModem:
    """Stand-in with the four methods decode_capture_sharded uses, computed by the CPU oracle on
    CPU tensors, so that the sharded control flow (window ranges, all-gather of class words,
    state machine over the whole sequence) runs on two gloo ranks without a GPU."""

    def __init__(self):
        from oracle import modem_oracle as mo
        self.mo, self.sch = mo, mo.scheme_b()
        self.lut = mo.bin_classes(self.sch.win_n, self.sch)

    def detect(self, audio, n_streams, stream_stride, n_windows, first_sample=0):
        import numpy as np
        assert n_streams == 1
        x = audio.numpy()[first_sample:first_sample + n_windows * self.sch.win_n].astype(np.float64)
        bins, mags = self.mo.window_bins(x, self.sch.win_n)
        cls = torch.from_numpy(self.lut[bins].astype(np.int16)).view(1, -1)
        return cls, torch.from_numpy(mags.astype(np.float32)).view(1, -1), torch.from_numpy(bins).view(1, -1)

    def timeout_windows(self, timeout):
        return self.mo.timeout_to_windows(timeout, self.sch)

    def slice(self, cls, timeout_windows, max_calls, frame_cap):
        import numpy as np
        calls = self.mo.slice_classes(cls.view(-1).numpy().astype(np.uint16), timeout_windows, max_calls)
        by = torch.zeros((1, max_calls, frame_cap), dtype=torch.uint8)
        lens, start, end, reason = (torch.zeros((1, max_calls), dtype=torch.int32) for _ in range(4))
        for j, (s, e, b, r) in enumerate(calls):
            by[0, j, :len(b)] = torch.tensor(list(b), dtype=torch.uint8)
            lens[0, j], start[0, j], end[0, j], reason[0, j] = len(b), s, e, r
        return by, lens, start, end, reason, torch.tensor([len(calls)], dtype=torch.int32)

    def parse_frames(self, rows, nbytes):
        out = torch.zeros((rows.shape[0], 8), dtype=torch.uint8)
        for i in range(rows.shape[0]):
            raw = bytes(rows[i, :int(nbytes[i])].tolist())
            st = self.mo.parse_status(raw)
            out[i, 4] = st
            if st == 0:
                out[i, 5] = int(self.mo.parse_frame(raw).crc_ok)
        return out


def _capture_two_frames():
    import numpy as np
    from oracle import modem_oracle as mo
    sch = mo.scheme_b()
    mac_o, mac_d = bytes.fromhex("aabbccddeeff"), bytes.fromhex("112233445566")
    frames = [mo.build_frame(1, mac_o, mac_d, 3, b"a"), mo.build_frame(1, mac_o, mac_d, 4, b"zz")]
    parts = [np.zeros(5000), mo.emit_frame(frames[0], sch), np.zeros(12345), mo.emit_frame(frames[1], sch),
             np.zeros(3 * sch.win_n)]
    return frames, np.concatenate(parts).astype(np.float32), sch


def _worker_capture(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames, audio, sch = _capture_two_frames()
    n_windows = len(audio) // sch.win_n
    lo, hi = SH.shard_range(n_windows, rank, world)
    piece = torch.from_numpy(audio[lo * sch.win_n:hi * sch.win_n].copy())
    m = _HostModem()
    out = SH.decode_capture_sharded(m, piece, n_windows, 123.0, max_calls=4)
    # the same capture decoded whole by the oracle's own loop
    whole_cls, _, _ = m.detect(torch.from_numpy(audio), 1, len(audio), n_windows)
    want = m.mo.slice_classes(whole_cls.view(-1).numpy().astype("uint16"), m.timeout_windows(123.0), 4)
    k = int(out["n_calls"][0])
    got = [(int(out["start"][0, j]), int(out["end"][0, j]), bytes(out["bytes"][0, j, :int(out["lens"][0, j])].tolist()),
            int(out["reason"][0, j])) for j in range(k)]
    ok = (got == want and [g[2] for g in got[:2]] == frames and torch.equal(out["cls"].view(-1), whole_cls.view(-1))
          and bool((out["rec"][:2, 5] == 1).all()))
    q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_decode_capture_sharded_two_ranks():
    """One capture of two frames cut into two window ranges: each rank detects its own range,
    the class words are all-gathered and both ranks slice the same two frames, with the start
    and end windows of the unsharded decode."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_capture, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert [q.get(timeout=10) for _ in procs] == [True, True]
