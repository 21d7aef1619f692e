"""Sharding of independent streams over the GPUs of one node, and the optional gather of
decoded frames.

Streams never interact: rank r owns a contiguous block of stream indices and runs the whole
encode/decode path on them with no data-path collective.  The only exchange is the collection
of fixed-size decoded-frame records on one rank (RCCL over xGMI when the tensors live on
GPUs, gloo on CPU tensors in the tests): 128 frame bytes + 8 bytes of parse record +
4 bytes of length per stream, about 14 MB for 100 000 streams, latency-bound.

One long capture shards by window ranges instead: windows are independent for detection and
for the ping search, only the receive state machine is sequential, and it runs over class
words (2 bytes per window: 45 KB for an hour of audio), which every rank collects with one
all-gather.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

RECORD_BYTES = 128 + 8 + 4


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank `rank`; the first n_total % world ranks get one more."""
    if world <= 0 or not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_records(frame_bytes: torch.Tensor, lens: torch.Tensor, rec: torch.Tensor) -> torch.Tensor:
    """[n,128] u8 frame bytes, [n] i32 lengths, [n,8] u8 parse records -> [n,140] u8."""
    n = frame_bytes.shape[0]
    assert frame_bytes.shape == (n, 128) and rec.shape == (n, 8) and lens.shape == (n,)
    out = torch.empty((n, RECORD_BYTES), dtype=torch.uint8, device=frame_bytes.device)
    out[:, :128] = frame_bytes
    out[:, 128:136] = rec
    out[:, 136:140] = lens.to(torch.int32).contiguous().view(torch.uint8).view(n, 4)
    return out


def unpack_records(packed: torch.Tensor):
    n = packed.shape[0]
    lens = packed[:, 136:140].contiguous().view(torch.int32).view(n)
    return packed[:, :128], lens, packed[:, 128:136]


def _wire_device(t: torch.Tensor) -> torch.device:
    """Where a tensor has to live for the collective: RCCL moves device memory; gloo is used
    with host memory (device tensors are staged through it)."""
    if t.is_cuda and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return t.device


def all_reduce_max(value: float, device: torch.device, even_alone: bool = False) -> float:
    """Largest `value` over the ranks (the benchmark's max-over-ranks step time)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not even_alone):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    t = t.to(_wire_device(t))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(packed: torch.Tensor, n_total: int, dst: int = 0, even_alone: bool = False):
    """Gathers every rank's packed records on rank `dst` in stream order.  Returns the
    [n_total,140] tensor there (on the device of `packed`) and None elsewhere.  Shards may
    differ by one stream, so each rank pads to the largest shard and rank dst trims.  A single
    rank returns its records as they are, unless `even_alone` asks for the collective anyway
    (the same code path on a communicator of one)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not even_alone):
        return packed
    world, rank = dist.get_world_size(), dist.get_rank()
    biggest = -(-n_total // world)
    wire = _wire_device(packed)
    pad = torch.zeros((biggest, RECORD_BYTES), dtype=torch.uint8, device=wire)
    pad[:packed.shape[0]] = packed.to(wire)
    # an all-gather rather than a gather: the same bytes on the wire for rank dst, a ring
    # collective every backend implements natively, and about 1.4 MB per rank per 10 000 streams
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(bufs[r][:hi - lo])
    return torch.cat(parts, dim=0).to(packed.device)


def gather_window_classes(cls_local: torch.Tensor, n_windows: int) -> torch.Tensor:
    """One capture cut into contiguous window ranges (shard_range over its n_windows): every
    rank passes the class words of its own range, [hi-lo] int16, and gets the whole sequence
    [n_windows] back, ready for the state machine.  Ranges may differ by one window, so each
    rank pads to the largest."""
    assert cls_local.dtype == torch.int16 and cls_local.dim() == 1
    if not dist.is_initialized() or dist.get_world_size() == 1:
        assert cls_local.numel() == n_windows
        return cls_local
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(n_windows, rank, world)
    assert cls_local.numel() == hi - lo
    biggest = -(-n_windows // world)
    dev = _wire_device(cls_local)
    pad = torch.zeros(biggest, dtype=torch.int16, device=dev)
    pad[:hi - lo] = cls_local.to(dev)
    wire = pad.view(torch.uint8)          # bytes on the wire: every backend moves uint8
    bufs = [torch.empty_like(wire) for _ in range(world)]
    dist.all_gather(bufs, wire)
    parts = []
    for r in range(world):
        rlo, rhi = shard_range(n_windows, r, world)
        parts.append(bufs[r].view(torch.int16)[:rhi - rlo])
    return torch.cat(parts).to(cls_local.device)


def decode_capture_sharded(modem, audio_local: torch.Tensor, n_windows: int, timeout: float, max_calls: int,
                           frame_cap: int = 128):
    """The listen loop over one long capture held in pieces: rank r's `audio_local` starts at
    the first sample of its window range (shard_range(n_windows, r, world) windows of
    win_n samples, padded to a multiple of 4).  Detection runs on the local range, the class
    words are all-gathered, and every rank runs the state machine and the frame parse over the
    whole sequence (they are tiny), so all ranks return the same frames."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(n_windows, rank, world)
    cls_local, _, _ = modem.detect(audio_local, 1, audio_local.numel(), hi - lo)
    cls = gather_window_classes(cls_local.view(-1), n_windows).view(1, n_windows).contiguous()
    by, lens, start, end, reason, n_calls = modem.slice(cls, modem.timeout_windows(timeout), max_calls, frame_cap)
    rec = modem.parse_frames(by.view(max_calls, frame_cap), lens.view(-1))
    return dict(cls=cls, bytes=by, lens=lens, start=start, end=end, reason=reason, n_calls=n_calls, rec=rec)
