"""The lane-parallel checksum kernel (WN_CRC_COMBINE=1: per-distance tables in LDS, eight lanes
per 128-byte row, partial registers combined by DPP) against zlib, the oracle, the golden frames
and the one-lane-per-row kernel.  Bit-exact."""
import zlib

import numpy as np
import pytest
import torch

from oracle import modem_oracle as mo
from wavenet_amd import _lib as L
from wavenet_amd import modem as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    """(context with the combine kernel, context with the default kernel): the switch is read when
    a context is made."""
    import os
    old = os.environ.get("WN_CRC_COMBINE")
    os.environ["WN_CRC_COMBINE"] = "1"
    try:
        mc = M.Modem(M.esquema_b())
    finally:
        if old is None:
            del os.environ["WN_CRC_COMBINE"]
        else:
            os.environ["WN_CRC_COMBINE"] = old
    return mc, M.Modem(M.esquema_b())


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_combine_crc32_against_zlib(pair):
    mc, _ = pair
    rng = np.random.default_rng(17)
    n = 5003                                   # not a multiple of the 32 rows a wave takes at a time
    buf = rng.integers(0, 256, (n, 128), dtype=np.uint8)
    lens = rng.integers(0, 129, n).astype(np.int32)
    lens[:12] = [0, 1, 128, 127, 15, 16, 17, 31, 32, 33, 112, 113]
    buf[12, :9] = np.frombuffer(b"123456789", dtype=np.uint8)
    lens[12] = 9
    got = mc.crc32(dev(buf), dev(lens)).cpu().numpy()
    want = np.array([zlib.crc32(buf[i, :lens[i]].tobytes()) for i in range(n)], dtype=np.int64)
    assert np.array_equal(got, want)
    assert got[12] == 0xCBF43926 and got[0] == 0
    for small in (1, 7, 8, 9, 33):             # fewer rows than one wave's tile
        g = mc.crc32(dev(buf[:small]), dev(lens[:small])).cpu().numpy()
        assert np.array_equal(g, want[:small])


def test_combine_parse_golden_frames_and_cases(pair, golden):
    mc, _ = pair
    frames = [bytes.fromhex(f["frame"]) for f in golden["frames"]]
    cases = [c for c in golden["parse_cases"] if len(bytes.fromhex(c["raw"])) <= 128]
    raws = frames + [bytes.fromhex(c["raw"]) for c in cases]
    buf = np.zeros((len(raws), 128), dtype=np.uint8)
    lens = np.zeros(len(raws), dtype=np.int32)
    for i, f in enumerate(raws):
        buf[i, :len(f)] = np.frombuffer(f, dtype=np.uint8)
        lens[i] = len(f)
    rec = M.Modem.records_to_numpy(mc.parse_frames(dev(buf), dev(lens)))
    for f, r in zip(frames, rec):
        assert r["status"] == L.ST_OK and r["crc_ok"] == 1 and int(r["crc"]) == zlib.crc32(f[:-4])
    assert len(cases) >= 6
    for c, r in zip(cases, rec[len(frames):]):
        raw = bytes.fromhex(c["raw"])
        assert int(r["status"]) == mo.parse_status(raw), c["name"]
        if c["error"] is None:
            assert (int(r["tipo"]), int(r["length"]), bool(r["crc_ok"])) == (c["tipo"], c["length"], c["crc_ok"])
            assert int(r["crc"]) == zlib.crc32(raw[:15 + c["length"]])


def test_combine_parse_random_mixed_equals_the_oracle_and_the_row_kernel(pair):
    mc, md = pair
    rng = np.random.default_rng(23)
    n = 30000
    frames = []
    for i in range(n):
        ln = int(rng.integers(0, 108))
        f = bytearray(mo.build_frame(1, rng.bytes(6), rng.bytes(6), int(rng.integers(0, 6)), rng.bytes(ln)))
        kind = rng.random()
        if kind < 0.02:
            f[int(rng.integers(0, len(f)))] ^= 1 << int(rng.integers(0, 8))     # a flipped bit anywhere
        elif kind < 0.04:
            f = f[:int(rng.integers(0, len(f)))]                                   # cut short
        elif kind < 0.06:
            f = f + rng.bytes(int(rng.integers(1, 129 - len(f)))) if len(f) < 128 else f   # trailing bytes are ignored
        frames.append(bytes(f))
    buf = rng.integers(0, 256, (n, 128), dtype=np.uint8)      # whatever follows a frame in its row must not matter
    lens = np.zeros(n, dtype=np.int32)
    for i, f in enumerate(frames):
        buf[i, :len(f)] = np.frombuffer(f, dtype=np.uint8)
        lens[i] = len(f)
    a = mc.parse_frames(dev(buf), dev(lens))
    b = md.parse_frames(dev(buf), dev(lens))
    assert torch.equal(a, b)
    rec = M.Modem.records_to_numpy(a)
    for f, r in zip(frames, rec):
        st = mo.parse_status(f)
        assert int(r["status"]) == st
        if st == mo.ST_OK:
            p = mo.parse_frame(f)
            assert bool(r["crc_ok"]) == p.crc_ok and int(r["crc"]) == zlib.crc32(f[:15 + p.length])


def test_combine_two_million_rows_equal_the_row_kernel(pair):
    """Full-width check without a CPU loop: both kernels on the same 2 M random rows and lengths;
    and the affine property crc(a ^ b ^ c) = crc(a) ^ crc(b) ^ crc(c) for rows of equal length."""
    mc, md = pair
    g = torch.Generator(device="cuda").manual_seed(3)
    n = 2_000_003
    rows = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda", generator=g)
    lens = torch.randint(0, 129, (n,), dtype=torch.int32, device="cuda", generator=g)
    assert torch.equal(mc.crc32(rows, lens), md.crc32(rows, lens))
    rows[:, 14] = torch.randint(0, 108, (n,), dtype=torch.uint8, device="cuda", generator=g)
    assert torch.equal(mc.parse_frames(rows, lens), md.parse_frames(rows, lens))
    m = 500_000
    a, b, c = rows[:m], rows[m:2 * m], rows[2 * m:3 * m]
    ln = lens[:m].contiguous()
    x = mc.crc32((a ^ b ^ c).contiguous(), ln)
    assert torch.equal(x, mc.crc32(a.contiguous(), ln) ^ mc.crc32(b.contiguous(), ln) ^ mc.crc32(c.contiguous(), ln))
