"""The benchmark configurations at their full sizes, checked through properties that do not
need the CPU oracle at that size: encode -> decode round trips, exactly the corrupted frames
failing, the affine property of the checksum, and frames recovered from an hour of noisy audio."""
import numpy as np
import pytest
import torch

from wavenet_amd import modem as M

pytestmark = pytest.mark.gpu
WIN = 7055


@pytest.fixture(scope="module")
def mb():
    return M.Modem(M.esquema_b())


def frames_for(m, n, seed):
    g = torch.Generator().manual_seed(seed)
    fields = torch.zeros((n, 16), dtype=torch.uint8)
    fields[:, 0] = 1
    fields[:, 1:13] = torch.randint(0, 256, (n, 12), dtype=torch.uint8, generator=g)
    fields[:, 13] = 3
    fields[:, 14] = 107
    pay = torch.randint(0, 256, (n, 112), dtype=torch.uint8, generator=g)
    return m.build_frames(fields.cuda(), pay.cuda(), 128)


def test_ten_thousand_streams_round_trip(mb):
    """Configuration 2: 10 000 streams, one 107-byte-payload frame each (122 GB of float32 audio)."""
    n = 10000
    free, _ = torch.cuda.mem_get_info()
    stride = M.pad4(mb.esquema.max_frame_samples(126))
    if free < n * stride * 4 * 1.05:
        pytest.skip("needs 128 GB of free HBM")
    frames, lens = frames_for(mb, n, 2)
    audio, n_samples = mb.encode_streams(frames, lens, stride)
    assert int(n_samples.max()) <= stride and int(n_samples.min()) > stride - 127 * 2646
    out = mb.decode_streams(audio, n, stride, stride // WIN, 123.0)
    del audio
    assert bool((out["n_calls"] == 1).all())
    assert torch.equal(out["lens"].view(-1), lens) and bool((lens == 126).all())
    assert torch.equal(out["bytes"].view(n, 128)[:, :126], frames[:, :126])
    rec = out["rec"].cpu().numpy().reshape(n, 8).view(M.REC_DTYPE).reshape(-1)
    assert (rec["status"] == 0).all() and (rec["crc_ok"] == 1).all() and (rec["length"] == 107).all()


def test_ten_million_frames_checksum_and_parse(mb):
    """Configuration 5: 10 M frames at stride 128, payload length uniform 0..107, 1 % corrupted:
    exactly the corrupted ones fail; and the checksum is affine over equal-length messages,
    crc(a) ^ crc(b) ^ crc(c) == crc(a ^ b ^ c)."""
    n = 10_000_000
    g = torch.Generator(device="cuda").manual_seed(5)
    rows = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda", generator=g)
    plen = torch.randint(0, 108, (n,), dtype=torch.int32, device="cuda", generator=g)
    rows[:, 14] = plen.to(torch.uint8)
    nbytes = plen + 19
    crc = mb.crc32(rows, plen + 15)
    ar = torch.arange(n, device="cuda")
    idx = (15 + plen).to(torch.int64)
    for k in range(4):                                      # stored big-endian after the payload
        rows[ar, idx + k] = ((crc >> (8 * (3 - k))) & 0xFF).to(torch.uint8)
    bad = torch.rand(n, device="cuda", generator=g) < 0.01
    pos = (torch.rand(n, device="cuda", generator=g) * nbytes).to(torch.int64)
    rows[ar[bad], pos[bad]] ^= 0x10
    rec = mb.parse_frames(rows, nbytes).cpu().numpy().view(M.REC_DTYPE).reshape(-1)
    failed = (rec["status"] != 0) | (rec["crc_ok"] == 0)    # a flipped length byte can also end in a parse error
    assert np.array_equal(failed, bad.cpu().numpy())
    third = n // 3
    a, b, c = rows[:third], rows[third:2 * third], rows[2 * third:3 * third]
    ln = torch.randint(0, 129, (third,), dtype=torch.int32, device="cuda", generator=g)
    lhs = mb.crc32(a, ln) ^ mb.crc32(b, ln) ^ mb.crc32(c, ln)
    assert torch.equal(lhs, mb.crc32(a ^ b ^ c, ln))


def test_one_hour_noisy_capture(mb):
    """Configuration 3: 158.76 M samples, 51 max-payload frames after random gaps, noise 0.1:
    every frame comes back exact, in order, and nothing else is heard."""
    total, nf = 158_760_000, 51
    frames, lens = frames_for(mb, nf, 9)
    lay = mb.layout(frames, lens)
    g = torch.Generator().manual_seed(3)
    gaps = torch.randint(0, 30001, (nf,), generator=g, dtype=torch.int64)
    ns = lay[3].cpu()
    dst = torch.cumsum(gaps + torch.cat([torch.zeros(1, dtype=torch.int64), ns[:-1]]), 0)
    assert int(dst[-1] + ns[-1]) <= total
    audio = torch.zeros(M.pad4(total), dtype=torch.float32, device="cuda")
    stride = M.pad4(mb.esquema.max_frame_samples(126))
    mb.synth(lay, dst.cuda(), audio, stride)
    mb.awgn(audio, 0.1, 77)
    out = mb.decode_streams(audio, 1, audio.numel(), total // WIN, 123.0, max_calls=nf + 4)
    k = int(out["n_calls"][0])
    got = [bytes(out["bytes"][0, j, :int(out["lens"][0, j])].cpu().numpy()) for j in range(k)]
    whole = [x for x in got if len(x) == 126]
    assert whole == [bytes(frames[j, :126].cpu().numpy()) for j in range(nf)]
    assert all(len(x) == 0 for x in got if len(x) != 126)      # the calls after the last frame time out empty
    rec = out["rec"].cpu().numpy().reshape(-1, 8).view(M.REC_DTYPE).reshape(-1)[:k]
    assert (rec["crc_ok"][[len(x) == 126 for x in got]] == 1).all()
    # positions: over the first 1500 windows (three frames) the class word of every window and the
    # (start window, end window, bytes, reason) of every listen call are the CPU oracle's on the same samples
    from oracle import modem_oracle as mo
    head = 1500
    sch = mo.scheme_b()
    bins, _ = mo.window_bins(audio[:head * WIN].cpu().numpy().astype(np.float64), WIN)
    cls_o = mo.bin_classes(WIN, sch)[bins]
    assert np.array_equal(out["cls"][0, :head].cpu().numpy().astype(np.uint16), cls_o)
    assert np.array_equal(out["bin"][0, :head].cpu().numpy(), bins)
    calls_o = [c for c in mo.slice_classes(cls_o, mo.timeout_to_windows(123.0, sch), nf + 4) if c[1] < head - 1]
    assert len(calls_o) >= 3
    for j, (w0, w1, b, r) in enumerate(calls_o):
        assert (int(out["start"][0, j]), int(out["end"][0, j]), got[j], int(out["reason"][0, j])) == (w0, w1, b, r)
    # and every later frame sits where its samples were put: a call ends within two windows of the
    # end of the frame's EOF tone (the EOF tone is 3 windows long, a call ends at the first EOF-class one)
    ends = out["end"][0, :k].cpu().numpy()
    starts = out["start"][0, :k].cpu().numpy()
    assert (starts[1:] == ends[:-1]).all() and starts[0] == 0
    frame_end = ((dst + ns).numpy() + WIN - 1) // WIN
    whole_idx = [j for j, x in enumerate(got) if len(x) == 126]
    assert all(0 <= frame_end[i] - ends[j] <= 3 for i, j in enumerate(whole_idx))
