#!/usr/bin/env python3
"""Headline benchmark: audio-seconds demodulated per second on the round-trip workload.

Workload (per GPU): 10 000 independent streams, each one type-3 frame with a 107-byte
payload (126 bytes on the wire, ~69.3 s of audio at 44.1 kHz, float32).  One step =
build the frames, synthesise every stream's audio into HBM, then detect every 7055-sample
window, run the receive state machine and parse/validate the frames, all on the GPU.
With N > 1 every rank does the same amount of work on its own block of streams (weak
scaling) and the decoded-frame records are gathered on rank 0 inside the timed region.

Launching.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment
starts N rank processes itself (fresh children, started before this process has touched
the GPU), one per device, and passes rank 0's line through.  Under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` every process is
already a rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).
`--backend gloo --share-devices` runs the same multi-rank branch with several ranks on one
device and the record gather over gloo (a rehearsal of the code path, not a measurement).

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# torch, the modem and the sharding helpers are imported by the rank code only: the launching
# parent must stay clear of the GPU runtime
np = torch = dist = M = SH = None


def _imports():
    global np, torch, dist, M, SH
    import numpy
    import torch as _torch
    import torch.distributed as _dist

    from wavenet_amd import modem, sharding
    np, torch, dist, M, SH = numpy, _torch, _dist, modem, sharding


PEAK_F32_MFMA_TFLOPS = 157.3                # MI355X dense f32 matrix peak
WIN = 7055


def detect_flop(n1: int, n2: int):
    """Algorithmic flop of the folded prime-factor DFT (multiply-adds of the matrix products
    only, without tile padding) per unit the kernel counts, as (stage-1 flop per unit, stage-1
    units per window, stage-2 flop per unit, stage-2 units per window).  Stage 1: for each of n1
    rows, real and imaginary part of h2 columns from h2 folded samples; its rows come in
    2*ceil(h1/16) tiles (sums and differences of mirrored rows); unit = (row tile, real or
    imaginary part, column).  Stage 2: four real sums over h1 folded rows for each of h1 output
    pairs of a column, in ceil(h1/16) row tiles; unit = (row tile, column).  The kernel computes
    only the units that can still change the decision."""
    h1, h2 = n1 // 2 + 1, n2 // 2 + 1
    u1 = 2 * 2 * ((h1 + 15) // 16) * h2
    u2 = ((h1 + 15) // 16) * h2
    return 2.0 * n1 * h2 * h2 * 2 / u1, u1, 2.0 * h1 * h1 * 4 * h2 / u2, u2


def detect_flop_per_window(n1: int, n2: int) -> float:
    """The same with every unit computed (the whole spectrum)."""
    f1, u1, f2, u2 = detect_flop(n1, n2)
    return f1 * u1 + f2 * u2


def make_inputs(n, first_index, device):
    """Frame fields and payloads of streams [first_index, first_index+n): MACs from the
    stream index, payload bytes from a seeded generator."""
    idx = torch.arange(first_index, first_index + n, dtype=torch.int64)
    fields = torch.zeros((n, 16), dtype=torch.uint8)
    fields[:, 0] = 1
    for b in range(6):
        fields[:, 1 + b] = ((idx >> (8 * b)) & 0xFF).to(torch.uint8)
        fields[:, 7 + b] = (((idx * 2654435761) >> (8 * b)) & 0xFF).to(torch.uint8)
    fields[:, 13] = 3
    fields[:, 14] = 107
    g = torch.Generator().manual_seed(1000003 + first_index)
    pay = torch.randint(0, 256, (n, 112), dtype=torch.uint8, generator=g)
    return fields.to(device), pay.to(device)


def cpu_baseline(n_streams, seed=7):
    """The oracle (the reference's algorithm, NumPy FFT per window, one thread) on the same
    kind of stream: synthesise one max-payload frame and listen to it until EOF."""
    from oracle import modem_oracle as mo
    sch = mo.scheme_b()
    rng = np.random.default_rng(seed)
    audio_s = 0.0
    t0 = time.perf_counter()
    for i in range(n_streams):
        frame = mo.build_frame(1, rng.bytes(6), rng.bytes(6), 3, rng.bytes(107))
        x = mo.emit_frame(frame, sch)
        got, reason, used = mo.listen_frame(mo.Capture(x), 123.0, sch)
        assert bytes(got) == frame and mo.parse_frame(bytes(got)).crc_ok
        audio_s += used * sch.win_n / sch.sample_rate
    dt = time.perf_counter() - t0
    return audio_s / dt, dt


def cpu_baseline_all_cores(n_streams):
    """The same sample on every host core this process may use (at most 16, the share of a
    one-GPU box): one forked single-threaded worker per core, all started together.  Must run
    before this process touches the GPU.  Returns (aggregate audio-s/s, cores, wall seconds)."""
    import multiprocessing as mp
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.starmap(cpu_baseline, [(n_streams, 100 + i) for i in range(cores)])
    wall = time.perf_counter() - t0
    audio_s = sum(rate * dt for rate, dt in res)
    return audio_s / max(dt for _, dt in res), cores, wall


def profile_traffic(kernel, units, units_key, suffix="_pmc.json"):
    """(HBM bytes per launch, source text) of `kernel` from the committed counter summaries under
    profiles/: the last one by name (tags are rNNx) that names the kernel and records the launch
    size it was taken at, scaled from that size to `units`; (None, reason) when there is none."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)), reverse=True)
    files = [f for f in files if suffix != "_pmc.json" or not f.endswith("_kernelbench_pmc.json")]
    for path in files:
        with open(path) as f:
            pmc = json.load(f)
        k = pmc.get("kernels", {}).get(kernel, {})
        base = pmc.get(units_key)
        if "hbm_bytes" in k and base:
            return k["hbm_bytes"] * units / base, (
                "from committed profile profiles/%s (rocprofv3 --pmc FETCH_SIZE, x2 for 16-byte streaming reads on "
                "gfx950, + WRITE_SIZE passes of this command at commit %s), scaled by launch size %s/%s; not measured "
                "by this run" % (os.path.basename(path), pmc.get("source_commit_or_parent", "?"), units, base))
    return None, "no committed counter summary under profiles/ has per-launch bytes and the launch size of " + kernel


def timed_launches(fn, reps=3):
    """Mean milliseconds of `fn` over `reps` launches, by events on the stream it launches on."""
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


def kernelbench_traffic(kernel):
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernelbench_pmc.json")), reverse=True):
        with open(path) as f:
            k = json.load(f).get("kernels", {}).get(kernel, {})
        if "hbm_bytes" in k:
            return k["hbm_bytes"], "profiles/" + os.path.basename(path)
    return None, None


def side_measurements(m, audio, n_streams, stride, nwin):
    """Outside the timed steps: the two HBM-bound kernels that are not on the round trip's critical
    path, each timed live over its full-size workload.  `sync`: the ping search over the audio of
    the last step.  `crc`: parse + validate of 10 M frames at stride 128, payload 0..107, 1 % corrupted.
    `traffic` is the counter figure of the same launch size from tools/kernel_bench.py's committed passes."""
    out = {}
    ms = timed_launches(lambda: m.sync_scan(audio, n_streams, stride, nwin))
    by = n_streams * nwin * (WIN * 4 + 9)
    tr, src = kernelbench_traffic("sync_kernel<7055>")
    out["sync"] = {"bound": "hbm", "kernel": "sync_kernel<7055>", "ms_per_launch": ms, "algorithmic_bytes": by,
                   "achieved_GBps": by / ms / 1e6, "peak_GBps": 8000.0, "frac": by / ms / 1e6 / 8000.0,
                   "traffic": tr if n_streams * nwin == 4330000 else None, "traffic_source": src}
    n = 10_000_000
    g = torch.Generator(device="cuda").manual_seed(5)
    rows = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda", generator=g)
    plen = torch.randint(0, 108, (n,), dtype=torch.int32, device="cuda", generator=g)
    rows[:, 14] = plen.to(torch.uint8)
    nbytes = plen + 19
    crc = m.crc32(rows, plen + 15)
    ar = torch.arange(n, device="cuda")
    idx = (15 + plen).to(torch.int64)
    for k in range(4):
        rows[ar, idx + k] = ((crc >> (8 * (3 - k))) & 0xFF).to(torch.uint8)
    bad = torch.rand(n, device="cuda", generator=g) < 0.01
    pos = (torch.rand(n, device="cuda", generator=g) * nbytes).to(torch.int64)
    rows[ar[bad], pos[bad]] ^= 0x10
    r = m.parse_frames(rows, nbytes).cpu().numpy().view(M.REC_DTYPE).reshape(-1)
    assert np.array_equal((r["status"] != 0) | (r["crc_ok"] == 0), bad.cpu().numpy()), "exactly the corrupted frames fail"
    ms = timed_launches(lambda: m.parse_frames(rows, nbytes))
    by = int(nbytes.sum().item()) + 12 * n
    tr, src = kernelbench_traffic("crc_rows_kernel<1, 8>")
    out["crc"] = {"bound": "hbm", "kernel": "crc_rows_kernel<parse+validate>", "ms_per_launch": ms, "frames": n,
                  "algorithmic_bytes": by, "achieved_GBps": by / ms / 1e6, "peak_GBps": 8000.0,
                  "frac": by / ms / 1e6 / 8000.0, "frames_per_s": n / ms * 1e3, "traffic": tr, "traffic_source": src}
    del rows
    out["c3"] = long_capture(m)
    out["scheme_a"] = prototype_scheme()
    # the round trip's own audio through a noisy channel (last: it changes the buffer)
    out["noisy_c2"] = noisy_detection(m, audio, n_streams, stride, nwin)
    return out


def long_capture(m):
    """C3: one capture of one hour (158 760 000 samples): 51 back-to-back max-payload frames, each after
    a random gap of 0..30 000 samples, white noise of sigma 0.1 over everything; detection of every
    window, then the receive state machine call after call.  Every frame must come back exact."""
    total, nf = 158_760_000, 51
    fields, pay = make_inputs(nf, 900_000, "cuda")
    frames, lens = m.build_frames(fields, pay, 128)
    lay = m.layout(frames, lens)
    g = torch.Generator().manual_seed(3)
    gaps = torch.randint(0, 30001, (nf,), generator=g, dtype=torch.int64)
    ns = lay[3].cpu()
    dst = torch.cumsum(gaps + torch.cat([torch.zeros(1, dtype=torch.int64), ns[:-1]]), 0)
    assert int(dst[-1] + ns[-1]) <= total
    audio = torch.zeros(M.pad4(total), dtype=torch.float32, device="cuda")
    m.synth(lay, dst.cuda(), audio, M.pad4(m.esquema.max_frame_samples(126)))
    m.awgn(audio, 0.1, 77)
    nw = total // WIN

    def run():
        cls, _, _ = m.detect(audio, 1, audio.numel(), nw)
        return m.slice(cls, m.timeout_windows(123.0), nf + 4, 128)
    ms = timed_launches(run, reps=5)
    by, ln, start, end, reason, ncalls = run()
    got = [bytes(by[0, j, :int(ln[0, j])].cpu().numpy()) for j in range(int(ncalls[0]))]
    want = [bytes(frames[j, :126].cpu().numpy()) for j in range(nf)]
    assert [x for x in got if len(x) == 126] == want, "every frame of the long capture comes back exact, in order"
    return {"workload": "C3: one capture of %d samples (1 h), %d frames after random gaps, AWGN sigma 0.1; "
                        "detect + state machine" % (total, nf), "ms": ms, "audio_s_per_s": total / 44100 / ms * 1e3,
            "windows": nw, "listen_calls": int(ncalls[0]), "frames_exact": nf,
            "algorithmic_bytes": nw * (WIN * 4 + 10), "hbm_frac": nw * (WIN * 4 + 10) / ms / 1e6 / 8000.0}


def noisy_detection(m, audio, n_streams, stride, nwin):
    """The audio of the last step after white noise of sigma 0.1 (SNR 16.8 dB against the 0.975-peak
    tones): what detection costs per window when the channel is not clean, and which way the windows go."""
    m.awgn(audio, 0.1, 99)
    m.detect_stats()
    ms = timed_launches(lambda: m.detect(audio, n_streams, stride, nwin))
    w, top2, full, cells, units, redone, whole, two, chosen, sep_alone, tone_alone = m.detect_stats(paths=True)
    cls, _, _ = m.detect(audio, n_streams, stride, nwin)
    by, ln, _, _, reason, _ = m.slice(cls, m.timeout_windows(123.0), 1, 128)
    rec = m.parse_frames(by.view(n_streams, 128), ln.view(-1)).cpu().numpy().view(M.REC_DTYPE).reshape(-1)
    ok = int(((rec["status"] == 0) & (rec["crc_ok"] == 1)).sum())
    return {"workload": "the C2 audio of the last step + AWGN sigma 0.1", "detect_ms": ms,
            "audio_s_per_s_detect_only": n_streams * nwin * WIN / 44100.0 / ms * 1e3,
            "hbm_frac": n_streams * nwin * (WIN * 4 + 10) / ms / 1e6 / 8000.0,
            "stage1_tile_columns_computed_per_window": units / max(w, 1),
            "stage2_tile_columns_computed_per_window": cells / max(w, 1),
            "passes_the_long_way_per_window": whole / max(w, 1), "of_them_without_the_weakest_column_tile": two / max(w, 1),
            "windows_done_twice": redone / max(w, 1),
            "float64_two_or_three_bins_per_window": top2 / max(w, 1), "float64_whole_spectrum_per_window": full / max(w, 1),
            "frames_valid": ok, "frames": n_streams}


def prototype_scheme():
    """The prototype scheme (capa1: 2205-sample tones and windows, no separators): 20 000 messages of
    127 bytes, synthesis and detection; every byte must come back."""
    ma = M.Modem(M.esquema_a())
    n, msg = 20000, 127
    g = torch.Generator().manual_seed(4)
    rows = torch.randint(0, 128, (n, 128), dtype=torch.uint8, generator=g).cuda()
    lens_a = torch.full((n,), msg, dtype=torch.int32, device="cuda")
    win_a = ma.esquema.win_n
    stride_a = M.pad4((msg + 1) * win_a)
    audio_a = torch.empty((n, stride_a), dtype=torch.float32, device="cuda")
    ms_s = timed_launches(lambda: ma.encode_streams(rows, lens_a, stride_a, audio=audio_a))
    nw = stride_a // win_a
    ms_d = timed_launches(lambda: ma.detect(audio_a, n, stride_a, nw))
    cls, _, _ = ma.detect(audio_a, n, stride_a, nw)
    assert bool(((cls[:, :msg] & 0x1FF).to(torch.uint8) == rows[:, :msg]).all()), "every byte of the prototype scheme"
    return {"workload": "capa1 scheme: %d messages of %d bytes, windows of %d" % (n, msg, win_a),
            "synth_ms": ms_s, "synth_GBps": n * stride_a * 4 / ms_s / 1e6, "synth_frac": n * stride_a * 4 / ms_s / 1e6 / 8000.0,
            "detect_ms": ms_d, "detect_GBps": n * nw * (win_a * 4 + 10) / ms_d / 1e6,
            "detect_hbm_frac": n * nw * (win_a * 4 + 10) / ms_d / 1e6 / 8000.0, "windows": n * nw, "bytes_exact": True}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks = devices of this node to use")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=10000, help="streams per GPU")
    ap.add_argument("--cpu-streams", type=int, default=200, help="streams of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--sub-batches", type=int, default=0, help="split the streams of a step (0 = only if memory requires)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="collective backend of the record gather (nccl = RCCL; gloo stages through host memory)")
    ap.add_argument("--share-devices", action="store_true",
                    help="ranks may share a device (rank r uses device r mod device count): rehearsal of the "
                         "multi-rank branch on a box with fewer devices than ranks; needs --backend gloo")
    ap.add_argument("--dist-alone", action="store_true",
                    help="with one rank, still form the process group and run the record gather through it "
                         "(exercises RCCL on a one-GPU box)")
    ap.add_argument("--no-pick", action="store_true",
                    help="take the audio buffer as it comes instead of trying two placements in HBM")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the side measurements after the timed steps (sync search, checksum batch)")
    ap.add_argument("--rank-info", action="store_true",
                    help="each rank prints its rank environment as one JSON line and exits (no GPU work)")
    return ap.parse_args(argv)


def launch_ranks(args):
    """Parent of a self-launched multi-rank run: starts args.gpus children of this same script,
    one rank each, and returns the first non-zero exit code (0 if all succeed).  The parent
    makes no GPU call before or after (it has not even imported torch), so every rank is a
    fresh process that initialises the GPU itself; nothing is re-executed."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = set(range(args.gpus))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print("bench.py: rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr)
                for o in pending:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rank_info:
        print(json.dumps({"rank": rank, "local_rank": local, "world": world, "pid": os.getpid(),
                          "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))}), flush=True)
        return
    _imports()
    launcher = "single process" if world == 1 else (
        "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "bench.py --gpus")
    all_cores = None
    if args.cpu_streams > 0 and world == 1:
        all_cores = cpu_baseline_all_cores(max(1, args.cpu_streams // 2))
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py: no GPU visible; the modem runs as HIP kernels only")
    if local >= n_dev:
        if not args.share_devices:
            raise SystemExit("bench.py: rank %d needs device %d but this node shows %d device(s) "
                             "(--share-devices --backend gloo rehearses the multi-rank path on fewer)" % (rank, local, n_dev))
        local %= n_dev
    if args.share_devices and world > n_dev and args.backend != "gloo":
        raise SystemExit("bench.py: ranks sharing a device cannot form an RCCL communicator; use --backend gloo")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    use_dist = world > 1 or args.dist_alone
    if use_dist:
        if world == 1:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                free_port = sk.getsockname()[1]
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
        # communicator set-up is not part of a step: run the collectives the steps use once,
        # on empty-sized data, before anything is timed (also when --warmup is 0)
        SH.all_reduce_max(0.0, device, even_alone=True)
        SH.gather_records(torch.zeros((1, SH.RECORD_BYTES), dtype=torch.uint8, device=device), world, dst=0,
                          even_alone=True)
        dist.barrier()

    m = M.Modem(M.esquema_b())
    n1, n2 = M.host_dft_plan(m.esquema)
    S = args.streams
    stride = M.pad4(m.esquema.max_frame_samples(126))
    nwin = stride // WIN
    free, _ = torch.cuda.mem_get_info()
    per_stream = stride * 4 + nwin * 10 + 4096
    nb = max(1, args.sub_batches)
    while (S + nb - 1) // nb * per_stream > 0.6 * free:
        nb += 1
    sub = (S + nb - 1) // nb
    audio = torch.empty((sub, stride), dtype=torch.float32, device=device)
    fields, pay = make_inputs(S, rank * S, device)
    # Where the buffer lands in HBM matters: of two allocations of this size in one process, one is
    # written about 3 % and read about 1 % faster than the other, and which one a process gets
    # first alternates from process to process (tools/synth_two_buffers.py).  If a second buffer
    # fits, both are tried with the step's own synthesis launch and the faster one is kept.
    placement = "single allocation"
    if not args.no_pick and 2 * sub * per_stream <= 0.85 * free:
        try:
            other = torch.empty((sub, stride), dtype=torch.float32, device=device)
        except RuntimeError:   # out of memory: keep the one buffer
            other = None
        if other is not None:
            pf, pl = m.build_frames(fields[:sub], pay[:sub], 128)
            times = []
            for buf in (audio, other):
                times.append(timed_launches(lambda: m.encode_streams(pf, pl, stride, audio=buf), reps=2))
            if times[1] < times[0]:
                audio, other = other, audio
            placement = "faster of two allocations (synthesis %.2f ms against %.2f ms)" % (min(times), max(times))
            del other, pf, pl
            torch.cuda.empty_cache()
    tw_timeout = 123.0
    # one-time set-up is not part of a step: with --warmup 0 code objects are loaded by a two-stream
    # pass over every kernel of the path (warm-up steps do the same otherwise)
    if args.warmup == 0:
        pf, pl = m.build_frames(fields[:2], pay[:2], 128)
        pa, _ = m.encode_streams(pf, pl, stride)
        m.decode_streams(pa, 2, stride, nwin, tw_timeout)
        del pa
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range((args.steps + args.warmup) * nb * 2)]
    ev_used = []

    def step(timed):
        outs = []
        for b in range(nb):
            lo, hi = b * sub, min(S, (b + 1) * sub)
            n = hi - lo
            frames, lens = m.build_frames(fields[lo:hi], pay[lo:hi], 128)
            e0 = ev[len(ev_used)]
            e0[0].record()
            m.encode_streams(frames, lens, stride, audio=audio)
            e0[1].record()
            e1 = ev[len(ev_used) + 1]
            e1[0].record()
            cls, mag, kbin = m.detect(audio, n, stride, nwin)
            e1[1].record()
            ev_used.extend([("synth", e0, n, timed), ("detect", e1, n, timed)])
            by, ln, start, end, reason, ncalls = m.slice(cls, m.timeout_windows(tw_timeout), 1, 128)
            rec = m.parse_frames(by.view(n, 128), ln.view(-1))
            outs.append((frames, lens, by.view(n, 128), ln.view(-1), rec, end.view(-1)))
        packed = SH.pack_records(torch.cat([o[2] for o in outs]), torch.cat([o[3] for o in outs]),
                                 torch.cat([o[4] for o in outs]))
        gathered = SH.gather_records(packed, S * world, dst=0, even_alone=use_dist)
        return outs, gathered

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    sync()
    m.detect_stats()                      # counters of the timed steps only
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs, gathered = step(True)
    sync()
    elapsed = time.perf_counter() - t0
    counters = m.detect_stats(paths=True)
    if use_dist:
        elapsed = SH.all_reduce_max(elapsed, device, even_alone=True)

    # correctness of what was timed: every stream's frame came back bit-exact and valid
    for frames, lens, by, ln, rec, end in outs:
        assert torch.equal(ln, lens) and bool((ln == 126).all())
        assert torch.equal(by[:, :126], frames[:, :126])
        r = rec.cpu().numpy().view(M.REC_DTYPE).reshape(-1)
        assert (r["status"] == 0).all() and (r["crc_ok"] == 1).all()
    gathered_ok = None
    if rank == 0 and use_dist:
        # the gathered records are every rank's frames in stream order: rank 0 rebuilds the frames
        # of all S * world streams from their indices (inputs are a function of the index) and compares
        assert gathered.shape == (S * world, SH.RECORD_BYTES)
        g_bytes, g_lens, g_rec = SH.unpack_records(gathered)
        assert bool((g_lens == 126).all())
        for r in range(world):
            f_r, p_r = make_inputs(S, r * S, device)
            want, _ = m.build_frames(f_r, p_r, 128)
            assert torch.equal(g_bytes[r * S:(r + 1) * S, :126], want[:, :126]), "records of rank %d" % r
        gr = g_rec.contiguous().cpu().numpy().view(M.REC_DTYPE).reshape(-1)
        assert (gr["status"] == 0).all() and (gr["crc_ok"] == 1).all() and (gr["length"] == 107).all()
        gathered_ok = True

    if rank == 0:
        audio_s_per_step = world * S * nwin * WIN / 44100.0
        value = audio_s_per_step * args.steps / elapsed
        det = [e[0].elapsed_time(e[1]) for k, e, n, timed in ev_used if k == "detect" and timed]
        syn = [e[0].elapsed_time(e[1]) for k, e, n, timed in ev_used if k == "synth" and timed]
        det_ms = float(np.mean(det))
        syn_ms = float(np.mean(syn))
        win_per_launch = sub * nwin
        # flop of what the launches computed: stage 1 of every window, stage 2 of the spectrum
        # columns the kernel did not leave out (counted by the kernel itself over the timed steps)
        st_windows, _, _, st_cells, st_units, st_redone, st_whole, st_two, st_chosen, st_sep, st_tone = counters
        s1_flop_per_unit, s1_units_full, s2_flop_per_unit, s2_units_full = detect_flop(n1, n2)
        cells_per_window = st_cells / max(st_windows, 1)
        units_per_window = st_units / max(st_windows, 1)
        flop = (s1_flop_per_unit * units_per_window + s2_flop_per_unit * cells_per_window) * win_per_launch
        achieved = flop / (det_ms * 1e-3) / 1e12
        # HBM traffic of a detection launch: not measurable from inside this process; taken from the
        # newest committed counter passes of this same command, only if they are of this kernel and
        # scaled by the windows per launch stored with them
        traffic, traffic_source = profile_traffic("detect_kernel<83, 85, false>", win_per_launch, "windows_per_launch")
        extras = {}
        if not args.no_extras and world == 1:
            extras = side_measurements(m, audio, sub, stride, nwin)
        line = {
            "metric": "audio-seconds demodulated/sec", "value": value, "unit": "audio-s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: %d streams/GPU x one 107-byte-payload frame (126 B, %d windows of 7055), "
                                   "encode+decode round trip, float32 audio in HBM" % (S, nwin),
                       "streams_per_gpu": S, "sub_batches": nb, "stream_stride_samples": stride,
                       "audio_buffer": placement,
                       "parallelism": "streams sharded by contiguous index blocks, gather of 140-B records"},
            "frames_per_s": world * S * args.steps / elapsed,
            "launcher": launcher, "devices_visible": n_dev,
            "gather": None if not use_dist else {
                "backend": "RCCL all-gather of %d-byte records on device tensors" % SH.RECORD_BYTES
                if args.backend == "nccl" else "gloo all-gather, records staged through host memory",
                "records": S * world, "verified_in_stream_order": gathered_ok,
                "ranks_share_devices": bool(world > n_dev)},
            # arithmetic intensity of what the kernel runs (flop per window / bytes per window) is below the
            # ridge of the chip (f32 matrix peak / HBM peak = 19.7 flop/B): HBM is the binding roof
            "roofline": {"bound": "hbm", "kernel": "detect_kernel<83,85> (+ float64 re-evaluation passes)",
                         "achieved": win_per_launch * (WIN * 4 + 10) / (det_ms * 1e-3) / 1e9, "peak": 8000.0,
                         "unit": "GB/s", "frac": win_per_launch * (WIN * 4 + 10) / (det_ms * 1e-3) / 1e9 / 8000.0,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes": win_per_launch * (WIN * 4 + 10),
                         "bytes_per_window": WIN * 4 + 10, "windows_per_launch": win_per_launch,
                         "ms_per_launch": det_ms,
                         "arithmetic_intensity_flop_per_byte": flop / win_per_launch / (WIN * 4 + 10),
                         "ridge_flop_per_byte": PEAK_F32_MFMA_TFLOPS * 1e12 / 8e12,
                         "mfma": {"achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                  "frac": achieved / PEAK_F32_MFMA_TFLOPS, "flop_per_window": flop / win_per_launch,
                                  "flop_per_window_whole_spectrum": detect_flop_per_window(n1, n2),
                                  "note": "flop of what the kernel ran: the parts of both DFT stages it could not "
                                          "rule out by the energy bound (counted by the kernel)"},
                         "stage2_tile_columns_computed_per_window": cells_per_window,
                         "stage2_tile_columns_per_window": s2_units_full,
                         "stage1_tile_columns_computed_per_window": units_per_window,
                         "stage1_tile_columns_per_window": s1_units_full,
                         "passes_the_long_way_per_window": st_whole / max(st_windows, 1),
                         "of_them_without_the_weakest_column_tile": st_two / max(st_windows, 1),
                         "windows_on_sixteen_chosen_columns": st_chosen / max(st_windows, 1),
                         "windows_decided_from_the_separator_bin_alone": st_sep / max(st_windows, 1),
                         "windows_decided_from_another_scheme_tone_bin_alone": st_tone / max(st_windows, 1),
                         "windows_done_twice": st_redone / max(st_windows, 1)},
            "synth": {"bound": "hbm", "kernel": "synth_kernel<float, false>", "ms_per_launch": syn_ms,
                      "algorithmic_bytes": sub * stride * 4,
                      "achieved_GBps": sub * stride * 4 / (syn_ms * 1e-3) / 1e9, "peak_GBps": 8000.0,
                      "frac": sub * stride * 4 / (syn_ms * 1e-3) / 1e9 / 8000.0,
                      "traffic": profile_traffic("synth_kernel<float, false>", sub * stride, "samples_per_launch")[0]},
        }
        line.update(extras)
        if args.cpu_streams > 0 and world == 1:
            v, dt = cpu_baseline(args.cpu_streams)
            line["cpu_baseline"] = {"value": v, "unit": "audio-s/s", "cores": 1, "kind": "port",
                                    "sample": "%d streams of the same workload (synthesise + listen until EOF, "
                                              "NumPy FFT per window), %.1f s of CPU time" % (args.cpu_streams, dt)}
            line["speedup_vs_cpu_core"] = value / v
            line["cpu_baseline_all_cores"] = {
                "value": all_cores[0], "unit": "audio-s/s", "cores": all_cores[1], "kind": "port",
                "sample": "%d streams per core, one single-threaded process per core, %.1f s wall"
                          % (max(1, args.cpu_streams // 2), all_cores[2])}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
