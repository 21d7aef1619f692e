#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the
committed summaries under profiles/:

  <tag>_bench.json                    the bench line of the plain run
  <tag>_bench_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the bench command
  <tag>_pmc.json                      counter passes over one step of the bench command
  <tag>_kernel_bench.jsonl            tools/kernel_bench.py, event-timed
  <tag>_kernelbench_kernel_stats.csv  rocprofv3 --kernel-trace --stats of tools/kernel_bench.py --reps 1
  <tag>_kernelbench_pmc.json          FETCH_SIZE / WRITE_SIZE passes of the same, per launch, largest launch per kernel

    python tools/summarise_profiles.py <tag> [output directory [commit]]

The rocprofv3 databases are too large to bring back from the GPU box, so the script is run there
with gpurun_out/<tag>_profiles as output directory and the commit given on the command line; the
files are then copied into profiles/ here.

HBM bytes per kernel follow the rocprofv3 recipe for this chip: FETCH_SIZE and WRITE_SIZE are
KiB; FETCH_SIZE counts 16-byte-per-lane streaming reads at half their size on gfx950, so
kernels that read that way (named in STREAMING_X4) get their fetch bytes doubled.
"""
import csv
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAMING_X4 = ("detect_kernel", "sync_kernel", "crc_rows_kernel")


def db(path):
    files = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    assert files, "no rocpd database under " + path
    return sqlite3.connect(files[0])


def short(name):
    n = name.split("(")[0]
    return n[5:] if n.startswith("void ") else n


def stats_csv(src_dir, dst):
    rows = db(src_dir).execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                               "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], "%.1f" % r[3], r[4], r[5], "%.3f" % (100.0 * r[2] / total)])


def counters(src, subdirs, per="mean"):
    """{kernel: {counter: value per launch}}.  per="mean": average over the launches of a kernel
    (all launches alike); per="max": the launch with the largest value (runs whose launches differ
    in size: the full-size launch is the one the figures are quoted on)."""
    kernels = {}
    for sub in subdirs:
        if not os.path.isdir(os.path.join(src, sub)):
            continue
        c = db(os.path.join(src, sub))
        q = ("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by 1, 2, 3")
        per_launch = {}
        for name, counter, _, value in c.execute(q):
            per_launch.setdefault((short(name), counter), []).append(value)
        for (name, counter), vals in per_launch.items():
            k = kernels.setdefault(name, {"launches": len(vals)})
            k[counter] = max(vals) if per == "max" else sum(vals) / len(vals)
    for name, k in kernels.items():
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            fetch = k["FETCH_SIZE"] * 1024.0
            if name.startswith(STREAMING_X4):
                fetch *= 2.0
            k["fetch_bytes_corrected"] = fetch
            k["write_bytes"] = k["WRITE_SIZE"] * 1024.0
            k["hbm_bytes"] = fetch + k["write_bytes"]
    return kernels


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    outdir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
    os.makedirs(outdir, exist_ok=True)
    dst = os.path.join(outdir, tag)
    if len(sys.argv) > 3:
        commit = sys.argv[3]
    else:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    shutil.copy(os.path.join(src, "bench.json"), dst + "_bench.json")
    shutil.copy(os.path.join(src, "kernel_bench.jsonl"), dst + "_kernel_bench.jsonl")
    stats_csv(os.path.join(src, "stats"), dst + "_bench_kernel_stats.csv")
    with open(os.path.join(src, "bench.json")) as f:
        bench = json.loads(f.read().strip().splitlines()[-1])
    notes = ("values are per launch, of the largest launch of each kernel (the full-size one); FETCH_SIZE and WRITE_SIZE "
             "are KiB; fetch_bytes_corrected doubles FETCH_SIZE for the kernels that stream 16 bytes per lane (gfx950 "
             "counts those reads at half size)")
    # the one-step run launches every kernel twice: a two-stream set-up pass, then the full-size step
    kernels = counters(src, ("pmc_fetch", "pmc_write", "pmc_sq"), per="max")
    out = {
        "command": "rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 1 --warmup 0 --cpu-streams 0 "
                   "--no-extras (one run per counter group: FETCH_SIZE; WRITE_SIZE; seven SQ counters)",
        "workload": bench["config"]["workload"],
        "windows_per_launch": bench["roofline"]["windows_per_launch"],
        "samples_per_launch": bench["config"]["streams_per_gpu"] // bench["config"]["sub_batches"]
        * bench["config"]["stream_stride_samples"],
        "source_commit_or_parent": commit,
        "notes": notes,
        "kernels": kernels,
    }
    with open(dst + "_pmc.json", "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    if os.path.isdir(os.path.join(src, "kb_stats")):
        stats_csv(os.path.join(src, "kb_stats"), dst + "_kernelbench_kernel_stats.csv")
        kb = counters(src, ("kb_fetch", "kb_write"), per="max")
        with open(os.path.join(src, "kernel_bench.jsonl")) as f:
            units = {}
            for line in f:
                if line.startswith("{"):
                    o = json.loads(line)
                    units[o["kernel"]] = {k: o[k] for k in ("units", "algorithmic_bytes", "windows") if k in o}
        with open(dst + "_kernelbench_pmc.json", "w") as f:
            json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/kernel_bench.py --reps 1 "
                                  "(one run per counter)",
                       "notes": notes + "; a kernel launched at several sizes is reported at its largest launch",
                       "source_commit_or_parent": commit, "workloads": units, "kernels": kb}, f, indent=1, sort_keys=True)
        kernels = dict(kb, **kernels)
    for name in sorted(kernels):
        k = kernels[name]
        if "hbm_bytes" in k:
            print("%-44s %8.3f GB/launch" % (name[:44], k["hbm_bytes"] / 1e9))


if __name__ == "__main__":
    main()
