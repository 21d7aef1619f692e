#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the measurements behind profiles/<tag>_*.
#   tools/collect_profiles.sh <tag>
# Leaves under gpurun_out/<tag>/:
#   bench.json                       plain run of bench.py
#   stats/                           rocprofv3 --kernel-trace --stats of the same command
#   pmc_fetch/ pmc_write/ pmc_sq/    one counter pass each over one step of bench.py (kernel trace only)
#   kernel_bench.jsonl               plain run of tools/kernel_bench.py (K1-K4, long capture, prototype scheme)
#   kb_stats/ kb_fetch/ kb_write/    the same three kinds of pass over tools/kernel_bench.py --reps 1
# tools/summarise_profiles.py turns them into profiles/<tag>_*.
set -e -o pipefail
tag=${1:?tag}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
bench="python3 bench.py --steps 3 --warmup 1"
one="python3 bench.py --steps 1 --warmup 0 --cpu-streams 0 --no-extras"
kb="python3 tools/kernel_bench.py --reps 1"
timeout -k 10 300 $bench > "$out/bench.json"
echo "bench done" >&2
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/stats" -o s -- $bench --cpu-streams 0 --no-extras > "$out/stats.log" 2>&1
echo "stats done" >&2
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" -o f -- $one > "$out/pmc_fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write" -o w -- $one > "$out/pmc_write.log" 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d "$out/pmc_sq" -o q -- $one > "$out/pmc_sq.log" 2>&1
echo "pmc done" >&2
timeout -k 10 400 python3 tools/kernel_bench.py > "$out/kernel_bench.jsonl"
echo "kernel bench done" >&2
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$out/kb_stats" -o s -- $kb > "$out/kb_stats.log" 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/kb_fetch" -o f -- $kb > "$out/kb_fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/kb_write" -o w -- $kb > "$out/kb_write.log" 2>&1
echo "kernel bench profiles done" >&2
