#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the measurements behind profiles/<tag>_*.
#   tools/collect_profiles.sh <tag>
# Leaves under gpurun_out/<tag>/: bench.json (plain run), stats/ (rocprofv3 --kernel-trace --stats
# of the same command), pmc_fetch/ pmc_write/ pmc_sq/ (one counter pass each, kernel trace
# only) and kernel_bench.jsonl.  tools/summarise_profiles.py turns them into profiles/<tag>_*.
set -e -o pipefail
tag=${1:?tag}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
bench="python3 bench.py --steps 3 --warmup 1"
one="python3 bench.py --steps 1 --warmup 0 --cpu-streams 0"
timeout -k 10 300 $bench > "$out/bench.json"
echo "bench done" >&2
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/stats" -o s -- $bench --cpu-streams 0 > "$out/stats.log" 2>&1
echo "stats done" >&2
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" -o f -- $one > "$out/pmc_fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write" -o w -- $one > "$out/pmc_write.log" 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d "$out/pmc_sq" -o q -- $one > "$out/pmc_sq.log" 2>&1
echo "pmc done" >&2
timeout -k 10 400 python3 tools/kernel_bench.py > "$out/kernel_bench.jsonl"
echo "kernel bench done" >&2
