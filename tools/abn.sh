#!/bin/bash
# Runs ON THE GPU BOX: bench.py with several builds of the library in turn, twice over (A B C ... C B A),
# so that every build gets both placements of the 122 GB buffer.   tools/abn.sh lib1.so lib2.so ... [-- steps]
set -e -o pipefail
libs=()
steps=6
while [ $# -gt 0 ]; do
    if [ "$1" = "--" ]; then steps=$2; shift 2; else libs+=("$1"); shift; fi
done
order=("${libs[@]}")
for ((i=${#libs[@]}-1; i>=0; i--)); do order+=("${libs[$i]}"); done
for lib in "${order[@]}"; do
    line=$(WN_HIP_LIB=$lib timeout -k 10 200 python3 -u bench.py --cpu-streams 0 --no-extras --steps $steps)
    echo "$(basename $lib) $(echo "$line" | grep -o '"ms_per_step": [0-9.]*\|"ms_per_launch": [0-9.]*\|"share_of_windows_on_every_column_tile": [0-9.]*\|"windows_done_twice": [0-9.]*' | head -5 | paste - - - - -)"
done
