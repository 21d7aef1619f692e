#!/bin/bash
# Runs ON THE GPU BOX: bench.py with two builds of the library in the order A B B A A B B A (fresh
# processes alternate between a faster and a slower placement of the 122 GB buffer, so each
# build gets both).   tools/ab.sh <other_lib.so> [steps]
set -e -o pipefail
other=$1
steps=${2:-6}
for which in A B B A A B B A; do
    if [ $which = A ]; then
        line=$(WN_HIP_LIB=$other timeout -k 10 200 python3 -u bench.py --cpu-streams 0 --no-extras --steps $steps)
    else
        line=$(timeout -k 10 200 python3 -u bench.py --cpu-streams 0 --no-extras --steps $steps)
    fi
    echo "$which $(echo "$line" | grep -o '"ms_per_step": [0-9.]*\|"ms_per_launch": [0-9.]*' | paste - - -)"
done
