#!/bin/bash
# Runs ON THE GPU BOX: the round trip with the committed library and with a build that leaves out the weakest
# column tile, at several thresholds.
for rep in 1 2; do
line=$(WN_HIP_LIB=tools/ab/lib_head.so timeout -k 10 200 python3 -u bench.py --cpu-streams 0 --no-extras --steps 6 2>/dev/null)
echo "head: $(echo "$line" | grep -o '"ms_per_launch": [0-9.]*' | head -1)"
for t in 0 0.1 0.2 0.3 0.333; do
line=$(WN_DETECT_TWO=$t WN_HIP_LIB=$1 timeout -k 10 200 python3 -u bench.py --cpu-streams 0 --no-extras --steps 6 2>/dev/null)
echo "two $t: $(echo "$line" | grep -o '"ms_per_launch": [0-9.]*\|"of_them_without_the_weakest_column_tile": [0-9.]*\|"windows_done_twice": [0-9.]*' | head -3 | paste - - -)"
done; done
