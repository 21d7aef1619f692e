#!/bin/bash
# Runs ON THE GPU BOX: the round trip at several numbers of streams per step.
for n in 250 500 1000 2000 5000 10000 15000; do
line=$(timeout -k 10 200 python3 -u bench.py --streams $n --cpu-streams 0 --no-extras --steps 5 2>/dev/null)
echo "streams $n: $(echo "$line" | grep -o '"value": [0-9.e+]*\|"ms_per_step": [0-9.]*\|"ms_per_launch": [0-9.]*' | head -4 | paste - - - -)"
done
