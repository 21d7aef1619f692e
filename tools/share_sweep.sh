#!/bin/bash
# Runs ON THE GPU BOX: the round trip at several values of the one-tile threshold (WN_DETECT_SHARE), twice.
for rep in 1 2; do
for t in ${@:-0.6 0.7 0.8 0.9}; do
line=$(WN_DETECT_SHARE=$t timeout -k 10 200 python3 -u bench.py --cpu-streams 0 --no-extras --steps 6 2>/dev/null)
echo "share $t: $(echo "$line" | grep -o '"ms_per_launch": [0-9.]*\|"passes_the_long_way_per_window": [0-9.]*\|"windows_done_twice": [0-9.]*' | head -3 | paste - - -)"
done; done
