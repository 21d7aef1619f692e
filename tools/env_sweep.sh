#!/bin/bash
# Runs ON THE GPU BOX: the round trip once per setting of environment switches; a setting is a comma-separated
# list of VAR=value (or "-" for none).   tools/env_sweep.sh "- WN_DETECT_PICK=0.004 WN_DETECT_PICK=0.016,WN_DETECT_SHARE=0.8 -" [steps]
set -e -o pipefail
steps=${2:-6}
for cfg in $1; do
    envs=()
    if [ "$cfg" != "-" ]; then IFS=',' read -ra envs <<< "$cfg"; fi
    line=$(env "${envs[@]}" timeout -k 10 200 python3 -u bench.py --cpu-streams 0 --no-extras --steps $steps)
    echo "$cfg $(echo "$line" | grep -o '"ms_per_step": [0-9.]*\|"ms_per_launch": [0-9.]*\|"windows_done_twice": [0-9.]*\|"windows_on_sixteen_chosen_columns": [0-9.]*\|"passes_the_long_way_per_window": [0-9.]*' | head -6 | paste - - - - - -)"
done
