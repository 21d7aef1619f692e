for part in 0 35280 70560 141120 282240; do for pc in 0 4 8 16 32; do
echo -n "part $part per_cu $pc: "; WN_SYNTH_PART=$part WN_SYNTH_PER_CU=$pc timeout -k 10 100 python tools/kernel_bench.py --only protoA --reps 5 2>&1 | tail -1 | grep -o '"synth_ms": [0-9.]*'
done; done
