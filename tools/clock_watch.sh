#!/bin/bash
# clock_watch.sh -- sample the GPU's shader clock and socket power while bench.py runs (is the fp32 MFMA rate we
# price against the one the box sustains under the real instruction mix?).  Output: gpurun_out/clock_watch.txt
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
python bench.py --steps 60 --warmup 5 --no-cpu-baseline > "$OUT/clock_bench.json" 2> /dev/null &
BP=$!
: > "$OUT/clock_watch.txt"
while kill -0 $BP 2> /dev/null; do
    rocm-smi --showclocks --showpower 2> /dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> "$OUT/clock_watch.txt"
    echo >> "$OUT/clock_watch.txt"
    sleep 0.2
done
wait $BP
sort "$OUT/clock_watch.txt" | uniq -c | sort -rn | head -12
tail -c 300 "$OUT/clock_bench.json"
