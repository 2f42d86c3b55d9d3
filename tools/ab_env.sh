#!/bin/bash
# ab_env.sh KNOB A B [RUNS] -- A/B of an environment knob on ONE box: rows_digest for both values (must be identical when the knob
# only moves work between kernels), then RUNS alternating bench.py runs per value with the per-launch table.
#   gpurun -- 'bash tools/ab_env.sh BYOLO_P1_WIDE 0 1 2'
set -u
KNOB=$1; A=$2; B=$3; RUNS=${4:-2}
OUT=$PWD/gpurun_out/ab_$KNOB
mkdir -p "$OUT"
for v in $A $B; do env $KNOB=$v timeout 300 python tools/rows_digest.py > "$OUT/dig_$v.txt" 2> "$OUT/dig_$v.err"; done
if cmp -s "$OUT/dig_$A.txt" "$OUT/dig_$B.txt"; then echo "digests identical ($(wc -l < "$OUT/dig_$A.txt") lines)"; else echo "DIGESTS DIFFER"; diff "$OUT/dig_$A.txt" "$OUT/dig_$B.txt"; fi
for i in $(seq 1 "$RUNS"); do
    for v in $A $B; do
        env $KNOB=$v timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-configs --fp32-steps 0 --entry-frames 0 \
            --dump-steps "$OUT/steps_${v}_$i.md" > "$OUT/line_${v}_$i.json" 2> "$OUT/err_${v}_$i.txt"
        python - "$KNOB=$v run $i" "$OUT/line_${v}_$i.json" <<'PY'
import sys, json
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-24s %.1f img/s  %.3f ms/step" % (sys.argv[1], d["value"], d["ms_per_step"]))
PY
    done
done
