#!/bin/bash
# ab_libs.sh LIB... -- A/B experiment builds of libbyolo on ONE box (box-to-box variance is +-2 %, more than most
# single changes): one bench.py run per library (BYOLO_LIB), img/s and milliseconds per step by kernel variant
# (130 fused Winograd GEMM, 129 streaming GEMM, 128/64/32 direct convolution tiles, -1 stem, -2/-3 Winograd transforms).
#   gpurun -- 'bash tools/ab_libs.sh bayesian-yolov3_amd/byolo/libbyolo.so bayesian-yolov3_amd/byolo/libbyolo_exp1.so'
set -u
OUT=$PWD/gpurun_out/ab_libs
mkdir -p "$OUT"
i=0
for lib in "$@" "$1"; do            # the first library runs again at the end: drift check
    i=$((i + 1))
    BYOLO_LIB=$PWD/$lib timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --dump-steps "$OUT/steps_$i.md" > "$OUT/line_$i.json" 2> "$OUT/err_$i.txt"
    python - "$lib" "$OUT/line_$i.json" "$OUT/steps_$i.md" <<'PY'
import sys, json, collections
lib, line, steps = sys.argv[1:4]
d = json.loads(open(line).read().strip().splitlines()[-1])
by = collections.defaultdict(float)
for l in open(steps):
    c = [x.strip() for x in l.split("|")]
    if len(c) > 9 and c[1].isdigit():
        by[int(c[3])] += float(c[7])
print("%-28s %.1f img/s  " % (lib.split("/")[-1], d["value"]) + "  ".join("%d: %.2f" % (k, v) for k, v in sorted(by.items(), reverse=True)))
PY
done
