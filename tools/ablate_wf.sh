#!/bin/bash
# ablate_wf.sh -- time the fused Winograd kernel's ablation builds (csrc/build.py --ablate-wf N) on the GPU box:
# one bench.py run per library, the kernel's executed TFLOP/s per K class from the per-launch table.
set -u
OUT=$PWD/gpurun_out/ablate_wf
mkdir -p "$OUT"
for n in 0 "$@"; do
    lib=$PWD/bayesian-yolov3_amd/byolo/libbyolo_wf$n.so
    [ "$n" = 0 ] && lib=$PWD/bayesian-yolov3_amd/byolo/libbyolo.so
    BYOLO_LIB=$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-steps "$OUT/steps_$n.md" > "$OUT/line_$n.json" 2> "$OUT/err_$n.txt"
    python - "$n" "$OUT" <<'PY'
import sys, json
n, out = sys.argv[1], sys.argv[2]
line = json.loads(open("%s/line_%s.json" % (out, n)).read().strip().splitlines()[-1])
by = {}
for l in open("%s/steps_%s.md" % (out, n)):
    c = [x.strip() for x in l.split("|")]
    if len(c) > 9 and c[3] == "130":
        k = int(c[6]); ms = float(c[7]); tf = float(c[8])
        a = by.setdefault(k, [0.0, 0.0]); a[0] += ms; a[1] += ms * tf
print("wf%-3s img/s %.1f  fused: %s" % (n, line["value"], "  ".join("K=%d %.2f ms %.1f TF" % (k, v[0], v[1] / v[0]) for k, v in sorted(by.items()))))
PY
done
