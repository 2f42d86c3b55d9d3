#!/bin/bash
# ab_trees.sh OLD_TREE [ROUNDS] -- A/B of two whole source trees (their own bench.py + libbyolo.so) on ONE box, interleaved:
# for changes the library's ABI check would refuse under BYOLO_LIB (round 5: the dropout stream, ABI 5 -> 6).  OLD_TREE is a copy
# of an earlier commit with its library built (git worktree + csrc/build.py), placed inside the repo so that gpurun ships it.
#   gpurun -- 'bash tools/ab_trees.sh _ab_old 3'
set -u
OLD=$1
N=${2:-3}
OUT=$PWD/gpurun_out/ab_trees
mkdir -p "$OUT"
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 --entry-frames 0 --no-other-configs"
for i in $(seq 1 $N); do
    for t in old new; do
        if [ $t = old ]; then B=$PWD/$OLD/bench.py; else B=$PWD/bench.py; fi
        timeout 300 python $B $ARGS --dump-steps "$OUT/steps_${t}_$i.md" > "$OUT/line_${t}_$i.json" 2> "$OUT/err_${t}_$i.txt"
        python - $t $i "$OUT/line_${t}_$i.json" <<'PY'
import sys, json
t, i, p = sys.argv[1:4]
d = json.loads(open(p).read().strip().splitlines()[-1])
bk = d["roofline"]["by_kernel"]
print("%s %s: %.1f img/s  %.3f ms/step  " % (t, i, d["value"], d["ms_per_step"]) + "  ".join("%s %.2f" % (k.split("<")[0][-12:] + k[k.find("<"):][:14], v["ms"]) for k, v in list(bk.items())[:4]))
PY
    done
done
