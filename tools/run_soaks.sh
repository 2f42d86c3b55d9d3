#!/bin/bash
# run_soaks.sh [SEED] -- the randomised sweeps of the hot path (parity, NMS, decode) and of the graph lowering (random layer graphs / random
# builder call sequences through the C-ABI: the views the 1x1 loop, the padded detection rows and the split-precision plans read) once, on the
# GPU box (about eight minutes):
#   gpurun --timeout 1800 -- 'bash tools/run_soaks.sh 7'
# Logs under gpurun_out/soaks_SEED/; the last line of each log is its verdict, the exit code is the number of sweeps
# that reported a failure.
set -u
SEED=${1:-1}
OUT=$PWD/gpurun_out/soaks_$SEED
mkdir -p "$OUT"
bad=0
run() {
    name=$1; shift
    timeout 1500 python "$@" > "$OUT/$name.md" 2> "$OUT/$name.err" || bad=$((bad + 1))
    echo "$name: $(tail -1 "$OUT/$name.md")"
}
run parity      tools/soak_parity.py --cases 60 --seed "$SEED"
run parity_big  tools/soak_parity.py --cases 12 --seed "$SEED" --max-cells 20 --max-T 30 --max-B 8 --budget 12000
run nms         tools/soak_nms.py --cases 400 --seed "$SEED"
run decode      tools/soak_decode.py --cases 400 --seed "$SEED"
run fuzz_graph  tools/fuzz_graph.py --cases 60 --seed "$SEED"
BYOLO_PRECISION=f32 run fuzz_graph_f32 tools/fuzz_graph.py --cases 40 --seed "$SEED"
run fuzz_builder tools/fuzz_builder.py --runs 100 --seed "$SEED"
echo "$bad sweep(s) failed"
exit $bad
