#!/bin/bash
# profile_round.sh TAG -- the evidence runs behind bench.py's `roofline` object, on the GPU box:
#   1. rocprofv3 --kernel-trace --stats over the headline bench command (per-kernel time; the average
#      launch duration of the dominant kernel must agree with bench.py's hipEvent figure),
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: fabric/HBM bytes per launch),
#   3. one --pmc pass with the SQ counters (matrix-pipe busy, VALU per MFMA, waits).
# Everything lands under gpurun_out/prof_TAG/; tools/rocprof_summary.py and tools/pmc_traffic.py turn it
# into the files committed under profiles/.
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r1_h'
set -u
TAG=${1:-round}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --fp32-steps 0 --entry-frames 0 --no-other-configs"
ONE="python $PWD/bench.py --steps 1 --warmup 1 --no-profile --no-cpu-baseline --fp32-steps 0 --pipeline 1 --entry-frames 0 --no-other-configs"

cd /tmp
timeout 600 $BENCH --dump-steps "$OUT/per_launch.md" > "$OUT/bench_line.json" 2> "$OUT/bench.err"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench -- $BENCH > "$OUT/stats_bench_line.json" 2> "$OUT/stats.err"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -o pmc -- $ONE > /dev/null 2> "$OUT/$c.err"
done
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES \
    --output-format csv -d "$OUT/SQ_A" -o pmc -- $ONE > /dev/null 2> "$OUT/SQ_A.err"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/SQ_B" -o pmc -- $ONE > /dev/null 2> "$OUT/SQ_B.err"
# keep the merge-back small: the sqlite database and the counter CSVs are what the summaries read
find "$OUT" -name "*.db" -size +60M -delete
du -sh "$OUT"
ls -R "$OUT" | head -40
