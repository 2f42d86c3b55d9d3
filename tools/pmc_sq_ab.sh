#!/bin/bash
# pmc_sq_ab.sh TAG KERNEL LAUNCHES [env assignments...] -- the two SQ counter passes of tools/profile_round.sh over ONE bench step
# under the given environment, summarised for KERNEL (tools/pmc_sq_summary.py): clock, matrix-pipe busy, vector instructions per
# MFMA, wait fractions.  For same-box A/B of kernel variants:
#   gpurun -- 'bash tools/pmc_sq_ab.sh big "conv_kx3_big_kernel" 3 BYOLO_B2B=0 BYOLO_KX3_BIG=1'
set -u
TAG=$1; KERNEL=$2; N=$3; shift 3
OUT=$PWD/gpurun_out/pmcab_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
ONE="python $PWD/bench.py --steps 1 --warmup 1 --no-profile --no-cpu-baseline --fp32-steps 0 --pipeline 1 --entry-frames 0 --no-other-configs"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES \
    --output-format csv -d "$OUT/SQ_A" -o pmc -- $ONE > /dev/null 2> "$OUT/SQ_A.err"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/SQ_B" -o pmc -- $ONE > /dev/null 2> "$OUT/SQ_B.err"
find "$OUT" -name "*.db" -delete
python "$OLDPWD/tools/pmc_sq_summary.py" "$OUT" --kernel "$KERNEL" --launches "$N" > "$OUT/summary.json"
cat "$OUT/summary.json"
