#!/bin/bash
# pmc_mem_probe.sh TAG -- memory-path counters of ONE bench step (separate --pmc passes; kernel trace only), for the question
# "what does a kernel wait for": LDS, the texture-address path (vector L1), the L2 return path.  Output: gpurun_out/prof_TAG/MEM_*/
set -u
TAG=${1:-mem}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ONE="python $PWD/bench.py --steps 1 --warmup 1 --no-profile --no-cpu-baseline --fp32-steps 0 --pipeline 1 --entry-frames 0 --no-other-configs"
cd /tmp
pass() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- $ONE > /dev/null 2> "$OUT/$name.err"; tail -2 "$OUT/$name.err" | cut -c1-200; }
pass MEM_A SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass MEM_B TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE
pass MEM_C TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES GRBM_GUI_ACTIVE
pass MEM_D TD_TD_BUSY TD_TC_STALL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
du -sh "$OUT"
