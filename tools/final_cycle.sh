#!/bin/bash
# final_cycle.sh TAG -- what a tree is measured by before its numbers go into the documents, in ONE gpurun call (one box): the GPU
# suite, the profile round (tools/profile_round.sh: rocprofv3 --kernel-trace --stats + the PMC passes) turned into the files under
# profiles/ (tools/profile_commit.sh: fresh traffic_cfg4_*.json, stamped with the kernel sources' hashes), then the default bench.py.
#   gpurun --timeout 3000 -- 'bash tools/final_cycle.sh r6_n'
TAG=$1; mkdir -p gpurun_out/$TAG
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.log 2>&1; tail -3 gpurun_out/$TAG/pytest_gpu.log
cp gpurun_out/parity_table.json gpurun_out/$TAG/parity_table.json 2>/dev/null      # (tests/conftest.py writes it there)
bash tools/profile_round.sh $TAG > gpurun_out/$TAG/profile_round.log 2>&1
bash tools/profile_commit.sh $TAG > gpurun_out/$TAG/profile_commit.log 2>&1
timeout 900 python bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err; tail -c 300 gpurun_out/$TAG/bench_default.json
