#!/bin/bash
# profile_commit.sh TAG -- turn gpurun_out/prof_TAG (tools/profile_round.sh TAG) into the files committed under profiles/:
#   profiles/TAG_bench_line.json, TAG_per_launch.md, TAG_bench_kernel_stats.md (rocprofv3 --stats of the same command),
#   profiles/traffic_cfg4_{wino,b2b}.json (PMC FETCH_SIZE / WRITE_SIZE per launch), TAG_pmc_sq_summary_{wino,b2b}.json.
# Runs on the GPU box right after the round (the sqlite database is too big to travel), or here on the merged CSVs.
set -u
TAG=$1
D=gpurun_out/prof_$TAG
cp "$D/bench_line.json" profiles/${TAG}_bench_line.json
cp "$D/per_launch.md" profiles/${TAG}_per_launch.md
DB=$(find "$D/stats" -name "*.db" | head -1)
if [ -n "$DB" ]; then
    python tools/rocprof_summary.py "$DB" --steps 5 --launches-per-step 6 --kernel "wino_split_kernel" > profiles/${TAG}_bench_kernel_stats.md
    echo >> profiles/${TAG}_bench_kernel_stats.md
    python tools/rocprof_summary.py "$DB" --steps 5 --launches-per-step 3 --kernel "conv_igemm_kernel<128, 256" | sed -n '/timed region/,$p' >> profiles/${TAG}_bench_kernel_stats.md
fi
C=bayesian-yolov3_amd/csrc
# (--sources: bench.py refuses a traffic file once the kernel it was measured on has changed)
python tools/pmc_traffic.py "$D" --kernel "wino_split_kernel" --launches 6 --tag "$TAG" --sources $C/wino_split.hip $C/mfma_pipe.h $C/epilogue.h --out profiles/traffic_cfg4_wino.json > /dev/null
python tools/pmc_traffic.py "$D" --kernel "wino_split_input2_kernel" --launches 6 --tag "$TAG" --sources $C/wino_split.hip --out profiles/traffic_cfg4_wino_input.json > /dev/null
python tools/pmc_traffic.py "$D" --kernel "conv_igemm_kernel<128, 256" --launches 3 --tag "$TAG" --sources $C/conv_igemm.hip $C/mfma_pipe.h $C/epilogue.h --out profiles/traffic_cfg4_b2b.json > /dev/null
python tools/pmc_sq_summary.py "$D" --kernel "wino_split_kernel" --launches 6 > profiles/${TAG}_pmc_sq_summary_wino.json
python tools/pmc_sq_summary.py "$D" --kernel "conv_igemm_kernel<128, 256" --launches 3 > profiles/${TAG}_pmc_sq_summary_b2b.json
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* profiles/traffic_cfg4_wino.json profiles/traffic_cfg4_wino_input.json profiles/traffic_cfg4_b2b.json gpurun_out/profiles_$TAG/
ls -la gpurun_out/profiles_$TAG
