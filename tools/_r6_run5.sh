set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6_e; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
bash tools/pmc_sq_ab.sh wino_persist0 "wino_split_kernel" 6 BYOLO_WINO_SPLIT_PERSIST=0 > $O/pmc0.log 2>&1
bash tools/pmc_sq_ab.sh wino_persist1 "wino_split_kernel" 6 BYOLO_WINO_SPLIT_PERSIST=1 > $O/pmc1.log 2>&1
cp gpurun_out/pmcab_wino_persist0/summary.json $O/pmc_sq_summary_wino_persist0.json; cp gpurun_out/pmcab_wino_persist1/summary.json $O/pmc_sq_summary_wino_persist1.json
cat $O/pmc_sq_summary_wino_persist0.json $O/pmc_sq_summary_wino_persist1.json
bash tools/ceiling_probe.sh $O/ceiling > $O/ceiling.log 2>&1; cat $O/ceiling.log
