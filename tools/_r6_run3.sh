set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6_c; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "launch_graph or persistent_unit or forward_vs_golden" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/pytest_new.log
bash tools/ab_env.sh BYOLO_WINO_SPLIT_PERSIST 0 1 3 > $O/ab_persist.log 2>&1
cp -r gpurun_out/ab_BYOLO_WINO_SPLIT_PERSIST $O/ 2>/dev/null
python -m pytest tests/test_gpu_bench_shapes.py -x -q -m gpu -k "reference_default_batched" > $O/pytest_shapes.log 2>&1; echo "pytest shapes rc=$?" | tee -a $O/pytest_shapes.log
tail -3 $O/pytest_new.log; cat $O/ab_persist.log; tail -5 $O/pytest_shapes.log
for v in 0 1; do grep "| 140 |" $O/ab_BYOLO_WINO_SPLIT_PERSIST/steps_${v}_1.md | awk -F'|' '{s+=$8; n++} END {print "persist='$v' variant 140:", n, "launches", s, "ms"}'; done
