set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "winograd_in_split") > gpurun_out/r3c/wino.log 2>&1
echo "wino rc=$?"; tail -8 gpurun_out/r3c/wino.log | cut -c1-300
for bm in 64 128; do
BYOLO_WINO_SPLIT_BM=$bm python bench.py --steps 10 --warmup 3 --fp32-steps 0 --no-cpu-baseline --dump-steps gpurun_out/r3c/per_launch_w$bm.md > gpurun_out/r3c/bench_w$bm.json 2> gpurun_out/r3c/bench_w$bm.err
echo "bench wino bm=$bm rc=$?"; cut -c1-200 gpurun_out/r3c/bench_w$bm.json; tail -3 gpurun_out/r3c/bench_w$bm.err
done
BYOLO_WINO_SPLIT=0 python bench.py --steps 10 --warmup 3 --fp32-steps 0 --no-cpu-baseline --dump-steps gpurun_out/r3c/per_launch_w0.md > gpurun_out/r3c/bench_w0.json 2> gpurun_out/r3c/bench_w0.err
echo "bench direct rc=$?"; cut -c1-200 gpurun_out/r3c/bench_w0.json
(time timeout 1500 python -m pytest tests/test_robustness.py tests/test_entry_points.py tests/test_gpu_bench_shapes.py -m gpu -q -s -k "split_range or detect_do_it_boxes or as_benched") > gpurun_out/r3c/changed.log 2>&1
echo "changed rc=$?"; tail -5 gpurun_out/r3c/changed.log | cut -c1-300
