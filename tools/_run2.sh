set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests/test_robustness.py -m gpu -q -s -k "not adversarial_weights_parity and not per_channel") > gpurun_out/r3b/robust.log 2>&1
echo "robust rc=$?"; tail -5 gpurun_out/r3b/robust.log
(time timeout 1500 python -m pytest tests/test_entry_points.py tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -m gpu -q -s -k "detect_do_it or forward_vs_golden or reference_default_frame or as_benched") > gpurun_out/r3b/changed.log 2>&1
echo "changed rc=$?"; tail -5 gpurun_out/r3b/changed.log
python bench.py --steps 20 --warmup 5 --dump-steps gpurun_out/r3b/per_launch_p2.md > gpurun_out/r3b/bench_p2.json 2> gpurun_out/r3b/bench_p2.err
echo "bench p2 rc=$?"; cut -c1-300 gpurun_out/r3b/bench_p2.json
python bench.py --steps 20 --warmup 5 --pipeline 1 --fp32-steps 0 --no-cpu-baseline > gpurun_out/r3b/bench_p1.json 2> gpurun_out/r3b/bench_p1.err
echo "bench p1 rc=$?"; cut -c1-300 gpurun_out/r3b/bench_p1.json
python bench.py --steps 20 --warmup 5 --pipeline 2 --fp32-steps 0 --no-cpu-baseline > gpurun_out/r3b/bench_p2b.json 2> gpurun_out/r3b/bench_p2b.err
cut -c1-300 gpurun_out/r3b/bench_p2b.json
for c in 2 3 5; do python bench.py --config $c --steps 20 --warmup 5 --fp32-steps 0 --no-cpu-baseline > gpurun_out/r3b/bench_c$c.json 2> gpurun_out/r3b/bench_c$c.err; cut -c1-200 gpurun_out/r3b/bench_c$c.json; python bench.py --config $c --pipeline 1 --steps 20 --warmup 5 --fp32-steps 0 --no-cpu-baseline > gpurun_out/r3b/bench_c${c}_p1.json 2>/dev/null; cut -c1-200 gpurun_out/r3b/bench_c${c}_p1.json; done
