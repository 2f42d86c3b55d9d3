cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
export TMPDIR=/tmp
(time timeout 2400 python -m pytest tests -m gpu -q) > gpurun_out/r3h/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -5 gpurun_out/r3h/gpu_suite.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3h/smoke.log 2>&1; tail -1 gpurun_out/r3h/smoke.log | cut -c1-300
python bench.py > gpurun_out/r3h/bench.json 2> gpurun_out/r3h/bench.err; echo "bench rc=$?"; cut -c1-200 gpurun_out/r3h/bench.json
for c in 2 3 5 6; do python bench.py --config $c --fp32-steps 0 --no-cpu-baseline > gpurun_out/r3h/bench_c$c.json 2> gpurun_out/r3h/bench_c$c.err; cut -c1-140 gpurun_out/r3h/bench_c$c.json; done
python bench.py --scaling strong --global-batch 64 --fp32-steps 0 --no-cpu-baseline > gpurun_out/r3h/bench_strong.json 2>/dev/null; cut -c1-140 gpurun_out/r3h/bench_strong.json
