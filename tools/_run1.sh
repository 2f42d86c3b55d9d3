set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests/test_robustness.py -m gpu -x -q -s) > gpurun_out/r3a/robust.log 2>&1
echo "robust rc=$?" 
tail -5 gpurun_out/r3a/robust.log
(time timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_robustness.py) > gpurun_out/r3a/gpu_suite.log 2>&1
echo "suite rc=$?"
tail -5 gpurun_out/r3a/gpu_suite.log
python bench.py --steps 20 --warmup 5 --dump-steps gpurun_out/r3a/per_launch.md > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/r3a/bench.json
