set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
export TMPDIR=/tmp
for n in 0 1 2 4 8 15 0; do
  if [ $n = 0 ]; then unset BYOLO_LIB; else export BYOLO_LIB=$PWD/bayesian-yolov3_amd/byolo/libbyolo_ws$n.so; fi
  python bench.py --steps 5 --warmup 2 --pipeline 1 --fp32-steps 0 --no-cpu-baseline --dump-steps gpurun_out/r3d/pl_abl$n.md > gpurun_out/r3d/bench_abl$n.json 2> gpurun_out/r3d/bench_abl$n.err
  echo "abl $n: $(cut -c1-110 gpurun_out/r3d/bench_abl$n.json)"; grep -E "\| 140 \|" gpurun_out/r3d/pl_abl$n.md | awk -F'|' '{print $7, $8}' | sort -u | head -4
done
unset BYOLO_LIB
(time timeout 1800 python -m pytest tests -m gpu -q -x) > gpurun_out/r3d/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -5 gpurun_out/r3d/gpu_suite.log | cut -c1-300
python bench.py --steps 20 --warmup 5 --dump-steps gpurun_out/r3d/per_launch.md > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err
echo "bench rc=$?"; cut -c1-200 gpurun_out/r3d/bench.json
