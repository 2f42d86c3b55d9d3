set -u
export TMPDIR=/tmp
bash tools/profile_round.sh r6_f > gpurun_out/r6_f_round.log 2>&1
bash tools/profile_commit.sh r6_f >> gpurun_out/r6_f_round.log 2>&1
tail -15 gpurun_out/r6_f_round.log
python bench.py > gpurun_out/r6_f_bench_default.json 2> gpurun_out/r6_f_bench_default.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r6_f_bench_default.json
