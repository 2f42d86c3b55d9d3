set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6_h; mkdir -p $O
rm -f gpurun_out/parity_table.json
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
cp gpurun_out/parity_table.json $O/parity_table.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/profile_round.sh r6_h > $O/round.log 2>&1
bash tools/profile_commit.sh r6_h >> $O/round.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 300 $O/bench_default.json
