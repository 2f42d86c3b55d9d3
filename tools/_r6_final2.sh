set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6_i; mkdir -p $O
python -m pytest tests/test_dist_cpu.py tests/test_gpu_layers.py tests/test_entry_points.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity_ok'], {k[:20]:round(v['img_s'],1) for k,v in d['other_configs'].items()})"
