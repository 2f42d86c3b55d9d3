set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6_d; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "launch_graph or persistent_unit" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/pytest_new.log
for v in 0 1 2; do env BYOLO_WINO_SPLIT_PERSIST=$v timeout 300 python tools/rows_digest.py > $O/dig_$v.txt 2> $O/dig_$v.err; done
cmp $O/dig_0.txt $O/dig_1.txt && cmp $O/dig_0.txt $O/dig_2.txt && echo "digests identical (persist 0 / 1 / 2)"
for i in 1 2 3; do for v in 0 2 1; do
  env BYOLO_WINO_SPLIT_PERSIST=$v timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-configs --fp32-steps 0 --entry-frames 0 --dump-steps $O/steps_${v}_$i.md > $O/line_${v}_$i.json 2> $O/err_${v}_$i.txt
  python - "persist=$v run $i" $O/line_${v}_$i.json $O/steps_${v}_$i.md <<'PY'
import sys, json
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
w = [float(l.split("|")[7]) for l in open(sys.argv[3]) if "| 140 |" in l]
print("%-18s %.1f img/s  %.3f ms/step   variant 140: %d launches %.4f ms (19x19 %.4f, 38x38 %.4f)" % (sys.argv[1], d["value"], d["ms_per_step"], len(w), sum(w), sum(w[:3]), sum(w[3:])))
PY
done; done
python tools/host_at_8_ranks.py --out $O/host_at_8_ranks.md > $O/host_at_8_ranks.json 2> $O/host_at_8_ranks.err; tail -3 $O/host_at_8_ranks.err; cat $O/host_at_8_ranks.md
