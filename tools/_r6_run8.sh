set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6_g; mkdir -p $O
python tools/rows_digest.py > $O/dig_new.txt 2>/dev/null; (cd _ab_old && python tools/rows_digest.py > $O/dig_old.txt 2>/dev/null); cmp $O/dig_new.txt $O/dig_old.txt && echo "digests identical to the round-5 tree"
for i in 1 2 3; do for v in 0 1; do
  env BYOLO_SERIALIZE_HEADS=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --fp32-steps 0 --entry-frames 0 --dump-steps $O/steps_${v}_$i.md > $O/line_${v}_$i.json 2> $O/err_${v}_$i.txt
  python - "serialize_heads=$v run $i" $O/line_${v}_$i.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-26s %.1f img/s  %.3f ms/step  roofline: %s frac %.4f avg launch %.4f ms (%s)" % (sys.argv[1], d["value"], d["ms_per_step"], r["kernel"].split(" ")[0], r["frac"], r["avg_launch_ms"], r["profiled_steps"][:8]))
PY
done; done
python -m pytest tests/test_entry_points.py tests/test_gpu_parity.py -x -q -m gpu -k "entry or concurrent or first_image or launch_graph or pipelin" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
