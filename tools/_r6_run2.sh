set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6_b; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "launch_graph or persistent_unit or winograd_in_split or drop_prob_zero or dropout_quirk or t_shard or tshard" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/pytest_new.log
(BYOLO_LIB=$PWD/_ab_old/bayesian-yolov3_amd/byolo/libbyolo.so; cd _ab_old && python tools/rows_digest.py > $O/digest_old.txt 2> $O/digest_old.err)
python tools/rows_digest.py > $O/digest_new.txt 2> $O/digest_new.err
diff $O/digest_old.txt $O/digest_new.txt > $O/digest_diff.txt && echo "digests identical" | tee -a $O/digest_diff.txt
bash tools/ab_trees.sh _ab_old 3 > $O/ab_trees.log 2>&1
cp -r gpurun_out/ab_trees $O/ 2>/dev/null
for c in 1 2; do
  python tools/small_cfg_profile.py --cfg $c --steps 300 --no-table > $O/small_cfg${c}_graph.json 2> $O/small_cfg${c}_graph.err
done
python -m pytest tests/test_gpu_bench_shapes.py -x -q -m gpu -k "config4_as_benched and split or reference_default_batched" > $O/pytest_shapes.log 2>&1; echo "pytest shapes rc=$?" | tee -a $O/pytest_shapes.log
tail -3 $O/pytest_new.log; cat $O/digest_diff.txt | head; cat $O/ab_trees.log; cat $O/small_cfg1_graph.json $O/small_cfg2_graph.json; tail -5 $O/pytest_shapes.log
