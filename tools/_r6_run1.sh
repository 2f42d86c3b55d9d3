set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6_a; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for c in 1 2 3; do
  python tools/small_cfg_profile.py --cfg $c --steps 200 --md $O/small_cfg${c}_per_launch.md > $O/small_cfg${c}.json 2> $O/small_cfg${c}.err
done
cd /tmp
for c in 1 2 3; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_cfg$c -o small -- python $GRAFT_REPO_ROOT/tools/small_cfg_profile.py --cfg $c --steps 50 --no-table --graph 0 > $O/stats_cfg$c.json 2> $O/stats_cfg$c.err
done
find $O -name "*.db" -size +20M -delete
ls -R $O | head -50
tail -c 600 $O/small_cfg1.json $O/small_cfg2.json $O/small_cfg3.json
