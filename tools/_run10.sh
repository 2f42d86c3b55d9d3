cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" python bench.py --steps 10 --warmup 3 --pipeline 1 --fp32-steps 0 --no-cpu-baseline --dump-steps gpurun_out/r3i/pl_$tag.md > gpurun_out/r3i/bench_$tag.json 2> gpurun_out/r3i/bench_$tag.err; echo "$tag: $(cut -c52-110 gpurun_out/r3i/bench_$tag.json)"; tail -1 gpurun_out/r3i/bench_$tag.err | cut -c1-200; grep -E "\| (140|-4) \|" gpurun_out/r3i/pl_$tag.md | awk -F'|' '{a[$4" K="$7]+=$8; n[$4" K="$7]++} END {for (k in a) print "   ", k, "launches", n[k], "total ms", a[k]}' | sort; }
run base X=1
run r1 BYOLO_WINO_SPLIT_ROUNDS=1
run r2 BYOLO_WINO_SPLIT_ROUNDS=2
run r3 BYOLO_WINO_SPLIT_ROUNDS=3
run base2 X=1
