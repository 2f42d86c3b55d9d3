cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" python bench.py --steps 10 --warmup 3 --pipeline 1 --fp32-steps 0 --no-cpu-baseline --dump-steps gpurun_out/r3g/pl_$tag.md > gpurun_out/r3g/bench_$tag.json 2> gpurun_out/r3g/bench_$tag.err; echo "$tag: $(cut -c52-110 gpurun_out/r3g/bench_$tag.json)"; tail -1 gpurun_out/r3g/bench_$tag.err | cut -c1-200; grep -E "\| (140|-4|-1) \|" gpurun_out/r3g/pl_$tag.md | awk -F'|' '{a[$4" K="$7]+=$8; n[$4" K="$7]++} END {for (k in a) print "   ", k, "launches", n[k], "total ms", a[k]}' | sort; }
run bn128 X=1
run bn256 BYOLO_WINO_SPLIT_BN=256
run bn128b X=1
run bn256b BYOLO_WINO_SPLIT_BN=256
BYOLO_WINO_SPLIT_BN=256 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "winograd_in_split or forward_vs_golden" 2>&1 | tail -2
