cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" python bench.py --steps 5 --warmup 2 --pipeline 1 --fp32-steps 0 --no-cpu-baseline --dump-steps gpurun_out/r3e/pl_$tag.md > gpurun_out/r3e/bench_$tag.json 2> gpurun_out/r3e/bench_$tag.err; echo "$tag: $(cut -c52-110 gpurun_out/r3e/bench_$tag.json)"; grep -E "\| (140|-4) \|" gpurun_out/r3e/pl_$tag.md | awk -F'|' '{a[$4" K="$7]+=$8; n[$4" K="$7]++} END {for (k in a) print "   ", k, "launches", n[k], "total ms", a[k]}' | sort; }
run nset4 X=1
run chunk140 BYOLO_WINO_SPLIT_CHUNK_MB=140
run chunk280 BYOLO_WINO_SPLIT_CHUNK_MB=280
run chunk560 BYOLO_WINO_SPLIT_CHUNK_MB=560
run minc128 BYOLO_WINO_SPLIT_MIN_C=128
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "winograd_in_split" 2>&1 | tail -2
