cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" python bench.py --steps 10 --warmup 3 --pipeline 1 --fp32-steps 0 --no-cpu-baseline --dump-steps gpurun_out/r3f/pl_$tag.md > gpurun_out/r3f/bench_$tag.json 2> gpurun_out/r3f/bench_$tag.err; echo "$tag: $(cut -c52-110 gpurun_out/r3f/bench_$tag.json)"; tail -2 gpurun_out/r3f/bench_$tag.err | cut -c1-200; }
run p1 X=1
run p0 BYOLO_P1=0
run p1b X=1
python - <<'PY'
import re
def load(f):
    rows=[]
    for l in open(f):
        c=[x.strip() for x in l.split('|')]
        if len(c)>8 and c[1].isdigit(): rows.append((int(c[2]),int(c[3]),int(c[4]),int(c[5]),int(c[6]),float(c[7])))
    return rows
a=load('gpurun_out/r3f/pl_p1.md'); b=load('gpurun_out/r3f/pl_p0.md')
tot={}
for tag,rows in (('p1',a),('p0',b)):
    for (layer,var,M,N,K,ms) in rows:
        tot.setdefault(tag,{}).setdefault(var,0.0); tot[tag][var]+=ms
print(tot)
bl={ (r[0],r[2],r[4]):r for r in b}
for r in a:
    if r[1] in (2128,2064):
        o=bl.get((r[0],r[2],r[4]))
        print("layer %3d M=%8d N=%4d K=%5d  p1 %.4f ms  old %.4f ms" % (r[0],r[2],r[3],r[4],r[5],o[5] if o else -1))
PY
(time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layers.py -m gpu -q -x) > gpurun_out/r3f/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r3f/tests.log | cut -c1-300
