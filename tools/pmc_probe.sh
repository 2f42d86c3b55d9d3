#!/bin/bash
# pmc_probe.sh TAG [bench args] -- counter passes over ONE forward of the bench configuration, summed per kernel name:
# where a kernel's wave-cycles go (SQ), what the LDS / load path / L2 report.  Output: gpurun_out/pmc_TAG/summary.md
#   gpurun --timeout 1500 -- 'BYOLO_PRECISION=split bash tools/pmc_probe.sh kx3'
set -u
TAG=${1:-probe}; shift || true
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ONE="python $PWD/bench.py --steps 1 --warmup 1 --no-profile --no-cpu-baseline $*"
cd /tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS" ; do      # (the TCP_* / TCC_* sets abort rocprofv3 on this image)
    i=$((i + 1))
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o pmc -- $ONE > /dev/null 2> "$OUT/p$i.err"
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:90]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add((k, r["Dispatch_Id"]))
    for k, _ in seen: cnt[k] = max(cnt[k], sum(1 for kk, _ in seen if kk == k))
with open(os.path.join(out, "summary.md"), "w") as f:
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
        f.write("## %s  (%d dispatches)\n" % (k, cnt[k]))
        for c in sorted(acc[k]): f.write("%-40s %.4g\n" % (c, acc[k][c]))
        f.write("\n")
print(open(os.path.join(out, "summary.md")).read()[:6000])
PY
find "$OUT" -name "*.db" -delete
