#!/bin/bash
# ceiling_probe.sh OUTDIR -- the evidence behind DESIGN.md's "bound / measured / gap" table:
#   1. the SUSTAINED v_mfma_f32_32x32x16_f16 rate under the socket power cap (tools/mfma_f16_power_probe.hip: an MFMA-only loop,
#      constant and random operands, ~1 s each) with the shader clock and the socket power sampled beside it (rocm-smi, 5 Hz);
#   2. the same two readings while bench.py runs the real instruction mix of BASELINE configs[3];
#   3. the achievable HBM rate (tools/hbm_probe.py).
#   gpurun -- 'bash tools/ceiling_probe.sh gpurun_out/r6_e'
set -u
OUT=${1:-gpurun_out/ceiling}; mkdir -p "$OUT"
mkdir -p tools/_build
[ -x tools/_build/mfma_f16_power_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/mfma_f16_power_probe.hip -o tools/_build/mfma_f16_power_probe
sample() {   # sample NAME PID: clock + power until PID ends
    : > "$OUT/$1_smi.txt"
    while kill -0 $2 2> /dev/null; do
        rocm-smi --showclocks --showpower 2> /dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> "$OUT/$1_smi.txt"; echo >> "$OUT/$1_smi.txt"
        sleep 0.2
    done
}
tools/_build/mfma_f16_power_probe > "$OUT/mfma_f16_power_probe.txt" 2>&1 &
P=$!; sample probe $P; wait $P
python bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-other-configs --fp32-steps 0 --entry-frames 0 > "$OUT/bench_line.json" 2> "$OUT/bench.err" &
P=$!; sample bench $P; wait $P
python tools/hbm_probe.py > "$OUT/hbm_probe.txt" 2>&1
cat "$OUT/mfma_f16_power_probe.txt"
for n in probe bench; do echo "== $n: most frequent (clock, power) readings"; sed 's/  */ /g' "$OUT/${n}_smi.txt" | sort | uniq -c | sort -rn | head -6; done
tail -5 "$OUT/hbm_probe.txt"
