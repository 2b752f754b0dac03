#!/bin/bash
# round-3 run 1: wgrad3 32x32x16 A/B, baseline bench, hipGraph vs eager traces
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3a; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== m32 A/B"; timeout -k 10 300 python3 "$R/tools/gemm_bench.py" --what knob-wgrad --knob wgrad3_m32=0,1 --only down1.3,down2.3,down3.3,down4.3,conv1.0,conv2.0,conv3.0 2>&1 | tee "$OUT/m32_ab.txt" || exit 1
echo "== bench"; timeout -k 10 400 python3 "$R/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
cut -c1-300 "$OUT/bench.json"
echo "== bench m32"; INSAR_TUNE=wgrad3_m32=1 timeout -k 10 400 python3 "$R/bench.py" --no-cpu-baseline > "$OUT/bench_m32.json" 2> "$OUT/bench_m32.err" || exit 1
cut -c1-300 "$OUT/bench_m32.json"
echo "== trace eager"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/pT" -- python3 "$R/bench.py" --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timing > "$OUT/pT.log" 2>&1 || exit 1
cp $(ls "$OUT"/pT/*/*_kernel_trace.csv | head -n 1) "$OUT/kernel_trace_eager.csv"
python3 "$R/tools/timeline.py" "$OUT/kernel_trace_eager.csv" | tee "$OUT/timeline_eager.txt"
echo "== trace graph"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/pG" -- python3 "$R/bench.py" --graph on --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timing > "$OUT/pG.log" 2>&1 || exit 1
cp $(ls "$OUT"/pG/*/*_kernel_trace.csv | head -n 1) "$OUT/kernel_trace_graph.csv"
python3 "$R/tools/timeline.py" "$OUT/kernel_trace_graph.csv" | tee "$OUT/timeline_graph.txt"
rm -rf "$OUT/pT" "$OUT/pG"
echo done
