#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3g; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== config 5 kernel stats + timeline"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p5" -- python3 "$R/bench.py" --model deeplab --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > "$OUT/p5.log" 2>&1 || { tail "$OUT/p5.log"; exit 1; }
cp $(ls "$OUT"/p5/*/*_kernel_stats.csv | head -n 1) "$OUT/cfg5_kernel_stats.csv"
cp $(ls "$OUT"/p5/*/*_kernel_trace.csv | head -n 1) "$OUT/cfg5_kernel_trace.csv"
python3 "$R/tools/timeline.py" "$OUT/cfg5_kernel_trace.csv" | head -40 | tee "$OUT/cfg5_timeline.txt"
rm -rf "$OUT/p5"
head -30 "$OUT/cfg5_kernel_stats.csv" | cut -c1-160
echo done
