#!/bin/bash
# Launch-by-launch durations inside the step for the default environment and for an INSAR_* setting, same box: rocprofv3
# kernel traces of 8 steps each, listed by tools/timeline.py --list. usage: trace_env_ab.sh "NAME=VALUE[;..]" <outdir> <regex>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$2; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/base" -- python3 "$R/bench.py" --allow-switches --steps 8 --warmup 4 --settle 0 --no-cpu-baseline --no-kernel-timing --no-other-configs > "$OUT/base.log" 2>&1 || exit 1
env ${1//;/ } timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/exp" -- python3 "$R/bench.py" --allow-switches --steps 8 --warmup 4 --settle 0 --no-cpu-baseline --no-kernel-timing --no-other-configs > "$OUT/exp.log" 2>&1 || exit 1
for v in base exp; do cp "$(ls "$OUT"/$v/*/*kernel_trace.csv | head -1)" "$OUT/${v}_trace.csv"; rm -rf "$OUT/$v"; echo "== $v"; python3 "$R/tools/timeline.py" "$OUT/${v}_trace.csv" | head -3; python3 "$R/tools/timeline.py" "$OUT/${v}_trace.csv" --list "$3"; rm -f "$OUT/${v}_trace.csv"; done
