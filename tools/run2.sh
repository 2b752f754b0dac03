#!/bin/bash
# round-3 run 2: schedule sweep of the weight-gradient stream (issue order x fill factor), interleaved rounds
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3b; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "$pick" "$label" | tee -a "$OUT/sweep.txt" || exit 1
}
for round in 1 2; do
  run base X=1
  run after_fill1.0_nocap INSAR_WGRAD_ORDER=after INSAR_WGRAD_FILL=1.0 INSAR_WGRAD_FILL_T=1.0 INSAR_WGRAD_GRID_CAP=0
  run after_fill0.7 INSAR_WGRAD_ORDER=after INSAR_WGRAD_FILL=0.7 INSAR_WGRAD_FILL_T=0.7
  run after_fill0.5 INSAR_WGRAD_ORDER=after
  run before_fill1.0_nocap INSAR_WGRAD_FILL=1.0 INSAR_WGRAD_FILL_T=1.0 INSAR_WGRAD_GRID_CAP=0
  run single_stream INSAR_SIDE_STREAM=0
  run main_high_prio INSAR_MAIN_PRIORITY=-1
  run main_high_after_fill1 INSAR_MAIN_PRIORITY=-1 INSAR_WGRAD_ORDER=after INSAR_WGRAD_FILL=1.0 INSAR_WGRAD_FILL_T=1.0 INSAR_WGRAD_GRID_CAP=0
  run main_high_before_fill1 INSAR_MAIN_PRIORITY=-1 INSAR_WGRAD_FILL=1.0 INSAR_WGRAD_FILL_T=1.0 INSAR_WGRAD_GRID_CAP=0
done
echo done
