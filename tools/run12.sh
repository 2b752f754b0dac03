#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3i; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
run() { local label=$1; shift
  env "$@" timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "$label" | tee -a "$OUT/sweep.txt" || exit 1
}
for round in 1 2; do
  run base X=1
  run fill0.4 INSAR_WGRAD_FILL=0.4
  run fill0.6 INSAR_WGRAD_FILL=0.6
  run cap160 INSAR_WGRAD_GRID_CAP=160
  run cap256 INSAR_WGRAD_GRID_CAP=256
  run fillT0.35 INSAR_WGRAD_FILL_T=0.35
  run fillT0.75 INSAR_WGRAD_FILL_T=0.75
  run persist2 INSAR_FLAT_PERSIST=2
  run persist0 INSAR_FLAT_PERSIST=0
done
echo done
