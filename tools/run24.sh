#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3o; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
run() { local label=$1; shift
  env "$@" timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>"$OUT/err.txt" | python3 -c "$pick" "$label" | tee -a "$OUT/sweep.txt" || { tail -5 "$OUT/err.txt"; echo "$label FAILED" | tee -a "$OUT/sweep.txt"; }
}
for round in 1 2; do
  run base X=1
  run mask128 INSAR_SIDE_CU_MASK=128
  run mask128_fill1 INSAR_SIDE_CU_MASK=128 INSAR_WGRAD_FILL=1.0 INSAR_WGRAD_FILL_T=1.0 INSAR_WGRAD_GRID_CAP=0
  run mask128s2 INSAR_SIDE_CU_MASK=128:2
  run mask96 INSAR_SIDE_CU_MASK=96
  run mask160 INSAR_SIDE_CU_MASK=160
  run mask64_fill1 INSAR_SIDE_CU_MASK=64 INSAR_WGRAD_FILL=1.0 INSAR_WGRAD_GRID_CAP=0
done
echo done
