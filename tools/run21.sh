#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3m; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
run() { local label=$1; shift
  env "$@" timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "$label" | tee -a "$OUT/sweep.txt" || exit 1
}
for round in 1 2 3; do
  run base X=1
  run wide128 INSAR_TUNE=igemm_wide_min=128
  run wide64 INSAR_TUNE=igemm_wide_min=64
done
echo "== microbench 16x16 level"
for v in 0 128; do INSAR_TUNE=igemm_wide_min=$v timeout -k 10 200 python3 $R/tools/gemm_bench.py --only down4.0,down4.3 --what fwd,dgrad 2>/dev/null | sed "s/^/wide_min=$v /" | tee -a "$OUT/micro.txt"; done
echo done
