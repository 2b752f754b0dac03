#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3q; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
run() { local label=$1; shift
  env "$@" timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "$label" | tee -a "$OUT/cmp.txt"
}
for round in 1 2 3; do
  (cd $R/_r02 && timeout -k 10 150 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null) | python3 -c "$pick" r02_tree | tee -a "$OUT/cmp.txt"
  run r03_all_off INSAR_BSTAT_FUSE=0 INSAR_FLAT_PERSIST=1 INSAR_COEF_SIMPLE=0 INSAR_TAPE=0
  run r03_all_off_tape_on INSAR_BSTAT_FUSE=0 INSAR_FLAT_PERSIST=1 INSAR_COEF_SIMPLE=0
  run r03_default X=1
  run r03_default_tape_off INSAR_TAPE=0
done
echo done
