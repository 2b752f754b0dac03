#!/bin/bash
# round-3 run 3: bstat fusion tests + A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3c; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== tests"; timeout -k 10 600 python3 -m pytest "$R/tests/test_bstat_gpu.py" -q -p no:cacheprovider 2>&1 | tail -15 | tee "$OUT/pytest_bstat.log"
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
run() { local label=$1; shift
  env "$@" timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "$pick" "$label" | tee -a "$OUT/ab.txt" || exit 1
}
for round in 1 2 3; do
  run bstat_off INSAR_BSTAT_FUSE=0
  run bstat_on INSAR_BSTAT_FUSE=1
done
echo done
