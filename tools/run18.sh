#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3l; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== tests"; timeout -k 10 600 python3 -m pytest "$R/tests/test_tape_gpu.py" -x -q -p no:cacheprovider 2>&1 | tail -25 | tee "$OUT/pytest.log"
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d["host_enqueue_ms_per_step"])'
run() { local label=$1; shift
  env "$@" timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "$label" | tee -a "$OUT/ab.txt" || exit 1
}
run5() { local label=$1; shift
  env "$@" timeout -k 10 150 python3 $R/bench.py --model deeplab --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "$pick" "$label" | tee -a "$OUT/ab.txt" || exit 1
}
for round in 1 2; do
  run tape_off INSAR_TAPE=0
  run tape_on INSAR_TAPE=1
  run5 cfg5_tape_off INSAR_TAPE=0
  run5 cfg5_tape_on INSAR_TAPE=1
done
echo done
