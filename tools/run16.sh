#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3k; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== tests"; timeout -k 10 600 python3 -m pytest "$R/tests/test_optim_in_backward_gpu.py" "$R/tests/test_graph_gpu.py" -q -p no:cacheprovider 2>&1 | tail -12 | tee "$OUT/pytest.log"
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d["host_enqueue_ms_per_step"])'
run() { local label=$1; shift
  timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs "$@" 2>/dev/null | python3 -c "$pick" "$label" | tee -a "$OUT/ab.txt" || exit 1
}
for round in 1 2 3; do
  run plain --adam-in-backward off
  run fused --adam-in-backward on
done
for round in 1 2; do
  run cfg5_plain --model deeplab --adam-in-backward off
  run cfg5_fused --model deeplab --adam-in-backward on
done
echo done
