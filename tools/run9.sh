#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3f; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== deeplab tests"; timeout -k 10 900 python3 -m pytest "$R/tests/test_deeplab_gpu.py" "$R/tests/test_deeplab_kernels_gpu.py" -q -s -p no:cacheprovider 2>&1 | grep -E "config-5|gradient rel-L2|convergence|passed|failed|Error|assert" | tee "$OUT/pytest_deeplab.log"
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d["host_enqueue_ms_per_step"])'
for round in 1 2; do
  INSAR_BSTAT_FUSE=0 INSAR_COEF_FUSE=0 timeout -k 10 200 python3 $R/bench.py --model deeplab --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "$pick" cfg5_base | tee -a "$OUT/cfg5.txt"
  timeout -k 10 200 python3 $R/bench.py --model deeplab --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "$pick" cfg5_fused | tee -a "$OUT/cfg5.txt"
done
echo "== parity subset (tightened gates)"; timeout -k 10 600 python3 -m pytest "$R/tests/test_parity_gpu.py" -q -p no:cacheprovider -k "bf16 or given_equal or tile_sizes" 2>&1 | tail -3
echo done
