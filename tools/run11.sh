#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3h; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== tests"; timeout -k 10 900 python3 -m pytest "$R/tests/test_parity_gpu.py" "$R/tests/test_deeplab_gpu.py" "$R/tests/test_bstat_gpu.py" -q -p no:cacheprovider -k "coefficient or double_conv or unet_golden or deeplab or bstat or five_adam or split_backward" 2>&1 | tail -5 | tee "$OUT/pytest.log"
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d["host_enqueue_ms_per_step"])'
for round in 1 2 3; do
  INSAR_COEF_SIMPLE=0 timeout -k 10 200 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" unet_ticket | tee -a "$OUT/ab.txt"
  timeout -k 10 200 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" unet_simple | tee -a "$OUT/ab.txt"
done
for round in 1 2; do
  INSAR_COEF_SIMPLE=0 timeout -k 10 200 python3 $R/bench.py --model deeplab --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "$pick" cfg5_ticket | tee -a "$OUT/ab.txt"
  timeout -k 10 200 python3 $R/bench.py --model deeplab --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "$pick" cfg5_simple | tee -a "$OUT/ab.txt"
done
echo done
