#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3n; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== tests"; timeout -k 10 900 python3 -m pytest "$R/tests/test_deeplab_kernels_gpu.py" "$R/tests/test_deeplab_gpu.py" -q -p no:cacheprovider 2>&1 | tail -4 | tee "$OUT/pytest.log"
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
for round in 1 2 3; do
  INSAR_HIP_LIB=$R/insar_unet_ca_amd/libinsar_hip_base.so timeout -k 10 200 python3 $R/bench.py --model deeplab --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "$pick" cfg5_base | tee -a "$OUT/ab.txt"
  timeout -k 10 200 python3 $R/bench.py --model deeplab --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "$pick" cfg5_skip_dead_taps | tee -a "$OUT/ab.txt"
done
echo done
