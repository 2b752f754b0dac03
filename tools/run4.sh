#!/bin/bash
# round-3 run 4: flat-kernel cross-tile prefetch: tests, microbench, A/B vs the base library
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3d; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== tests"; timeout -k 10 800 python3 -m pytest "$R/tests/test_parity_gpu.py" "$R/tests/test_bstat_gpu.py" -q -p no:cacheprovider -k "flat or bstat or c64 or double_conv or unet_golden_fp32 or unet_bf16 or reproducible or full_size" 2>&1 | tail -8 | tee "$OUT/pytest.log"
echo "== microbench"
for lib in libinsar_hip_base.so libinsar_hip.so; do
  echo "-- $lib"
  INSAR_HIP_LIB=$R/insar_unet_ca_amd/$lib timeout -k 10 200 python3 "$R/tools/gemm_bench.py" --what fpers --only down1.0,down1.3,conv4.0 2>/dev/null | tee -a "$OUT/flat_micro.txt"
done
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
run() { local label=$1; shift
  env "$@" timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "$pick" "$label" | tee -a "$OUT/ab.txt" || exit 1
}
for round in 1 2 3; do
  run base INSAR_HIP_LIB=$R/insar_unet_ca_amd/libinsar_hip_base.so
  run pf_persist1 X=1
  run pf_persist2 INSAR_FLAT_PERSIST=2
done
echo done
