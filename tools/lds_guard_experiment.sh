#!/bin/bash
# Is the LDS store guard of csrc/common.h (LDS_PIN / LDS_DRAIN / LDS_KEEP) still needed now that the K-step barrier WAR race
# is fixed? Build the library WITHOUT the guard next to the shipped one and run the multi-run determinism screens on both
# (GPU box). The guard-free library is a diagnostic build: gpurun_out/libinsar_guard.so, never loaded by the package.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=/tmp/guard_build; rm -rf $B; mkdir -p $B
cd $R/insar_unet_ca_amd/csrc
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$R/include -DINSAR_LDS_GUARD -fno-gpu-rdc -c $f -o $B/${f%.hip}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libinsar_guard.so $B/*.o || exit 1
echo "== shipped library (no guard)"
RUNS=${RUNS:-40} STEPS=24 python3 $R/tools/debug_race_steps.py 2>/dev/null
echo "== with the guard"
INSAR_HIP_LIB=$B/libinsar_guard.so RUNS=${RUNS:-40} STEPS=24 python3 $R/tools/debug_race_steps.py 2>/dev/null
INSAR_HIP_LIB=$B/libinsar_guard.so RUNS=${RUNS:-40} STEPS=24 python3 $R/tools/debug_race_steps.py 2>/dev/null
