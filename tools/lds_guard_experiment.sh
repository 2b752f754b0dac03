#!/bin/bash
# Is the LDS store guard of csrc/common.h (LDS_PIN / LDS_DRAIN / LDS_KEEP) still needed now that the K-step barrier WAR race
# is fixed? Build the library WITHOUT the guard next to the shipped one and run the multi-run determinism screens on both
# (GPU box). The guard-free library is a diagnostic build: gpurun_out/libinsar_noguard.so, never loaded by the package.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=/tmp/noguard_build; rm -rf $B; mkdir -p $B
cd $R/insar_unet_ca_amd/csrc
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$R/include -DINSAR_NO_LDS_GUARD -fno-gpu-rdc -c $f -o $B/${f%.hip}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libinsar_noguard.so $B/*.o || exit 1
echo "== with the guard (shipped library)"
RUNS=${RUNS:-40} STEPS=24 python3 $R/tools/debug_race_steps.py 2>/dev/null
echo "== without the guard"
INSAR_HIP_LIB=$B/libinsar_noguard.so RUNS=${RUNS:-40} STEPS=24 python3 $R/tools/debug_race_steps.py 2>/dev/null
INSAR_HIP_LIB=$B/libinsar_noguard.so RUNS=${RUNS:-40} STEPS=24 python3 $R/tools/debug_race_steps.py 2>/dev/null
