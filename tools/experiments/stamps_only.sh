set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
INSAR_HIP_LIB=$R/insar_unet_ca_amd/libinsar_hip_stamps.so timeout -k 10 300 python3 $R/tools/stamp_gemm.py 2>&1 | tee $OUT/stamps_gemm.txt | tail -n 3
INSAR_HIP_LIB=$R/insar_unet_ca_amd/libinsar_hip_stamps.so timeout -k 10 200 python3 $R/tools/stamp_flat.py 2>&1 | tee $OUT/stamps_flat.txt | tail -n 3
