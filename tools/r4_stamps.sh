set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4f; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
INSAR_HIP_LIB=$R/insar_unet_ca_amd/libinsar_hip_stamps.so timeout -k 10 300 python3 $R/tools/stamp_gemm.py 2>/dev/null | tee $OUT/stamps_gemm.txt
