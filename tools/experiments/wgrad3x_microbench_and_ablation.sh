set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4b; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 python3 $R/tools/gemm_bench.py --what x3 --only down2.3,conv2.0,down3.0,down3.3,conv1.0,down4.0,down4.3,conv3.0,down2.0 2>&1 | tee $OUT/x3_v2.txt
INSAR_HIP_LIB=$R/insar_unet_ca_amd/libinsar_hip_wx.so timeout -k 10 500 python3 $R/tools/gemm_bench.py --what x3var --only conv2.0,down3.3,down4.3 2>&1 | tee $OUT/x3var_v2.txt
