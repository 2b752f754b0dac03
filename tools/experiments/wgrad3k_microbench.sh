set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4d; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest $R/tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k "wgrad_conv3k" 2>&1 | tail -n 5
timeout -k 10 500 python3 $R/tools/gemm_bench.py --what k3 --only inc.3,conv4.0,down1.0,down1.3 2>&1 | tee $OUT/k3.txt
