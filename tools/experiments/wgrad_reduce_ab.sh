set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4h; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest $R/tests/test_parity_gpu.py $R/tests/test_deeplab_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "wgrad or igemm_family or conv_transpose or deeplab" 2>&1 | tail -n 3
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
for i in 1 2 3; do
  INSAR_HIP_LIB=$R/insar_unet_ca_amd/libinsar_hip_base.so timeout -k 10 150 python3 $R/bench.py --allow-switches --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "old-reduce" | tee -a $OUT/ab.txt
  timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "vec-reduce" | tee -a $OUT/ab.txt
  INSAR_WGRAD_K=0 timeout -k 10 150 python3 $R/bench.py --allow-switches --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "vec-reduce K=0" | tee -a $OUT/ab.txt
done
