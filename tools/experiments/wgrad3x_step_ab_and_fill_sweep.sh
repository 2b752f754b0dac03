set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest $R/tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "wgrad_conv3x or unet_golden_fp32" > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?"; grep -E "passed|failed|AssertionError" $OUT/pytest_new.log | cut -c1-700
timeout -k 10 900 python3 -m pytest $R/tests -m gpu -q -p no:cacheprovider --deselect tests/test_parity_gpu.py::test_unet_golden_fp32 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 6 $OUT/pytest.log
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
for i in 1 2 3; do
  timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "x=1 fill=0.6" | tee -a $OUT/ab.txt
  INSAR_WGRAD_X=0 timeout -k 10 150 python3 $R/bench.py --allow-switches --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "x=0 fill=0.6" | tee -a $OUT/ab.txt
  for f in 0.4 0.5 0.7 0.8; do
    timeout -k 10 150 python3 $R/bench.py --wgrad-fill $f --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "x=1 fill=$f" | tee -a $OUT/ab.txt
  done
done
timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-other-configs > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
