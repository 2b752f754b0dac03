set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4a; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest $R/tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 15 $OUT/pytest.log
timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-other-configs > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/pT -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-other-configs > $OUT/pT.log 2>&1 && python3 $R/tools/timeline.py $(ls $OUT/pT/*/*_kernel_trace.csv | head -n 1) | tee $OUT/timeline.txt
cp $(ls $OUT/pT/*/*_kernel_trace.csv | head -n 1) $OUT/kernel_trace.csv; rm -rf $OUT/pT
