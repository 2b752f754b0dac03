set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4e; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest $R/tests/test_bstat_gpu.py $R/tests/test_tape_gpu.py $R/tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_part.log 2>&1; echo "part rc=$?"; tail -n 4 $OUT/pytest_part.log
timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-other-configs > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["gemm_total_alone"], d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["alone"])
for k,v in d["gemm_kernels_alone"].items():
    if "c64" in k or "wgrad3k" in k: print(k, v)
PY
