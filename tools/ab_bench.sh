#!/bin/bash
# A/B of two builds of libinsar_hip.so on ONE box: alternating bench runs (timed region only), then one run each
# with the per-kernel event pass. usage: ab_bench.sh <exp.so> [rounds]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
EXP=$R/$1; N=${2:-3}
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], " ".join("%s=%.1f" % (k.replace("_kernel","").replace("bf16_t, ",""), v["avg_us"]) for k, v in d.get("gemm_kernels", {}).items()))'
for i in $(seq $N); do
  timeout -k 10 120 python3 $R/bench.py --allow-switches --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "$pick" base || exit 1
  INSAR_HIP_LIB=$EXP timeout -k 10 120 python3 $R/bench.py --allow-switches --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "$pick" exp || exit 1
done
timeout -k 10 200 python3 $R/bench.py --allow-switches --no-cpu-baseline 2>/dev/null | python3 -c "$pick" base || exit 1
INSAR_HIP_LIB=$EXP timeout -k 10 200 python3 $R/bench.py --allow-switches --no-cpu-baseline 2>/dev/null | python3 -c "$pick" exp
