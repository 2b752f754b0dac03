#!/bin/bash
# Same-box A/B of several builds of the library: `rounds` interleaved bench runs (timed region only) of the product library
# and of every experiment library named. usage: [STEPS=40 WARMUP=10] ab_multi.sh <rounds> <exp1.so> [exp2.so ...]   (paths relative to the repo root)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=$1; shift
STEPS=${STEPS:-40}; WARMUP=${WARMUP:-10}
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
for i in $(seq $N); do
  timeout -k 10 150 python3 $R/bench.py --allow-switches --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" base || exit 1
  for L in "$@"; do
    INSAR_HIP_LIB=$R/$L timeout -k 10 150 python3 $R/bench.py --allow-switches --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "$(basename $L)" || exit 1
  done
done
