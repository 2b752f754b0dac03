#!/bin/bash
# Same-box A/B of environment settings: `rounds` interleaved bench runs of the default and of every "NAME=VALUE[,NAME=VALUE]"
# setting named. usage: [STEPS=30 WARMUP=8] ab_env.sh <rounds> "<bench args>" <env1> [env2 ...]     e.g. ab_env.sh 3 "--model deeplab" INSAR_TUNE=wgrad_tile_max=128
# (30-step runs scatter by +-0.04 ms on one box; STEPS=150 WARMUP=20 resolves 0.03 ms)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=$1; ARGS=$2; shift 2
STEPS=${STEPS:-30}; WARMUP=${WARMUP:-8}
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
for i in $(seq $N); do
  timeout -k 10 200 python3 $R/bench.py --allow-switches $ARGS --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" default || exit 1
  for E in "$@"; do
    env ${E//;/ } timeout -k 10 200 python3 $R/bench.py --allow-switches $ARGS --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "$E" || exit 1
  done
done
