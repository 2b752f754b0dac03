#!/bin/bash
# The driver's command (--steps 20 --warmup 5) with and without the settling steps, interleaved, starting on a fresh box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
for i in 1 2 3; do
for st in 0 100; do
  timeout -k 10 200 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --settle $st --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "settle=$st"
done; done
