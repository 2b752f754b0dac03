#!/bin/bash
# same-box comparison of the round-2 tree (git worktree under _r02/, commit 8bafea8) and the current tree
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3p; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
for round in 1 2 3 4; do
  (cd $R/_r02 && timeout -k 10 150 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null) | python3 -c "$pick" round2_tree_8bafea8 | tee -a "$OUT/r02_vs_r03.txt"
  timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" round3_tree | tee -a "$OUT/r02_vs_r03.txt"
done
for round in 1 2; do
  (cd $R/_r02 && timeout -k 10 150 python3 bench.py --model deeplab --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null) | python3 -c "$pick" cfg5_round2_tree | tee -a "$OUT/r02_vs_r03.txt"
  timeout -k 10 150 python3 $R/bench.py --model deeplab --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "$pick" cfg5_round3_tree | tee -a "$OUT/r02_vs_r03.txt"
done
echo done
