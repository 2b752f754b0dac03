set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4d; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
for i in 1 2 3; do
  for cfg in "INSAR_WGRAD_K=0" "INSAR_WGRAD_K=1" "INSAR_WGRAD_K=1 INSAR_WGRAD_K_TILES=128x64,64x128,128x128,64x64" "INSAR_WGRAD_K=1 INSAR_WGRAD_K_TILES=128x128" "INSAR_WGRAD_K=1 INSAR_WGRAD_K_TILES=128x64,128x128" "INSAR_WGRAD_K=1 INSAR_WGRAD_K_TILES=128x64"; do
    env $cfg timeout -k 10 150 python3 $R/bench.py --allow-switches --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "$cfg" | tee -a $OUT/k3ab.txt
  done
done
