#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3e; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== tests skipped"
echo "== default bench"; SECONDS=0; timeout -k 10 600 python3 "$R/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }; echo "default bench took $SECONDS s"
python3 -c "
import json,sys
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline_step'])
print(json.dumps(d.get('other_configs'), indent=1)[:3000])"
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d.get("input"))'
for round in 1 2 3; do
  timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" resident | tee -a "$OUT/stream.txt"
  timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs --stream-input 2>/dev/null | python3 -c "$pick" streamed | tee -a "$OUT/stream.txt"
done
echo "== dp2 gloo rehearsal"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 "$R/bench.py" --gpus 2 --steps 3 --warmup 1 --backend gloo --no-cpu-baseline --no-kernel-timing 2> "$OUT/dp2.err" | tee "$OUT/dp2.json" | cut -c1-600
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 "$R/bench.py" --gpus 2 --steps 3 --warmup 1 --backend gloo --shard-optimizer --no-cpu-baseline --no-kernel-timing 2> "$OUT/dp2s.err" | tee "$OUT/dp2s.json" | cut -c1-600
echo "== PMC"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pF" -- python3 "$R/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-configs > "$OUT/pF.log" 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pW" -- python3 "$R/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-configs > "$OUT/pW.log" 2>&1 || exit 1
cp $(ls "$OUT"/pF/*/*_counter_collection.csv | head -n 1) "$OUT/pmc_fetch.csv"
cp $(ls "$OUT"/pW/*/*_counter_collection.csv | head -n 1) "$OUT/pmc_write.csv"
python3 "$R/tools/pmc_traffic.py" "$OUT/pmc_fetch.csv" "$OUT/pmc_write.csv" "$OUT/pmc_traffic.json" "timed region (weight gradients split for half the work-group slots, as beside the dgrad chain)" | tee "$OUT/pmc_traffic.txt"
rm -rf "$OUT/pF" "$OUT/pW" "$OUT/pmc_fetch.csv" "$OUT/pmc_write.csv"
echo done
