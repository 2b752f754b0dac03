#!/bin/bash
# Per-kernel durations of two builds on one box: rocprofv3 kernel stats of the timed region (two streams) for the product
# library and for an experiment library. usage: prof_ab.sh <exp.so> <outdir under gpurun_out>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$2; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/base" -- python3 "$R/bench.py" --allow-switches --steps 20 --warmup 5 --settle 0 --no-cpu-baseline --no-kernel-timing --no-other-configs > "$OUT/base.log" 2>&1 || exit 1
INSAR_HIP_LIB=$R/$1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/exp" -- python3 "$R/bench.py" --allow-switches --steps 20 --warmup 5 --settle 0 --no-cpu-baseline --no-kernel-timing --no-other-configs > "$OUT/exp.log" 2>&1 || exit 1
for v in base exp; do cp "$(ls "$OUT"/$v/*/*kernel_stats.csv | head -1)" "$OUT/${v}_kernel_stats.csv"; rm -rf "$OUT/$v"; done
python3 - "$OUT" <<'P'
import csv, sys
o = sys.argv[1]
def load(p):
    return {r["Name"].replace("void ", "").split("(")[0]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3) for r in csv.DictReader(open(p))}
b, e = load(o + "/base_kernel_stats.csv"), load(o + "/exp_kernel_stats.csv")
rows = []
for k in sorted(set(b) | set(e)):
    cb, ab, tb = b.get(k, (0, 0, 0)); ce, ae, te = e.get(k, (0, 0, 0))
    rows.append((te - tb, k, cb, ab, ce, ae))
print("kernel".ljust(66) + "base calls   avg us   exp calls   avg us   d(total)/step us  (exp - base, 25 steps profiled)")
for d, k, cb, ab, ce, ae in sorted(rows):
    if abs(d) / 25 > 0.5: print(f"{k[:64]:66s}{cb:6d} {ab:9.1f} {ce:10d} {ae:9.1f} {d / 25:12.1f}")
P
