"""Per-kernel SQ counter ratios from a rocprofv3 --pmc counter_collection CSV (one pass, kernels alone on the chip):
wait / active shares of the wave cycles, MFMA-pipe busy share, LDS activity and bank conflicts.
usage: sq_counters.py counter_collection.csv [min_total_wave_cycles]"""
import collections, csv, re, sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    seen = set()
    for r in rows:
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (name, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key); calls[name] += 1
    print("# kernel | launches | wait_any / wave_cycles | wait_inst | active_inst | MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)) |"
          " LDS active / CU busy | bank conflicts / LDS active")
    for name, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = c.get("SQ_WAVE_CYCLES", 0)
        if wc < float(sys.argv[2]) if len(sys.argv) > 2 else wc <= 0:
            continue
        cu = c.get("SQ_BUSY_CU_CYCLES", 0) or 1.0
        lds = c.get("SQ_LDS_IDX_ACTIVE", 0)
        print(f"{name[:58]:58s} n={calls[name]:4d}  wait_any {c.get('SQ_WAIT_ANY', 0) / wc:.2f}  wait_inst {c.get('SQ_WAIT_INST_ANY', 0) / wc:.2f}"
              f"  active {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f}  mfma_busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * cu):.3f}"
              f"  lds_active {lds / cu:.3f}  lds_conflict {(c.get('SQ_LDS_BANK_CONFLICT', 0) / lds) if lds else 0:.3f}")


if __name__ == "__main__":
    main()
