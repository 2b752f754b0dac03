"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into per-kernel average HBM
traffic per launch: bytes = 2*FETCH_SIZE*1024 (gfx950 reports half of wide coalesced reads,
MI355X_MICROARCH.md §HBM) + WRITE_SIZE*1024. usage: pmc_traffic.py fetch.csv write.csv out.json"""
import collections, csv, json, re, sys

def load(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
    return acc

def main():
    f, w, out = sys.argv[1:4]
    fa, wa = load(f, "FETCH_SIZE"), load(w, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fa) | set(wa)):
        if not any(t in k for t in ("igemm_kernel", "wgrad_kernel", "wgrad3_kernel", "conv3x3_flat_kernel", "conv3x3_c64_kernel")):
            continue
        fk = fa[k][0] / max(fa[k][1], 1); wk = wa[k][0] / max(wa[k][1], 1)
        res[k] = {"launches_profiled": fa[k][1], "FETCH_SIZE_KB_avg": round(fk, 1), "WRITE_SIZE_KB_avg": round(wk, 1),
                  "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024),
                  "note": "2*FETCH_SIZE (wide-read correction) + WRITE_SIZE, KiB -> bytes"}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:50s} {v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch  (n={v['launches_profiled']})")

if __name__ == "__main__":
    main()
