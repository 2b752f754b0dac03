"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into HBM traffic figures:
  * per kernel (EVERY kernel of the step, not only the GEMM-class ones): average bytes per launch
    = 2*FETCH_SIZE*1024 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md §HBM) + WRITE_SIZE*1024;
  * per STEP: the sum over all dispatches between two consecutive Adam launches (steady-state steps only: plan
    construction and its zero fills are before the first one), averaged over those steps -> key "__step__",
    which bench.py reports as roofline_step.traffic next to the algorithmic bytes.
usage: pmc_traffic.py fetch.csv write.csv out.json [launch-configuration note]"""
import collections, csv, json, re, sys


def load(path, counter):
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        rows.append((int(r["Dispatch_Id"]), name, float(r["Counter_Value"])))
    rows.sort()
    return rows


def per_kernel(rows):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for _, name, v in rows:
        acc[name][0] += v
        acc[name][1] += 1
    return acc


def per_step(rows):
    """(KiB per steady-state step, steps, launches per step): dispatches after the first adam_kernel up to the last."""
    adam = [i for i, (_, n, _) in enumerate(rows) if n.startswith("adam_kernel")]
    if len(adam) < 2:
        return None
    seg = rows[adam[0] + 1:adam[-1] + 1]
    steps = len(adam) - 1
    return sum(v for _, _, v in seg) / steps, steps, len(seg) / steps


def main():
    f, w, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    fr, wr = load(f, "FETCH_SIZE"), load(w, "WRITE_SIZE")
    fa, wa = per_kernel(fr), per_kernel(wr)
    res = {}
    for k in sorted(set(fa) | set(wa)):
        fk = fa[k][0] / max(fa[k][1], 1); wk = wa[k][0] / max(wa[k][1], 1)
        res[k] = {"launches_profiled": fa[k][1], "FETCH_SIZE_KB_avg": round(fk, 1), "WRITE_SIZE_KB_avg": round(wk, 1),
                  "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024)}
    fs, ws = per_step(fr), per_step(wr)
    if fs and ws:
        res["__step__"] = {"hbm_bytes_per_step": int(2 * fs[0] * 1024 + ws[0] * 1024), "fetch_bytes_per_step": int(2 * fs[0] * 1024),
                           "write_bytes_per_step": int(ws[0] * 1024), "steady_steps_profiled": min(fs[1], ws[1]),
                           "launches_per_step": round(fs[2], 1),
                           "note": "sum over every dispatch between consecutive Adam launches; 2*FETCH_SIZE (wide-read correction) + "
                                   "WRITE_SIZE, KiB -> bytes; memory-side requests, Infinity-Cache hits included"}
    res["__config__"] = {"launch_configuration": note or "unspecified"}
    json.dump(res, open(out, "w"), indent=1)
    top = sorted(((v["hbm_bytes_per_launch"] * v["launches_profiled"], k) for k, v in res.items() if not k.startswith("__")), reverse=True)
    for _, k in top[:25]:
        v = res[k]
        print(f"{k[:70]:70s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  (n={v['launches_profiled']})")
    if "__step__" in res:
        s = res["__step__"]
        print(f"step: {s['hbm_bytes_per_step'] / 1e9:.2f} GB (read {s['fetch_bytes_per_step'] / 1e9:.2f} + write {s['write_bytes_per_step'] / 1e9:.2f}), "
              f"{s['launches_per_step']} launches, {s['steady_steps_profiled']} steady steps")


if __name__ == "__main__":
    main()
