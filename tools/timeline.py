"""Timeline of the last full training step in a rocprofv3 --kernel-trace CSV of bench.py: wall span, busy time of the
main and the weight-gradient queue, their overlap, GPU idle time, and the kernel-time sums of the forward pass and of
the rest of the step. usage: timeline.py kernel_trace.csv [--list REGEX]
--list: every launch of the step whose kernel name matches, in launch order: queue, start within the step, duration averaged
over the complete steps of the trace, and how much of it overlapped a kernel of the other queue (last step)."""
import collections, csv, re, sys


def union(iv):
    iv = sorted(iv)
    if not iv:
        return 0
    tot, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["n"] = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    rows.sort(key=lambda r: r["s"])
    adam = [i for i, r in enumerate(rows) if r["n"].startswith("adam_kernel")]
    step = rows[adam[-2] + 1:adam[-1] + 1]
    t0, t1 = step[0]["s"], step[-1]["e"]
    qs = collections.Counter(r["Queue_Id"] for r in step)
    mainq = qs.most_common(1)[0][0]
    m = [(r["s"], r["e"]) for r in step if r["Queue_Id"] == mainq]
    o = [(r["s"], r["e"]) for r in step if r["Queue_Id"] != mainq]
    busy = union(m + o)
    loss = [r for r in step if r["n"].startswith("dicece") or r["n"].startswith("ce_")]
    tf = loss[0]["s"] if loss else t0
    fw = [r for r in step if r["e"] <= tf]
    bw = [r for r in step if r["s"] >= tf]
    print(f"last step (under the profiler): {len(step)} kernels, wall {1e-6 * (t1 - t0):.3f} ms; forward {1e-6 * (tf - t0):.3f} ms, "
          f"loss + backward + Adam {1e-6 * (t1 - tf):.3f} ms")
    print(f"main queue busy {1e-6 * union(m):.3f} ms ({len(m)} kernels), side queue busy {1e-6 * union(o):.3f} ms ({len(o)} kernels), "
          f"both busy {1e-6 * (union(m) + union(o) - busy):.3f} ms, GPU idle {1e-6 * (t1 - t0 - busy):.3f} ms")
    print(f"sum of kernel durations: forward {1e-6 * sum(r['e'] - r['s'] for r in fw):.3f} ms, rest {1e-6 * sum(r['e'] - r['s'] for r in bw):.3f} ms")
    # kernels of the step that are not this library's (ATen element-wise kernels, runtime copies / fills): should be none
    fw_k = collections.Counter()
    fw_t = 0
    for r in step:
        if "at::" in r["n"] or "rocclr" in r["n"] or r["n"].startswith("__amd"):
            m_ = re.search(r"(\w+Functor)", r["n"])
            fw_k[m_.group(1) if m_ else r["n"][:40]] += 1
            fw_t += r["e"] - r["s"]
    print(f"framework kernels in the step (at::native / __amd_rocclr): {sum(fw_k.values())} launches, {1e-3 * fw_t:.1f} us"
          + (": " + ", ".join(f"{k} x{v}" for k, v in fw_k.most_common()) if fw_k else ""))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in step:
        agg[(r["n"][:64], "side" if r["Queue_Id"] != mainq else "main")][0] += r["e"] - r["s"]
        agg[(r["n"][:64], "side" if r["Queue_Id"] != mainq else "main")][1] += 1
    for (k, q), v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
        print(f"  {k:64s} {q:4s} {1e-3 * v[0]:8.1f} us  n={v[1]}")


def listing(pattern):
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["n"] = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    rows.sort(key=lambda r: r["s"])
    adam = [i for i, r in enumerate(rows) if r["n"].startswith("adam_kernel")]
    steps = [rows[adam[i] + 1:adam[i + 1] + 1] for i in range(len(adam) - 1)]
    steps = [st for st in steps if len(st) == len(steps[-1])]
    last = steps[-1]
    mainq = collections.Counter(r["Queue_Id"] for r in last).most_common(1)[0][0]
    rx = re.compile(pattern)
    for i, r in enumerate(last):
        if not rx.search(r["n"]):
            continue
        durs = [st[i]["e"] - st[i]["s"] for st in steps if st[i]["n"] == r["n"]]
        other = [(o["s"], o["e"]) for o in last if (o["Queue_Id"] == mainq) != (r["Queue_Id"] == mainq)]
        ov = sum(max(0, min(r["e"], e) - max(r["s"], s)) for s, e in other)
        print(f"  {i:4d} {'main' if r['Queue_Id'] == mainq else 'side'} +{1e-3 * (r['s'] - last[0]['s']):8.1f} us  "
              f"{1e-3 * sum(durs) / len(durs):7.1f} us (n={len(durs)})  beside the other queue {100.0 * ov / max(1, r['e'] - r['s']):3.0f} %  {r['n'][:70]}")


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "--list":
        listing(sys.argv[3])
        sys.exit(0)
    main()
