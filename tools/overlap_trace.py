"""Dev tool: what ran beside what.  Reads a rocprofv3 kernel trace (csv) of a two-lane run and prints, per kernel kind, its
average duration and the share of that duration during which a kernel of each other kind (any stream) was running too.
usage: python tools/overlap_trace.py <dir with *kernel_trace.csv> [out.json [from to]]  (from / to: fractions of the launch sequence; default 0.5 1.0)"""
import csv, glob, json, sys, collections

KINDS = ("k_setup", "k_chunks", "k_bin", "k_raster_slow", "k_raster", "k_clear", "k_shade", "k_vertex")


def kind_of(name):
    if "k_clear_tune" in name:  # (the one-thread kernel that ends a measurement render)
        return None
    for k in KINDS:
        if k in name:
            if k == "k_shade":
                return "k_shade_generic" if "k_shade<false, 0," in name else "k_shade"
            return k
    return None


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        k = kind_of(r["Kernel_Name"])
        if k:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Stream_Id") or r.get("Queue_Id")))
    rows.sort()
    # keep the second half of the run (the timed region, warm) — or the fractions of it given as 3rd / 4th argument
    lo_f, hi_f = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (0.5, 1.0)
    rows = rows[int(len(rows) * lo_f):int(len(rows) * hi_f)]
    dur = collections.defaultdict(float)
    cnt = collections.Counter()
    ov = collections.defaultdict(lambda: collections.defaultdict(float))
    n = len(rows)
    for i, (s, e, k, q) in enumerate(rows):
        dur[k] += e - s
        cnt[k] += 1
        per = collections.defaultdict(list)
        for j in range(max(0, i - 64), min(n, i + 64)):
            if j == i:
                continue
            s2, e2, k2, q2 = rows[j]
            lo, hi = max(s, s2), min(e, e2)
            if hi > lo:
                per[k2].append((lo, hi))
        for k2, iv in per.items():  # union of the intervals of kind k2
            iv.sort()
            tot, cur_lo, cur_hi = 0, None, None
            for lo, hi in iv:
                if cur_hi is None or lo > cur_hi:
                    if cur_hi is not None:
                        tot += cur_hi - cur_lo
                    cur_lo, cur_hi = lo, hi
                else:
                    cur_hi = max(cur_hi, hi)
            tot += cur_hi - cur_lo
            ov[k][k2] += tot
    res = {}
    for k in dur:
        res[k] = {"launches": cnt[k], "avg_us": dur[k] / cnt[k] / 1e3,
                  "beside": {k2: round(v / dur[k], 3) for k2, v in sorted(ov[k].items()) if v / dur[k] >= 0.005}}
        print(f"{k:16s} n={cnt[k]:4d} avg={res[k]['avg_us']:8.1f} us  beside: {res[k]['beside']}")
    span = (max(r[1] for r in rows) - min(r[0] for r in rows)) / 1e3
    res["_span_us"] = span
    res["_busy_sum_us"] = sum(dur.values()) / 1e3
    print(f"span {span:.0f} us, sum of kernel durations {res['_busy_sum_us']:.0f} us")
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], "w"), indent=1)


main()
