"""Dev tool: average per-launch PMC values per kernel from a rocprofv3 --pmc csv directory."""
import collections, csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "<true>" in n: continue
    if "k_shade<true" in n: continue  # (the counting run)
    if "k_shade<false, 0," in n:      # (the generic build: a near-empty launch in the bench configs — kept apart, or it halves every average)
        agg["k_shade_generic"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        continue
    for k in ("k_setup", "k_bin", "k_raster_slow", "k_raster", "k_clear", "k_shade", "k_vertex"):
        if k in n:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            break
for k, d in agg.items():
    print(k, {c: (round(sum(v) / len(v) / 1e6, 2) if sum(v) / len(v) > 1e4 else round(sum(v) / len(v), 3)) for c, v in sorted(d.items())})  # (millions, or raw when small: derived metrics)
