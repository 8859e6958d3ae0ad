"""Dev tool: print the kernel timeline of three consecutive renders from a rocprofv3 --kernel-trace csv (times relative
to the first k_setup)."""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
setups = [i for i, r in enumerate(rows) if "k_setup" in r["Kernel_Name"]]
i0 = setups[-4]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:setups[-1]]:
    print(f'{r["Kernel_Name"][:40]:40s} {(int(r["Start_Timestamp"])-t0)/1e3:9.1f} -> {(int(r["End_Timestamp"])-t0)/1e3:9.1f} us')
