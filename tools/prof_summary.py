"""Condenses rocprofv3 output (kernel stats + PMC csv) into a small text summary for profiles/."""
import csv, glob, os, sys
root = sys.argv[1]
def find(pat):
    return sorted(glob.glob(os.path.join(root, "**", pat), recursive=True))
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        name = row.get("Name", "")
        if "rtfe" in name or "k_" in name:
            print(f'{name[:60]:60s} calls {row.get("Calls")} total_ns {row.get("TotalDurationNs")} avg_ns {row.get("AverageNs")} pct {row.get("Percentage")}')
for kind in ("pmc_fetch", "pmc_write"):
    print(f"== {kind} ==")
    for f in find(f"{kind}/**/*counter_collection.csv") or find(f"{kind}*/*counter_collection.csv"):
        agg = {}
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "")[:40], row.get("Counter_Name"))
            agg.setdefault(k, []).append(float(row.get("Counter_Value", 0)))
        for (kn, cn), v in sorted(agg.items()):
            if "k_" in kn:
                print(f"{kn:40s} {cn} n={len(v)} mean={sum(v)/len(v):.1f}")
