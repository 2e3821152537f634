"""Condenses rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into a small text summary for profiles/.
usage: prof_summary.py <dir with trace/ pmc_fetch/ pmc_write/ sub-directories>"""
import glob, os, sqlite3, sys
root = sys.argv[1]

def dbs(sub):
    return sorted(glob.glob(os.path.join(root, sub, "**", "*.db"), recursive=True))

cmd = sys.argv[2] if len(sys.argv) > 2 else "--steps 5 --warmup 2"
print(f"== kernel stats: rocprofv3 --kernel-trace --stats -- python bench.py {cmd} --no-cpu-baseline --no-e2e ==")
for f in dbs("trace"):
    db = sqlite3.connect(f)
    for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name.split("(")[0][:48]
        if "rtfe" in name or "copy" in name.lower():
            print(f"{short:48s} calls {calls:4d}  total_us {total:12.1f}  avg_us {avg:10.1f}  {pct:5.1f}%")
for sub, label in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    print(f"== rocprofv3 --pmc {label} (own pass; KiB per dispatch, mean over dispatches) ==")
    for f in dbs(sub):
        db = sqlite3.connect(f)
        q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
        for kn, cn, n, v in db.execute(q):
            if "rtfe" in kn:
                note = ""
                if cn == "FETCH_SIZE":
                    note = f"  -> x2 (gfx950 wide-stream correction, MI355X_MICROARCH.md HBM) = {2 * v * 1024 / 1e9:.3f} GB"
                else:
                    note = f"  = {v * 1024 / 1e9:.3f} GB (uncalibrated on gfx950)"
                print(f"{kn.split('(')[0][:32]:32s} {cn} n={n} mean_KiB={v:.1f}{note}")
