"""Condenses rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into a small text summary for profiles/.
usage: prof_summary.py <dir with a trace/ sub-directory> (tools/gpu_profile.sh)"""
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
