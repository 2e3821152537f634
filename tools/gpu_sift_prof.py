"""GPU box: where k_sift_s's cycles go (a build with -DRTFE_SIFT_PROF: RTFE_LIB_PATH=readtape_amd/librtfe_prof.so).  One C2 scan; per phase the mean cycles a
wave spends in it per tile step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from readtape_amd import frontend

conf = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
dev = torch.device("cuda:0")
wl = bench.Workload(conf, 0, 1, dev, None, float(conf["rows"]), 5e6)
for i in range(3):
    res = wl.step(i)
torch.cuda.synchronize()
st = wl.fe.scan_stats(res)
names = ["tile->LDS (+wait rows)", "barrier 1", "loads + copy-out", "quiet", "strips", "compaction", "rounds + tail", "barrier 2"]
if "prof2" in os.environ.get("RTFE_LIB_PATH", ""):
    names = ["next tile's loads", "quiet word out", "pairs' lists out", "split head's list out", "round: derivation", "round: placement", "tail", "everything else"]
tiles = (wl.nrows + 895) // 896
waves = 4 if wl.cfg.ntrks == 9 else 3
pc = st["phase_cycles"]
tot = sum(pc)
if os.environ.get("RTFE_DEBUG") == "3":      # (the product build with its counters on: deferred candidates and rounds)
    print("deferred candidates", pc[4], "rounds", pc[5], "per wave-step %.2f" % (pc[5] / (tiles * waves)), "tile steps of wave 0", pc[7], "redone", st["redone"], "of", st["bursts"])
    sys.exit(0)
print("rows", wl.nrows, "tiles", tiles, "wave-steps", tiles * waves, "sum of cycles per wave-step", round(tot / (tiles * waves)))
for n, c in zip(names, pc):
    print(f"  {n:26s} {c / (tiles * waves):9.0f} cycles per wave and tile step  {100.0 * c / tot:5.1f} %")
