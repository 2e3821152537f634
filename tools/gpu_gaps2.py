"""GPU box: bench.Workload.step() wall time per step against plain fe.scan() on the same rows."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import torch
import torch.distributed as dist
import bench
dev = torch.device("cuda:0")
wl = bench.Workload(bench.CONFIGS["C2"], 0, 1, dev, dist, 1e8, 5e6)
for name, fn in (("step", lambda i: wl.step(i, timed=True)), ("scan", lambda i: wl.fe.scan(wl.sr.buf[:wl.nrows], row_base=0, first_is_tape_start=True, own_rows=wl.nrows)), ("step", lambda i: wl.step(i, timed=True))):
    for i in range(3): fn(i)
    torch.cuda.synchronize(); wl.collect()
    t0 = time.perf_counter()
    for i in range(10): fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for k in wl.kms: wl.kms[k] = 0.0
    wl.collect()
    print(name, "%.3f ms per step, cpu side %.3f ms per step, spans %.3f" % ((t2 - t0) / 10 * 1e3, (t1 - t0) / 10 * 1e3, sum(wl.kms.values()) / 10), {k: round(v / 10, 3) for k, v in wl.kms.items() if v > 0.1})
