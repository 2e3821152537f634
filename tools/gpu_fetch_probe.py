import os, sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from readtape_amd import frontend
tape = bench.make_base_tape(1000, 5_000_000)
rows = torch.from_numpy(tape.rows).cuda()[: 1 << 21].contiguous()
fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(tape.spec.header(), nparmsets=1))
for _ in range(3):
    r = fe.scan(rows); torch.cuda.synchronize()
    t0 = time.perf_counter(); r.fetch(); t1 = time.perf_counter()
    b = r.bufs
    t2 = time.perf_counter(); x = b["bursts"].cpu(); t3 = time.perf_counter(); y = b["counts"].cpu(); t4 = time.perf_counter()
    used = int((r.bursts["event_base"].astype(np.int64) + 9 * r.bursts["event_cap"].astype(np.int64)).max())
    z = b["events"][: used * 16].cpu(); t5 = time.perf_counter()
    print(json.dumps({"fetch_ms": (t1 - t0) * 1e3, "bursts_MB": x.numel() / 1e6, "bursts_ms": (t3 - t2) * 1e3, "counts_MB": y.numel() / 1e6, "counts_ms": (t4 - t3) * 1e3,
                      "events_MB": z.numel() / 1e6, "events_ms": (t5 - t4) * 1e3, "nevents": int(r.counts.sum())}))
