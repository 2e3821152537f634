"""GPU box: the NRZI parameter sweep (-m: 8 sets, three window widths) on the three paths, ms per scan of ~5e7 rows."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import torch
import bench
from readtape_amd import frontend
tape = bench.make_base_tape(seed=1000, target_rows=int(5e6), kind="nrzi")
hdr = tape.spec.header()
base = torch.from_numpy(tape.rows).cuda()
rows = base.repeat(max(1, int(round(5e7 / base.shape[0]))), 1).contiguous()
for nset in (8, 2, 1):
    for pp in ("0", "1", "0d"):                         # sample path / peak path / the dense sample path by force
        os.environ["RTFE_PEAK_PATH"] = pp[0]
        os.environ["RTFE_DENSE_PATH"] = "1" if pp == "0d" else "0"
        fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, parmsets=frontend.DEFAULT_PARMSETS[frontend.NRZI][:nset]))
        fe.set_timing(True)
        for i in range(2): r = fe.scan(rows)
        torch.cuda.synchronize(); fe.kernel_ms()
        t0 = time.perf_counter()
        for i in range(3): r = fe.scan(rows)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3 * 1e3
        ms, n = fe.kernel_ms()
        r.fetch(events=False)
        st = fe.scan_stats(r)
        print("sets", nset, "peak_path", pp, "rows", rows.shape[0], "%.2f ms" % dt, {k: round(v / n, 2) for k, v in ms.items() if v / n > 0.02}, "events", int(r.counts.sum()), {k: st[k] for k in ("bursts", "redone", "parallel", "sequential")}, st["gave_up"][:7])
        fe.close()
