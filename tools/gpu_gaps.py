"""GPU box: what the gaps between the kernels of a scan cost - wall time per scan with and without the timing events, against the sum
of the kernel spans."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import torch
import bench
from readtape_amd import frontend
tape = bench.make_base_tape(seed=1000, target_rows=int(5e6))
hdr = tape.spec.header()
base = torch.from_numpy(tape.rows).cuda()
rows = base.repeat(max(1, int(round(float(sys.argv[1]) if len(sys.argv) > 1 else 1e8) / base.shape[0])), 1).contiguous()
fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, nparmsets=1))
for timing in (False, True, False):
    fe.set_timing(timing)
    for i in range(3): fe.scan(rows)
    torch.cuda.synchronize()
    if timing: fe.kernel_ms()
    t0 = time.perf_counter()
    for i in range(20): fe.scan(rows)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20 * 1e3
    extra = ""
    if timing:
        ms, n = fe.kernel_ms()
        extra = " sum of spans %.3f ms" % (sum(ms.values()) / n)
    print("timing events", timing, "%.3f ms per scan" % dt, extra)
