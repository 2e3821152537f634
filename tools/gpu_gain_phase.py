"""k_gain phase split (RTFE_DEBUG=4: cycle counters of lane 0 of every wave, summed) on the bench tape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
os.environ["RTFE_DEBUG"] = "4"
os.environ.setdefault("RTFE_PEAK_PATH", "1")
import torch
import bench
from readtape_amd import frontend
rows_target = float(sys.argv[1]) if len(sys.argv) > 1 else 1e8
tape = bench.make_base_tape(seed=1000, target_rows=int(5e6))
hdr = tape.spec.header()
base = torch.from_numpy(tape.rows).cuda()
rows = base.repeat(max(1, int(round(rows_target / base.shape[0]))), 1).contiguous()
cfg = frontend.FrontEndConfig.from_header(hdr, nparmsets=1)
fe = frontend.FrontEnd(cfg)
fe.set_timing(True)
for i in range(2):
    r = fe.scan(rows)
    ms = fe.kernel_ms()[0]
st = fe.scan_stats(r)
ph = st["phase_cycles"]
waves = max(ph[5], 1)
print({k: round(v, 3) for k, v in ms.items() if v > 0.01})
print("per wave: cycles in steps %d, in the general step %d, at chunk ends %d; chunks %d, of which with a general step %d; waves %d" % (ph[0] // waves, ph[1] // waves, ph[2] // waves, ph[3] // waves, ph[4] // waves, waves))
print({k: st[k] for k in ("bursts", "redone", "parallel", "sequential", "gave_up")})
