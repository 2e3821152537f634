"""k_sift phase split (RTFE_DEBUG=3 cycle counters, summed over the waves) and k_gain statistics on the bench tape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
os.environ.setdefault("RTFE_DEBUG", "3")
os.environ.setdefault("RTFE_PEAK_PATH", "1")
import torch
import bench
from readtape_amd import frontend
rows_target = float(sys.argv[1]) if len(sys.argv) > 1 else 2e7
tape = bench.make_base_tape(seed=1000, target_rows=int(5e6))
hdr = tape.spec.header()
base = torch.from_numpy(tape.rows).cuda()
rows = base.repeat(max(1, int(round(rows_target / base.shape[0]))), 1).contiguous()
cfg = frontend.FrontEndConfig.from_header(hdr, nparmsets=1)
fe = frontend.FrontEnd(cfg)
fe.set_timing(True)
for i in range(3):
    r = fe.scan(rows)
    ms = fe.kernel_ms()[0]
st = fe.scan_stats(r)
ph = st["phase_cycles"]
tiles = max(ph[7], 1)
print("rows", rows.shape[0], {k: round(v, 3) for k, v in ms.items() if v > 0.01})
print("k_sift wave-cycles per tile (5 waves): copy+quiet %d dense %d owners %d; deferred candidates %d, rounds %d, tiles %d" % (ph[0] // tiles, ph[1] // tiles, ph[2] // tiles, ph[4], ph[5], tiles))
print("raw phase counters", ph)
print({k: st[k] for k in ("bursts", "redone", "parallel", "sequential", "gave_up")})
print("record bytes per row %.2f" % (ph[3] / rows.shape[0]))
