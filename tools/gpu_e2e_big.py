"""GPU box helper: end to end (file -> .tap) on a tape of ~1e8 rows with 1 / 8 / 16 / 32 replay threads (readtape_amd/ingest.py)."""
import os, sys, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from readtape_amd import ingest, tbin
tape = bench.make_base_tape(1000, 5_000_000)
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 18
with tempfile.TemporaryDirectory() as wd:
    path = os.path.join(wd, "big.tbin")
    tbin.write_tbin(path, tape.spec.header(), np.tile(tape.rows, (copies, 1)))
    ref = None
    for th, wr in ((1, 21), (8, 22), (12, 22), (16, 22), (24, 22)):
        st = ingest.decode_file_streaming(path, os.path.join(wd, f"o{th}.tap"), window_rows=1 << wr, halo_rows=1 << 18, replay_threads=th)
        data = open(os.path.join(wd, f"o{th}.tap"), "rb").read()
        ref = ref or data
        print(json.dumps({"threads": th, "window_rows_log2": wr, "rows": st["rows"], "seconds": round(st["seconds"], 3), "msamples_per_s": round(st["msamples_per_s"], 1),
                          "replay_seconds_summed": round(st["replay_seconds"], 2), "scan_wait": round(st["scan_wait_seconds"], 3), "read": round(st["read_seconds"], 3),
                          "blocks": st["blocks"], "same_tap": data == ref}), flush=True)
