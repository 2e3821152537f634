"""GPU box helper: the streaming reader's stage timeline (RT_INGEST_TRACE) on the bench's C2 sample.  usage: gpu_e2e_trace.py [copies] [window log2] [read threads] [replay threads] [split]"""
import os, sys, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_INGEST_TRACE"] = "1"
import numpy as np
import bench
from readtape_amd import ingest, tbin
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 16
wr = int(sys.argv[2]) if len(sys.argv) > 2 else 23
rt = int(sys.argv[3]) if len(sys.argv) > 3 else 4
th = int(sys.argv[4]) if len(sys.argv) > 4 else 32
sp = int(sys.argv[5]) if len(sys.argv) > 5 else 4
tape = bench.make_base_tape(1000, 5_000_000)
with tempfile.TemporaryDirectory() as wd:
    path = os.path.join(wd, "big.tbin")
    tbin.write_tbin(path, tape.spec.header(), np.tile(tape.rows, (copies, 1)))
    wpath = os.path.join(wd, "w.tbin")
    tbin.write_tbin(wpath, tape.spec.header(), tape.rows)
    ingest.decode_file_streaming(wpath, os.path.join(wd, "w.tap"), window_rows=1 << wr, halo_rows=1 << 18, replay_threads=th, read_threads=rt, replay_split=sp)
    for rep in range(2):
        st = ingest.decode_file_streaming(path, os.path.join(wd, "o.tap"), window_rows=1 << wr, halo_rows=1 << 18, replay_threads=th, read_threads=rt, replay_split=sp)
    tr = st.pop("trace")
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}))
    for e in tr:
        print("%-12s w%-3d %8.2f -> %8.2f ms  (%.2f)" % (e[0], e[1], e[2] * 1e3, e[3] * 1e3, (e[3] - e[2]) * 1e3))
