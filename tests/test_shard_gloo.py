"""The N > 1 path on CPU: two processes (torch.distributed, gloo) each own half of a tape's rows, exchange
the seam halo, scan their slice with the emulated kernels, and together must reproduce the single-scan
burst table and event lists exactly — wherever the seam falls (inside a block, inside a gap)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, pickle
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
import numpy as np, torch, torch.distributed as dist
from emul_util import emul_frontend
from golden_util import load_case
from parity_util import config_for
from readtape_amd import shard
case, cut_frac, out = sys.argv[2], float(sys.argv[3]), sys.argv[4]
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = load_case(case)
rows = np.ascontiguousarray(g["rows"])
n = rows.shape[0]
align = int(sys.argv[5])
cut = int(n * cut_frac) // align * align
spans = [(0, cut), (cut, n)]
lo, hi = spans[rank]
own = torch.from_numpy(rows[lo:hi].copy())
with_halo, own_rows = shard.exchange_halo(own, 4096, rank, world, dist)
fe = emul_frontend(config_for(g["hdr"], g["oracle_opts"]))
res = fe.scan(with_halo.numpy(), row_base=lo, first_is_tape_start=(rank == 0), own_rows=own_rows).fetch()
b = shard.absolute_bursts(res, lo)
mine = dict(bursts=b, events=shard.flatten_events(res, b, 0))
allr = [None] * world
dist.all_gather_object(allr, mine)
if rank == 0:
    pickle.dump(allr, open(out, "wb"))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("case,cut_frac,align", [("nrzi9", 0.30, 512), ("nrzi9", 0.52, 8), ("pe", 0.40, 8)])
def test_two_rank_time_shards_equal_single_scan(case, cut_frac, align, tmp_path):
    import pickle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emul_util import build_emul, emul_frontend
    from golden_util import load_case
    from parity_util import config_for
    from readtape_amd import shard
    build_emul()
    out = str(tmp_path / "res.pkl")
    wfile = tmp_path / "worker.py"
    wfile.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 1000), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(wfile), ROOT, case, str(cut_frac), out, str(align)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    parts = pickle.load(open(out, "rb"))
    g = load_case(case)
    fe = emul_frontend(config_for(g["hdr"], g["oracle_opts"]))
    whole = fe.scan(g["rows"]).fetch()
    wb = shard.absolute_bursts(whole, 0)
    we = shard.flatten_events(whole, wb, 0)
    got_b = np.concatenate([p["bursts"] for p in parts])
    got_e = np.concatenate([p["events"] for p in parts])
    # shard starts on the 512-row grid see the same quiet map, hence the same zones / restarts / extents
    if align == 512:
        for f in ("zone_end", "reset_sample", "safe_last", "end_sample"):
            assert list(got_b[f]) == list(wb[f]), f
    assert not (got_b["flags"] & ~np.uint32(1)).any()
    # any cut: the union of the ranks' events is the single-scan event list, bit for bit
    key = lambda e: e[np.lexsort((e[:, 1], e[:, 0]))]
    assert got_e.shape == we.shape and (key(got_e) == key(we)).all()


TAP_WORKER = r'''
import os, sys, pickle
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
import numpy as np, torch, torch.distributed as dist
from emul_util import emul_frontend
from golden_util import load_case
from readtape_amd import shard
case, out, cuts = sys.argv[2], sys.argv[3], [int(x) for x in sys.argv[4].split(",") if int(x)]
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = load_case(case)
rows = np.ascontiguousarray(g["rows"])
n = rows.shape[0]
spans = list(zip([0] + cuts, cuts + [n])) if cuts else shard.plan_shards(n, world, align=64)
lo, hi = spans[rank]
if os.environ.get("TEST_FAIL_RANK") == str(rank):          # (a fatal reference condition in one rank's replay)
    from readtape_amd import pipeline
    def boom(*a, **k): raise pipeline.ReferenceFatal("injected")
    pipeline.decode_fragment = boom
table = shard.decode_sharded(g["hdr"], torch.from_numpy(rows[lo:hi].copy()), lo, n, rank, world, dist, out if rank == 0 else None,
                             fe_factory=emul_frontend, halo_rows=1024)
if rank == 0:
    pickle.dump(table, open(out + ".tab", "wb"))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("case,cutrow", [("nrzi9", 0), ("nrzi9", 2496), ("gcr", 0), ("pe", 0)])      # (2496: inside the first block of nrzi9; 0: nrows / 2)
def test_two_ranks_write_the_single_rank_tap(case, cutrow, tmp_path):
    """Two gloo ranks decode ONE tape: halo exchange, scans, all-gather of the per-rank tables, per-rank replay, rank 0 assembles the
    .tap - byte for byte the reference's (the golden), wherever the seam falls.  The starting halo (1024 rows) is shorter than a
    block + gap, so the halo-growing round trip runs too."""
    import pickle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emul_util import build_emul
    from golden_util import load_case
    build_emul()
    out = str(tmp_path / "sharded.tap")
    wfile = tmp_path / "tap_worker.py"
    wfile.write_text(TAP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 1000), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(wfile), ROOT, case, out, str(cutrow)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    g = load_case(case)
    assert open(out, "rb").read() == g["tap"]
    table = pickle.load(open(out + ".tab", "rb"))
    assert [t["rank"] for t in table] == [0, 1] and table[1]["tap_offset"] == table[0]["tap_len"]
    assert sum(t["blocks"] + t["tapemarks"] for t in table) > 0 and all(t["bursts"] > 0 for t in table)


def test_three_ranks_with_a_shard_shorter_than_the_halo(tmp_path):
    """The middle rank owns 640 rows: the first rank's burst runs through all of them into the third rank's rows, so its halo must span
    two ranks (the retry loop of decode_sharded once waited for ever for rows its neighbour did not have)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emul_util import build_emul
    from golden_util import load_case
    build_emul()
    out = str(tmp_path / "sharded.tap")
    wfile = tmp_path / "tap_worker.py"
    wfile.write_text(TAP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + os.getpid() % 1000), WORLD_SIZE="3")
    procs = [subprocess.Popen([sys.executable, str(wfile), ROOT, "nrzi9", out, "2496,3136"], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(3)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    assert open(out, "rb").read() == load_case("nrzi9")["tap"]


def test_a_failing_rank_stops_every_rank(tmp_path):
    """One rank's replay raises: the ranks agree on the failure before the next collective, and every process ends with an error
    instead of one waiting in a gather for ever."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emul_util import build_emul
    build_emul()
    out = str(tmp_path / "sharded.tap")
    wfile = tmp_path / "tap_worker.py"
    wfile.write_text(TAP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 1000), WORLD_SIZE="2", TEST_FAIL_RANK="1")
    procs = [subprocess.Popen([sys.executable, str(wfile), ROOT, "nrzi9", out, "0"], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stderr=subprocess.PIPE, text=True) for r in range(2)]
    errs = [p.communicate(timeout=600)[1] for p in procs]
    assert all(p.returncode != 0 for p in procs)
    assert "rank 1 failed" in errs[1] and "another rank failed" in errs[0]


BENCH_WORKER = r'''
import os, sys, pickle
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
import numpy as np, torch, torch.distributed as dist
import bench
from emul_util import emul_frontend
from readtape_amd import shard, synth
config, out, halo = sys.argv[2], sys.argv[3], int(sys.argv[4])
copies = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
tape = synth.nrzi_tape(seed=77, nblocks=5, minlen=64, maxlen=300, marks_every=3, gap_samples=3000)
wl = bench.Workload(bench.CONFIGS[config], rank, world, torch.device("cpu"), dist, total_rows=copies * tape.rows.shape[0], base_rows=0,
                    fe_factory=emul_frontend, halo=halo, tape=tape)
parts = []
for i in range(2):                                   # two steps: the halo exchange repeats into the same buffer
    res = wl.step(i).fetch()
    b = shard.absolute_bursts(res, wl.row_base)
    parts.append(dict(bursts=b, events=shard.flatten_events(res, b, 0), lo=wl.row_base, n=wl.nrows, got=wl.sr.got))
allr = [None] * world
dist.all_gather_object(allr, parts)                  # (test plumbing only)
if rank == 0:
    pickle.dump(allr, open(out, "wb"))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("config,halo", [("C5", 1 << 14), ("C2", 1 << 14)])
def test_two_ranks_drive_the_step_of_bench_py(config, halo, tmp_path):
    """bench.py's own Workload.step() - what `bench.py --gpus N` times - on two gloo ranks with the emulated kernels.  C5 (the default
    for N > 1): ONE tape cut by plan_shards, the seam halo received into the tail of the rank's one buffer, rtfe_scan with the ownership
    rule; the ranks' bursts and events together are the single scan of the whole tape.  C2 with N > 1 (weak): every rank's own tape."""
    import pickle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import bench
    from emul_util import build_emul, emul_frontend
    from readtape_amd import frontend, shard, synth
    build_emul()
    out = str(tmp_path / "res.pkl")
    wfile = tmp_path / "bench_worker.py"
    wfile.write_text(BENCH_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 1000), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(wfile), ROOT, config, out, str(halo)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    parts = pickle.load(open(out, "rb"))
    tape = synth.nrzi_tape(seed=77, nblocks=5, minlen=64, maxlen=300, marks_every=3, gap_samples=3000)
    fe = emul_frontend(frontend.FrontEndConfig.from_header(tape.spec.header()))
    key = lambda e: e[np.lexsort((e[:, 1], e[:, 0]))]
    if config == "C5":
        rows = np.tile(tape.rows, (3, 1))
        assert [(p[0]["lo"], p[0]["lo"] + p[0]["n"]) for p in parts] == shard.plan_shards(rows.shape[0], 2)
        assert parts[0][0]["got"] == halo and parts[1][0]["got"] == 0
        whole = fe.scan(rows).fetch()
        wb = shard.absolute_bursts(whole, 0)
        we = shard.flatten_events(whole, wb, 0)
        for i in range(2):
            got_b = np.concatenate([p[i]["bursts"] for p in parts])
            got_e = np.concatenate([p[i]["events"] for p in parts])
            for f in ("zone_end", "reset_sample", "safe_last", "end_sample"):
                assert list(got_b[f]) == list(wb[f]), f
            assert got_e.shape == we.shape and (key(got_e) == key(we)).all()
        assert we.shape[0] > 5000
    else:
        # weak: rank r scans its own 3 copies (+ the first rows of rank r + 1's tape as halo); its own bursts are the single scan's
        # except the one that straddles the end of its rows
        assert parts[0][0]["n"] == parts[1][0]["n"] == 3 * tape.rows.shape[0]
        whole = fe.scan(np.tile(tape.rows, (3, 1))).fetch()
        assert parts[1][0]["bursts"].shape[0] == whole.nbursts and parts[0][0]["bursts"].shape[0] >= whole.nbursts - 1
        assert parts[0][0]["got"] == halo and parts[1][0]["got"] == 0 and parts[1][0]["lo"] == parts[0][0]["n"]
        assert parts[0][1]["events"].shape == parts[0][0]["events"].shape and (parts[0][1]["events"] == parts[0][0]["events"]).all()


def test_eight_ranks_drive_the_step_of_bench_py_on_c5s_shape(tmp_path):
    """What `bench.py --gpus 8` runs: ONE tape cut into eight time shards by plan_shards, every rank but the last receives its seam halo
    from the rank behind it (neighbour isend / irecv, no collective on the data path), rtfe_scan with the ownership rule - eight gloo ranks with the
    emulated kernels, two steps (the halo lands in the same buffer again).  The ranks' bursts and events together are the single scan of the whole tape."""
    import pickle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    from emul_util import build_emul, emul_frontend
    from readtape_amd import frontend, shard, synth
    build_emul()
    world, halo, copies = 8, 1 << 13, 8
    out = str(tmp_path / "res.pkl")
    wfile = tmp_path / "bench_worker.py"
    wfile.write_text(BENCH_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(30100 + os.getpid() % 1000), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(wfile), ROOT, "C5", out, str(halo), str(copies)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=1500) == 0
    parts = pickle.load(open(out, "rb"))
    tape = synth.nrzi_tape(seed=77, nblocks=5, minlen=64, maxlen=300, marks_every=3, gap_samples=3000)
    rows = np.tile(tape.rows, (copies, 1))
    assert [(p[0]["lo"], p[0]["lo"] + p[0]["n"]) for p in parts] == shard.plan_shards(rows.shape[0], world)
    assert all(p[0]["got"] == halo for p in parts[:-1]) and parts[-1][0]["got"] == 0
    fe = emul_frontend(frontend.FrontEndConfig.from_header(tape.spec.header()))
    whole = fe.scan(rows).fetch()
    wb = shard.absolute_bursts(whole, 0)
    we = shard.flatten_events(whole, wb, 0)
    key = lambda e: e[np.lexsort((e[:, 1], e[:, 0]))]
    for i in range(2):
        got_b = np.concatenate([p[i]["bursts"] for p in parts])
        got_e = np.concatenate([p[i]["events"] for p in parts])
        for f in ("zone_end", "reset_sample", "safe_last", "end_sample"):
            assert list(got_b[f]) == list(wb[f]), f
        assert got_e.shape == we.shape and (key(got_e) == key(we)).all()
    assert we.shape[0] > 12000 and all(p[0]["bursts"].shape[0] > 0 for p in parts)


def test_every_bench_configuration_is_complete():
    """bench.py's CONFIGS: each entry carries every key the bench legs read (a comment once swallowed C4's ref_opts / port_opts and the
    CPU baseline of `--config C4` died on the missing key)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    need = {"kind", "rows", "nparmsets", "find_zeros", "window_rows", "ref_opts", "port_opts", "workload"}
    for name, conf in bench.CONFIGS.items():
        assert need <= set(conf), (name, need - set(conf))
        assert conf["workload"].startswith(name), name
        assert isinstance(conf["ref_opts"], list) and isinstance(conf["port_opts"], list), name
