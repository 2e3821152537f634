"""GPU parity at BASELINE.json's full sizes (run with -m gpu on an MI355X), through properties that need no oracle run of that length:

* the base tape (a few 1e6 rows, the very tape bench.py tiles) is checked against the CPU oracle event for event;
* tiling k copies of a tape gives k copies of its events (the front end is shift invariant: every burst starts from reset state inside
  a dead-quiet zone) - so the FIRST, a MIDDLE and the LAST copy of the full-size scan must carry the base scan's events bit for bit,
  also where the byte offsets into the rows pass 2^32;
* the .tap of the tiled tape through the host replay is k times the base tape's records;
* a tape cut into fragments at row 2^28 (what bench.py does for C3 / C4 and shard.py for C5) gives the unfragmented scan's bursts and
  events.
"""
import os
import sys

import numpy as np
import pytest

from parity_util import check_tape, config_for, oracle_attempts
from readtape_amd import frontend, pipeline, shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the workload generator of the benchmark line)

pytestmark = pytest.mark.gpu


@pytest.fixture()
def gpu():
    import gc
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    yield torch
    gc.collect()                                   # (each test holds GBs of rows, workspace and event arena: hand them back before the next)
    torch.cuda.empty_cache()


def burst_lists(r, b):
    """The event lists of burst b, all parameter sets and tracks, copied from the device arena on their own (a full-size scan's arena
    holds tens of GB: only the bursts a check looks at travel) -> [P, T] lists of event records."""
    fe = r.fe
    P, T = len(fe.cfg.parmsets), fe.cfg.ntrks
    B = r.bursts[b]
    base, cap = int(B["event_base"]), int(B["event_cap"])
    it = frontend.EVENT_DTYPE.itemsize
    reg = fe.backend.to_numpy(r.bufs["events"][base * it: (base + P * T * cap) * it], frontend.EVENT_DTYPE)
    return [[reg[(p * T + t) * cap: (p * T + t) * cap + int(r.counts[b, p, t])] for t in range(T)] for p in range(P)]


def copy_events(r, lo, hi, nparm=1):
    """Per parameter set: [(row - lo, trk, v_peak bits, agc_gain bits, left_distance, flags)] of the events at rows [lo, hi), sorted."""
    rs = r.bursts["reset_sample"].astype(np.int64)
    out = [[] for _ in range(nparm)]
    for b in np.nonzero((rs >= lo - (1 << 17)) & (rs < hi))[0]:       # (a copy's first burst may restart in the gap that ends the copy in front)
        lists = burst_lists(r, int(b))
        for p in range(nparm):
            ev = np.concatenate(lists[p])
            a = rs[b] + ev["sample"].astype(np.int64)
            m = (a >= lo) & (a < hi)
            out[p].append(np.stack([a[m] - lo, ev["trk"][m].astype(np.int64), ev["v_peak"][m].view("u4").astype(np.int64), ev["agc_gain"][m].view("u4").astype(np.int64),
                                    ev["left_distance"][m].astype(np.int64), ev["flags"][m].astype(np.int64)], 1))
    res = []
    for p in range(nparm):
        e = np.concatenate(out[p]) if out[p] else np.zeros((0, 6), np.int64)
        res.append(e[np.lexsort((e[:, 1], e[:, 0]))])
    return res


def tiled(torch, tape, total_rows):
    base = torch.from_numpy(tape.rows).cuda()
    k = max(3, int(round(total_rows / base.shape[0])))
    return base.repeat(k, 1).contiguous(), k, int(base.shape[0])


def check_copies(fe, rows, k, n, nparm=1, copies=None):
    """The events of copy 1 of a three-copy scan in copies 1, k // 2 and k - 1 (or `copies`) of the full scan, bit for bit; no burst
    flagged.  (Copy 0 begins at the tape's start - its first burst is an exact-start burst - so it is not the yardstick.)"""
    r3 = fe.scan(rows[: 3 * n]).fetch(events=False)
    ref = copy_events(r3, n, 2 * n, nparm)
    assert all(x.shape[0] > 1000 for x in ref)
    rk = fe.scan(rows).fetch(events=False)
    assert not (rk.bursts["flags"] & ~np.uint32(frontend.F_EXACT_START | frontend.F_STATE_AT_END)).any()
    for j in (copies or (1, k // 2, k - 1)):
        got = copy_events(rk, j * n, (j + 1) * n, nparm)
        for p in range(nparm):
            assert got[p].shape == ref[p].shape and (got[p] == ref[p]).all(), f"copy {j} parmset {p}"
    return rk


def test_c2_full_size(tmp_path, gpu):
    """BASELINE configs[1]: 9-track NRZI, 1e8 rows, one parameter set - the tape bench.py times."""
    torch = gpu
    tape = bench.make_base_tape(seed=1000, target_rows=5e6, kind="nrzi")
    hdr = tape.spec.header()
    fe = frontend.FrontEnd(config_for(hdr, []))
    msgs, stats = check_tape(fe, hdr, tape.rows, oracle_attempts(hdr, tape.rows, [], str(tmp_path)))
    assert not msgs, "\n".join(msgs[:12])
    assert stats["speculative"] == stats["attempts"] and stats["flags"] == 0
    rows, k, n = tiled(torch, tape, 1e8)
    assert rows.shape[0] >= 1e8
    rk = check_copies(fe, rows, k, n)
    st = fe.scan_stats(rk)
    assert st["redone"] == 0 and st["parallel"] > 0.95 * int(rk.counts.sum()) and st["parallel"] + st["sequential"] == int(rk.counts.sum()), st      # the peak path did the work, not a fallback
    # the .tap through the host replay: k times the base tape's records, one end mark
    base_tap, full_tap = str(tmp_path / "base.tap"), str(tmp_path / "full.tap")
    pipeline.decode_tape(hdr, tape.rows, base_tap)
    pipeline.decode_tape(hdr, rows, full_tap)
    b = open(base_tap, "rb").read()
    assert b.endswith(b"\xff\xff\xff\xff") and open(full_tap, "rb").read() == b[:-4] * k + b"\xff\xff\xff\xff"


def test_nrzi_rows_beyond_4_gib(gpu):
    """3e8 rows of 9 tracks = 5.4 GB: the byte offsets of the last copies do not fit 32 bits (C5 on few GPUs holds as much per rank)."""
    torch = gpu
    tape = bench.make_base_tape(seed=1001, target_rows=5e6, kind="nrzi")
    fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(tape.spec.header()))
    rows, k, n = tiled(torch, tape, 3e8)
    assert rows.shape[0] * rows.shape[1] * 2 > (1 << 32) + (1 << 30)
    check_copies(fe, rows, k, n, copies=(k // 2, k - 2, k - 1))


def test_c3_full_size(gpu):
    """BASELINE configs[2]: 9-track PE -zeros, 1e9 rows in ONE scan (18 GB of rows: every row offset beyond 2^32 bytes after the first
    quarter) through k_zeros - the first, a middle and the last copies carry the three-copy scan's events bit for bit."""
    torch = gpu
    tape = bench.make_base_tape(seed=1002, target_rows=5e6, kind="pe")
    fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(tape.spec.header(), nparmsets=1, find_zeros=True))
    rows, k, n = tiled(torch, tape, 1e9)
    assert rows.shape[0] >= 9.9e8
    rk = check_copies(fe, rows, k, n, copies=(1, k // 3, k // 2, k - 2, k - 1))
    assert int(rk.counts.sum()) > 5e8


@pytest.mark.parametrize("kind,nparm,zeros", [("pe", 1, True), ("gcr", 8, False)])
def test_c3_c4_one_full_fragment(kind, nparm, zeros, gpu):
    """BASELINE configs[2] and [3] at the size of one bench.py fragment (2^28 rows; 1e9 rows are four of them): PE -zeros through
    k_zeros, GCR with the 8-set sweep through k_decode."""
    torch = gpu
    conf = bench.CONFIGS["C3" if kind == "pe" else "C4"]
    tape = bench.make_base_tape(seed=1002, target_rows=5e6, kind=kind)
    hdr = tape.spec.header()
    parmsets = None
    if nparm > len(frontend.DEFAULT_PARMSETS[hdr.mode]):
        extra = [(1.4, 0.20, 0.2, 0.5, 0, 0.0), (1.6, 0.14, 0.0, 0.5, 0, 0.0), (1.5, 0.10, 0.1, 0.5, 0, 0.0), (1.3, 0.25, 0.2, 0.5, 0, 0.0)]
        parmsets = (list(frontend.DEFAULT_PARMSETS[hdr.mode]) + extra)[:nparm]
    fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, nparmsets=conf["nparmsets"], find_zeros=zeros, parmsets=parmsets))
    rows, k, n = tiled(torch, tape, 1 << 28)
    rows = rows[: 1 << 28].contiguous() if rows.shape[0] > (1 << 28) else rows
    k = rows.shape[0] // n
    check_copies(fe, rows, k, n, nparm=nparm, copies=(1, k - 1))


def test_fragments_cut_at_row_2_to_the_28(gpu):
    """A 2^28 + 2^23-row NRZI tape scanned as bench.py / shard.py scan a long one - the fragment [0, 2^28) with its halo, then the
    fragment from row 2^28 on with row_base = 2^28 - against the unfragmented scan: the same bursts, the same events."""
    torch = gpu
    tape = bench.make_base_tape(seed=1003, target_rows=5e6, kind="nrzi")
    fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(tape.spec.header()))
    rows, k, n = tiled(torch, tape, (1 << 28) + (1 << 23))
    total, cut, halo = int(rows.shape[0]), 1 << 28, 1 << 18
    assert total > cut + (1 << 22)
    lo_w = cut - (1 << 23)                                     # compare the 2^24 rows around the cut (the rest is test_c2_full_size's business)

    def near_cut(res, base):
        ab = shard.absolute_bursts(res, base)
        keep = np.nonzero(ab["reset_sample"] >= lo_w)[0]
        ev = []
        for b in keep:
            e = np.concatenate(burst_lists(res, int(b))[0])
            ev.append(np.stack([ab[b]["reset_sample"] + e["sample"].astype(np.int64), e["trk"].astype(np.int64), e["flags"].astype(np.int64), e["v_peak"].view("u4").astype(np.int64),
                                e["agc_gain"].view("u4").astype(np.int64), e["left_distance"].astype(np.int64)], 1))
        return ab[keep], (np.concatenate(ev) if ev else np.zeros((0, 6), np.int64))

    wb, we = near_cut(fe.scan(rows).fetch(events=False), 0)
    parts_b, parts_e = [], []
    for a, b, first in ((0, cut, True), (cut, total, False)):
        end = min(total, b + halo) if b < total else total
        res = fe.scan(rows[a:end], row_base=a, first_is_tape_start=first, own_rows=b - a).fetch(events=False)
        pb, pe = near_cut(res, a)
        parts_b.append(pb); parts_e.append(pe)
    got_b, got_e = np.concatenate(parts_b), np.concatenate(parts_e)
    for f in ("zone_end", "reset_sample", "safe_last", "end_sample"):
        assert list(got_b[f]) == list(wb[f]), f
    key = lambda e: e[np.lexsort((e[:, 1], e[:, 0]))]
    assert got_e.shape == we.shape and (key(got_e) == key(we)).all() and we.shape[0] > 100000


C4_EXTRA = [(1.4, 0.20, 0.2, 0.5, 0, 0.0), (1.6, 0.14, 0.0, 0.5, 0, 0.0), (1.5, 0.10, 0.1, 0.5, 0, 0.0), (1.3, 0.25, 0.2, 0.5, 0, 0.0)]


def test_bench_tape_c3_against_the_oracle(tmp_path, gpu):
    """The very PE tape bench.py --config C3 tiles (seed 1000 of make_base_tape, -zeros) against the CPU oracle (round 3 checked the bench's
    C3 / C4 tapes only against themselves).  With -zeros the device's crossings only mean something behind the replay's slope gate, so the
    check is end to end: the oracle's .tap, byte for byte, no exact rescan, no burst flagged."""
    import subprocess
    from parity_util import ORACLE, build_oracle
    from readtape_amd import tbin
    tape = bench.make_base_tape(seed=1000, target_rows=5e6, kind="pe")
    hdr = tape.spec.header()
    build_oracle()
    tbin.write_tbin(str(tmp_path / "t.tbin"), hdr, tape.rows)
    subprocess.run([ORACLE, "-zeros", f"-out={tmp_path}/o", str(tmp_path / "t.tbin")], check=True)
    stats, r1 = pipeline.decode_tape(hdr, tape.rows, str(tmp_path / "g.tap"), find_zeros=True)
    want = open(tmp_path / "o.tap", "rb").read()
    assert open(tmp_path / "g.tap", "rb").read() == want and len(want) > 100000
    assert stats["blocks"] > 50 and stats["exact_scans"] == 0, stats
    assert int(r1.counts.sum()) > 1e6 and not (r1.bursts["flags"] & ~np.uint32(frontend.F_EXACT_START | frontend.F_STATE_AT_END)).any()


def test_bench_tape_p1_against_the_oracle(tmp_path, gpu):
    """The PE tape of bench.py's P1 line (seed 1000 of make_base_tape, PEAK detection - the dense path, not -zeros) against the CPU oracle,
    attempt by attempt, event for event (P1's shape had only met k_decode)."""
    tape = bench.make_base_tape(seed=1000, target_rows=5e6, kind="pe")
    hdr = tape.spec.header()
    att = oracle_attempts(hdr, tape.rows, [], str(tmp_path))
    fe = frontend.FrontEnd(config_for(hdr, []))
    msgs, stats = check_tape(fe, hdr, tape.rows, att)
    assert not msgs, "\n".join(msgs[:8])
    assert stats["events"] > 2e6, stats


@pytest.mark.parametrize("kind,noise", [("nrzi", 60.0), ("gcr", 30.0)])
def test_bench_noise_tapes_against_the_oracle(kind, noise, tmp_path, gpu):
    """The noisy lines of bench.py (N1: C2's tape at 60 mV rms, N2: G1's at 30 mV - other_configs): whatever the speculation costs there,
    the events are the oracle's."""
    tape = bench.make_base_tape(seed=1000, target_rows=2e6, kind=kind, noise_mv=noise)
    hdr = tape.spec.header()
    att = oracle_attempts(hdr, tape.rows, [], str(tmp_path))
    fe = frontend.FrontEnd(config_for(hdr, []))
    msgs, stats = check_tape(fe, hdr, tape.rows, att)
    assert not msgs, "\n".join(msgs[:8])
    assert stats["events"] > 1e5, stats


def test_bench_tape_c4_all_eight_sets_against_the_oracle(tmp_path, gpu):
    """The GCR tape bench.py --config C4 tiles (seed 1000) with the bench's eight parameter sets: every set's events of the ONE eight-set
    scan equal the single-set front end's, and that one is checked event for event against the oracle run with the set as its only one -
    all eight, not two."""
    tape = bench.make_base_tape(seed=1000, target_rows=5e6, kind="gcr")
    hdr = tape.spec.header()
    sets = (list(frontend.DEFAULT_PARMSETS[hdr.mode]) + C4_EXTRA)[:8]
    fe8 = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, nparmsets=8, parmsets=sets))
    r8 = fe8.scan(tape.rows).fetch()
    assert r8.nbursts > 20
    done = {}
    for p, ps in enumerate(sets):
        bf, rise, mp, al, aw, _ = ps
        key = (bf, rise, mp, al, aw)
        fe1 = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, parmsets=[ps]))
        if key not in done:                                  # (four of the reference's five GCR sets differ only in what the host decoder reads: one oracle run serves them)
            parms = ("parms active, clk_window, clk_alpha, agc_window, agc_alpha, min_peak, pulse_adj, pkww_bitfrac, pkww_rise, z1pt, z2pt, id\n"
                     f"{{1, 0, 0.015, {aw}, {al}, {mp}, 0.3, {bf}, {rise}, 1.45, 2.35, PRM}}\n")
            (tmp_path / f"p{p}.parms").write_text(parms)
            att = oracle_attempts(hdr, tape.rows, [f"-parms={tmp_path}/p{p}.parms"], str(tmp_path))
            msgs, stats = check_tape(fe1, hdr, tape.rows, att)
            assert not msgs, f"set {p}: " + "\n".join(msgs[:8])
            done[key] = fe1.scan(tape.rows).fetch()
        r1 = done[key]
        assert r1.nbursts == r8.nbursts
        for b in range(r8.nbursts):
            for t in range(hdr.ntrks):
                a, c = r8.track_events(b, p, t).copy(), r1.track_events(b, 0, t).copy()
                a["parmset"] = 0                                  # (a burst's restart row - the earliest over sets and tracks - differs between the sweep and a single set: absolute rows)
                a["sample"] = (a["sample"].astype(np.int64) + int(r8.bursts[b]["reset_sample"]) - int(r1.bursts[b]["reset_sample"])).astype(np.uint32)
                assert a.tobytes() == c.tobytes(), (p, b, t)


def test_c4_full_size_through_the_bench_step(gpu):
    """BASELINE configs[3] exactly as bench.py times it: 1e9 rows of GCR, eight sets, ONE launch (round 6: the event arena sized for a burst per 1e4 rows instead
    of the worst case - bench.CONFIGS["C4"]["bursts_per_row"]; an arena that is too small would flag bursts RTFE_F_EVENT_OVERFLOW) through bench.Workload.step().
    Shift invariance gives the expected totals: the events of the first, a middle and the last copy of a three-copy scan, the middle
    one times (copies - 2); and no burst may be flagged."""
    torch = gpu
    conf = bench.CONFIGS["C4"]
    wl = bench.Workload(conf, 0, 1, torch.device("cuda:0"), None, 1e9, 5e6)
    assert len(wl.frags) == 1 and wl.nrows >= 9.9e8 and wl.fe.bursts_hint >= 9e4
    n, k = int(wl.tape.rows.shape[0]), wl.copies
    fe3 = frontend.FrontEnd(wl.cfg)
    r3 = fe3.scan(wl.sr.buf[: 3 * n]).fetch(events=False)
    rs = r3.bursts["reset_sample"].astype(np.int64)
    per = np.zeros((3, 8), np.int64)                      # events per (copy, set), a burst counted where it restarts (as the fragments' union does)
    for b in range(r3.nbursts):
        c = 0 if rs[b] < n - (1 << 17) else (1 if rs[b] < 2 * n - (1 << 17) else 2)      # (a copy's first burst restarts in the gap that ends the copy in front)
        per[c] += r3.counts[b].sum(axis=1).astype(np.int64)
    fe3.close(); del r3
    torch.cuda.empty_cache()
    tot = np.zeros(8, np.int64)
    tally = dict(bursts=0, bad=0)

    def each(r):
        r.fetch(events=False)
        tot[:] += r.counts.sum(axis=(0, 2)).astype(np.int64)
        tally["bursts"] += int(r.nbursts)
        tally["bad"] += int(((r.bursts["flags"] & ~np.uint32(frontend.F_EXACT_START | frontend.F_STATE_AT_END)) != 0).sum())
    wl.step(0, each=each)
    assert tally["bad"] == 0 and tally["bursts"] > 10000
    expect = per[0] + (k - 2) * per[1] + per[2]
    assert (tot == expect).all(), (tot, expect)


def test_c5_shape_eight_shards_on_one_gpu(gpu):
    """BASELINE configs[4]'s shape: ONE 5.55e8-row (10 GB) NRZI tape cut by shard.plan_shards(n, 8), every shard scanned with its halo and the
    ownership rule on this one GPU (what eight ranks do side by side): the union of the shards' bursts and counts is the whole scan's, and
    around every cut the events are the whole scan's byte for byte."""
    torch = gpu
    tape = bench.make_base_tape(seed=1000, target_rows=5e6, kind="nrzi")
    fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(tape.spec.header()))
    rows, k, n = tiled(torch, tape, 10e9 / 18)
    total = int(rows.shape[0])
    assert total >= 5.5e8
    whole = fe.scan(rows).fetch(events=False)
    wb = shard.absolute_bursts(whole, 0)
    spans = shard.plan_shards(total, 8)
    halo = 1 << 18
    pos = 0
    for r, (lo, hi) in enumerate(spans):
        end = min(total, hi + halo) if hi < total else total
        res = fe.scan(rows[lo:end], row_base=lo, first_is_tape_start=(lo == 0), own_rows=hi - lo).fetch(events=False)
        ab = shard.absolute_bursts(res, lo)
        m = res.nbursts
        assert m > 0
        for f in ("zone_first", "zone_end", "reset_sample", "safe_last", "end_sample"):
            k0 = 1 if (f == "zone_first" and lo > 0) else 0           # (a shard sees its first zone only from its own first row on)
            assert (ab[f][k0:m] == wb[f][pos + k0: pos + m]).all(), (r, f)
        assert lo == 0 or lo <= ab["zone_first"][0] or wb["zone_first"][pos] <= lo
        assert (res.bursts["flags"][:m] & ~np.uint32(frontend.F_EXACT_START)).max() == 0
        assert (res.counts[:m] == whole.counts[pos: pos + m]).all(), r
        for b in (0, 1, m - 2, m - 1):                          # the bursts next to the cuts: their events, every track
            if 0 <= b < m:
                got, ref = burst_lists(res, b)[0], burst_lists(whole, pos + b)[0]
                for t in range(9):
                    assert got[t].tobytes() == ref[t].tobytes(), (r, b, t)
        pos += m
    assert pos == whole.nbursts
