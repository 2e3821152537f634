"""Streaming ingest: a .tbin file larger than the device window is decoded window by window.

Replaces the reference's read loop (two fread()s per row into a stack buffer, src/readtape.c:1405-1414) with

    disk --(reader thread, readinto)--> pinned host buffer --(hipMemcpyAsync, copy stream)--> device window
         --(rtfe_scan, compute stream)--> events --(host replay of the window's own bursts)--> piece of the .tap

A three-stage software pipeline: window k + 2 is being read (three pinned buffers; the read itself in `read_threads` pieces side by
side), window k + 1 is being copied and scanned behind window k on the device (two scan contexts), window k is fetched - event arena
packed on the device first - and handed to the replay threads.  A window is a FRAGMENT in the sense of pipeline.decode_fragment: it
carries a halo of the following rows, owns the bursts whose zone ends inside it, and its .tap piece concatenates with its
neighbours' (DESIGN.md 6).  PyTorch supplies the pinned allocation (hipHostMalloc), the streams and the events; nothing here
computes on the CPU except the block decoders.
"""
from __future__ import annotations

import os
import threading
import time

import numpy as np

from . import frontend, pipeline, tbin


def _payload_geometry(path):
    """(header, payload offset, whole rows in the file).  The data end at the first row whose head-0 sample is 0x8000
    (src/readtape.c:1410) - normally the lone int16 behind the last row; the reader also looks for it inside every window."""
    hdr, off = tbin.read_header(path)
    nrows = (os.path.getsize(path) - off) // (2 * hdr.ntrks)
    return hdr, off, int(nrows)


class _All:
    """Several futures waited for as one (a device window is free again when all replays of its sub-fragments are done)."""

    def __init__(self, futs):
        self.futs = futs

    def result(self):
        for f in self.futs:
            f.result()


def decode_file_streaming(path, tap_path, window_rows=1 << 24, halo_rows=1 << 17, opts: pipeline.DecodeOptions | None = None, cfgkw=None,
                          device="cuda:0", replay_threads: int = 1, read_threads: int = 4, replay_split: int = 1):
    """Decodes the .tbin file `path` to the SIMH file `tap_path` through device windows of `window_rows` rows (a multiple of 1024).
    Returns statistics incl. the end-to-end rate (disk -> .tap), the time spent in the host replay and the rows the halos re-read.
    replay_threads > 1: the windows' host replays run side by side (fragments are independent: each has its own decoder context
    and writes its own piece of the .tap; the pieces are concatenated in window order).  A window's device buffer is kept until
    its replay is done (exact rescans read it), so replay_threads + 2 device windows are held.
    replay_split > 1: a window's bursts are replayed as that many sub-fragments side by side (cut at burst boundaries, exactly as the
    windows themselves are): the last window's replay is what the pipeline drains into, and a 4 M-row window takes one thread ~50 ms."""
    import torch
    assert window_rows % 1024 == 0
    t_enter = time.perf_counter()
    opts = opts or pipeline.DecodeOptions()
    hdr, off, nrows = _payload_geometry(path)
    if hdr.mode == tbin.MODE_WW:
        raise NotImplementedError("Whirlwind tapes are one chain (no independent fragments): read the file and use pipeline.decode_tape_ww")
    ntrks = hdr.ntrks
    full = pipeline.default_parmsets(hdr.mode, opts.nparmsets or (15 if opts.multiple_tries else 1))
    cfg = frontend.FrontEndConfig.from_header(hdr, parmsets=pipeline.frontend_parmsets(full), **(cfgkw or {}))
    fes = [frontend.FrontEnd(cfg, device=device) for _ in range(2)]      # two scans are in flight: window k is fetched while k + 1 is copied and scanned
    fe = fes[0]
    dev = torch.device(device)
    spans = [(lo, min(nrows, lo + window_rows)) for lo in range(0, nrows, window_rows)]
    cap = window_rows + halo_rows
    from concurrent.futures import ThreadPoolExecutor
    nthreads = max(1, int(replay_threads))
    depth = 2 if nthreads == 1 else nthreads + 2
    NP = 3                                                # pinned buffers: one being read into, one being copied from, one in between
    pinned = [torch.empty((cap, ntrks), dtype=torch.int16, pin_memory=True) for _ in range(NP)]      # hipHostMalloc
    dwin = [torch.empty((cap, ntrks), dtype=torch.int16, device=dev) for _ in range(depth)]
    pool = ThreadPoolExecutor(nthreads) if nthreads > 1 else None
    read_threads = max(1, int(read_threads))
    readers = ThreadPoolExecutor(read_threads) if read_threads > 1 else None
    fe_exact = frontend.FrontEnd(cfg, device=device)       # exact rescans of the replays: their own context, one at a time
    exact_lock = threading.Lock()
    copy_stream, scan_stream = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    copied = [torch.cuda.Event() for _ in range(NP)]
    fd = os.open(path, os.O_RDONLY)
    t_read = [0.0]
    data_end = [nrows]
    end_lock = threading.Lock()

    def read_span(arr, r0, r1, lo):
        """rows [lo + r0, lo + r1) of the payload -> arr[r0:r1]; returns the first row (relative to lo) whose head-0 sample is the end
        marker, or -1 (looked for here, by the thread that has just read the rows: a strided pass over the whole window in one
        thread had become the longest stage of the pipeline)."""
        view = memoryview(arr).cast("B")[r0 * 2 * ntrks: r1 * 2 * ntrks]
        pos, done = off + (lo + r0) * 2 * ntrks, 0
        while done < len(view):
            got = os.preadv(fd, [view[done: done + (1 << 30)]], pos + done)
            if got <= 0:
                raise IOError("short read")
            done += got
        marks = np.flatnonzero(arr[r0:r1, 0] == tbin.END_MARK)
        return r0 + int(marks[0]) if marks.size else -1

    def read_rows(dst, lo, end):
        """rows [lo, end) of the payload -> the pinned tensor dst (positional reads: safe beside the other thread's), in `read_threads`
        pieces side by side (from the page cache one thread copies ~13 GB/s).  Returns the first end-marker row (relative to lo) or -1."""
        t0 = time.perf_counter()
        arr = dst.numpy()
        n = end - lo
        if readers is None or n * 2 * ntrks < (8 << 20):
            first = read_span(arr, 0, n, lo)
        else:
            step = -(-n // read_threads)
            futs = [readers.submit(read_span, arr, a, min(n, a + step), lo) for a in range(0, n, step)]
            hits = [h for h in (f.result() for f in futs) if h >= 0]
            first = min(hits) if hits else -1
        t_read[0] += time.perf_counter() - t0
        return first

    def read_window(k):
        lo, hi = spans[k]
        end = min(nrows, hi + halo_rows)
        first = read_rows(pinned[k % NP], lo, end)
        with end_lock:                                    # (the reader thread writes it for window k + 2 while the main thread clamps window k with it)
            if first >= 0:                                # an end marker inside the payload: the tape ends there
                data_end[0] = min(data_end[0], lo + first)
            return min(end, data_end[0])

    def launch(k, end):
        lo, hi = spans[k]
        hi = min(hi, data_end[0])
        if hi <= lo:
            return None, None, lo
        if busy[k % depth] is not None:                   # the replay that still reads this device window
            busy[k % depth].result()
            busy[k % depth] = None
        with torch.cuda.stream(copy_stream):
            dwin[k % depth][: end - lo].copy_(pinned[k % NP][: end - lo], non_blocking=True)
            copied[k % NP].record(copy_stream)
        scan_stream.wait_event(copied[k % NP])
        piece = dwin[k % depth][: end - lo]
        fin = pipeline.scan_fragment(fes[k & 1], piece, hi - lo, lo, lo == 0, hi >= data_end[0], stream=scan_stream.cuda_stream)
        return piece, fin, end

    busy = [None] * depth
    retired = []                                          # scan contexts replaced by ones with a calibrated screen floor (closed at the end: a scan of theirs may still be in flight)
    floor_state = {"tries": 2, "floor": None}
    pieces = []                                           # per window: bytes, or the future that returns (bytes, replay stats, seconds)

    def replay(k, res, piece, lo, bound, start=None, tag=""):
        t0 = time.perf_counter()
        frag = f"{tap_path}.frag{k}{tag}"
        if start is None:
            start = 0 if lo == 0 else int(res.bursts[0]["zone_first"])
        st = pipeline.decode_fragment(hdr, cfg, fe_exact, res, piece, lo, start, bound, frag, full, opts, exact_lock=exact_lock if pool else None)
        with open(frag, "rb") as g:
            data = g.read()
        os.remove(frag)
        return data, st, time.perf_counter() - t0

    stats = dict(rows=nrows, windows=len(spans), halo_rows_read=0, blocks=0, tapemarks=0, events_delivered=0, exact_scans=0, retries=0, replay_threads=nthreads)
    t_replay = t_wait = 0.0
    for f in fes:                                         # set-up, like the pinned buffers and the device windows: the scan contexts' workspaces
        f._buffers(cap)
    if hasattr(fe.backend, "gather_lists"):               # (and the first use of the device-side packing: PyTorch loads its kernels then)
        one = np.ones(1, np.int64)
        fe.backend.gather_lists(fe._buffers(cap)["events"], 0 * one, 4 * one, 2 * one, 3, frontend.EVENT_DTYPE)
    torch.cuda.synchronize(dev)
    t_start = time.perf_counter()
    total = 0
    try:
        with open(tap_path, "wb") as tapf:
            # software pipeline: while window k is fetched and handed to its replay, window k + 1 is being copied and scanned (queued
            # behind k on the device) and window k + 2 is being read
            pend = {0: launch(0, read_window(0))} if spans else {}
            reader, nxt, reading = None, [0], -1

            def start_read(kk):
                def work():
                    nxt[0] = read_window(kk)
                th = threading.Thread(target=work)
                th.start()
                return th
            if len(spans) > 1:
                reader, reading = start_read(1), 1
            for k, (lo, hi) in enumerate(spans):
                if lo >= data_end[0]:
                    break
                hi = min(hi, data_end[0])
                if reader is not None and reading == k + 1:     # window k + 1 has been read: queue its copy and scan behind window k's
                    reader.join()
                    reader = None
                    pend[k + 1] = launch(k + 1, nxt[0])          # (its device window and scan context were window k - 1's: fetched and replayed, or waited for in launch)
                    if k + 2 < len(spans):
                        reader, reading = start_read(k + 2), k + 2      # (its pinned buffer was window k - 1's: copied long ago)
                piece, fin, end_k = pend.pop(k)
                if fin is None:
                    continue
                t0 = time.perf_counter()
                res, nb, bound = fin()
                t_wait += time.perf_counter() - t0
                if floor_state["tries"] > 0 and cfg.peak_detection_floor_applies():
                    # The candidate screen is built for the loosest thresholds any AGC state could ask for - a learned peak height of 1 V -, and on a
                    # noisy tape every wiggle above THAT becomes a record (lists outgrow their slots, the bursts are redone on the samples).  The first
                    # windows say how high the tape's peaks really are: the scans behind them screen against half the smallest height a chain learned
                    # (a later chain below that floor is flagged RTFE_F_SCREEN_UNDERFLOW and rescanned exactly: slower, never wrong).
                    floor_state["tries"] -= 1
                    st_k = fes[k & 1].scan_stats(res)
                    if st_k["bursts"] > 0 and st_k["redone"] * 4 > st_k["bursts"] and st_k.get("min_learned_height"):
                        import dataclasses
                        cfg2 = dataclasses.replace(cfg, screen_floor_height=min(4.0, 0.5 * st_k["min_learned_height"]))
                        retired.extend(fes)
                        fes[:] = [frontend.FrontEnd(cfg2, device=device) for _ in range(2)]
                        for f in fes: f._buffers(cap)
                        floor_state["tries"] = 0; floor_state["floor"] = cfg2.screen_floor_height
                halo = halo_rows
                while nb is None and end_k < data_end[0]:    # the last own burst runs past the halo: read more (rare; synchronous)
                    halo *= 4
                    stats["retries"] += 1
                    end_k = min(data_end[0], hi + halo)
                    host = torch.empty((end_k - lo, ntrks), dtype=torch.int16, pin_memory=True)
                    first = read_rows(host, lo, end_k)
                    if first >= 0:                            # an end marker inside the longer halo: the tape ends there (src/readtape.c:1410)
                        with end_lock:
                            data_end[0] = min(data_end[0], lo + first)
                        end_k = min(end_k, data_end[0])
                    piece = host[: end_k - lo].to(dev)
                    res, nb, bound = pipeline.scan_fragment(fes[k & 1], piece, hi - lo, lo, lo == 0, hi >= data_end[0])()
                stats["halo_rows_read"] += end_k - hi
                if res.nbursts:
                    if pool:
                        # sub-fragments: cut at the zone starts of evenly spaced bursts
                        nsub = max(1, min(int(replay_split), res.nbursts // 8))
                        cuts = [int(res.bursts[(res.nbursts * j) // nsub]["zone_first"]) for j in range(1, nsub)]
                        first = 0 if lo == 0 else int(res.bursts[0]["zone_first"])
                        starts, stops = [first] + cuts, cuts + [bound]
                        futs = [pool.submit(replay, k, res, piece, lo, stops[j], starts[j], f".{j}") for j in range(nsub) if stops[j] is None or stops[j] > starts[j]]
                        busy[k % depth] = _All(futs)
                        pieces.extend(futs)
                    else:
                        pieces.append(replay(k, res, piece, lo, bound))
            if reader is not None:
                reader.join()
            for pc in pieces:                                 # in window order
                data, st, secs = pc.result() if hasattr(pc, "result") else pc
                t_replay += secs
                tapf.write(data)
                total += len(data)
                for key in ("blocks", "tapemarks", "events_delivered", "exact_scans"):
                    stats[key] += int(st[key])
            if total > 0:
                tapf.write(b"\xff\xff\xff\xff")                   # src/readtape.c:1885
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t_start
    finally:                                              # (also when a replay raised: no thread, file or device context is left behind)
        if readers:
            readers.shutdown(wait=True)
        os.close(fd)
        if pool:
            pool.shutdown(wait=True, cancel_futures=True)
        fe_exact.close()
        for f in list(fes) + retired:
            f.close()
    stats.update(setup_seconds=t_start - t_enter, rows=data_end[0], seconds=dt, msamples_per_s=data_end[0] / dt / 1e6, replay_seconds=t_replay, read_seconds=t_read[0], scan_wait_seconds=t_wait,
                 replay_events_per_s=(stats["events_delivered"] / t_replay) if t_replay > 0 else None, tap_bytes=total + (4 if total else 0), screen_floor_height=floor_state["floor"])
    return stats
