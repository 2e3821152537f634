"""Streaming ingest: a .tbin file larger than the device window is decoded window by window.

Replaces the reference's read loop (two fread()s per row into a stack buffer, src/readtape.c:1405-1414) with

    disk --(reader thread, readinto)--> pinned host buffer --(hipMemcpyAsync, copy stream)--> device window
         --(rtfe_scan, compute stream)--> events --(host replay of the window's own bursts)--> piece of the .tap

A software pipeline of threads (the interpreter lock is the scarce resource: whatever runs per row or per event runs in native code):
  - the producer reads window after window into a ring of pinned buffers (rt_read_mt: native threads, positional reads) and queues each
    window's upload (copy stream) and, behind it on the scan stream, rtfe_find_end_mark (the reader's end-of-data check, on the device),
    rtfe_scan, rtfe_pack_events and the copies of the small tables into page-locked memory - `scan_contexts` windows are in flight;
  - a fetcher per context waits for its window's event, copies the packed event lists to the host and hands the window to the replay pool;
  - a replay task per window runs its sub-fragments on native threads (rt_replay_run_fragments) and collects their pieces of the .tap;
  - the caller's thread concatenates the pieces in window order.
A window is a FRAGMENT in the sense of pipeline.decode_fragment: it carries a halo of the following rows, owns the bursts whose zone ends
inside it, and its .tap piece concatenates with its neighbours' (DESIGN.md 6).  PyTorch supplies the pinned allocation (hipHostMalloc), the
streams and the events; nothing here computes on the CPU except the block decoders.
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
                          device="cuda:0", replay_threads: int = 1, read_threads: int = 4, replay_split: int = 1, scan_contexts: int = 3, ramp: bool = True):
    """Decodes the .tbin file `path` to the SIMH file `tap_path` through device windows of `window_rows` rows (a multiple of 1024).
    Returns statistics incl. the end-to-end rate (disk -> .tap), the time spent in the host replay and the rows the halos re-read.
    replay_threads > 1: the windows' host replays run side by side (fragments are independent: each has its own decoder context
    and writes its own piece of the .tap; the pieces are concatenated in window order).  A window's device buffer is kept until
    its replay is done (exact rescans read it), so replay_threads + 2 device windows are held.
    replay_split > 1: a window's bursts are replayed as that many sub-fragments side by side (cut at burst boundaries, exactly as the
    windows themselves are); the last two windows - what the pipeline drains into - as four times as many.
    scan_contexts: scans in flight (a context = a front end with its workspace and output buffers; a context is free again once its window's
    results have been fetched).  ramp: the first two windows are a quarter and a half of `window_rows` (the pipeline fills sooner)."""
    import queue
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
    nctx = max(2, int(scan_contexts))
    fes = [frontend.FrontEnd(cfg, device=device) for _ in range(nctx)]      # window k is fetched while k + 1 .. are copied and scanned
    fe = fes[0]
    dev = torch.device(device)
    spans, lo = [], 0
    sizes = [window_rows // 4 & ~1023, window_rows // 2 & ~1023] if (ramp and window_rows >= (1 << 20) and nrows > 2 * window_rows) else []
    while lo < nrows:
        w = sizes.pop(0) if sizes else window_rows
        if ramp and not sizes and window_rows >= (1 << 20) and len(spans) >= 2 and nrows - lo <= 2 * window_rows and nrows - lo > (1 << 20):
            # ... and drains sooner behind short last windows: what follows the last upload - its scan, its fetch, its replay - is in nobody's shadow.  The last
            # two windows' worth of rows go as a half, a quarter and two eighths of what is left.
            rest = nrows - lo
            for frac in (2, 4, 8):
                w2 = max(1 << 18, rest // frac & ~1023)
                if lo + w2 < nrows:
                    spans.append((lo, lo + w2)); lo += w2
            spans.append((lo, nrows)); lo = nrows
            break
        spans.append((lo, min(nrows, lo + w)))
        lo += w
    cap = window_rows + halo_rows
    from concurrent.futures import ThreadPoolExecutor
    nthreads = max(1, int(replay_threads))
    depth = max(2 if nthreads == 1 else nthreads + 2, nctx + 2)
    NP = nctx + 1                                         # pinned buffers: one being read into, the others being copied from / waiting for their context
    pinned = [torch.empty((cap, ntrks), dtype=torch.int16, pin_memory=True) for _ in range(NP)]      # hipHostMalloc
    dwin = [torch.empty((cap, ntrks), dtype=torch.int16, device=dev) for _ in range(depth)]
    pool = ThreadPoolExecutor(max(2, -(-nthreads // max(1, int(replay_split))))) if nthreads > 1 else None      # a task is a window: its sub-fragments are native threads
    read_threads = max(1, int(read_threads))
    readers = ThreadPoolExecutor(read_threads) if read_threads > 1 else None
    fe_exact = frontend.FrontEnd(cfg, device=device)       # exact rescans of the replays: their own context, one at a time
    exact_lock = threading.Lock()
    copy_stream, scan_stream = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    copied = [torch.cuda.Event() for _ in range(NP)]
    fd = os.open(path, os.O_RDONLY)
    dlib = pipeline._load_decode_lib()
    t_read = [0.0]
    trace = [] if os.environ.get("RT_INGEST_TRACE") else None      # (stage, window, start, end) in seconds since the timed region began
    t_base = [0.0]

    def mark(stage, k, t0):
        if trace is not None:
            trace.append((stage, k, round(t0 - t_base[0], 5), round(time.perf_counter() - t_base[0], 5)))
    data_end = [nrows]
    end_lock = threading.Lock()

    gpu_mark = hasattr(fe.backend, "pinned") and os.environ.get("RTFE_PACK_EVENTS") != "0" and not os.environ.get("RT_INGEST_HOST_MARK")
    for f in fes:                                          # the end-of-data marker is looked for on the device, in front of each window's scan (rtfe_find_end_mark)
        f.pack_on_scan, f.find_end_mark = True, gpu_mark

    def read_span(arr, r0, r1, lo, check=True):
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
        if not check:
            return -1
        marks = np.flatnonzero(arr[r0:r1, 0] == tbin.END_MARK)
        return r0 + int(marks[0]) if marks.size else -1

    def read_rows(dst, lo, end, check=True):
        """rows [lo, end) of the payload -> the pinned tensor dst (positional reads: safe beside the other thread's), in `read_threads`
        pieces side by side (from the page cache one thread copies ~13 GB/s).  Returns the first end-marker row (relative to lo) or -1."""
        t0 = time.perf_counter()
        arr = dst.numpy()
        n = end - lo
        if not check:                                      # (the end marker is looked for on the device: the read is one call into native threads)
            if dlib.rt_read_mt(fd, arr.ctypes.data, off + lo * 2 * ntrks, n * 2 * ntrks, read_threads) != 0:
                raise IOError("short read")
            first = -1
        elif readers is None or n * 2 * ntrks < (8 << 20):
            first = read_span(arr, 0, n, lo, check)
        else:
            step = -(-n // read_threads)
            futs = [readers.submit(read_span, arr, a, min(n, a + step), lo, check) for a in range(0, n, step)]
            hits = [h for h in (f.result() for f in futs) if h >= 0]
            first = min(hits) if hits else -1
        t_read[0] += time.perf_counter() - t0
        return first

    def read_window(k):
        lo, hi = spans[k]
        end = min(nrows, hi + halo_rows)
        t0 = time.perf_counter()
        first = read_rows(pinned[k % NP], lo, end, check=not gpu_mark)
        mark("read", k, t0)
        with end_lock:                                    # (the producer writes it while the main thread clamps a window with it)
            if first >= 0:                                # an end marker inside the payload: the tape ends there
                data_end[0] = min(data_end[0], lo + first)
            return min(end, data_end[0])

    busy = [None] * depth
    ctx_free = [threading.Semaphore(1) for _ in range(nctx)]
    stop = threading.Event()

    def launch(k, end):
        lo, hi = spans[k]
        hi = min(hi, data_end[0])
        if hi <= lo:
            return None, None, lo
        t0 = time.perf_counter()
        ctx_free[k % nctx].acquire()                      # the results of the window that used this context have been fetched
        if busy[k % depth] is not None:                   # the replay that still reads this device window
            busy[k % depth].result()
            busy[k % depth] = None
        mark("wait_ctx", k, t0)
        t0 = time.perf_counter()
        with torch.cuda.stream(copy_stream):
            if trace is not None:
                gpu_ev[("copy0", k)] = torch.cuda.Event(enable_timing=True); gpu_ev[("copy0", k)].record(copy_stream)
            dwin[k % depth][: end - lo].copy_(pinned[k % NP][: end - lo], non_blocking=True)
            copied[k % NP].record(copy_stream)
            if trace is not None:
                gpu_ev[("copy1", k)] = torch.cuda.Event(enable_timing=True); gpu_ev[("copy1", k)].record(copy_stream)
        mark("l_h2d", k, t0)
        scan_stream.wait_event(copied[k % NP])
        piece = dwin[k % depth][: end - lo]
        fin = pipeline.scan_fragment(fes[k % nctx], piece, hi - lo, lo, lo == 0, hi >= data_end[0], stream=scan_stream.cuda_stream)
        if trace is not None:
            gpu_ev[("scan1", k)] = torch.cuda.Event(enable_timing=True); gpu_ev[("scan1", k)].record(scan_stream)
        mark("launch", k, t0)
        if trace is not None and getattr(fin.res, "launch_times", None):
            lt = fin.res.launch_times
            for name, a, b in (("l_scan", lt[0], lt[1]), ("l_pack", lt[1], lt[2]), ("l_mirror", lt[2], lt[3])) if len(lt) == 4 else ():
                trace.append((name, k, round(a - t_base[0], 5), round(b - t_base[0], 5)))
        return piece, fin, end

    launched = queue.Queue()

    def producer():
        """reads window after window into the ring of pinned buffers and queues each one's copy and scan behind its predecessor's"""
        try:
            for k, (lo, hi) in enumerate(spans):
                if stop.is_set() or lo >= data_end[0]:
                    break
                if k >= NP:
                    copied[k % NP].synchronize()           # the pinned buffer's last copy (window k - NP) has left it
                end = read_window(k)
                if stop.is_set():
                    break
                item = launch(k, end)
                if item[1] is None:
                    marks_seen[k].set()
                launched.put((k, fetchers.submit(stage2, k, item) if item[1] is not None else None))
            launched.put((None, None))
        except BaseException as e:                         # the main thread raises it
            launched.put((None, e))
        finally:
            for ev in marks_seen:                          # (windows that were never launched hold nobody up)
                if stop.is_set():
                    ev.set()

    retired = []                                          # scan contexts replaced by ones with a calibrated screen floor (closed at the end: a scan of theirs may still be in flight)
    floor_state = {"tries": 2, "floor": None}
    pieces = []                                           # per window: bytes, or the future that returns (bytes, replay stats, seconds)

    def replay(k, res, piece, lo, bound, start=None, tag=""):
        t0 = time.perf_counter()
        frag = f"{tap_path}.frag{k}{tag}"
        if start is None:
            start = 0 if lo == 0 else int(res.bursts[0]["zone_first"])
        st = pipeline.decode_fragment(hdr, cfg, fe_exact, res, piece, lo, start, bound, frag, full, opts, exact_lock=exact_lock)      # (always: the fetcher threads replay side by side on ONE exact-rescan handle - ADVICE r5)
        with open(frag, "rb") as g:
            data = g.read()
        os.remove(frag)
        mark("replay" + tag, k, t0)
        return data, st, time.perf_counter() - t0

    def replay_window(k, res, piece, lo, starts, stops):
        """-> [(bytes, replay statistics, seconds), ...] of the window's sub-fragments, in order"""
        t0 = time.perf_counter()
        frags = [f"{tap_path}.frag{k}.{j}" for j in range(len(starts))]
        sts = pipeline.decode_fragments(hdr, cfg, fe_exact, res, piece, lo, starts, stops, frags, full, opts, exact_lock=exact_lock)
        out = []
        for frag, (st, secs) in zip(frags, sts):
            with open(frag, "rb") as g:
                out.append((g.read(), st, secs))
            os.remove(frag)
        mark("replay", k, t0)
        return out

    stats = dict(rows=nrows, windows=len(spans), halo_rows_read=0, blocks=0, tapemarks=0, events_delivered=0, exact_scans=0, retries=0, replay_threads=nthreads)
    t_replay = 0.0
    for f in fes:                                         # set-up, like the pinned buffers and the device windows: the scan contexts' workspaces
        f._pack_buffers(f._buffers(cap), f._buffers(cap)["cap"] // 2)
    # (and page-locked blocks for the windows' event lists in the host allocator's cache.  A window's packed events are ~1 / 40 of its arena - C2: 27 MB of events in
    #  31 - 35 MB of packed lists -, the allocator's blocks are powers of two, and a request one byte over 32 MB is a hipHostMalloc of 64 MB in the middle of the
    #  pipeline: ~12 ms during which every thread that talks to the runtime waits (round 6's trace: the producer's event wait, the fetchers).  Blocks of the NEXT power of two.)
    blk = [torch.empty(max(1 << 20, cap * ntrks * 16 // 20), dtype=torch.uint8, pin_memory=True) for _ in range(min(depth, 12))]
    del blk
    torch.cuda.synchronize(dev)
    gpu_ev = {}
    if trace is not None:
        gpu_ev["base"] = torch.cuda.Event(enable_timing=True); gpu_ev["base"].record(copy_stream); copy_stream.synchronize()
    t_start = time.perf_counter()
    t_base[0] = t_start
    total = 0
    calib_lock = threading.Lock()
    marks_seen = [threading.Event() for _ in spans]
    fetchers = ThreadPoolExecutor(nctx)                    # a window's results are fetched (and its replays queued) beside its neighbours': a fetch is ~6 ms of host-side work
    fetch_streams = [torch.cuda.Stream(dev) for _ in range(nctx)]
    wait_s = [0.0]

    def stage2(k, item):
        """window k: its results to the host, its bursts to the replay threads -> the replays' futures (or results), in order"""
        piece, fin, end_k = item
        out = []
        try:
            lo, hi = spans[k]
            hi = min(hi, data_end[0])
            t0 = time.perf_counter()
            with torch.cuda.stream(fetch_streams[k % nctx]):
                res, nb, bound = fin()
            wait_s[0] += time.perf_counter() - t0
            mark("fetch", k, t0)
            if gpu_mark:
                # the window's end-of-data check came back with its tables.  The windows are looked at in order (a marker ends the tape for every window
                # behind it); a window that holds one - rows behind the data's end were scanned with it - is scanned again up to the marker.
                if k > 0:
                    marks_seen[k - 1].wait()
                try:
                    assert res.end_mark_valid
                    if lo >= data_end[0]:
                        return []
                    if res.end_mark is not None and res.end_mark < end_k - lo:
                        with end_lock:
                            data_end[0] = min(data_end[0], lo + res.end_mark)
                        end_k = min(end_k, data_end[0])
                        hi = min(hi, data_end[0])
                        if hi <= lo:
                            return []
                        piece = piece[: end_k - lo]
                        with torch.cuda.stream(fetch_streams[k % nctx]):
                            # (a marker inside this window's HALO leaves the window its own rows [lo, hi) only: it keeps its bounding burst, and the
                            #  window behind it decodes [hi, data_end) - ADVICE r5: is_last = True there decoded that range twice)
                            res, nb, bound = pipeline.scan_fragment(res.fe, piece, hi - lo, lo, lo == 0, hi >= data_end[0])()
                finally:
                    marks_seen[k].set()
            if trace is not None and getattr(res, "fetch_times", None):
                ft = res.fetch_times
                for name, a, b in (("f_wait", ft[0], ft[1]), ("f_tables", ft[1], ft[2]), ("f_events", ft[2], ft[3])):
                    trace.append((name, k, round(a - t_base[0], 5), round(b - t_base[0], 5)))
            if floor_state["tries"] > 0 and cfg.peak_detection_floor_applies():
                # The candidate screen is built for the loosest thresholds any AGC state could ask for - a learned peak height of 1 V -, and on a
                # noisy tape every wiggle above THAT becomes a record (lists outgrow their slots, the bursts are redone on the samples).  The first
                # windows say how high the tape's peaks really are: the scans behind them screen against half the smallest height a chain learned
                # (a later chain below that floor is flagged RTFE_F_SCREEN_UNDERFLOW and rescanned exactly: slower, never wrong).
                with calib_lock:
                    if floor_state["tries"] > 0:
                        floor_state["tries"] -= 1
                        st_k = res.fe.scan_stats(res)
                        if floor_state["floor"] is None:      # (what the window's own first scan estimated from its samples, k_scan_begin)
                            floor_state["floor"] = st_k.get("screen_floor_used")
                        if st_k["bursts"] > 0 and st_k["redone"] * 4 > st_k["bursts"] and st_k.get("min_learned_height"):
                            import dataclasses
                            cfg2 = dataclasses.replace(cfg, screen_floor_height=min(4.0, 0.5 * st_k["min_learned_height"]))
                            new = [frontend.FrontEnd(cfg2, device=device) for _ in range(nctx)]
                            for f in new: f._pack_buffers(f._buffers(cap), f._buffers(cap)["cap"] // 2); f.pack_on_scan, f.find_end_mark = True, gpu_mark
                            retired.extend(fes)
                            fes[:] = new                     # (the producer takes a window's context from the list when it launches it)
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
                res, nb, bound = pipeline.scan_fragment(res.fe, piece, hi - lo, lo, lo == 0, hi >= data_end[0])()
            stats["halo_rows_read"] += end_k - hi
            if res.nbursts:
                if pool:
                    # sub-fragments: cut at the zone starts of evenly spaced bursts; one task (a Python thread) per window, native threads for its sub-fragments
                    split = int(replay_split) * (4 if k + 3 >= len(spans) else (2 if k + 5 >= len(spans) else 1))      # (the last windows' replays have the threads to themselves)
                    nsub = max(1, min(split, nthreads, res.nbursts // 8))
                    cuts = [int(res.bursts[(res.nbursts * j) // nsub]["zone_first"]) for j in range(1, nsub)]
                    first = 0 if lo == 0 else int(res.bursts[0]["zone_first"])
                    starts, stops = [first] + cuts, cuts + [bound]
                    keep = [j for j in range(nsub) if stops[j] is None or stops[j] > starts[j]]
                    out = [pool.submit(replay_window, k, res, piece, lo, [starts[j] for j in keep], [stops[j] for j in keep])]
                    busy[k % depth] = _All(out)
                else:
                    out = [replay(k, res, piece, lo, bound)]
        finally:
            marks_seen[k].set()
            ctx_free[k % nctx].release()
        return out

    prod = threading.Thread(target=producer, name="rt-ingest-producer")
    import sys
    switch = sys.getswitchinterval()
    import gc
    gc_was = gc.isenabled()
    if not os.environ.get("RT_INGEST_GC"):
        gc.disable()                                      # a full collection walks every object of the process (PyTorch is loaded: tens of ms) with the interpreter lock held - every stage stalls at once
    sys.setswitchinterval(float(os.environ.get("RT_INGEST_SWITCH", "0.0002")))      # a dozen threads hand the interpreter lock around between system calls: at the default 5 ms a stage waits longer for the lock than it works
    try:
        with open(tap_path, "wb") as tapf:
            # software pipeline: the producer thread reads window after window and queues each one's copy and scan on the device; the fetchers bring
            # the windows' results to the host and hand them to the replay threads; this thread collects the pieces in window order
            prod.start()
            while True:
                k, fut = launched.get()
                if k is None:
                    if isinstance(fut, BaseException):
                        raise fut
                    break
                if fut is not None:
                    pieces.extend(fut.result())
            for pc in pieces:                                 # in window order
                got = pc.result() if hasattr(pc, "result") else [pc]
                for data, st, secs in got:
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
        sys.setswitchinterval(switch)
        if gc_was:
            gc.enable()
        stop.set()
        for sem in ctx_free:                              # (a producer waiting for a context wakes up and sees the stop flag)
            sem.release()
        if prod.is_alive():
            prod.join()
        fetchers.shutdown(wait=True)
        if readers:
            readers.shutdown(wait=True)
        os.close(fd)
        if pool:
            pool.shutdown(wait=True, cancel_futures=True)
        torch.cuda.synchronize(dev)
        fe_exact.close()
        for f in list(fes) + retired:
            f.close()
    if trace is not None:
        for k in range(len(spans)):                       # the device's own clock: a window's copy and the end of its scan, ms behind the start of the timed region
            if ("copy0", k) in gpu_ev:
                c0, c1, s1 = (gpu_ev["base"].elapsed_time(gpu_ev[(n, k)]) for n in ("copy0", "copy1", "scan1"))
                trace.append(("gpu_copy", k, c0 / 1e3, c1 / 1e3)); trace.append(("gpu_scan_done", k, c1 / 1e3, s1 / 1e3))
        stats["trace"] = sorted(trace, key=lambda x: x[2])
    stats.update(setup_seconds=t_start - t_enter, rows=data_end[0], seconds=dt, msamples_per_s=data_end[0] / dt / 1e6, replay_seconds=t_replay, read_seconds=t_read[0], scan_wait_seconds=wait_s[0],
                 replay_events_per_s=(stats["events_delivered"] / t_replay) if t_replay > 0 else None, tap_bytes=total + (4 if total else 0), screen_floor_height=floor_state["floor"])
    return stats
