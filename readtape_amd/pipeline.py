"""End-to-end decode: TBIN rows -> HIP front end (librtfe.so) -> event replay + block decoders
(librtdecode.so) -> SIMH .tap.  The counterpart of the reference's `process_file()` for one tape
(src/readtape.c:1564-1889), with `readblock()` replaced by the device scan + replay.

The analog front end runs only on the GPU; the host code below never looks at a sample."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import frontend, tbin

HERE = os.path.dirname(os.path.abspath(__file__))


class _Options(C.Structure):          # struct rt_options (csrc/host/rt_decode.h)
    _fields_ = [("mode", C.c_int), ("ntrks", C.c_int), ("bpi", C.c_float), ("ips", C.c_float),
                ("specified_parity", C.c_int), ("revparity", C.c_int), ("do_correction", C.c_int),
                ("find_zeros", C.c_int), ("do_differentiate", C.c_int), ("multiple_tries", C.c_int),
                ("tap_format", C.c_int), ("add_parity", C.c_int), ("verbose", C.c_int),
                ("ww_fluxdir", C.c_int), ("ww_reverse", C.c_int), ("ww_order", C.c_char * 24)]


class _Parms(C.Structure):            # struct rt_parms
    _fields_ = [("active", C.c_int), ("clk_window", C.c_int), ("clk_alpha", C.c_float), ("agc_window", C.c_int),
                ("agc_alpha", C.c_float), ("min_peak", C.c_float), ("clk_factor", C.c_float), ("pulse_adj", C.c_float),
                ("pkww_bitfrac", C.c_float), ("pkww_rise", C.c_float), ("midbit", C.c_float), ("z1pt", C.c_float),
                ("z2pt", C.c_float), ("tried", C.c_int), ("chosen", C.c_int)]


class _Stats(C.Structure):            # struct rt_replay_stats
    _fields_ = [("attempts", C.c_int64), ("exact_scans", C.c_int64), ("chained", C.c_int64),
                ("events_delivered", C.c_int64), ("agc_mismatches", C.c_int64),
                ("blocks", C.c_int32), ("tapemarks", C.c_int32), ("blocks_with_errors", C.c_int32),
                ("blocks_with_warnings", C.c_int32), ("blocks_unusable", C.c_int32), ("all_ok", C.c_int32),
                ("data_bytes", C.c_int64), ("device_failures", C.c_int64),
                ("reference_fatal", C.c_int64), ("fatal_row", C.c_int64), ("fatal_trk", C.c_int64)]


_EXACT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.POINTER(C.c_uint32),
                        C.POINTER(C.c_void_p), C.POINTER(C.c_uint32))
_FREE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
_WW_SCAN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32)


_DECODE_LIB = None


def _load_decode_lib():
    global _DECODE_LIB
    if _DECODE_LIB is None:
        _DECODE_LIB = _open_decode_lib()
    return _DECODE_LIB


def _open_decode_lib():
    path = os.path.join(HERE, "librtdecode.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    lib.rt_default_parmsets.argtypes = [C.c_int, C.POINTER(_Parms)]
    lib.rt_parse_parms_text.argtypes = [C.c_int, C.c_char_p, C.POINTER(_Parms)]
    lib.rt_replay_run.argtypes = [C.POINTER(_Options), C.POINTER(_Parms), C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                  C.POINTER(C.c_int), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, _EXACT_FN, _FREE_FN, C.c_void_p,
                                  C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(_Stats)]
    lib.rt_replay_run_after_deskew.argtypes = lib.rt_replay_run.argtypes
    lib.rt_replay_run_ww.argtypes = [C.POINTER(_Options), C.POINTER(_Parms), C.c_int64, C.c_int64, C.c_int64, C.c_int, _WW_SCAN_FN, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(_Stats), C.c_int, C.POINTER(C.c_int)]
    lib.rt_replay_run_named.argtypes = lib.rt_replay_run.argtypes[:15] + [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(_Stats)]
    lib.rt_replay_run_fragment.argtypes = lib.rt_replay_run.argtypes + [C.c_int64, C.c_int64]
    lib.rt_replay_run_fragments.argtypes = lib.rt_replay_run.argtypes[:15] + [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(_Stats), C.POINTER(C.c_double)]
    lib.rt_read_mt.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int]
    lib.rt_replay_deskew.argtypes = lib.rt_replay_run.argtypes[:15] + [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.rt_replay_density.argtypes = lib.rt_replay_run.argtypes[:15] + [C.c_char_p, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return lib


@dataclass
class DecodeOptions:
    """The reference's command-line switches that matter after the front end (src/readtape.c:936-1022)."""
    multiple_tries: bool = False      # -m
    correct: bool = False             # -correct
    even_parity: bool = False         # -even
    revparity: int = 0                # -revparity=n
    verbose: bool = True              # -v
    nparmsets: int | None = None      # how many built-in parameter sets to scan (default: 1, or all with -m)


def default_parmsets(mode, n):
    lib = _load_decode_lib()
    arr = (_Parms * 15)()
    lib.rt_default_parmsets(mode, arr)
    return [arr[i] for i in range(15) if arr[i].active][:n]


def frontend_parmsets(full):
    """The front-end half of full parameter sets."""
    return [(p.pkww_bitfrac, p.pkww_rise, p.min_peak, p.agc_alpha, p.agc_window, p.clk_factor) for p in full]


def _exact_callbacks(fe, rows, ntrks, fe_factory=None, lock=None):
    """The callbacks through which the host replay asks for an exact device scan of one attempt (rtfe_scan_exact)."""
    import contextlib
    import dataclasses
    keep = {}
    big = []                                          # a front end with room for an event every other row (noise): built on demand

    def exact(user, reset_row, end_row, parmset, burst_out, counts_out, events_out, cap_out):
        with (lock or contextlib.nullcontext()):
            return exact_locked(reset_row, end_row, parmset, burst_out, counts_out, events_out, cap_out)

    def exact_locked(reset_row, end_row, parmset, burst_out, counts_out, events_out, cap_out):
        try:
            ex = fe.scan_exact(rows, reset_row, end_row, parmset_mask=1 << parmset).fetch()
            if int(ex.bursts[0]["flags"]) & frontend.F_SCREEN_UNDERFLOW:
                ex = fe.scan_exact(rows, reset_row, end_row, parmset_mask=1 << parmset, screen_off=True).fetch()
            if int(ex.bursts[0]["flags"]) & frontend.F_EVENT_OVERFLOW:
                # far more events than a recording holds (e.g. noise above the dead band on a differentiated signal): the
                # reference plods through them, so does the device - with event regions sized for the worst case
                if not big:
                    big.append((fe_factory or frontend.FrontEnd)(dataclasses.replace(fe.cfg, events_per_sample_cap=0.6)))
                under = bool(int(ex.bursts[0]["flags"]) & frontend.F_SCREEN_UNDERFLOW)
                ex = big[0].scan_exact(rows, reset_row, end_row, parmset_mask=1 << parmset, screen_off=under).fetch()
                if int(ex.bursts[0]["flags"]) & frontend.F_SCREEN_UNDERFLOW and not under:
                    ex = big[0].scan_exact(rows, reset_row, end_row, parmset_mask=1 << parmset, screen_off=True).fetch()
            B = ex.bursts[0]
            if int(B["flags"]) & (frontend.F_EVENT_OVERFLOW | frontend.F_DETECTOR_FATAL):
                return 1
            cap = int(B["event_cap"])
            base = int(B["event_base"]) + parmset * ntrks * cap
            ev = np.ascontiguousarray(ex._events[base: base + ntrks * cap])
            C.memmove(burst_out, B.tobytes(), frontend.BURST_DTYPE.itemsize)
            for t in range(ntrks):
                counts_out[t] = int(ex.counts[0, parmset, t])
            events_out[0] = ev.ctypes.data
            cap_out[0] = cap
            keep[ev.ctypes.data] = ev
            return 0
        except Exception:
            return 2

    def free(user, ptr):
        keep.pop(ptr, None)

    keep["__big__"] = big
    return _EXACT_FN(exact), _FREE_FN(free), keep


def _release(keep):
    """Ends what the exact-scan callbacks built on demand (the worst-case front end holds device buffers)."""
    for b in keep.pop("__big__", []):
        b.close()
    keep.clear()


def calibrate_deskew(hdr, rows, full, o, fe_factory, invert=False, find_zeros=False, differentiate=False,
                     log_path=None, evt_path=None, first_prefix_rows=1 << 22, append=False, **cfgkw):
    """The -deskew pre-pass (src/readtape.c:1675-1717) on the device front end: scans a prefix of the resident tape with
    NO deskew delays and the first parameter set, lets the host decoders record where each track's transitions fall,
    and returns the per-track delays in samples.  The prefix grows until the reference's stopping rule (1000
    transitions on every track, or 100 blocks) is met inside it, or it is the whole tape."""
    lib = _load_decode_lib()
    nrows = int(rows.shape[0])
    n0 = min(nrows, first_prefix_rows)
    cfg0 = frontend.FrontEndConfig.from_header(hdr, parmsets=frontend_parmsets(full[:1]), skew=None, invert=invert,
                                               find_zeros=find_zeros, differentiate=differentiate, **cfgkw)
    fe0 = (fe_factory or frontend.FrontEnd)(cfg0)
    parr = (_Parms * 1)(full[0])
    W = (C.c_int * 1)(fe0.widths[0])
    while True:
        prefix = rows[:n0]
        res = fe0.scan(prefix).fetch()
        exact, free, keep = _exact_callbacks(fe0, prefix, hdr.ntrks, fe_factory)
        bursts = np.ascontiguousarray(res.bursts)
        counts = np.ascontiguousarray(res.counts)
        delays = (C.c_int * 19)()
        nblks, hit_end = C.c_int(0), C.c_int(0)
        rc = lib.rt_replay_deskew(C.byref(o), parr, 1, hdr.tdelta_ns, hdr.tstart_ns, n0, 0, W,
                                  bursts.ctypes.data, len(bursts), counts.ctypes.data, res._events.ctypes.data,
                                  exact, free, None, log_path.encode() if log_path else None,
                                  evt_path.encode() if evt_path else None, int(append), delays, C.byref(nblks), C.byref(hit_end))
        _release(keep)
        if rc == -3:
            raise ReferenceFatal("a fatal assert of the reference's detector inside the -deskew pre-pass", {"reference_fatal": 1, "events_delivered": 0})
        if rc != 0:
            raise RuntimeError("rt_replay_deskew failed")
        if hit_end.value and n0 < nrows:
            n0 = min(nrows, n0 * 4)
            continue
        if nblks.value < 0:
            raise RuntimeError("deskew: some tracks have no transitions (is ntrks right?)")
        return [int(delays[t]) for t in range(hdr.ntrks)]


def detect_density(hdr, rows, full, o, fe_factory, invert=False, log_path=None, evt_path=None, first_prefix_rows=1 << 22):
    """Density detection (src/readtape.c:1656-1672) when the tape's header carries no bpi: a prefix of the resident tape
    is scanned with the front end in its bpi = 0 configuration (window of 8 samples, no AGC feedback), the host builds
    the histogram of transition distances from the first 9999 of them and picks the standard density.  NRZI only
    (for PE and GCR the reference's own pre-pass degenerates: with a zero clock estimate every track goes idle at each
    peak, src/decoder.c:868,880; give bpi for those)."""
    if hdr.mode != tbin.MODE_NRZI:
        raise NotImplementedError("density detection is built for NRZI only: pass bpi")
    lib = _load_decode_lib()
    nrows = int(rows.shape[0])
    n0 = min(nrows, first_prefix_rows)
    cfg0 = frontend.FrontEndConfig.from_header(hdr, parmsets=frontend_parmsets(full[:1]), skew=None, invert=invert, bpi=0.0)
    fe0 = (fe_factory or frontend.FrontEnd)(cfg0)
    parr = (_Parms * 1)(full[0])
    W = (C.c_int * 1)(fe0.widths[0])
    o0 = _Options.from_buffer_copy(o)
    o0.bpi = 0.0
    while True:
        prefix = rows[:n0]
        res = fe0.scan(prefix).fetch()
        exact, free, keep = _exact_callbacks(fe0, prefix, hdr.ntrks, fe_factory)
        bursts = np.ascontiguousarray(res.bursts)
        counts = np.ascontiguousarray(res.counts)
        bpi, implied, nblks, hit_end = C.c_float(0), C.c_float(0), C.c_int(0), C.c_int(0)
        rc = lib.rt_replay_density(C.byref(o0), parr, 1, hdr.tdelta_ns, hdr.tstart_ns, n0, 0, W,
                                   bursts.ctypes.data, len(bursts), counts.ctypes.data, res._events.ctypes.data,
                                   exact, free, None, log_path.encode() if log_path else None,
                                   evt_path.encode() if evt_path else None, C.byref(bpi), C.byref(implied), C.byref(nblks), C.byref(hit_end))
        _release(keep)
        if rc == -3:
            raise ReferenceFatal("a fatal assert of the reference's detector inside the density pre-pass", {"reference_fatal": 1, "events_delivered": 0})
        if rc != 0:
            raise RuntimeError("rt_replay_density failed")
        if hit_end.value and n0 < nrows and bpi.value >= 0:
            n0 = min(nrows, n0 * 4)
            continue
        if bpi.value < 0:
            raise RuntimeError("density detection met a non-positive transition distance (or too many distinct ones): non-standard input, fatal in the reference too; please specify bpi")
        if bpi.value == 0:
            raise RuntimeError(f"the detected density of {implied.value:.0f} BPI is non-standard; please specify it")
        return float(bpi.value)


class ReferenceFatal(RuntimeError):
    """The input drives the reference into one of its fatal asserts (exit 99): the decode stops where the reference stops."""

    def __init__(self, msg, stats):
        super().__init__(msg)
        self.stats = stats


def decode_tape(hdr, rows, tap_path, log_path=None, opts: DecodeOptions | None = None, fe_factory=None,
                skew=None, invert=False, parms_text: str | None = None, find_zeros=False, evt_path=None, differentiate=False,
                subsample: int = 1, deskew: bool = False, deskew_prefix_rows: int = 1 << 22, trkorder: str | None = None,
                out_base: str | None = None, in_name: str | None = None, tap_format: bool = True, fluxdir: str = "neg", reverse: bool = False):
    """Decodes one tape; returns (stats dict, ScanResult).  `fe_factory(cfg)` builds the front end
    (default: the GPU one; tests/cpu_emul passes the emulated library).
    Output: tap_path = one SIMH .tap file; or out_base = the reference's own naming - <out_base>.tap, or with tap_format=False
    the numbered <out_base>.NNN.bin data files, one per tape file (src/readtape.c:1091-1111) - with the files' creation/closing
    and the end-of-run summary (src/readtape.c:2021-2044; in_name = the input's name in it) in the log."""
    opts = opts or DecodeOptions()
    if hdr.mode == tbin.MODE_WW:                                           # one chain per tape, detector state handed back and forth: its own path
        st = decode_tape_ww(hdr, rows, tap_path, log_path=log_path, order=trkorder, verbose=opts.verbose, evt_path=evt_path, fe_factory=fe_factory,
                            invert=invert, out_base=out_base, in_name=in_name, deskew=deskew, fluxdir=fluxdir, reverse=reverse)      # (-fluxdir, -reverse: Whirlwind only)
        return st, None
    lib = _load_decode_lib()
    if trkorder:                                                           # -order= wins over the header's TBINORD extension (src/readtape.c:1346-1355)
        import dataclasses
        hdr = dataclasses.replace(hdr, trkorder=trkorder)
    if subsample > 1:
        # -subsample=n (src/readtape.c:1407-1414): of every n rows the LAST one is used and the time base is not
        # stretched (timenow_ns still advances by tdelta per used row, :1424) - a strided view of the resident tape
        rows = rows[subsample - 1::subsample][: rows.shape[0] // subsample]
        if hasattr(rows, "contiguous"):
            rows = rows.contiguous()
        else:
            rows = np.ascontiguousarray(rows)
    mode = hdr.mode
    nsets = opts.nparmsets or (15 if opts.multiple_tries else 1)
    if parms_text:
        arr = (_Parms * 15)()
        n = lib.rt_parse_parms_text(mode, parms_text.encode(), arr)
        if n <= 0:
            raise ValueError("bad .parms text")
        full = [arr[i] for i in range(n)][:nsets]
    else:
        full = default_parmsets(mode, nsets)
    bpi_kw = {}
    detected = False
    if hdr.mode != tbin.MODE_GCR and not hdr.bpi > 0:                   # density unknown: estimate it from the first transitions
        o_probe = _Options(mode=mode, ntrks=hdr.ntrks, bpi=0.0, ips=hdr.ips or 50.0, specified_parity=0 if opts.even_parity else 1,
                           revparity=opts.revparity, do_correction=int(opts.correct), find_zeros=int(find_zeros), do_differentiate=int(differentiate),
                           multiple_tries=int(opts.multiple_tries), tap_format=1, add_parity=0, verbose=int(opts.verbose))
        bpi_kw = {"bpi": detect_density(hdr, rows, full, o_probe, fe_factory, invert=invert, log_path=log_path, evt_path=evt_path,
                                        first_prefix_rows=deskew_prefix_rows)}
        detected = True
    cfg = frontend.FrontEndConfig.from_header(hdr, parmsets=frontend_parmsets(full), skew=skew, invert=invert, find_zeros=find_zeros, differentiate=differentiate, **bpi_kw)
    o = _Options(mode=mode, ntrks=hdr.ntrks, bpi=cfg.bpi, ips=cfg.ips, specified_parity=0 if opts.even_parity else 1,
                 revparity=opts.revparity, do_correction=int(opts.correct), find_zeros=int(find_zeros), do_differentiate=int(differentiate),
                 multiple_tries=int(opts.multiple_tries), tap_format=1, add_parity=0, verbose=int(opts.verbose))
    calibrated = deskew and skew is None and mode != tbin.MODE_PE        # "-deskew option is ignored for PE", src/readtape.c:1677
    if calibrated:
        skew = calibrate_deskew(hdr, rows, full, o, fe_factory, invert=invert, find_zeros=find_zeros, differentiate=differentiate,
                                log_path=log_path, evt_path=evt_path, first_prefix_rows=deskew_prefix_rows, append=detected, **bpi_kw)
        cfg = frontend.FrontEndConfig.from_header(hdr, parmsets=frontend_parmsets(full), skew=skew, invert=invert, find_zeros=find_zeros, differentiate=differentiate, **bpi_kw)
    fe = (fe_factory or frontend.FrontEnd)(cfg)
    res = fe.scan(rows).fetch()
    nrows = int(rows.shape[0])

    o = _Options(mode=mode, ntrks=hdr.ntrks, bpi=cfg.bpi, ips=cfg.ips, specified_parity=0 if opts.even_parity else 1,
                 revparity=opts.revparity, do_correction=int(opts.correct), find_zeros=int(find_zeros), do_differentiate=int(differentiate),
                 multiple_tries=int(opts.multiple_tries), tap_format=int(tap_format), add_parity=0, verbose=int(opts.verbose))
    parr = (_Parms * len(full))(*full)
    W = (C.c_int * len(full))(*fe.widths)
    exact, free, keep = _exact_callbacks(fe, rows, hdr.ntrks, fe_factory)

    st = _Stats()
    bursts = np.ascontiguousarray(res.bursts)
    counts = np.ascontiguousarray(res.counts)
    events = res._events
    if out_base:
        rc = lib.rt_replay_run_named(C.byref(o), parr, len(full), hdr.tdelta_ns, hdr.tstart_ns, nrows, 0, W,
                                     bursts.ctypes.data, len(bursts), counts.ctypes.data, events.ctypes.data, exact, free, None,
                                     out_base.encode(), (in_name or out_base + ".tbin").encode(), log_path.encode() if log_path else None,
                                     evt_path.encode() if evt_path else None, int(calibrated or detected), C.byref(st))
    else:
        run = lib.rt_replay_run_after_deskew if (calibrated or detected) else lib.rt_replay_run      # (continues the pre-passes' log / event dump)
        rc = run(C.byref(o), parr, len(full), hdr.tdelta_ns, hdr.tstart_ns, nrows, 0, W,
                 bursts.ctypes.data, len(bursts), counts.ctypes.data, events.ctypes.data,
                 exact, free, None,
                 tap_path.encode() if tap_path else None, log_path.encode() if log_path else None,
                 evt_path.encode() if evt_path else None, C.byref(st))
    _release(keep)
    if rc != 0:
        raise RuntimeError("rt_replay_run failed")
    stats = {k: getattr(st, k) for k, _ in _Stats._fields_}
    if stats["reference_fatal"]:
        raise ReferenceFatal(f"AGC gain bad in lookfor_peak on track {stats['fatal_trk']} at sample {stats['fatal_row']}: the reference asserts and exits "
                             "here (src/decoder.c:782); the blocks in front of it are in the .tap", stats)
    if stats["device_failures"]:
        raise RuntimeError(f"the device front end could not deliver {stats['device_failures']} attempt(s) (event regions overflowed even at "
                           "their largest, or an exact rescan failed): the decode is incomplete")
    stats["bursts"] = res.nbursts
    stats["skew"] = list(skew) if skew is not None else None
    stats["bpi"] = cfg.bpi
    return stats, res


FLUX = {"neg": 0, "pos": 1, "auto": 2}


def decode_tape_ww(hdr, rows, tap_path, log_path=None, order: str | None = None, fluxdir: str = "neg", reverse: bool = False, verbose: bool = True,
                   evt_path=None, fe_factory=None, invert=False, chunk_rows: int = 4096, out_base=None, in_name=None, deskew: bool = False):
    """Decodes a Whirlwind tape (6 tracks, 100 BPI; mode WW in the header or by the caller).  order = the heads' roles (-order=CMLcml,
    default: the header's TBINORD string); fluxdir neg / pos / auto.  The device detector keeps its state across block attempts, so
    the host replay fetches its events in chunks and hands the state back in (rtfe_ww_scan).  deskew: the reference's
    -deskew (a pre-pass over the first blocks learns every head's delay and the pulse heights, src/readtape.c:1676-1716; the delays
    come back as stats["skew_delays"]).  Returns the statistics."""
    lib = _load_decode_lib()
    order = order or hdr.trkorder or "CMLcml"
    if len(order) != hdr.ntrks:
        raise ValueError("the Whirlwind order string must name every head of the file")
    if "x" in order:
        # heads that are not Whirlwind tracks ('x', src/readtape.c:891: the reference parks their samples in a spare track nobody reads): their
        # columns never reach the device - the detector sees the used heads side by side, in file order, as the reference numbers its tracks
        used = [i for i, ch in enumerate(order) if ch != "x"]
        rows = rows[:, used]
        rows = rows.contiguous() if hasattr(rows, "contiguous") else np.ascontiguousarray(rows)
        order = "".join(order[i] for i in used)
        import dataclasses as _dc
        hdr = _dc.replace(hdr, ntrks=len(used))
    bpi, ips = (hdr.bpi or 100.0), (hdr.ips or 50.0)
    full = default_parmsets(tbin.MODE_WW, 1)
    import dataclasses
    h2 = dataclasses.replace(hdr, mode=tbin.MODE_WW, bpi=bpi, ips=ips, trkorder="", flags=hdr.flags & ~tbin.FLAG_NO_REORDER)
    cfg = frontend.FrontEndConfig.from_header(h2, parmsets=frontend_parmsets(full), invert=invert)
    fe = (fe_factory or frontend.FrontEnd)(cfg)
    o = _Options(mode=tbin.MODE_WW, ntrks=hdr.ntrks, bpi=bpi, ips=ips, specified_parity=1, revparity=0, do_correction=0, find_zeros=0, do_differentiate=0,
                 multiple_tries=0, tap_format=1, add_parity=0, verbose=int(verbose), ww_fluxdir=FLUX[fluxdir], ww_reverse=int(reverse), ww_order=order.encode())
    d_rows = fe.backend.rows(rows)
    cap = int(chunk_rows)
    bad = []

    def scan(user, first_row, nscan, seed_row0, state_in, state_out, counts_out, events_out, cap_in):
        try:
            st_in = C.string_at(state_in, hdr.ntrks * fe.WW_TRACK_BYTES)
            counts, events, st, flags = fe.ww_scan(d_rows, first_row, nscan, seed_row0, st_in, int(cap_in))
            if flags & (frontend.F_DETECTOR_FATAL | frontend.F_EVENT_OVERFLOW):
                bad.append(flags)
                return int(flags) or 1
            for t in range(hdr.ntrks):
                counts_out[t] = int(counts[t])
            C.memmove(events_out, events.ctypes.data, events.nbytes)
            C.memmove(state_out, st, len(st))
            return 0
        except Exception as e:          # (a ctypes callback must not raise)
            bad.append(repr(e))
            return 2

    cb = _WW_SCAN_FN(scan)
    st = _Stats()
    init = fe.ww_initial_state()
    delays = (C.c_int * hdr.ntrks)()
    rc = lib.rt_replay_run_ww(C.byref(o), (_Parms * 1)(full[0]), hdr.tdelta_ns, hdr.tstart_ns, int(d_rows.shape[0]), fe.widths[0], cb, None, init, cap,
                              tap_path.encode() if tap_path and not out_base else None, out_base.encode() if out_base else None,
                              (in_name or "").encode() if (out_base and in_name) else None, log_path.encode() if log_path else None,
                              evt_path.encode() if evt_path else None, C.byref(st), int(bool(deskew)), delays)
    fe.close()
    if rc == -2 and not bad:
        raise RuntimeError("the -deskew pre-pass found tracks without flux transitions (is the order string right?)")
    if rc != 0:
        raise RuntimeError(f"rt_replay_run_ww failed ({bad[:2]})")
    stats = {k: getattr(st, k) for k, _ in _Stats._fields_}
    if deskew:
        stats["skew_delays"] = list(delays)
    if stats["reference_fatal"]:
        raise ReferenceFatal(f"AGC gain bad in lookfor_peak on track {stats['fatal_trk']} at sample {stats['fatal_row']}", stats)
    if stats["device_failures"]:
        raise RuntimeError(f"the device front end could not deliver an attempt ({bad[:2]})")
    return stats


# ---------------------------------------------------------------------------------------------------------------------
# Fragments: time shards of one tape (one per GPU) and the windows of a tape streamed from disk are decoded the same way.
# A fragment owns the bursts whose zone ends in its own rows (rtfe_scan's own_rows rule); its decode starts at the start of the
# zone in front of its first own burst and stops where an attempt would start in the first zone its right neighbour owns.  Blocks
# are independent (every attempt restarts all state, src/decoder.c:425-455), so the fragments' .tap files concatenate to the
# whole tape's; the end-of-medium marker goes behind the last one.  (SURVEY.md 8e "Host side".)
# ---------------------------------------------------------------------------------------------------------------------
I64MAX = (1 << 63) - 1


def scan_fragment(fe, rows_with_halo, own_rows, lo, is_first, is_last, stream=None):
    """Scans one fragment (async) -> a function that fetches (result, absolute bursts incl. the bounding one, stop_row or None)."""
    res = fe.scan(rows_with_halo, row_base=lo, first_is_tape_start=is_first, own_rows=own_rows, stream=stream)

    def finish():
        res.fetch()
        be = fe.backend
        nb = res.nbursts
        bound = None
        if not is_last:
            truncated = nb > 0 and bool(int(res.bursts[nb - 1]["flags"]) & frontend.F_TRUNCATED)
            if truncated or res.next_burst.shape[0] < 1:
                return res, None, None                 # the halo holds no further zone: the caller retries with a longer one
            bound = int(res.next_burst[0]["zone_first"])          # relative to the fragment
        return res, nb, bound
    finish.res = res
    return finish


def decode_fragment(hdr, cfg, fe, res, rows_with_halo, lo, start_row, stop_row, tap_path, full, opts, fe_factory=None, log_path=None, evt_path=None,
                    exact_lock=None):
    """Host replay of one scanned fragment -> its piece of the .tap (no end marker).  Returns the replay statistics.
    exact_lock: several fragments are replayed side by side and share `fe` for their exact rescans: one at a time."""
    from readtape_amd import shard
    lib = _load_decode_lib()
    o = _Options(mode=hdr.mode, ntrks=hdr.ntrks, bpi=cfg.bpi, ips=cfg.ips, specified_parity=0 if opts.even_parity else 1,
                 revparity=opts.revparity, do_correction=int(opts.correct), find_zeros=int(cfg.find_zeros), do_differentiate=int(cfg.differentiate),
                 multiple_tries=int(opts.multiple_tries), tap_format=1, add_parity=0, verbose=int(opts.verbose))
    parr = (_Parms * len(full))(*full)
    W = (C.c_int * len(full))(*fe.widths)
    exact, free, keep = _exact_callbacks(fe, rows_with_halo, hdr.ntrks, fe_factory, lock=exact_lock)
    st = _Stats()
    bursts = np.ascontiguousarray(shard.absolute_bursts(res, lo))
    counts = np.ascontiguousarray(res.counts)
    nrows = int(rows_with_halo.shape[0])
    rc = lib.rt_replay_run_fragment(C.byref(o), parr, len(full), hdr.tdelta_ns, hdr.tstart_ns, nrows, lo, W,
                                    bursts.ctypes.data, len(bursts), counts.ctypes.data, res._events.ctypes.data, exact, free, None,
                                    tap_path.encode(), log_path.encode() if log_path else None, evt_path.encode() if evt_path else None, C.byref(st),
                                    int(start_row), I64MAX if stop_row is None else int(stop_row))
    _release(keep)
    if rc != 0:
        raise RuntimeError("rt_replay_run_fragment failed")
    stats = {k: getattr(st, k) for k, _ in _Stats._fields_}
    if stats["reference_fatal"] or stats["device_failures"]:
        raise RuntimeError(f"fragment at row {lo}: the decode stopped early ({stats})")
    return stats


def decode_fragments(hdr, cfg, fe, res, rows_with_halo, lo, start_rows, stop_rows, tap_paths, full, opts, fe_factory=None, exact_lock=None):
    """decode_fragment for several fragments of ONE scan side by side - native threads inside librtdecode.so, this thread takes the first
    (rt_replay_run_fragments): the streaming reader hands a window's sub-fragments over in one call.  Returns [(statistics, seconds), ...]."""
    from readtape_amd import shard
    lib = _load_decode_lib()
    o = _Options(mode=hdr.mode, ntrks=hdr.ntrks, bpi=cfg.bpi, ips=cfg.ips, specified_parity=0 if opts.even_parity else 1,
                 revparity=opts.revparity, do_correction=int(opts.correct), find_zeros=int(cfg.find_zeros), do_differentiate=int(cfg.differentiate),
                 multiple_tries=int(opts.multiple_tries), tap_format=1, add_parity=0, verbose=int(opts.verbose))
    parr = (_Parms * len(full))(*full)
    W = (C.c_int * len(full))(*fe.widths)
    exact, free, keep = _exact_callbacks(fe, rows_with_halo, hdr.ntrks, fe_factory, lock=exact_lock)
    n = len(tap_paths)
    st = (_Stats * n)()
    secs = (C.c_double * n)()
    paths = (C.c_char_p * n)(*[p.encode() for p in tap_paths])
    starts = (C.c_int64 * n)(*[int(a) for a in start_rows])
    stops = (C.c_int64 * n)(*[I64MAX if b is None else int(b) for b in stop_rows])
    bursts = np.ascontiguousarray(shard.absolute_bursts(res, lo))
    counts = np.ascontiguousarray(res.counts)
    rc = lib.rt_replay_run_fragments(C.byref(o), parr, len(full), hdr.tdelta_ns, hdr.tstart_ns, int(rows_with_halo.shape[0]), lo, W,
                                     bursts.ctypes.data, len(bursts), counts.ctypes.data, res._events.ctypes.data, exact, free, None,
                                     n, paths, starts, stops, st, secs)
    _release(keep)
    if rc != 0:
        raise RuntimeError("rt_replay_run_fragments failed")
    out = []
    for i in range(n):
        stats = {k: getattr(st[i], k) for k, _ in _Stats._fields_}
        if stats["reference_fatal"] or stats["device_failures"]:
            raise RuntimeError(f"fragment at row {lo}: the decode stopped early ({stats})")
        out.append((stats, float(secs[i])))
    return out


def decode_tape_fragments(hdr, rows, tap_path, spans, opts: DecodeOptions | None = None, fe_factory=None, halo_rows=1 << 16, cfgkw=None):
    """Decodes one resident tape as the fragments `spans` = [(lo, hi), ...] (contiguous, cuts on the 64-row grid), one after the other
    on this process' device, and concatenates their .tap pieces: what N ranks do with shard.plan_shards + the halo exchange, and what
    the streaming reader does window by window.  Returns a list of per-fragment statistics."""
    opts = opts or DecodeOptions()
    full = default_parmsets(hdr.mode, opts.nparmsets or (15 if opts.multiple_tries else 1))
    cfg = frontend.FrontEndConfig.from_header(hdr, parmsets=frontend_parmsets(full), **(cfgkw or {}))
    fe = (fe_factory or frontend.FrontEnd)(cfg)
    n = int(rows.shape[0])
    out, total = [], 0
    with open(tap_path, "wb") as tapf:
        for k, (lo, hi) in enumerate(spans):
            if hi <= lo:
                continue
            is_last = hi >= n
            halo = halo_rows
            while True:
                end = n if is_last else min(n, hi + halo)
                piece = rows[lo:end]
                res, nb, bound = scan_fragment(fe, piece, hi - lo, lo, lo == 0, is_last)()
                if nb is not None or end >= n:
                    break
                halo *= 4                                   # the last own burst runs past the halo: ask the neighbour for more
            # (nb None with the halo at the end of the tape: no further zone exists, the last own burst runs to the end of the data)
            if res.nbursts == 0:
                continue
            start = 0 if lo == 0 else int(res.bursts[0]["zone_first"])
            frag = tap_path + f".frag{k}"
            st = decode_fragment(hdr, cfg, fe, res, piece, lo, start, bound, frag, full, opts, fe_factory)
            data = open(frag, "rb").read()
            os.remove(frag)
            tapf.write(data)
            total += len(data)
            st["span"] = (lo, hi)
            out.append(st)
        if total > 0:
            tapf.write(b"\xff\xff\xff\xff")                 # src/readtape.c:1885
    fe.close()
    return out
