"""Python binding of the C ABI in include/rt_frontend.h (librtfe.so, the HIP front end).

Host-side mirror of the reference's front-end seam: `FrontEnd.scan()` is what `readblock()` /
`process_sample()` (src/readtape.c:1396, src/decoder.c:817) do for a whole tape at once, and the
returned events are the `{mode}_top/_bot` calls they would make (src/decoder.c:574-609).

PyTorch is used only for device memory and streams.  There is no CPU path: if librtfe.so is missing
or no GPU is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

from . import tbin

MAXTRKS, MAXPARMSETS = 19, 15
PE, NRZI, GCR, WW = 1, 2, 4, 8

F_EXACT_START, F_UNSAFE, F_EVENT_OVERFLOW, F_SCREEN_UNDERFLOW, F_DETECTOR_FATAL, F_TRUNCATED, F_STATE_AT_END, F_AGC_FATAL = 1, 2, 4, 8, 16, 32, 64, 128
EV_FATAL = 0x80       # event flag: the reference's "AGC gain bad in lookfor_peak" assert fires here (include/rt_frontend.h)

EVENT_DTYPE = np.dtype([("sample", "<u4"), ("v_peak", "<f4"), ("agc_gain", "<f4"), ("trk", "u1"),
                        ("flags", "u1"), ("left_distance", "u1"), ("parmset", "u1")])
BURST_DTYPE = np.dtype([("zone_first", "<i8"), ("zone_end", "<i8"), ("reset_sample", "<i8"), ("safe_last", "<i8"),
                        ("end_sample", "<i8"), ("event_base", "<u8"), ("event_cap", "<u4"), ("flags", "<u4")])
PLAN_DTYPE = np.dtype([("event_base", "<u8"), ("event_cap", "<u4"), ("reserved", "<u4")])      # rtfe_pack_entry
assert EVENT_DTYPE.itemsize == 16 and BURST_DTYPE.itemsize == 56 and PLAN_DTYPE.itemsize == 16


class _Parmset(C.Structure):
    _fields_ = [("pkww_bitfrac", C.c_float), ("pkww_rise", C.c_float), ("min_peak", C.c_float),
                ("agc_alpha", C.c_float), ("agc_window", C.c_int32), ("clk_factor", C.c_float)]


class _Config(C.Structure):
    _fields_ = [("mode", C.c_int32), ("ntrks", C.c_int32), ("head_to_trk", C.c_int32 * MAXTRKS),
                ("invert", C.c_int32), ("differentiate", C.c_int32), ("find_zeros", C.c_int32),
                ("skew_delaycnt", C.c_int32 * MAXTRKS), ("maxvolts", C.c_float), ("bpi", C.c_float), ("ips", C.c_float),
                ("tdelta_ns", C.c_int64), ("tstart_ns", C.c_int64), ("nparmsets", C.c_int32),
                ("parmset", _Parmset * MAXPARMSETS), ("gap_min_samples", C.c_int32), ("quiet_volts", C.c_float),
                ("screen_floor_height", C.c_float), ("events_per_sample_cap", C.c_float)]


class _Burst(C.Structure):
    _fields_ = [("zone_first", C.c_int64), ("zone_end", C.c_int64), ("reset_sample", C.c_int64), ("safe_last", C.c_int64),
                ("end_sample", C.c_int64), ("event_base", C.c_uint64), ("event_cap", C.c_uint32), ("flags", C.c_uint32)]


assert C.sizeof(_Burst) == BURST_DTYPE.itemsize, (C.sizeof(_Burst), BURST_DTYPE.itemsize)

# Built-in parameter sets: the front-end half of src/parmsets.c:77-118
#                 bitfrac rise  min_peak agc_alpha agc_window clk_factor
DEFAULT_PARMSETS = {
    NRZI: [(0.7, 0.20, 1.0, 0.3, 0, 0.0), (0.6, 0.20, 1.0, 0.3, 0, 0.0), (0.7, 0.20, 1.0, 0.3, 0, 0.0), (0.6, 0.20, 1.0, 0.3, 0, 0.0),
           (0.9, 0.05, 0.5, 0.0, 1, 0.0), (0.7, 0.05, 1.0, 0.0, 1, 0.0), (0.7, 0.05, 0.5, 0.0, 1, 0.0), (0.6, 0.05, 0.5, 0.0, 1, 0.0)],
    PE: [(0.7, 0.10, 0.0, 0.0, 5, 1.5), (0.7, 0.10, 0.1, 0.0, 5, 1.5), (0.7, 0.10, 0.0, 0.0, 5, 1.4), (0.7, 0.10, 0.0, 0.0, 5, 1.4),
         (0.7, 0.10, 0.0, 0.0, 5, 1.4), (0.7, 0.10, 0.0, 0.0, 5, 1.5), (0.7, 0.10, 0.0, 0.0, 5, 1.4), (0.7, 0.10, 0.0, 0.0, 5, 1.4)],
    GCR: [(1.5, 0.20, 0.2, 0.5, 0, 0.0), (1.5, 0.20, 0.2, 0.5, 0, 0.0), (1.5, 0.20, 0.2, 0.5, 0, 0.0), (1.5, 0.14, 0.0, 0.5, 0, 0.0),
          (1.5, 0.20, 0.2, 0.5, 0, 0.0)],
}


def parse_track_order(order: str):
    """head -> track permutation of the reference's -order= string / TBINORD header extension for PE, NRZI and GCR
    (src/readtape.c:877-915): one character per head, a digit = that track (0 = msb), p/P = the parity track (last)."""
    n = len(order)
    h2t = []
    for ch in order:
        if ch in "pP":
            h2t.append(n - 1)
        elif ch.isdigit() and int(ch) <= n - 2:
            h2t.append(int(ch))
        else:
            raise ValueError(f"bad track order string {order!r}")
    if sorted(h2t) != list(range(n)):
        raise ValueError(f"track order {order!r} is not a permutation")
    return h2t


@dataclass
class FrontEndConfig:
    mode: int
    ntrks: int
    maxvolts: float
    bpi: float
    ips: float
    tdelta_ns: int
    tstart_ns: int = 0
    parmsets: list = field(default_factory=list)      # tuples (bitfrac, rise, min_peak, agc_alpha, agc_window, clk_factor)
    head_to_trk: list | None = None
    skew: list | None = None
    invert: bool = False
    differentiate: bool = False
    find_zeros: bool = False
    gap_min_samples: int = 0
    quiet_volts: float = 0.0
    screen_floor_height: float = 0.0
    events_per_sample_cap: float = 0.0

    def peak_detection_floor_applies(self) -> bool:
        """True where a scan learns peak heights and screens candidates against screen_floor_height: NRZI peak detection on the undifferentiated signal
        (the peak path; the caller has not set a floor of its own)."""
        return self.mode == NRZI and not self.find_zeros and not self.differentiate and self.bpi > 0 and not self.screen_floor_height

    @classmethod
    def from_header(cls, h: tbin.TbinHeader, nparmsets: int = 1, **kw) -> "FrontEndConfig":
        mode = kw.pop("mode", h.mode)
        bpi = kw.pop("bpi", 9042.0 if mode == GCR else h.bpi)     # src/readtape.c:1652-1654
        ps = kw.pop("parmsets", None) or DEFAULT_PARMSETS[mode][:nparmsets]
        # the header's TBINORD extension (src/readtape.c:1346-1355).  A file without TBIN_NO_REORDER "had a permutation applied to
        # it" when it was made: the reference then ignores every track order, the command line's included (src/readtape.c:1646-1648)
        # (an explicit head_to_trk= is the caller's business and is taken as given)
        if kw.get("head_to_trk") is None and h.trkorder and (h.flags & tbin.FLAG_NO_REORDER):
            kw["head_to_trk"] = parse_track_order(h.trkorder)
        return cls(mode=mode, ntrks=h.ntrks, maxvolts=h.maxvolts, bpi=bpi, ips=h.ips or 50.0,
                   tdelta_ns=h.tdelta_ns, tstart_ns=h.tstart_ns, parmsets=list(ps), **kw)

    def to_c(self) -> _Config:
        c = _Config()
        c.mode, c.ntrks = self.mode, self.ntrks
        h2t = self.head_to_trk or list(range(self.ntrks))
        sk = self.skew or [0] * self.ntrks
        for i in range(self.ntrks):
            c.head_to_trk[i] = h2t[i]
            c.skew_delaycnt[i] = sk[i]
        c.invert, c.differentiate, c.find_zeros = int(self.invert), int(self.differentiate), int(self.find_zeros)
        c.maxvolts, c.bpi, c.ips = self.maxvolts, self.bpi, self.ips
        c.tdelta_ns, c.tstart_ns = self.tdelta_ns, self.tstart_ns
        c.nparmsets = len(self.parmsets)
        for i, p in enumerate(self.parmsets):
            c.parmset[i] = _Parmset(*[float(x) if j != 4 else int(x) for j, x in enumerate(p)])
        c.gap_min_samples = self.gap_min_samples
        c.quiet_volts, c.screen_floor_height = self.quiet_volts, self.screen_floor_height
        c.events_per_sample_cap = self.events_per_sample_cap
        return c


class TorchBackend:
    """Device memory through PyTorch-ROCm (plumbing only)."""

    def __init__(self, device="cuda:0"):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("readtape_amd.frontend needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.torch = torch
        self.device = torch.device(device)
        # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be the one resident in
        # the process before librtfe.so is dlopen'ed, otherwise the extension binds a second, uninitialised
        # runtime and sees no device.  Touch the device here so the context exists.
        torch.cuda.init()
        torch.empty(1, device=self.device)

    def empty(self, nbytes):
        return self.torch.empty(max(int(nbytes), 16), dtype=self.torch.uint8, device=self.device)

    def ptr(self, t):
        return t.data_ptr()

    def rows(self, rows):
        t = rows
        if isinstance(rows, np.ndarray):
            t = self.torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int16)).to(self.device, non_blocking=True)
        assert t.dtype == self.torch.int16 and t.is_contiguous() and t.is_cuda
        return t

    def _host(self, t):
        """The tensor on the host.  Large ones through page-locked memory (PyTorch's caching host allocator: power-of-two blocks, reused), on the
        current stream: a pageable destination goes through the runtime's staging buffers at a third of the rate."""
        if t.numel() * t.element_size() < (1 << 12):
            return t.cpu()
        h = self.torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        self.torch.cuda.current_stream(self.device).synchronize()
        return h

    def to_numpy(self, t, dtype, count=None):
        a = self._host(t).numpy().view(dtype)
        return a if count is None else a[:count]

    def pinned(self, nbytes):
        return self.torch.empty(max(int(nbytes), 16), dtype=self.torch.uint8, pin_memory=True)

    def mirror_async(self, host, pieces, stream_handle):
        """Device buffers -> consecutive slices of the page-locked tensor `host`, queued on the stream (behind the scan that fills them): the tables
        are on the host when the scan's event fires - a copy issued by the fetching thread later would queue behind the next windows' uploads."""
        torch = self.torch
        st = torch.cuda.current_stream(self.device) if stream_handle is None else torch.cuda.ExternalStream(int(stream_handle), device=self.device)
        with torch.cuda.stream(st):
            off = 0
            for t in pieces:
                n = t.numel()
                host[off: off + n].copy_(t, non_blocking=True)
                off += n

    def upload(self, t, host_bytes):
        t[: len(host_bytes)].copy_(self.torch.frombuffer(bytearray(host_bytes), dtype=self.torch.uint8))

    def stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def record(self, stream_handle):
        """An event behind what has been queued on the stream so far (ScanResult.fetch waits for it instead of for the whole device:
        a second scan may already be queued behind the one that is fetched)."""
        torch = self.torch
        st = torch.cuda.current_stream(self.device) if stream_handle is None else torch.cuda.ExternalStream(int(stream_handle), device=self.device)
        ev = torch.cuda.Event()
        ev.record(st)
        return ev

    def wait(self, ev):
        ev.synchronize()

    def sync(self):
        self.torch.cuda.synchronize(self.device)


def _load_library(path=None):
    path = path or os.environ.get("RTFE_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "librtfe.so")      # (RTFE_LIB_PATH: tools/ only - a build with other compile-time knobs, side by side)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                           "The front end has no CPU fallback.")
    lib = C.CDLL(path)
    lib.rtfe_last_error.restype = C.c_char_p
    lib.rtfe_create.argtypes = [C.POINTER(_Config), C.POINTER(C.c_void_p)]
    lib.rtfe_destroy.argtypes = [C.c_void_p]
    lib.rtfe_pkww_width.argtypes = [C.c_void_p, C.c_int]
    lib.rtfe_workspace_bytes.argtypes = [C.c_void_p, C.c_int64]; lib.rtfe_workspace_bytes.restype = C.c_size_t
    lib.rtfe_max_bursts.argtypes = [C.c_void_p, C.c_int64]; lib.rtfe_max_bursts.restype = C.c_int64
    lib.rtfe_event_capacity.argtypes = [C.c_void_p, C.c_int64]; lib.rtfe_event_capacity.restype = C.c_int64
    lib.rtfe_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_size_t,
                              C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.rtfe_scan_exact.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_uint32, C.c_int,
                                    C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.rtfe_kernel_name.restype = C.c_char_p
    lib.rtfe_ww_initial_state.argtypes = [C.c_void_p, C.c_int]
    lib.rtfe_ww_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.rtfe_scan_stats.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    lib.rtfe_find_end_mark.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.rtfe_pack_events.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.rtfe_set_timing.argtypes = [C.c_void_p, C.c_int]
    lib.rtfe_set_graphs.argtypes = [C.c_void_p, C.c_int]
    lib.rtfe_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.rtfe_reset_floor.argtypes = [C.c_void_p, C.c_void_p]
    if lib.rtfe_abi_version() != 6:
        raise RuntimeError("librtfe.so ABI mismatch")
    return lib


class ScanResult:
    """Device-resident output of one scan; `fetch()` copies the small tables and the used part of
    the event regions to the host."""

    def __init__(self, fe, bufs, max_bursts, single=False):
        self.fe, self.bufs, self.max_bursts, self.single = fe, bufs, max_bursts, single
        self.bursts = self.counts = self._events = None

    def fetch(self, events=True):
        fe, be = self.fe, self.fe.backend
        import time
        tt = [time.perf_counter()]
        if getattr(self, "done", None) is not None:
            be.wait(self.done)
        else:
            be.sync()
        tt.append(time.perf_counter())
        P, T = len(fe.cfg.parmsets), fe.cfg.ntrks
        mirror = None
        if getattr(self, "mirrored", False):              # the tables came with the scan (TorchBackend.mirror_async)
            m, off, mirror = self.bufs["mirror"].numpy(), 0, {}
            for name, t in zip(("nbursts", "bursts", "counts", "plan", "endmark"), self.bufs["mirror_of"]):
                mirror[name] = m[off: off + int(t.numel())]
                off += int(t.numel())
        tab = (lambda name, dtype: mirror[name][: (len(mirror[name]) // dtype.itemsize) * dtype.itemsize].view(dtype)) if mirror else (lambda name, dtype: be.to_numpy(self.bufs[name], dtype))
        if mirror and getattr(self, "end_mark_checked", False):      # rtfe_find_end_mark ran in front of the scan: the first row with the reader's end marker, or None
            m = int(mirror["endmark"][:8].view(np.int64)[0])
            self.end_mark = None if m == np.iinfo(np.int64).max else m
            self.end_mark_valid = True
        nb = 1 if self.single else int(tab("nbursts", np.dtype(np.int32))[0])
        allb = tab("bursts", BURST_DTYPE)
        self.bursts = allb[:nb].copy()
        self.next_burst = allb[nb: nb + 1].copy()          # time shards: the burst that bounds the last own one (rtfe_scan writes it behind the table's own entries)
        self.counts = tab("counts", np.dtype(np.uint32))[: nb * P * T].reshape(nb, P, T).copy()
        # the used part of the event arena: bursts are laid out one after the other (event_base, P*T regions of event_cap each)
        used = int((self.bursts["event_base"].astype(np.int64) + P * T * self.bursts["event_cap"].astype(np.int64)).max()) if nb else 0
        if not events:
            self._events = None
            return self
        # The arena is laid out for the worst case (event_cap per list); what the lists hold is a fraction of it (C2: 6.7 of 38 MB
        # per 2^21 rows).  Where the backend can gather on the device, every burst's P*T lists are packed to the burst's longest
        # list before the copy, and the host copy of the burst table is re-based to the packed layout (same addressing rule:
        # event_base + (p * T + t) * event_cap) - the device table is left alone.
        tt.append(time.perf_counter())
        # The arena is laid out for the worst case (event_cap per list); what the lists hold is a fraction of it (C2: 6.7 of 38 MB
        # per 2^21 rows).  rtfe_pack_events - queued behind the scan - has packed every burst's P*T lists to the burst's longest
        # list; the host copy of the burst table is re-based to the packed layout (same addressing rule:
        # event_base + (p * T + t) * event_cap) - the device table is left alone.
        pack = os.environ.get("RTFE_PACK_EVENTS")                   # (tests: "1" packs whatever the size, "0" never)
        packed = False
        if nb and not getattr(self, "packed", False) and not self.single and pack != "0":
            # not packed behind the scan (a resident tape's scans are not fetched as a rule): now, if it pays - the lists' sizes are known
            cap2 = np.maximum(self.counts.reshape(nb, -1).max(axis=1), 1).astype(np.int64)
            dense = int(P * T * cap2.sum())
            if pack == "1" or (used >= (1 << 16) and 2 * dense < used):
                fe._pack_buffers(self.bufs, dense)
                fe._pack(self.bufs, None)
                be.sync()
                self.packed, mirror = True, None
                tab = lambda name, dtype: be.to_numpy(self.bufs[name], dtype)
        if nb and getattr(self, "packed", False) and pack != "0":
            plan = tab("plan", PLAN_DTYPE)[: nb + 1].copy()
            total = int(plan[nb]["event_base"])
            if int(plan[nb]["reserved"]) == nb and total <= self.bufs["packed_cap"] and (pack == "1" or (used >= (1 << 16) and 2 * total < used)):
                self._events = be.to_numpy(self.bufs["packed"][: max(total, 1) * EVENT_DTYPE.itemsize], EVENT_DTYPE)
                self.bursts["event_base"] = plan[:nb]["event_base"]
                self.bursts["event_cap"] = plan[:nb]["event_cap"]
                packed = True
        if not packed:
            self._events = be.to_numpy(self.bufs["events"][: max(used, 1) * EVENT_DTYPE.itemsize], EVENT_DTYPE)
        tt.append(time.perf_counter())
        self.fetch_times = tt                              # (wait for the scan, the tables, the events)
        return self

    @property
    def nbursts(self):
        return len(self.bursts)

    def track_events(self, b, p, t):
        B = self.bursts[b]
        base = int(B["event_base"]) + (p * self.fe.cfg.ntrks + t) * int(B["event_cap"])
        return self._events[base: base + int(self.counts[b, p, t])]

    def events(self, b, p):
        """All tracks of (burst b, parmset p) merged in the order the reference's per-sample loop
        produces them: by detection sample, then track number (src/decoder.c:847)."""
        parts = [self.track_events(b, p, t) for t in range(self.fe.cfg.ntrks)]
        ev = np.concatenate(parts) if parts else np.zeros(0, EVENT_DTYPE)
        order = np.lexsort((ev["trk"], ev["sample"]))
        return ev[order]


class FrontEnd:
    def __init__(self, cfg: FrontEndConfig, device="cuda:0", _lib_path=None, _backend=None):
        """_lib_path / _backend are hooks for tests/cpu_emul only; the product uses librtfe.so + a GPU."""
        self.cfg = cfg
        self.backend = _backend or TorchBackend(device)      # first: see TorchBackend.__init__
        self.lib = _load_library(_lib_path)
        self._c = cfg.to_c()
        h = C.c_void_p()
        rc = self.lib.rtfe_create(C.byref(self._c), C.byref(h))
        if rc != 0:
            raise ValueError(f"rtfe_create failed ({rc}): {self.lib.rtfe_last_error().decode()}")
        self.h = h
        self.widths = [self.lib.rtfe_pkww_width(self.h, p) for p in range(len(cfg.parmsets))]
        self._cache = {}

    def close(self):
        if getattr(self, "h", None):
            self.lib.rtfe_destroy(self.h)
            self.h = None
        if getattr(self, "_cache", None):                   # the workspaces and output buffers go with the handle (not when the garbage collector finds the object)
            self._cache.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def kernel_names(self):
        return [self.lib.rtfe_kernel_name(i).decode() for i in range(self.lib.rtfe_kernel_count())]

    def set_timing(self, enable=True):
        if self.lib.rtfe_set_timing(self.h, int(enable)) != 0:
            raise RuntimeError(self.lib.rtfe_last_error().decode())

    def set_graphs(self, enable=True):
        """rtfe_set_graphs: scans with the same arguments (the same buffers) replay one captured HIP graph instead of launching their ~20 kernels one by one."""
        if self.lib.rtfe_set_graphs(self.h, int(enable)) != 0:
            raise RuntimeError(self.lib.rtfe_last_error().decode())

    def reset_floor(self, stream=None):
        """rtfe_reset_floor: the candidate screen back to what the handle was made with - the next scan is a tape's first scan again (it estimates the floor
        from the samples before it screens)."""
        if self.lib.rtfe_reset_floor(self.h, C.c_void_p(stream) if stream else None) != 0:
            raise RuntimeError(self.lib.rtfe_last_error().decode())

    def kernel_ms(self):
        """Per-span elapsed ms (HIP events on the scans' stream) summed over the scans since the last call - at most 64 - and how many
        scans that was: ({span: ms}, scans).  Synchronises those scans."""
        out = (C.c_float * self.lib.rtfe_kernel_count())()
        n = self.lib.rtfe_kernel_ms(self.h, out)
        if n < 0:
            raise RuntimeError(self.lib.rtfe_last_error().decode())
        return dict(zip(self.kernel_names(), [float(x) for x in out])), int(n)

    def scan_stats(self, result):
        """{'bursts', 'redone', 'record_bytes'} of the scan that produced `result` (synchronises; diagnostics)."""
        self.backend.sync()
        out = (C.c_int64 * 24)()
        if self.lib.rtfe_scan_stats(self.h, self.backend.ptr(result.bufs["ws"]), out) != 0:
            raise RuntimeError(self.lib.rtfe_last_error().decode())
        key = int(out[21])
        min_height = float(np.array([0x7fffffff - key], dtype=np.uint32).view(np.float32)[0]) if key > 0 else None      # smallest v_avg_height a chain of the scan learned (peak path)
        return dict(bursts=int(out[0]), redone=int(out[1]), record_bytes=int(out[2]), parallel=int(out[3]), sequential=int(out[4]), gave_up=[int(out[5 + i]) for i in range(8)], phase_cycles=[int(out[13 + i]) for i in range(8)],
                    min_learned_height=min_height, screen_floor_now=float(np.array([int(out[22])], dtype=np.uint32).view(np.float32)[0]),
                    screen_floor_used=float(np.array([int(out[23])], dtype=np.uint32).view(np.float32)[0]))

    def _buffers(self, nrows, key="scan"):
        """Allocates (once per size) the workspace and output buffers for a scan of nrows rows.  Exact rescans share ONE
        grow-only set (their lengths are almost always distinct: a per-length cache would grow without bound)."""
        k = (key, nrows)
        if key == "exact":
            k = (key, 0)
            old = self._cache.get(k)
            if old is not None and old["nrows"] >= nrows:
                return old
            nrows = max(nrows, 2 * old["nrows"] if old is not None else 1 << 16)
            self._cache.pop(k, None)
        if key == "scan" and k not in self._cache:         # fragments of one tape differ in length: the largest buffer set serves them all
            fit = [kk for kk in self._cache if kk[0] == "scan" and kk[1] >= nrows]
            if fit:
                return self._cache[min(fit, key=lambda kk: kk[1])]
        if k not in self._cache:
            be, lib = self.backend, self.lib
            mb = int(lib.rtfe_max_bursts(self.h, nrows))
            cap = int(lib.rtfe_event_capacity(self.h, nrows))
            P, T = len(self.cfg.parmsets), self.cfg.ntrks
            hint = getattr(self, "bursts_hint", None)
            if key == "scan" and hint:
                # rtfe_event_capacity is the worst case: every run of gap-length quiet rows a burst (nrows / 512 of them), 128 events of slack each.  A caller who
                # knows its tape holds far fewer may bring a smaller arena - the scan flags a burst that no longer fits RTFE_F_EVENT_OVERFLOW (its lists stay
                # empty, nothing is written outside the arena) and the caller comes back with the full size.  1e9 rows x 8 sets: 288 GB of slack alone.
                frac = self.cfg.events_per_sample_cap or 0.125
                cap = min(cap, int((nrows * frac + 128.0 * int(hint) + 256) * P * T) + 4096)
            self._cache[k] = dict(
                ws=be.empty(lib.rtfe_workspace_bytes(self.h, nrows)), bursts=be.empty(mb * BURST_DTYPE.itemsize),
                nbursts=be.empty(16), counts=be.empty(mb * P * T * 4), events=be.empty(cap * 16), max_bursts=mb, cap=cap, nrows=nrows)
        return self._cache[k]

    def _pack_buffers(self, b, records):
        """The buffers of rtfe_pack_events beside a scan's: the plan, room for `records` packed events, and - where the backend has page-locked
        memory - the host mirror of the small tables (nbursts | bursts | counts | plan | end mark), filled behind the scan on its stream."""
        be = self.backend
        if "plan" not in b:
            b["plan"] = be.empty((b["max_bursts"] + 1) * PLAN_DTYPE.itemsize)
            b["endmark"] = be.empty(16)
            if hasattr(be, "pinned"):
                b["mirror_of"] = [b["nbursts"], b["bursts"], b["counts"], b["plan"], b["endmark"]]
                b["mirror"] = be.pinned(sum(int(t.numel()) for t in b["mirror_of"]))
        if b.get("packed_cap", 0) < records:
            b["packed"], b["packed_cap"] = be.empty(records * 16), int(records)
        return b

    def _pack(self, b, stream):
        be = self.backend
        rc = self.lib.rtfe_pack_events(self.h, be.ptr(b["bursts"]), be.ptr(b["nbursts"]), b["max_bursts"], be.ptr(b["counts"]), be.ptr(b["events"]),
                                       be.ptr(b["packed"]), b["packed_cap"], be.ptr(b["plan"]), stream if stream is not None else be.stream())
        if rc != 0:
            raise RuntimeError(f"rtfe_pack_events failed ({rc}): {self.lib.rtfe_last_error().decode()}")

    def _rows(self, rows):
        """The rows where the kernels can read them; the C ABI takes a pointer and a row count, so the row width is checked here."""
        d_rows = self.backend.rows(rows)
        if d_rows.ndim != 2 or int(d_rows.shape[1]) != self.cfg.ntrks:
            raise ValueError(f"rows must be [n, {self.cfg.ntrks}] int16, got {tuple(d_rows.shape)}")
        return d_rows

    def scan(self, rows, row_base=0, first_is_tape_start=True, stream=None, own_rows=None) -> ScanResult:
        """Launches the speculative scan of `rows` ([n, ntrks] int16, device tensor or numpy) — asynchronous.
        Time shards pass own_rows < n: the trailing rows are the right neighbour's halo (include/rt_frontend.h)."""
        be = self.backend
        d_rows = self._rows(rows)
        nrows = int(d_rows.shape[0])
        b = self._buffers(nrows)
        pack_now = getattr(self, "pack_on_scan", False) and os.environ.get("RTFE_PACK_EVENTS") != "0"      # (the streaming reader: every window's lists cross PCIe)
        if pack_now:
            self._pack_buffers(b, b["cap"] // 2)          # half the arena is room for every tape seen so far; a window that needs more is fetched as it is
        marked = False
        import time
        lt = [time.perf_counter()]
        if getattr(self, "find_end_mark", False) and pack_now:      # (the streaming reader: the window's end-of-data check on the device, in front of the scan)
            if self.lib.rtfe_find_end_mark(self.h, be.ptr(d_rows), nrows, be.ptr(b["endmark"]), stream if stream is not None else be.stream()) != 0:
                raise RuntimeError(f"rtfe_find_end_mark failed: {self.lib.rtfe_last_error().decode()}")
            marked = True
        rc = self.lib.rtfe_scan(self.h, be.ptr(d_rows), nrows, nrows if own_rows is None else int(own_rows), row_base, int(first_is_tape_start),
                                be.ptr(b["ws"]), b["ws"].numel() if hasattr(b["ws"], "numel") else b["ws"].size,
                                be.ptr(b["bursts"]), b["max_bursts"], be.ptr(b["nbursts"]), be.ptr(b["counts"]),
                                be.ptr(b["events"]), b["cap"], stream if stream is not None else be.stream())
        if rc != 0:
            raise RuntimeError(f"rtfe_scan failed ({rc}): {self.lib.rtfe_last_error().decode()}")
        lt.append(time.perf_counter())
        r = ScanResult(self, b, b["max_bursts"])
        r._rows_keepalive = d_rows
        r.end_mark_checked, r.end_mark, r.end_mark_valid = marked, None, False
        if pack_now:
            self._pack(b, stream)
            r.packed = True
            lt.append(time.perf_counter())
            if "mirror" in b:
                be.mirror_async(b["mirror"], b["mirror_of"], stream)
                r.mirrored = True
        if hasattr(be, "record"):
            r.done = be.record(stream)
        lt.append(time.perf_counter())
        r.launch_times = lt                                # (the scan queued, the packing queued, the mirror copies and the event queued)
        return r

    def scan_exact(self, rows, reset_row, end_row, parmset_mask=0xFFFFFFFF, screen_off=False, row_base=0, stream=None) -> ScanResult:
        be = self.backend
        d_rows = self._rows(rows)
        nrows = int(d_rows.shape[0])
        b = self._buffers(max(int(end_row - reset_row), 1), key="exact")
        rc = self.lib.rtfe_scan_exact(self.h, be.ptr(d_rows), nrows, row_base, int(reset_row), int(end_row), parmset_mask, int(screen_off),
                                      be.ptr(b["ws"]), b["ws"].numel() if hasattr(b["ws"], "numel") else b["ws"].size,
                                      be.ptr(b["bursts"]), be.ptr(b["counts"]), be.ptr(b["events"]), b["cap"],
                                      stream if stream is not None else be.stream())
        if rc != 0:
            raise RuntimeError(f"rtfe_scan_exact failed ({rc}): {self.lib.rtfe_last_error().decode()}")
        r = ScanResult(self, b, 1, single=True)
        r._rows_keepalive = d_rows
        return r

    # --- Whirlwind: the detector's state goes in and comes back (include/rt_frontend.h: rtfe_ww_scan) ---
    WW_TRACK_BYTES = 224

    def ww_initial_state(self) -> bytes:
        buf = (C.c_ubyte * (self.WW_TRACK_BYTES * self.cfg.ntrks))()
        self.lib.rtfe_ww_initial_state(buf, self.cfg.ntrks)
        return bytes(buf)

    def ww_scan(self, rows, first_row, nscan, seed_row0, state: bytes, cap: int):
        """-> (counts[ntrks], events[ntrks, cap] (EVENT_DTYPE, sample relative to first_row), state after the last row, flags)"""
        be, T = self.backend, self.cfg.ntrks
        d_rows = self._rows(rows)
        b = self._cache.get(("ww", cap))
        if b is None:
            b = self._cache[("ww", cap)] = dict(st_in=be.empty(T * self.WW_TRACK_BYTES), st_out=be.empty(T * self.WW_TRACK_BYTES), counts=be.empty(T * 4),
                                                events=be.empty(T * cap * 16), flags=be.empty(16))
        assert len(state) == T * self.WW_TRACK_BYTES
        be.upload(b["st_in"], state)
        be.upload(b["flags"], b"\0" * 16)
        rc = self.lib.rtfe_ww_scan(self.h, be.ptr(d_rows), int(d_rows.shape[0]), 0, int(first_row), int(nscan), int(seed_row0), be.ptr(b["st_in"]), be.ptr(b["st_out"]),
                                   be.ptr(b["counts"]), be.ptr(b["events"]), int(cap), be.ptr(b["flags"]), be.stream())
        if rc != 0:
            raise RuntimeError(f"rtfe_ww_scan failed ({rc}): {self.lib.rtfe_last_error().decode()}")
        be.sync()
        counts = be.to_numpy(b["counts"], np.uint32)[:T].copy()
        events = be.to_numpy(b["events"], EVENT_DTYPE)[: T * cap].reshape(T, cap).copy()
        st = bytes(be.to_numpy(b["st_out"], np.uint8)[: T * self.WW_TRACK_BYTES])
        flags = int(be.to_numpy(b["flags"], np.uint32)[0])
        return counts, events, st, flags

    # --- helpers that restate how the reference turns an event into times (src/decoder.c:732, src/readtape.c:1423) ---
    def time_of_row(self, row):
        return (self.cfg.tstart_ns + np.asarray(row, dtype=np.int64) * self.cfg.tdelta_ns).astype(np.float64) / 1e9

    def peak_times(self, burst, ev, parmset):
        W = self.widths[parmset]
        dt = np.float32(np.float32(self.cfg.tdelta_ns) / np.float32(1e9))
        adjc = (ev["flags"] >> 1) & 3
        adj = np.where(adjc == 1, np.float32(-0.5), np.where(adjc == 2, np.float32(0.5), np.float32(0))).astype(np.float32)
        back = ((W - ev["left_distance"].astype(np.int32)).astype(np.float32) - adj) * dt        # float product
        return self.time_of_row(int(burst["reset_sample"]) + ev["sample"].astype(np.int64)) - back.astype(np.float64)
