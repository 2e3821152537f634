"""CSV ingest: a logic-analyser export ("time, v0, v1, ..." behind two title lines) -> TBIN header + int16 rows,
with the numbers the reference's converter writes (src/csvtbin.c:619-716; parser and quantiser in csrc/host/rt_csv.c).
The decode then is the .tbin decode: the reference's direct CSV path (src/readtape.c:1426-1448) works on the unquantised floats and
on the file's rounded timestamps, which no int16 front end can follow bit for bit - its author's own recommended route is the
converter ("tbin is still smaller and faster", src/readtape.c:343)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import frontend, tbin

HERE = os.path.dirname(os.path.abspath(__file__))


class _Info(C.Structure):
    _fields_ = [("columns", C.c_int), ("rows", C.c_int64), ("tstart_ns", C.c_uint64), ("tdelta_ns", C.c_uint32), ("maxvolts", C.c_float)]


def _lib():
    lib = C.CDLL(os.path.join(HERE, "librtdecode.so"))
    lib.rt_csv_survey.argtypes = [C.c_char_p, C.c_int, C.c_float, C.c_int, C.c_float, C.POINTER(_Info)]
    lib.rt_csv_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_float, C.c_int, C.c_float, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    lib.rt_csv_load.restype = C.c_int64
    return lib


def read_csv(path: str, ntrks: int = 9, mode: int = tbin.MODE_NRZI, bpi: float = 0.0, ips: float = 0.0, order: str | None = None,
             invert: bool = False, scale: float = 1.0, subsample: int = 1, maxvolts: float = 0.0, descr: str = ""):
    """-> (TbinHeader, rows[n, ntrks] int16, {clipped_samples, columns}).  order = the converter's -order= string: column k of the file is that track, and
    goes to that column of the rows (src/csvtbin.c:330-352) - the file is then in track order and says so (no TBIN_NO_REORDER).  Without
    an order the header is marked TBIN_NO_REORDER (src/csvtbin.c:804-807) and a decode's trkorder= applies; a Whirlwind order string is
    kept in the header extension, columns unmoved (src/csvtbin.c:317-323)."""
    lib = _lib()
    info = _Info()
    rc = lib.rt_csv_survey(path.encode(), ntrks, scale, subsample, maxvolts, C.byref(info))
    if rc != 0:
        raise OSError(f"cannot read {path} as a CSV sample file ({rc})")
    perm = None
    flags = tbin.FLAG_INVERTED if invert else 0
    trkorder = ""
    if order and mode == tbin.MODE_WW:
        # Whirlwind: the string goes into the header extension as it is, no column moves (src/csvtbin.c:317-323)
        if len(order) != ntrks:
            raise ValueError(f"Whirlwind -order string {order!r} does not name {ntrks} tracks")
        trkorder = order
        flags |= tbin.FLAG_NO_REORDER                  # (write_tbin adds TRKORDER_INCLUDED for a header that carries a string)
    elif order:
        h2t = frontend.parse_track_order(order)
        perm = (C.c_int * ntrks)(*h2t)
    else:
        flags |= tbin.FLAG_NO_REORDER                  # "marking the .tbin file to show it wasn't given" (src/csvtbin.c:804-807): a later -order= applies
    rows = np.empty((max(int(info.rows), 1), ntrks), dtype=np.int16)
    clipped = C.c_int64()
    n = lib.rt_csv_load(path.encode(), ntrks, perm, int(invert), scale, subsample, info.maxvolts, rows.ctypes.data, rows.shape[0], C.byref(clipped))
    if n in (-3, -4):
        raise ValueError(f"ntrks {ntrks} or the track order is out of range for a CSV sample file")
    if n < 0:
        raise OSError(f"cannot read {path}")
    hdr = tbin.TbinHeader(ntrks=ntrks, tdelta_ns=int(info.tdelta_ns), maxvolts=float(info.maxvolts), mode=mode, bpi=bpi, ips=ips, flags=flags,
                          tstart_ns=int(info.tstart_ns), descr=descr, trkorder=trkorder)
    return hdr, rows[:n], dict(clipped_samples=int(clipped.value), columns=int(info.columns))
