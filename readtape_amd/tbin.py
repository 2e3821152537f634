"""TBIN container: reader and writer for the packed int16 tape-waveform format.

Host-side counterpart of the reference's `.tbin` framing (layout: src/csvtbin.h:50-105; reader:
src/readtape.c:1319-1376).  Only framing lives here; the sample payload is handed to the device
front end untouched (interleaved little-endian int16 rows, one row per sample instant).

Layout restated (all little endian):
  0    char[8]  "TBINHDR\\0"
  8    char[80] description
  88   u32 hdrsize(=240) | u32 format(=1) | 3 x 9 x i32 struct tm (written, read, converted)
  204  u32 flags | u32 ntrks | u32 tdelta_ns | f32 maxvolts | u32 rsvd1 | u32 rsvd2
  228  u32 mode (PE=1 NRZI=2 GCR=4 WW=8) | f32 bpi | f32 ips            -> 240 bytes
  [if flags & 2]  char[8] "TBINORD\\0" | char[20] order string           -> +28 bytes
  char[4] "DAT\\0" | u8 options | u8 sample_bits(=16) | u8 | u8 | u64 tstart_ns   -> +16 bytes
  rows of ntrks int16 ... terminated by a single int16 0x8000 in the head-0 slot.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

HDR_TAG = b"TBINHDR\0"
ORD_TAG = b"TBINORD\0"
DAT_TAG = b"DAT\0"
HDR_SIZE = 240
END_MARK = -32768

MODE_PE, MODE_NRZI, MODE_GCR, MODE_WW = 1, 2, 4, 8
MODE_NAMES = {MODE_PE: "PE", MODE_NRZI: "NRZI", MODE_GCR: "GCR", MODE_WW: "WW", 0: "UNKNOWN"}

FLAG_NO_REORDER = 0x01
FLAG_TRKORDER_INCLUDED = 0x02
FLAG_INVERTED = 0x04
FLAG_REVERSED = 0x08


@dataclass
class TbinHeader:
    ntrks: int
    tdelta_ns: int
    maxvolts: float
    mode: int = 0
    bpi: float = 0.0
    ips: float = 0.0
    flags: int = 0
    tstart_ns: int = 1_000_000
    descr: str = ""
    trkorder: str = ""
    times: tuple = field(default_factory=lambda: (0,) * 27)

    @property
    def sample_deltat(self) -> np.float32:
        # src/readtape.c:1345  sample_deltat = (float)sample_deltat_ns / 1e9f
        return np.float32(np.float32(self.tdelta_ns) / np.float32(1e9))


def pack_header(h: TbinHeader) -> bytes:
    flags = h.flags | (FLAG_TRKORDER_INCLUDED if h.trkorder else 0)
    out = bytearray()
    out += HDR_TAG
    out += h.descr.encode("ascii")[:79].ljust(80, b"\0")
    out += struct.pack("<II", HDR_SIZE, 1)
    out += struct.pack("<27i", *h.times)
    out += struct.pack("<IIIfII", flags, h.ntrks, h.tdelta_ns, h.maxvolts, 0, 0)
    out += struct.pack("<Iff", h.mode, h.bpi, h.ips)
    assert len(out) == HDR_SIZE
    if h.trkorder:
        out += ORD_TAG + h.trkorder.encode("ascii")[:19].ljust(20, b"\0")
    out += DAT_TAG + struct.pack("<BBBBQ", 0, 16, 0, 0, h.tstart_ns)
    return bytes(out)


def write_tbin(path: str, h: TbinHeader, rows: np.ndarray) -> None:
    """rows: int16 array [nsamples, ntrks] in head order."""
    rows = np.ascontiguousarray(rows, dtype="<i2")
    assert rows.ndim == 2 and rows.shape[1] == h.ntrks
    assert not (rows[:, 0] == END_MARK).any(), "0x8000 in head 0 is the end-of-data marker"
    with open(path, "wb") as f:
        f.write(pack_header(h))
        f.write(rows.tobytes())
        f.write(struct.pack("<h", END_MARK))


def parse_header(buf: bytes) -> tuple[TbinHeader, int]:
    """Returns (header, payload byte offset).  Raises ValueError like the reference's asserts
    (src/readtape.c:1322-1373) would exit."""
    if len(buf) < HDR_SIZE + 16 or buf[:8] != HDR_TAG:
        raise ValueError(".tbin file missing TBINHDR tag")
    descr = buf[8:88].split(b"\0", 1)[0].decode("ascii", "replace")
    hdrsize, fmt = struct.unpack_from("<II", buf, 88)
    if fmt != 1:
        raise ValueError("bad .tbin file header version")
    if hdrsize != HDR_SIZE:
        raise ValueError(f"bad .tbin hdr size: {hdrsize}, not {HDR_SIZE}")
    times = struct.unpack_from("<27i", buf, 96)
    flags, ntrks, tdelta, maxvolts, _, _ = struct.unpack_from("<IIIfII", buf, 204)
    mode, bpi, ips = struct.unpack_from("<Iff", buf, 228)
    off = HDR_SIZE
    order = ""
    if flags & FLAG_TRKORDER_INCLUDED:
        if buf[off:off + 8] != ORD_TAG:
            raise ValueError(".tbin file missing TBINORD tag")
        order = buf[off + 8:off + 28].split(b"\0", 1)[0].decode("ascii")
        off += 28
    if buf[off:off + 4] != DAT_TAG:
        raise ValueError(".tbin file missing DAT tag")
    _opts, bits, _, _, tstart = struct.unpack_from("<BBBBQ", buf, off + 4)
    if bits != 16:
        raise ValueError(f"we support only 16 bits/sample, not {bits}")
    off += 16
    h = TbinHeader(ntrks=ntrks, tdelta_ns=tdelta, maxvolts=maxvolts, mode=mode, bpi=bpi, ips=ips,
                   flags=flags & ~FLAG_TRKORDER_INCLUDED, tstart_ns=tstart, descr=descr,
                   trkorder=order, times=tuple(times))
    return h, off


def read_header(path: str) -> tuple[TbinHeader, int]:
    """(header, payload byte offset) of a .tbin file without reading the payload."""
    with open(path, "rb") as f:
        return parse_header(f.read(4096))


def read_tbin(path: str) -> tuple[TbinHeader, np.ndarray]:
    """Returns (header, rows[nsamples, ntrks] int16), rows cut at the 0x8000 end marker
    (src/readtape.c:1410: only the head-0 slot is tested)."""
    with open(path, "rb") as f:
        buf = f.read()
    h, off = parse_header(buf)
    payload = np.frombuffer(buf, dtype="<i2", offset=off)
    nfull = payload.size // h.ntrks
    rows = payload[: nfull * h.ntrks].reshape(nfull, h.ntrks)
    ends = np.flatnonzero(rows[:, 0] == END_MARK)
    if ends.size:
        rows = rows[: ends[0]]
    return h, rows
