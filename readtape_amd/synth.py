"""Synthetic magnetic-tape waveform generator (TBIN rows) for tests and benchmarks.

The reference ships no generator and its example inputs are absent from the mount, so parity is
pinned on tapes made here.  Waveform model: every flux transition on a track is read back as a
Lorentzian pulse  +-A / (1 + ((t - c) / w)^2)  with alternating sign, plus Gaussian noise and
per-transition jitter; samples are quantised exactly like the TBIN writer side
(src/csvtbin.h:98-101: round(v / maxvolts * 32767), clipped to +-32767).

Block formats follow what the reference's decoders accept:
  * 9/7-track NRZI with CRC/LRC trailer and tapemarks  (src/decode_nrzi.c:35-75, 97-101)
  * 9-track PE with 40-zero preamble / postamble        (src/decode_pe.c:127-155, 33-102)
  * 9-track GCR 6250 with 5-4 group coding and ECC      (src/decode_gcr.c:118-144, 430-486)
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import tbin


# --------------------------------------------------------------------------------------------
# waveform rendering
# --------------------------------------------------------------------------------------------

def _render_pulses(nsamp: int, centers: np.ndarray, amps: np.ndarray, w: float,
                   reach: float = 16.0) -> np.ndarray:
    """Sum of truncated Lorentzians.  centers in (fractional) sample units, amps signed volts.
    Each pulse is evaluated on |d| <= K = ceil(reach*w) and shifted so it is continuous at the
    truncation edge."""
    out = np.zeros(nsamp, dtype=np.float64)
    if centers.size == 0:
        return out
    K = int(np.ceil(reach * w))
    offs = np.arange(-K, K + 1)
    edge = 1.0 / (1.0 + (K / w) ** 2)
    # chunk to bound memory
    step = max(1, 2_000_000 // offs.size)
    for i in range(0, centers.size, step):
        c = centers[i:i + step]
        a = amps[i:i + step]
        base = np.rint(c).astype(np.int64)
        idx = base[:, None] + offs[None, :]
        d = (idx - c[:, None]) / w
        val = a[:, None] * (1.0 / (1.0 + d * d) - edge)
        ok = (idx >= 0) & (idx < nsamp)
        out += np.bincount(idx[ok], weights=val[ok], minlength=nsamp)[:nsamp]
    return out


def _quantise(v: np.ndarray, maxvolts: float) -> np.ndarray:
    q = np.rint(v / maxvolts * 32767.0)
    return np.clip(q, -32767, 32767).astype(np.int16)


@dataclass
class TapeSpec:
    """Physical parameters of a synthetic tape."""
    mode: int
    ntrks: int = 9
    bpi: float = 800.0
    ips: float = 50.0
    tdelta_ns: int = 1280
    maxvolts: float = 4.4
    amplitude: float = 2.5          # volts, track 0; rises slightly with track number
    amp_slope: float = 0.03         # per-track amplitude increment (fraction)
    pulse_w: float = 0.22           # Lorentzian half-width in bit cells
    noise_mv: float = 10.0
    jitter: float = 0.02            # sigma of transition position, in bit cells
    skew_cells: tuple = ()          # optional per-track static skew, in bit cells
    tstart_ns: int = 1_000_000
    seed: int = 1
    flags: int = 0                  # TBIN header flags (tbin.FLAG_NO_REORDER: the columns are in head order, -order= applies)
    trkorder: str = ""              # TBINORD header extension

    @property
    def samples_per_bit(self) -> float:
        return 1.0 / (self.bpi * self.ips * self.tdelta_ns * 1e-9)

    def header(self) -> tbin.TbinHeader:
        return tbin.TbinHeader(ntrks=self.ntrks, tdelta_ns=self.tdelta_ns, maxvolts=self.maxvolts,
                               mode=self.mode, bpi=self.bpi, ips=self.ips, tstart_ns=self.tstart_ns, flags=self.flags, trkorder=self.trkorder,
                               descr=f"synthetic {tbin.MODE_NAMES.get(self.mode)} seed {self.seed}")


@dataclass
class Tape:
    spec: TapeSpec
    rows: np.ndarray                       # int16 [nsamples, ntrks], head order == track order
    blocks: list = field(default_factory=list)   # [(kind, start_sample, end_sample, payload bytes)]

    def write(self, path: str) -> None:
        tbin.write_tbin(path, self.spec.header(), self.rows)


# --------------------------------------------------------------------------------------------
# NRZI
# --------------------------------------------------------------------------------------------

def _parity9(b: int) -> int:
    return bin(b).count("1") & 1


def nrzi_words(payload: bytes, ntrks: int = 9, odd_parity: bool = True) -> list[int]:
    """Data words (data<<1 | P) followed by the 8 trailer cells.
    9-trk: 0,0,0,CRC,0,0,0,LRC with CRC/LRC as the checker computes them
    (src/decode_nrzi.c:55-69); 7-trk: 0,0,0,LRC,0,0,0,0 (src/decode_nrzi.c:44,50-51)."""
    databits = ntrks - 1
    words = []
    for b in payload:
        b &= (1 << databits) - 1
        p = _parity9(b) ^ (1 if odd_parity else 0)
        words.append((b << 1) | p)
    crc = 0
    lrc = 0
    for w in words:
        lrc ^= w
        crc ^= w
        if crc & 2:
            crc ^= 0xF0
        lsb = crc & 1
        crc >>= 1
        if lsb:
            crc |= 0x100
    crc ^= 0x1AF
    if ntrks == 9:
        lrc ^= crc
        return words + [0, 0, 0, crc, 0, 0, 0, lrc]
    return words + [0, 0, 0, lrc, 0, 0, 0, 0]


def nrzi_tapemark_words(ntrks: int = 9) -> list[int]:
    # src/decode_nrzi.c:97-100
    if ntrks == 9:
        return [0x26, 0, 0, 0, 0, 0, 0, 0, 0x26]
    return [0x1E, 0, 0, 0, 0x1E]


def _words_to_transitions(words: list[int], ntrks: int) -> list[np.ndarray]:
    """NRZI: a 1 bit in cell k on track t = one flux transition at the centre of cell k."""
    w = np.asarray(words, dtype=np.int64)
    return [np.flatnonzero((w >> (ntrks - 1 - t)) & 1).astype(np.float64) for t in range(ntrks)]


def _render_block(spec: TapeSpec, rng, per_track_cells: list[np.ndarray], ncells: float,
                  lead_cells: float = 8.0, first_sign: int = +1,
                  per_track_first_sign=None) -> np.ndarray:
    """Render one block: per_track_cells[t] = fractional cell positions of flux transitions on
    track t.  Returns float volts [nsamp, ntrks] (no noise)."""
    spb = spec.samples_per_bit
    nsamp = int(np.ceil((ncells + 2 * lead_cells) * spb))
    out = np.zeros((nsamp, spec.ntrks), dtype=np.float64)
    for t, cells in enumerate(per_track_cells):
        if cells.size == 0:
            continue
        skew = spec.skew_cells[t] if t < len(spec.skew_cells) else 0.0
        jit = rng.normal(0.0, spec.jitter, size=cells.size) if spec.jitter > 0 else 0.0
        centers = (lead_cells + cells + 0.5 + skew + jit) * spb
        amp = spec.amplitude * (1.0 + spec.amp_slope * t)
        s0 = first_sign if per_track_first_sign is None else per_track_first_sign[t]
        signs = s0 * np.where(np.arange(cells.size) % 2 == 0, 1.0, -1.0)
        out[:, t] = _render_pulses(nsamp, centers, amp * signs, spec.pulse_w * spb)
    return out


def make_tape(spec: TapeSpec, items: list, gap_samples: int = 5000) -> Tape:
    """items: list of ("block", payload_bytes) | ("block", payload_bytes, gcr_flips) | ("mark",) | ("gap", nsamples) | ("raw", cells_per_track, ncells).
    A gap of `gap_samples` is placed before, between and after items."""
    rng = np.random.default_rng(spec.seed)
    pieces = []
    blocks = []
    pos = 0

    def add_gap(n):
        nonlocal pos
        pieces.append(np.zeros((n, spec.ntrks), dtype=np.float64))
        pos += n

    add_gap(gap_samples)
    for it in items:
        kind = it[0]
        if kind == "gap":
            add_gap(int(it[1]))
            continue
        if kind == "block":
            cells, ncells, fs = encode_block(spec, it[1]) if len(it) < 3 else gcr_encode(it[1], it[2])
        elif kind == "mark":
            cells, ncells, fs = encode_mark(spec)
        elif kind == "raw":
            cells, ncells, fs = it[1], it[2], (it[3] if len(it) > 3 else None)
        else:
            raise ValueError(kind)
        v = _render_block(spec, rng, cells, ncells, per_track_first_sign=fs)
        blocks.append((kind, pos, pos + v.shape[0], it[1] if kind == "block" else b""))
        pieces.append(v)
        pos += v.shape[0]
        add_gap(gap_samples)
    rows = np.empty((pos, spec.ntrks), dtype=np.int16)
    at = 0
    for p in pieces:
        n = p.shape[0]
        noise = rng.normal(0.0, spec.noise_mv * 1e-3, size=p.shape) if spec.noise_mv > 0 else 0.0
        rows[at:at + n] = _quantise(p + noise, spec.maxvolts)
        at += n
    return Tape(spec=spec, rows=rows, blocks=blocks)


def encode_block(spec: TapeSpec, payload: bytes):
    if spec.mode == tbin.MODE_NRZI:
        words = nrzi_words(payload, spec.ntrks)
        return _words_to_transitions(words, spec.ntrks), float(len(words)), None
    if spec.mode == tbin.MODE_PE:
        return pe_encode(payload, spec.ntrks)
    if spec.mode == tbin.MODE_GCR:
        return gcr_encode(payload)
    raise ValueError("unsupported mode for encode_block")


def encode_mark(spec: TapeSpec):
    if spec.mode == tbin.MODE_NRZI:
        words = nrzi_tapemark_words(spec.ntrks)
        return _words_to_transitions(words, spec.ntrks), float(len(words)), None
    if spec.mode == tbin.MODE_PE:
        return pe_tapemark(spec.ntrks)
    raise ValueError("tapemark not implemented for this mode")


# --------------------------------------------------------------------------------------------
# PE (1600 BPI phase encoding)
# --------------------------------------------------------------------------------------------

def _pe_track_transitions(bits: np.ndarray):
    """Manchester: a 1 = upward flux transition at the cell centre, 0 = downward; a phase
    transition at the cell boundary whenever two equal bits follow each other.
    Returns (cell positions, first pulse sign).  Pulse sign = transition direction."""
    pos = []
    prev = None
    for k, b in enumerate(bits):
        if prev is not None and b == prev:
            pos.append(float(k) - 0.5)      # boundary (phase) transition; cell centre is k+0.5 later
        pos.append(float(k))
        prev = b
    first_sign = +1 if bits[0] == 1 else -1
    return np.asarray(pos, dtype=np.float64), first_sign


def pe_encode(payload: bytes, ntrks: int = 9, pre: int = 40, post: int = 40):
    """Per track: `pre` zero bits, a one, data, a one, `post` zero bits (IBM PE preamble and
    postamble; src/decode_pe.c:127-145 waits for >70 peaks then a late 1)."""
    per = []
    signs = []
    words = [((b << 1) | (_parity9(b) ^ 1)) for b in payload]
    for t in range(ntrks):
        data = [(w >> (ntrks - 1 - t)) & 1 for w in words]
        bits = np.asarray([0] * pre + [1] + data + [1] + [0] * post, dtype=np.int64)
        p, s = _pe_track_transitions(bits)
        per.append(p)
        signs.append(s)
    ncells = pre + 1 + len(payload) + 1 + post
    return per, float(ncells), signs


def pe_tapemark(ntrks: int = 9, nflux: int = 90):
    """PE tapemark: >= 80 flux reversals on tracks 0,2,5,6,7,P and none on 1,3,4
    (src/decode_pe.c:38-53)."""
    per = []
    signs = []
    for t in range(ntrks):
        if t in (1, 3, 4):
            per.append(np.zeros(0))
            signs.append(+1)
        else:
            bits = np.zeros(nflux // 2, dtype=np.int64)
            p, s = _pe_track_transitions(bits)
            per.append(p)
            signs.append(s)
    return per, float(nflux // 2), signs


# --------------------------------------------------------------------------------------------
# GCR (6250)
# --------------------------------------------------------------------------------------------

# 4 -> 5 bit map, the inverse of the decoder's table (src/decode_gcr.c:430-436)
GCR_4TO5 = [0b11001, 0b11011, 0b10010, 0b10011, 0b11101, 0b10101, 0b10110, 0b10111,
            0b11010, 0b01001, 0b01010, 0b01011, 0b11110, 0b01101, 0b01110, 0b01111]
# ECC: bit i of the check character = parity of (56-bit big-endian data word AND row i)
# (the matrix the decoder checks against, src/decode_gcr.c:128-136)
_GCR_ECC_ROWS = [0x0f6a71994c5230, 0x70110840108004, 0x5a701108401080, 0x372be95d5a7011,
                 0xe95d5a70110840, 0x4c523001884412, 0x2be95d5a701108, 0x5d5a7011084010]


def gcr_ecc(seven: bytes) -> int:
    word = int.from_bytes(bytes(seven), "big")
    return sum((bin(word & row).count("1") & 1) << i for i, row in enumerate(_GCR_ECC_ROWS))


def _gcr_group_cells(chars8: list[int], flips=()) -> list[list[int]]:
    """8 characters (7 data + check) -> 10 cells per track (two 5-bit storage groups), as 9-bit words.
    flips: (character 0..7, track 0..8) pairs whose recorded bit is wrong (written after the ECC and parity were formed:
    what the decoder's -correct repair is for, src/decode_gcr.c:588-611)."""
    words = [((c & 0xFF) << 1) | (_parity9(c & 0xFF) ^ 1) for c in chars8]         # odd parity on track 8
    for ch, trk in flips:
        words[ch] ^= 1 << (8 - trk)
    cells = [0] * 10
    for trk in range(9):
        bits = [(w >> (8 - trk)) & 1 for w in words]
        for half in range(2):
            nib = (bits[4 * half] << 3) | (bits[4 * half + 1] << 2) | (bits[4 * half + 2] << 1) | bits[4 * half + 3]
            code = GCR_4TO5[nib]
            for k in range(5):                                                        # MSB first
                if (code >> (4 - k)) & 1:
                    cells[5 * half + k] |= 1 << (8 - trk)
    return cells


def gcr_encode(payload: bytes, flips=None):
    """One 6250 block: preamble, data groups, end mark, residual group, CRC group, postamble
    (what src/decode_gcr.c:503-674 walks through).  Same cells on every track
    for the control subgroups.  flips: {data group index: [(character, track), ...]} recorded-bit errors."""
    flips = flips or {}
    ALL = 0x1FF

    def ctl(code):                     # a 5-bit control subgroup on all nine tracks
        return [ALL if (code >> (4 - k)) & 1 else 0 for k in range(5)]

    cells = []
    cells += ctl(0b10101) + ctl(0b01111)
    for _ in range(14):
        cells += ctl(0b11111)
    cells += ctl(0b00111)                                            # MARK1: data starts
    nfull, left = divmod(len(payload), 7)
    for g in range(nfull):
        seven = payload[7 * g: 7 * g + 7]
        cells += _gcr_group_cells(list(seven) + [gcr_ecc(seven)], flips.get(g, ()))
    cells += ctl(0b11111)                                            # end of data groups
    resid = list(payload[7 * nfull:]) + [0] * (6 - left) + [0]       # H H H H H H N
    cells += _gcr_group_cells(resid + [gcr_ecc(bytes(resid))])
    crc = [0, 0, 0, 0, 0, 0, (left << 5) & 0xFF]                     # B C C C C C X ; X's top 3 bits = residual count
    cells += _gcr_group_cells(crc + [gcr_ecc(bytes(crc))])
    cells += ctl(0b11100)                                            # MARK2
    for _ in range(14):
        cells += ctl(0b11111)
    cells += ctl(0b11110) + ctl(0b10101)
    return _words_to_transitions(cells, 9), float(len(cells)), None


# --------------------------------------------------------------------------------------------
# convenience builders
# --------------------------------------------------------------------------------------------

def nrzi_spec(seed: int = 1, ntrks: int = 9, **kw) -> TapeSpec:
    return TapeSpec(mode=tbin.MODE_NRZI, ntrks=ntrks, bpi=800.0, ips=50.0, tdelta_ns=1280,
                    maxvolts=4.4, pulse_w=0.22, seed=seed, **kw)


def gcr_spec(seed: int = 1, **kw) -> TapeSpec:
    kw.setdefault("pulse_w", 0.4)
    kw.setdefault("amplitude", 1.8)
    return TapeSpec(mode=tbin.MODE_GCR, ntrks=9, bpi=9042.0, ips=50.0, tdelta_ns=160, maxvolts=3.3, seed=seed, **kw)


def gcr_tape(seed: int = 1, nblocks: int = 3, minlen: int = 100, maxlen: int = 1000, gap_samples: int = 6000, **kw) -> Tape:
    spec = gcr_spec(seed=seed, **kw)
    rng = np.random.default_rng(seed + 3000)
    items = [("block", p) for p in random_payloads(rng, nblocks, minlen, maxlen)]
    return make_tape(spec, items, gap_samples=gap_samples)


def pe_spec(seed: int = 1, **kw) -> TapeSpec:
    return TapeSpec(mode=tbin.MODE_PE, ntrks=9, bpi=1600.0, ips=50.0, tdelta_ns=640,
                    maxvolts=4.4, pulse_w=0.13, seed=seed, **kw)


def random_payloads(rng, nblocks: int, minlen: int, maxlen: int, databits: int = 8) -> list[bytes]:
    out = []
    for _ in range(nblocks):
        n = int(rng.integers(minlen, maxlen + 1))
        out.append(bytes(rng.integers(0, 1 << databits, size=n, dtype=np.int64).astype(np.uint8)))
    return out


def nrzi_tape(seed: int = 1, nblocks: int = 4, minlen: int = 64, maxlen: int = 512,
              marks_every: int = 0, ntrks: int = 9, gap_samples: int = 5000, **kw) -> Tape:
    spec = nrzi_spec(seed=seed, ntrks=ntrks, **kw)
    rng = np.random.default_rng(seed + 1000)
    items = []
    for i, p in enumerate(random_payloads(rng, nblocks, minlen, maxlen, databits=ntrks - 1)):
        items.append(("block", p))
        if marks_every and (i + 1) % marks_every == 0:
            items.append(("mark",))
    return make_tape(spec, items, gap_samples=gap_samples)


def pe_tape(seed: int = 1, nblocks: int = 3, minlen: int = 64, maxlen: int = 300,
            gap_samples: int = 5000, **kw) -> Tape:
    spec = pe_spec(seed=seed, **kw)
    rng = np.random.default_rng(seed + 2000)
    items = [("block", p) for p in random_payloads(rng, nblocks, minlen, maxlen)]
    return make_tape(spec, items, gap_samples=gap_samples)


# --------------------------------------------------------------------------------------------
# Whirlwind I (6 tracks, 100 BPI): two data tracks (the MSB and LSB of a 2-bit character) and a clock track, each recorded
# twice (src/decode_ww.c, src/readtape.c:367).  A bit is a PULSE - a flux change and its return, here 0.35 cell apart, the
# negative half first (-fluxdir=neg, the reference's default).  Every cell has a clock pulse; a data track has a pulse where its
# bit is one.  A block is a run of 16-bit words, eight characters each, most significant first; a block mark is a lone pulse on
# the LSB tracks while the clock is silent.
# --------------------------------------------------------------------------------------------
WW_ORDER = "CMLcml"


def ww_spec(seed: int = 1, **kw) -> TapeSpec:
    kw.setdefault("pulse_w", 0.07)
    kw.setdefault("amplitude", 2.0)
    kw.setdefault("amp_slope", 0.02)
    kw.setdefault("jitter", 0.01)
    return TapeSpec(mode=tbin.MODE_WW, ntrks=6, bpi=100.0, ips=50.0, tdelta_ns=5000, maxvolts=4.4, seed=seed,
                    flags=tbin.FLAG_NO_REORDER, trkorder=WW_ORDER, **kw)


def _ww_pulses(cells_with_pulse):
    c = np.asarray(cells_with_pulse, dtype=np.float64)
    return np.sort(np.concatenate([c, c + 0.35]))


def ww_block_cells(words: list[int]):
    """per-track transition positions (cell units) of a block of 16-bit words, tracks in WW_ORDER; -> (cells, ncells, first signs)"""
    chars = []
    for w in words:
        for k in range(8):
            chars.append((w >> (14 - 2 * k)) & 3)
    n = len(chars)
    clk = list(range(n))
    msb = [i for i, c in enumerate(chars) if c & 2]
    lsb = [i for i, c in enumerate(chars) if c & 1]
    by_type = {"C": clk, "c": clk, "M": msb, "m": msb, "L": lsb, "l": lsb}
    return [_ww_pulses(by_type[ch]) for ch in WW_ORDER], float(n), [-1] * 6


def ww_mark_cells():
    by_type = {"C": [], "c": [], "M": [], "m": [], "L": [0], "l": [0]}
    return [_ww_pulses(by_type[ch]) for ch in WW_ORDER], 1.0, [-1] * 6


def ww_tape(seed: int = 1, nblocks: int = 4, minwords: int = 4, maxwords: int = 24, marks_every: int = 0, gap_samples: int = 800, **kw) -> Tape:
    spec = ww_spec(seed=seed, **kw)
    rng = np.random.default_rng(seed + 4000)
    items = []
    for i in range(nblocks):
        words = [int(x) for x in rng.integers(0, 1 << 16, size=int(rng.integers(minwords, maxwords + 1)))]
        cells, n, fs = ww_block_cells(words)
        items.append(("raw", cells, n, fs))
        if marks_every and (i + 1) % marks_every == 0:
            cells, n, fs = ww_mark_cells()
            items.append(("raw", cells, n, fs))
    return make_tape(spec, items, gap_samples=gap_samples)
