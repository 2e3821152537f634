"""Shapes written over the peaks of clean tapes: the inputs of the shape fuzzer (tools/fuzz_shapes.py) and of the tests that pin what it found.

Gaussian noise almost never draws the shapes that decide whether a record may fire on the chains' lean step - two extremes of nearly the same height
inside one window, a narrow valley right behind a flat top, a notch in a shoulder, a stale minimum next to the true one.  shape_tape() writes such shapes
over a share of a clean tape's peaks: every sample of the window either side of a chosen peak drawn from a mixture of "a hair below the peak", "a little
below", "well below" (wild), or moved by a few per cent of the peak (mild: what a real head could deliver).  Test infrastructure."""
import numpy as np

from readtape_amd import synth

KINDS = ("nrzi9", "nrzi9_m", "nrzi7", "gcr", "gcr_m", "pe")


def base_tape(kind, seed, noise_mv):
    """(tape, oracle options) of one of the formats the peak path takes"""
    if kind.startswith("nrzi"):
        ntrks = 7 if kind.startswith("nrzi7") else 9
        tape = synth.nrzi_tape(seed=seed, nblocks=4, minlen=150, maxlen=400, gap_samples=3000, noise_mv=noise_mv, ntrks=ntrks)
    elif kind.startswith("gcr"):
        tape = synth.gcr_tape(seed=seed, nblocks=3, minlen=100, maxlen=300, gap_samples=4000, noise_mv=noise_mv)
    else:
        tape = synth.pe_tape(seed=seed, nblocks=3, minlen=64, maxlen=200, gap_samples=4000, noise_mv=noise_mv)
    return tape, (["-m"] if kind.endswith("_m") else [])


def shape_rows(rows0, rng, density=0.25, reach=14, wild=1.0, first=24):
    """rows0 with random shapes written over a share of its peaks (from peak `first` of a track's block on - from the 24th the chains are steady, before it they learn
    their baseline: the start-up path); returns (rows, sites).
    wild: the share of sites drawn from the full mixture; the others only move samples by a few per cent of the peak (what a real head could deliver)."""
    rows = rows0.copy()
    nrows, ntrks = rows.shape
    nsites = 0
    for t in range(ntrks):
        x = rows0[:, t].astype(np.int64)
        amp = np.abs(x).max()
        # local extremes well above the noise
        mid = x[1:-1]
        tops = np.flatnonzero((mid > x[:-2]) & (mid >= x[2:]) & (mid > 0.4 * amp)) + 1
        bots = np.flatnonzero((mid < x[:-2]) & (mid <= x[2:]) & (mid < -0.4 * amp)) + 1
        peaks = np.sort(np.concatenate([tops, bots]))
        if peaks.size < 40:
            continue
        # blocks: a gap of more than 200 rows between peaks
        starts = np.concatenate([[0], np.flatnonzero(np.diff(peaks) > 200) + 1, [peaks.size]])
        for a, b in zip(starts[:-1], starts[1:]):
            for k in range(a + first, b - 2):
                if rng.random() >= density:
                    continue
                P = int(peaks[k])
                v = int(x[P])
                s = 1 if v > 0 else -1
                mag = abs(v)
                mild = rng.random() >= wild
                right = int(rng.integers(2, reach + 1))
                left = int(rng.integers(0, reach // 2 + 1)) if rng.random() < 0.5 else 0
                for off in list(range(1, right + 1)) + [-o for o in range(1, left + 1)]:
                    r = P + off
                    if r < 0 or r >= nrows:
                        continue
                    if mild:                               # the sample where it was, give or take: plateaus, double tops, a stale minimum next to the true one
                        y = int(rows[r, t]) + int(round(mag * rng.normal(0.0, 0.03)))
                        if rng.random() < 0.3:
                            y = v - s * int(rng.integers(0, 4))
                        rows[r, t] = np.clip(y, -32767, 32767)
                        continue
                    kind_u = rng.random()
                    if kind_u < 0.30:
                        u = rng.uniform(0.0, 0.004)        # a hair below (ties and one-lsb steps included)
                    elif kind_u < 0.55:
                        u = rng.uniform(0.0, 0.05)         # under the screen, mostly
                    elif kind_u < 0.80:
                        u = rng.uniform(0.03, 0.30)        # around the rise threshold
                    else:
                        u = rng.uniform(0.3, 1.8)          # well below: the other polarity's territory
                    if rng.random() < 0.04:
                        u = -rng.uniform(0.0, 0.02)        # ... or a hair ABOVE: the peak is not the extreme after all
                    y = v - s * int(round(mag * u))
                    rows[r, t] = np.clip(y, -32767, 32767)
                nsites += 1
    return rows, nsites


def shape_tape(seed, density=0.25, noise_mv=5.0, reach=14, kind="nrzi9", wild=1.0, first=24):
    """A clean tape of one of KINDS with shape_rows() over it: (tape, rows, sites, oracle options)."""
    tape, opts = base_tape(kind, seed, noise_mv)
    rows, nsites = shape_rows(tape.rows, np.random.default_rng(seed * 7919 + 13), density=density, reach=reach, wild=wild, first=first)
    return tape, rows, nsites, opts


def draw(seed):
    """the parameters of tape `seed` (one place: the tool and tests/test_*fuzz* draw the same tapes)"""
    rng = np.random.default_rng(seed)
    d = dict(kind=str(rng.choice(KINDS)), density=float(rng.choice([0.05, 0.15, 0.4])), noise_mv=float(rng.choice([2.0, 10.0, 40.0])),
             wild=float(rng.choice([0.0, 0.5, 1.0])))
    if seed >= 100000:                                     # (later seeds: shapes over a block's first peaks too - the chains' start-up; the earlier seeds' tapes stay what they were)
        d["first"] = int(rng.choice([0, 3, 24]))
    return d
