"""-zeros: k_zeros (two tracks per lane on packed 16-bit arithmetic, rtfe_zeros.hip) against k_decode's zero-crossing mode (one track
per lane, 32-bit, rtfe_kernels.hip) and its purely sequential walk - two independent implementations of lookfor_zerocrossing
(src/decoder.c:617-649) that must agree byte for byte.  Shared by the emulator and the GPU test."""
import dataclasses

import numpy as np

from readtape_amd import frontend, synth

KNOBS = ("RTFE_ZEROS_KERNEL", "RTFE_TILE_ROWS", "RTFE_ZC_WARM", "RTFE_ZC_PARALLEL")


def zeros_rows(ntrks, nblocks, seed=77, clip=False, noise_mv=30.0):
    """Rows for a -zeros scan with `ntrks` columns: the first columns of a noisy 9-track PE tape whose amplitude drifts (what matters to
    the detector is that the columns carry flux changes, gaps and weak stretches - not that the tape decodes).  clip: a few samples at
    the ends of the int16 range, zeros at sign changes."""
    tape = synth.pe_tape(seed=seed, nblocks=nblocks, minlen=200, maxlen=1500, gap_samples=6000, noise_mv=noise_mv, amp_slope=0.15)
    rows = np.ascontiguousarray(np.concatenate([tape.rows] * 3, 1)[:, :ntrks])
    assert rows.shape[1] == ntrks
    if clip:                                      # (inside the blocks: the gaps stay dead quiet and the bursts apart)
        rng = np.random.default_rng(seed)
        flat = rows.reshape(-1)
        for sel, val in ((flat > 8000, 32767), (flat < -8000, -32768), (np.abs(flat) < 3000, 0)):
            at = np.flatnonzero(sel)
            flat[rng.choice(at, min(len(at), max(200, len(at) // 50)), replace=False)] = val
    hdr = dataclasses.replace(tape.spec.header(), ntrks=ntrks)
    return hdr, rows


def scan_variants(make_fe, hdr, rows, monkeypatch, variants, **cfgkw):
    """One scan per knob set -> [ScanResult]; the first is the yardstick of same_scan()."""
    out = []
    for knobs in variants:
        for k in KNOBS:
            monkeypatch.delenv(k, raising=False)
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        fe = make_fe(frontend.FrontEndConfig.from_header(hdr, find_zeros=True, **cfgkw))
        out.append(fe.scan(rows).fetch())
    return out


def same_scan(r0, r, ntrks):
    assert r.nbursts == r0.nbursts and (r.counts == r0.counts).all()
    for k in ("zone_first", "zone_end", "reset_sample", "safe_last", "end_sample", "flags"):
        assert (r.bursts[k] == r0.bursts[k]).all(), k
    for b in range(r0.nbursts):
        for t in range(ntrks):
            assert r.track_events(b, 0, t).tobytes() == r0.track_events(b, 0, t).tobytes(), (b, t)
