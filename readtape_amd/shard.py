"""Time sharding of one tape across the GPUs of a node.

The sample timeline is cut into contiguous row ranges, one per rank.  Bursts are independent (each
starts from reset state inside a dead-quiet zone), so the only coupling is at the seams: the burst that
straddles a seam is finished by the LEFT rank, which therefore needs the first rows of the right
rank's range — one neighbour halo (isend/irecv over RCCL/xGMI, or gloo on CPU), no collective on the
data path.  `rtfe_scan(..., own_rows)` then decodes exactly the bursts whose zone ends in the owned rows.
"""
from __future__ import annotations

import numpy as np


def plan_shards(nrows: int, world: int, align: int = 1024):
    """[start, end) per rank.  Starts are multiples of `align` rows (a multiple of 64): 16-byte aligned for every track count, and
    on the 64-row grid of the quiet map, so a sharded scan finds exactly the zones of the whole-tape scan.  A tape too short to
    give every rank rows leaves the last ranks empty: (n, n)."""
    assert align % 64 == 0
    cuts = [min(nrows, (nrows * r // world) // align * align) for r in range(world)] + [nrows]
    for r in range(1, world):                       # (monotone; a rank whose slice would be empty gets (nrows, nrows) at the end)
        cuts[r] = max(cuts[r], cuts[r - 1])
    spans = [(cuts[r], cuts[r + 1]) for r in range(world)]
    live = [sp for sp in spans if sp[1] > sp[0]]
    return live + [(nrows, nrows)] * (world - len(live))


def exchange_halo(own, halo_rows: int, rank: int, world: int, dist):
    """own: [n, ntrks] int16 tensor holding this rank's rows.  Sends its first `halo_rows` rows to the left
    neighbour, receives the right neighbour's, and returns (rows_with_halo, own_rows)."""
    import torch
    n = own.shape[0]
    if world == 1:
        return own, n
    h = min(halo_rows, n)
    send = own[:h].contiguous()
    recv = torch.empty((halo_rows, own.shape[1]), dtype=own.dtype, device=own.device) if rank < world - 1 else None
    # neighbours may own fewer rows than the halo: agree on the length first (tiny, host side)
    lens = [None] * world
    dist.all_gather_object(lens, int(h))
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, send, rank - 1))
    if rank < world - 1:
        recv = recv[: lens[rank + 1]]
        ops.append(dist.P2POp(dist.irecv, recv, rank + 1))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if rank == world - 1:
        return own, n
    return torch.cat([own, recv], 0).contiguous(), n


def absolute_bursts(res, start_row: int):
    """Burst table of one rank with row fields shifted to tape-absolute rows."""
    b = res.bursts.copy()
    for f in ("zone_first", "zone_end", "reset_sample", "safe_last", "end_sample"):
        m = b[f] >= 0
        b[f][m] += start_row
    return b


def flatten_events(res, bursts_abs, parmset: int):
    """[(absolute detection row, trk, kind, v_peak bits, agc bits, left_distance, adj)] over all bursts of one rank."""
    out = []
    for i in range(res.nbursts):
        ev = res.events(i, parmset)
        n0 = bursts_abs[i]["reset_sample"] + ev["sample"].astype(np.int64)
        out.append(np.stack([n0, ev["trk"].astype(np.int64), ev["flags"].astype(np.int64), ev["v_peak"].view("u4").astype(np.int64),
                             ev["agc_gain"].view("u4").astype(np.int64), ev["left_distance"].astype(np.int64)], 1))
    return np.concatenate(out) if out else np.zeros((0, 6), np.int64)


def decode_sharded(hdr, own, lo: int, n_total: int, rank: int, world: int, dist, tap_path: str | None, opts=None, fe_factory=None,
                   halo_rows: int = 1 << 16, cfgkw=None):
    """The whole multi-rank decode of ONE tape: `own` = this rank's rows [lo, lo + len(own)) of the n_total-row tape.
      1. neighbour halo exchange (isend/irecv; repeated with a longer halo while some rank's last own burst runs past it),
      2. rtfe_scan of own + halo with the own_rows ownership rule,
      3. all-gather of one small record per rank {own range, bursts, events per parameter set, .tap bytes} (the only collective),
      4. host replay of the own bursts -> this rank's piece of the .tap,
      5. rank 0 gathers the pieces in rank order, concatenates them and ends the file (tap_path; other ranks pass None or the same path).
    Returns the list of per-rank records (every rank) - their 'tap_offset' fields are the prefix sums of the pieces."""
    import torch
    from . import frontend, pipeline
    opts = opts or pipeline.DecodeOptions()
    full = pipeline.default_parmsets(hdr.mode, opts.nparmsets or (15 if opts.multiple_tries else 1))
    cfg = frontend.FrontEndConfig.from_header(hdr, parmsets=pipeline.frontend_parmsets(full), **(cfgkw or {}))
    fe = (fe_factory or frontend.FrontEnd)(cfg)
    n = int(own.shape[0])
    is_last = lo + n >= n_total
    halo = halo_rows
    while True:
        with_halo, own_rows = exchange_halo(own, halo, rank, world, dist)
        piece = with_halo if isinstance(with_halo, np.ndarray) or fe_factory is None else with_halo.numpy()
        res, nb, bound = pipeline.scan_fragment(fe, piece, own_rows, lo, lo == 0, is_last)() if n > 0 else (None, 0, None)
        got_all = lo + int(piece.shape[0]) >= n_total
        more = torch.tensor([1 if (n > 0 and nb is None and not got_all) else 0])
        dist.all_reduce(more, op=dist.ReduceOp.MAX)          # (host-side agreement, a few bytes)
        if int(more.item()) == 0:
            break
        halo *= 4
    tap_bytes = b""
    stats = dict(blocks=0, tapemarks=0, events_delivered=0)
    if n > 0 and res.nbursts > 0:
        import os, tempfile
        with tempfile.TemporaryDirectory() as wd:
            frag = os.path.join(wd, f"r{rank}.tap")
            start = 0 if lo == 0 else int(res.bursts[0]["zone_first"])
            stats = pipeline.decode_fragment(hdr, cfg, fe, res, piece, lo, start, bound, frag, full, opts, fe_factory)
            tap_bytes = open(frag, "rb").read()
    mine = dict(rank=rank, lo=lo, hi=lo + n, bursts=int(res.nbursts) if n > 0 else 0,
                events=[int(x) for x in res.counts.sum(axis=(0, 2))] if n > 0 and res.nbursts else [0] * len(full),
                blocks=int(stats["blocks"]), tapemarks=int(stats["tapemarks"]), tap_len=len(tap_bytes))
    table = [None] * world
    dist.all_gather_object(table, mine)
    off = 0
    for rec in table:
        rec["tap_offset"] = off
        off += rec["tap_len"]
    pieces = [None] * world if rank == 0 else None
    dist.gather_object(tap_bytes, pieces, dst=0)
    if rank == 0 and tap_path:
        with open(tap_path, "wb") as f:
            for pc in pieces:
                f.write(pc)
            if off > 0:
                f.write(b"\xff\xff\xff\xff")                # src/readtape.c:1885
    fe.close()
    return table
