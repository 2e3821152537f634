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
