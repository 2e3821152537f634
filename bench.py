#!/usr/bin/env python
"""bench.py — throughput of the MI355X analog front end on BASELINE.json's workload.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by the driver through torch.distributed.run, one rank per GPU, RCCL)

Workload at N = 1 (BASELINE.json configs[1], "C2"): synthetic 9-track 800 BPI NRZI, 781.25 kHz,
1e8 sample instants (1.8 GB of interleaved int16, resident in HBM before the timed region),
1 parameter set.  One "step" = one full pass of the hot path over that tape:
k_quiet -> k_bursts -> k_decode, events written to HBM.  Metric = sample instants per second
(the reference's own unit: "N samples were processed", src/readtape.c:2024), whole job.

N > 1: the sample timeline is time-sharded — every rank holds its own 1e8-row shard (weak scaling);
the only exchange is one neighbour halo per step (the rows a block straddling the seam needs), no
data-path collective.

Also on the line:  roofline (dominant kernel, algorithmic bytes / measured kernel time, HIP events)
and cpu_baseline (the reference compiled by oracle/Makefile when it travelled with the snapshot,
else the oracle port; single core; bounded sample of the same tape).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_base_tape(seed, target_rows):
    """Unique synthetic NRZI tape of about target_rows rows: 512..4096-byte blocks, >= 6 ms gaps,
    a tapemark every 16 blocks."""
    from readtape_amd import synth
    rng = np.random.default_rng(seed)
    spb = synth.nrzi_spec().samples_per_bit
    nblocks = max(4, int(target_rows / ((2304 + 8 + 16) * spb + 6000)))
    tape = synth.nrzi_tape(seed=seed, nblocks=nblocks, minlen=512, maxlen=4096, marks_every=16, gap_samples=6000)
    return tape


def cpu_baseline(tape, copies):
    """Times the CPU path on `copies` concatenated copies of the base tape (single core)."""
    from readtape_amd import tbin
    hdr = tape.spec.header()
    rows = np.tile(tape.rows, (copies, 1))
    ref = os.path.join(ROOT, "oracle", "_ref", "readtape_ref")
    port = os.path.join(ROOT, "oracle", "_build", "oracle_readtape")
    out = {}
    with tempfile.TemporaryDirectory() as wd:
        path = os.path.join(wd, "b.tbin")
        tbin.write_tbin(path, hdr, rows)
        nrows = rows.shape[0]
        del rows
        if not os.path.exists(port):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
        p = subprocess.run([port, "-time", f"-out={wd}/o", path], capture_output=True, text=True)
        j = json.loads(p.stdout.strip().splitlines()[-1])
        out["port_msamples_per_s"] = j["msamples_per_s"]
        port_tap = open(f"{wd}/o.tap", "rb").read()
        value, kind = j["msamples_per_s"], "port"
        if os.path.exists(ref) and os.access(ref, os.X_OK):
            t0 = time.perf_counter()
            p = subprocess.run([ref, "-nm", "-tap", "-nolabels", "-q", "b"], cwd=wd, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if p.returncode == 0 and os.path.exists(f"{wd}/b.tap"):
                value, kind = nrows / dt / 1e6, "reference"
                out["tap_identical_to_reference"] = open(f"{wd}/b.tap", "rb").read() == port_tap
    out.update(value=round(value, 3), unit="Msamples/s", cores=1, kind=kind,
               sample=f"{nrows} rows ({copies} copies of the base tape) of the same synthetic NRZI tape, whole pipeline incl. bit decoding, single thread")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=float, default=1e8, help="sample instants per GPU")
    ap.add_argument("--base-rows", type=float, default=5e6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", action="store_true", help="alternate two front-end contexts on two HIP streams (steps overlap; per-kernel times then include contention)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from readtape_amd import frontend

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    dev = torch.device(f"cuda:{local}")
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE {world}"

    # ---- synthetic tape, resident in HBM ----
    tape = make_base_tape(seed=1000 + rank, target_rows=int(args.base_rows))
    hdr = tape.spec.header()
    base = torch.from_numpy(tape.rows).to(dev)
    copies = max(1, int(round(args.rows / base.shape[0])))
    rows = base.repeat(copies, 1).contiguous()
    nrows = int(rows.shape[0])
    del base
    cfg = frontend.FrontEndConfig.from_header(hdr, nparmsets=1)
    fe = frontend.FrontEnd(cfg, device=str(dev))
    fe.set_timing(True)

    # time shards: this rank owns `nrows` rows; the tail of the buffer receives the right neighbour's first rows
    halo_rows = (1 << 18) if (world > 1 and rank < world - 1) else 0
    if halo_rows:
        buf = torch.empty((nrows + halo_rows, rows.shape[1]), dtype=rows.dtype, device=dev)
        buf[:nrows].copy_(rows)
        rows = buf
        del buf
    own_view = rows[:nrows]

    # Default: one stream, steps back to back (the per-kernel HIP-event times are then contention-free, which is what the
    # roofline line needs).  --pipeline alternates two front-end contexts (own HIP stream, workspace and outputs) so that
    # the latency-bound sequential pass of step i overlaps the dense pass of step i+1 (about 10 % more throughput).
    if args.pipeline:
        fes = [fe, frontend.FrontEnd(cfg, device=str(dev))]
        fes[1].set_timing(True)
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    else:
        fes = [fe, fe]
        streams = [torch.cuda.current_stream(dev)] * 2

    def step(i):
        s = streams[i & 1]
        with torch.cuda.stream(s):
            if world > 1:
                # the seam halo is the only exchange: neighbour isend/irecv over RCCL (xGMI), no collective on the data path
                ops = []
                if rank > 0: ops.append(dist.P2POp(dist.isend, own_view[:1 << 18], rank - 1))
                if rank < world - 1: ops.append(dist.P2POp(dist.irecv, rows[nrows:], rank + 1))
                for w in dist.batch_isend_irecv(ops): w.wait()
            return fes[i & 1].scan(rows, row_base=rank * nrows, first_is_tape_start=(rank == 0), own_rows=nrows, stream=s.cuda_stream)

    torch.cuda.synchronize(dev)
    for i in range(args.warmup):
        res = step(i)
    torch.cuda.synchronize(dev)
    if world > 1: dist.barrier()
    kms = {k: 0.0 for k in fe.kernel_names()}
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = step(i)
        if not args.pipeline:
            ms = fe.kernel_ms()                      # HIP events on the scan's stream (synchronises this scan)
            for k in kms: kms[k] += ms[k]
        elif i > 0:
            ms = fes[(i - 1) & 1].kernel_ms()    # the previous step's events, on its own stream (waits for that step only)
            for k in kms: kms[k] += ms[k]
    if args.pipeline and args.steps > 0:
        ms = fes[(args.steps - 1) & 1].kernel_ms()
        for k in kms: kms[k] += ms[k]
    torch.cuda.synchronize(dev)
    if world > 1: dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    res.fetch()
    nevents = int(res.counts.sum())
    bad = int((res.bursts["flags"] & ~np.uint32(frontend.F_EXACT_START | frontend.F_STATE_AT_END)).any())
    if world > 1:
        tot = torch.tensor([nevents, bad], device=dev, dtype=torch.int64)
        dist.all_reduce(tot)
        nevents_all, bad = int(tot[0].item()), int(tot[1].item())
    else:
        nevents_all = nevents
    if rank == 0:
        for k in kms: kms[k] /= max(args.steps, 1)
        dom = max(kms, key=kms.get)
        alg_bytes = 2 * cfg.ntrks * nrows + 16 * nevents        # 18 B per sample instant + 16 B per event
        achieved = alg_bytes / (kms[dom] * 1e-3) / 1e9 if kms[dom] > 0 else 0.0
        traffic = None
        try:                                         # HBM bytes per launch from the committed rocprofv3 --pmc passes
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            if dom in pm and abs(pm["workload_rows"] - nrows) < 0.01 * nrows:
                traffic = pm[dom]["fetch_bytes"] + pm[dom]["write_bytes"]
        except Exception:
            pass
        line = {
            "metric": "Msamples/sec (all tracks) 9-trk TBIN", "value": round(nrows * world * args.steps / dt / 1e6, 1),
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i16/f32", "data": "synthetic",
            "config": {"workload": "C2: synthetic 9-track 800 BPI NRZI, 781.25 kHz, 1 parmset", "rows_per_gpu": nrows,
                       "bytes_per_gpu": nrows * cfg.ntrks * 2, "events_per_gpu": nevents, "events_total": nevents_all, "bursts": int(res.nbursts),
                       "flagged_bursts": bad, "sharding": "time shards, neighbour halo only" if world > 1 else "none"},
            "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes": alg_bytes},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(tape, copies=max(1, min(copies, 4)))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
