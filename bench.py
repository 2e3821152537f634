#!/usr/bin/env python
"""bench.py — throughput of the MI355X analog front end on BASELINE.json's workload.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by the driver through torch.distributed.run, one rank per GPU, RCCL)

Workload at N = 1 (BASELINE.json configs[1], "C2"): synthetic 9-track 800 BPI NRZI, 781.25 kHz,
1e8 sample instants (1.8 GB of interleaved int16, resident in HBM before the timed region),
1 parameter set.  One "step" = one full pass of the hot path over that tape (peak path: k_sift -> k_prep -> k_bursts -> k_gain ->
k_gain_s -> k_emit -> k_decode for whatever the chains gave up), events written to HBM.  Metric = sample instants per second
(the reference's own unit: "N samples were processed", src/readtape.c:2024), whole job.

N > 1: the sample timeline is time-sharded — every rank holds its own 1e8-row shard (weak scaling);
the only exchange is one neighbour halo per step (the rows a block straddling the seam needs), no
data-path collective.

--config C3 | C4 | C5 selects BASELINE.json's other configurations (PE -zeros, 1e9 rows; GCR, 1e9 rows, 8 parameter sets; one 10 GB
NRZI tape time-sharded over the ranks, strong scaling).  The driver's default line is C2.

Also on the line:  e2e (a bounded sample of the same tape from a .tbin file through the pinned double-buffered reader to the .tap),
roofline (dominant kernel, algorithmic bytes / measured kernel time, HIP events)
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


# BASELINE.json configs[1..4].  window_rows: the resident tape is scanned as consecutive fragments of this many rows through ONE
# workspace (the ownership rule of rtfe_scan makes fragments exact) - at 1e9 rows a whole-tape workspace plus the
# 8-parmset event arena would not fit beside the tape (C4).  C3's single parameter set does: one scan of 1e9 rows (18 GB of rows,
# as much event arena) - four scans of 2^28 rows each ended on their longest bursts, 16.2 instead of 15.0 ms.
CONFIGS = {
    "C2": dict(kind="nrzi", rows=1e8, nparmsets=1, find_zeros=False, window_rows=None, ref_opts=["-nm"], port_opts=[], overlap=True,
               workload="C2: synthetic 9-track 800 BPI NRZI, 781.25 kHz, 1 parmset"),
    "C3": dict(kind="pe", rows=1e9, nparmsets=1, find_zeros=True, window_rows=None, ref_opts=["-nm", "-zeros"], port_opts=["-zeros"], overlap=True,
               workload="C3: synthetic 9-track 1600 BPI PE, 1.5625 MHz, -zeros (zero-crossing path), 1 parmset"),
    # (C4 in ONE launch since round 6.  rtfe_event_capacity is the worst case - every gap-length quiet run a burst, 128 events of slack per list and burst: 288 GB of slack
    #  alone at 1e9 rows x 8 sets, which forced three fragments (262 ms; a fragment's chains have a latency floor).  The arena is now sized for one burst per 1e4 rows
    #  (bursts_per_row: the tape holds one per ~4e5; an arena that is too small is flagged RTFE_F_EVENT_OVERFLOW per burst, never silent - flagged_bursts on the line): 159 GB)
    "C4": dict(kind="gcr", rows=1e9, nparmsets=8, find_zeros=False, window_rows=None, bursts_per_row=1e-4, ref_opts=[], port_opts=["-m"],
               workload="C4: synthetic 9-track 6250 BPI GCR (9042 fci), 6.25 MHz, 8-parmset batched sweep"),
    # not BASELINE.json configurations: the single-set shapes of C4 / C3's formats on the peak detector (VERDICT r3 item 1 asks for them), compact lines only
    "G1": dict(kind="gcr", rows=1e9, nparmsets=1, find_zeros=False, window_rows=None, ref_opts=["-nm"], port_opts=[],
               workload="G1 (extra): synthetic 9-track 6250 BPI GCR (9042 fci), 6.25 MHz, 1 parmset, peak detection, one scan"),
    "P1": dict(kind="pe", rows=1e9, nparmsets=1, find_zeros=False, window_rows=None, ref_opts=["-nm"], port_opts=[],
               workload="P1 (extra): synthetic 9-track 1600 BPI PE, 1.5625 MHz, 1 parmset, peak detection (not -zeros), one scan"),
    # The handle's candidate screen follows the tape: a handle's FIRST scan estimates the floor from the samples before it screens them (k_scan_begin), and behind
    # each scan the floor moves to half the smallest peak height the scan's chains learned (k_adapt_floor).  The ...f lines time a tape's FIRST scan: every step
    # resets the handle's screen to what rtfe_create made (rtfe_reset_floor, inside the timed region) and scans - what INTEGRATION.md's one rtfe_scan per tape costs.
    "C2f": dict(kind="nrzi", rows=1e8, nparmsets=1, find_zeros=False, window_rows=None, ref_opts=["-nm"], port_opts=[], overlap=True, fresh=True,
                workload="C2f (extra): C2, every step a tape's FIRST scan (rtfe_reset_floor + rtfe_scan: the screen floor estimated from the samples inside the scan)"),
    "M8": dict(kind="nrzi", rows=1e8, nparmsets=8, find_zeros=False, window_rows=None, overlap=True, ref_opts=[], port_opts=["-m"],
               workload="M8 (extra): C2's tape under the reference's default -m: the 8 built-in NRZI parameter sets (three window widths) in one scan; default configuration (the screen floor follows the tape)"),
    "M8f": dict(kind="nrzi", rows=1e8, nparmsets=8, find_zeros=False, window_rows=None, overlap=True, ref_opts=[], port_opts=["-m"], fresh=True,
                workload="M8f (extra): M8, every step a tape's FIRST scan (rtfe_reset_floor + rtfe_scan)"),
    "M8p": dict(kind="nrzi", rows=1e8, nparmsets=8, find_zeros=False, window_rows=None, overlap=True, ref_opts=[], port_opts=["-m"], fixed_floor=1.0,
                workload="M8p (extra): M8 with the candidate screen pinned at 1 V (round 5's first scan: the four 0.05 V-rise sets' screens pass every wiggle)"),
    # the same tapes with noise (every other line is 10 mV rms): what the speculation costs when the signal is not clean -
    # flagged bursts, bursts redone on the samples and what the chains left to the literal detector are on the line
    "N1": dict(kind="nrzi", rows=1e8, nparmsets=1, find_zeros=False, window_rows=None, overlap=True, ref_opts=["-nm"], port_opts=[], noise_mv=60.0,
               workload="N1 (extra): C2's tape with 60 mV rms of noise on 2-3 V peaks (C2: 10 mV), 1 parmset; default configuration (the screen floor follows the tape)"),
    "N1f": dict(kind="nrzi", rows=1e8, nparmsets=1, find_zeros=False, window_rows=None, ref_opts=["-nm"], port_opts=[], noise_mv=60.0, fresh=True,
                workload="N1f (extra): N1, every step a tape's FIRST scan (rtfe_reset_floor + rtfe_scan)"),
    "N1p": dict(kind="nrzi", rows=1e8, nparmsets=1, find_zeros=False, window_rows=None, ref_opts=["-nm"], port_opts=[], noise_mv=60.0, fixed_floor=1.0,
                workload="N1p (extra): N1 with the candidate screen pinned at 1 V (round 5's first scan: the lists outgrow their slots, the bursts are redone on the samples)"),
    "N2": dict(kind="gcr", rows=1e9, nparmsets=1, find_zeros=False, window_rows=None, ref_opts=["-nm"], port_opts=[], noise_mv=30.0,
               workload="N2 (extra): G1's tape with 30 mV rms of noise on 1.8 V peaks (G1: 10 mV), 1 parmset, one scan"),
    "C5": dict(kind="nrzi", rows=10e9 / 18, nparmsets=1, find_zeros=False, window_rows=None, ref_opts=["-nm"], port_opts=[], strong=True, no_events=True,
               workload="C5: ONE 10 GB synthetic 9-track 800 BPI NRZI tape, time-sharded over the ranks (strong scaling)"),
}


def make_base_tape(seed, target_rows, kind="nrzi", noise_mv=None):
    """Unique synthetic tape of about target_rows rows: 512..4096-byte blocks, >= 6 ms gaps (NRZI: a tapemark every 16 blocks)
    (SURVEY.md 8d)."""
    from readtape_amd import synth
    kw = {} if noise_mv is None else {"noise_mv": float(noise_mv)}
    make = {"nrzi": lambda n: synth.nrzi_tape(seed=seed, nblocks=n, minlen=512, maxlen=4096, marks_every=16, gap_samples=6000, **kw),
            "pe": lambda n: synth.pe_tape(seed=seed, nblocks=n, minlen=512, maxlen=4096, gap_samples=8000, **kw),
            "gcr": lambda n: synth.gcr_tape(seed=seed, nblocks=n, minlen=512, maxlen=4096, gap_samples=30000, **kw)}[kind]
    probe = make(4)
    nblocks = max(4, int(target_rows / (probe.rows.shape[0] / 4)))
    return make(nblocks)


def cpu_baseline(tape, copies, conf):
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
        p = subprocess.run([port, "-time", f"-out={wd}/o", *conf["port_opts"], path], capture_output=True, text=True)
        j = json.loads(p.stdout.strip().splitlines()[-1])
        out["port_msamples_per_s"] = j["msamples_per_s"]
        port_tap = open(f"{wd}/o.tap", "rb").read()
        value, kind = j["msamples_per_s"], "port"
        if os.path.exists(ref) and os.access(ref, os.X_OK):
            t0 = time.perf_counter()
            p = subprocess.run([ref, *conf["ref_opts"], "-tap", "-nolabels", "-q", "b"], cwd=wd, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if p.returncode == 0 and os.path.exists(f"{wd}/b.tap"):
                value, kind = nrows / dt / 1e6, "reference"
                out["tap_identical_to_reference"] = open(f"{wd}/b.tap", "rb").read() == port_tap
    out.update(value=round(value, 3), unit="Msamples/s", cores=1, kind=kind,
               sample=f"{nrows} rows ({copies} copies of the base tape) of the same synthetic tape, whole pipeline incl. bit decoding, single thread"
                      + (f", options {' '.join(conf['ref_opts'])}" if conf["ref_opts"] else ""))
    return out


def e2e_line(tape, copies, conf, dev):
    """End to end on a bounded sample of the same tape: .tbin file -> pipelined reader (pinned buffers) -> device windows -> events ->
    host replay -> SIMH .tap (readtape_amd/ingest.py).  The .tap must be the CPU port's."""
    from readtape_amd import ingest, pipeline, tbin
    hdr = tape.spec.header()
    rows = np.tile(tape.rows, (copies, 1))
    with tempfile.TemporaryDirectory() as wd:
        path = os.path.join(wd, "e.tbin")
        tbin.write_tbin(path, hdr, rows)
        del rows
        opts = pipeline.DecodeOptions(multiple_tries=conf["nparmsets"] > 1, verbose=False)      # (-m: the reference's built-in sets)
        threads = max(1, min(int(os.environ.get("RT_E2E_REPLAY_THREADS", "64")), (os.cpu_count() or 1) - 1))
        rthreads = max(1, min(int(os.environ.get("RT_E2E_READ_THREADS", "16")), (os.cpu_count() or 2) // 2))
        wrows = int(os.environ.get("RT_E2E_WINDOW_ROWS", str(1 << 23 if copies > 4 else 1 << 22)))      # (measured on the 9e7-row C2 sample: 2^22 / 2^23 / 2^24 rows -> 0.79 / 0.93 / 0.88 Gsamples/s)
        split = int(os.environ.get("RT_E2E_REPLAY_SPLIT", "8"))
        if True:                                        # one untimed pass over the same file first, as the device-resident line has its warm-up steps over the same tape
            # (first use of the replay pool, of the packing kernels, of the scan contexts - and of the host allocator's page-locked blocks at THIS file's sizes: after a pass
            #  over a short file the first windows of the long one still paid for them, 0.063 s against 0.072 - 0.081)
            ingest.decode_file_streaming(path, os.path.join(wd, "w.tap"), window_rows=wrows, halo_rows=1 << 18, opts=opts,
                                         cfgkw=dict(find_zeros=True) if conf["find_zeros"] else None, device=str(dev), replay_threads=threads, read_threads=rthreads, replay_split=split)
        os.sync()                                       # (the sample was written a moment ago: its dirty pages' write-back would run beside the timed decode)
        st = ingest.decode_file_streaming(path, os.path.join(wd, "e.tap"), window_rows=wrows, halo_rows=1 << 18, opts=opts,
                                          cfgkw=dict(find_zeros=True) if conf["find_zeros"] else None, device=str(dev), replay_threads=threads, read_threads=rthreads, replay_split=split)
        same = None
        port = os.path.join(ROOT, "oracle", "_build", "oracle_readtape")
        if os.path.exists(port):
            subprocess.run([port, f"-out={wd}/o", *conf["port_opts"], path], capture_output=True, text=True)
            same = open(f"{wd}/o.tap", "rb").read() == open(f"{wd}/e.tap", "rb").read()
    return {"value": round(st["msamples_per_s"], 2), "unit": "Msamples/s", "rows": st["rows"], "windows": st["windows"], "seconds": round(st["seconds"], 3), "setup_seconds_not_included": round(st["setup_seconds"], 3),
            "warmup": "one untimed pass over the same file",
            "host_replay_seconds_summed": round(st["replay_seconds"], 3), "host_replay_events_per_s_per_thread": round(st["replay_events_per_s"] or 0),
            "host_replay_threads": st["replay_threads"], "host_read_threads": rthreads, "window_rows": wrows, "replay_split": split, "host_cores": os.cpu_count(),
            "file_read_seconds_overlapped": round(st["read_seconds"], 3), "scan_wait_seconds": round(st["scan_wait_seconds"], 3),
            "blocks": st["blocks"], "tapemarks": st["tapemarks"], "exact_rescans": st["exact_scans"], "tap_identical_to_cpu_port": same,
            "path": ".tbin in the page cache -> a producer thread: native parallel positional reads into a ring of pinned buffers, hipMemcpyAsync on a copy stream, rtfe_find_end_mark + rtfe_scan + rtfe_pack_events (three contexts in flight), the tables mirrored to pinned memory on the scan's stream -> fetcher threads: the packed event lists to the host -> host replay of the windows (fragments), each as replay_split sub-fragments on native threads -> .tap"}


class Workload:
    """What one rank times: its rows resident in device memory (tail = room for the seam halo), the front end, and step() - one pass of
    the hot path over those rows, the halo exchange included.  main() builds it on cuda:LOCAL_RANK; tests/test_shard_gloo.py builds the
    same object on CPU tensors with the emulated kernels and two gloo ranks, and drives the same step()."""

    HALO = 1 << 18

    def __init__(self, conf, rank, world, dev, dist, total_rows, base_rows, window_rows=None, pipeline=False, fe_factory=None, halo=None, tape=None):
        import torch
        from readtape_amd import frontend, shard
        self.torch, self.dist, self.shard = torch, dist, shard
        self.conf, self.rank, self.world, self.dev, self.pipeline = conf, rank, world, dev, pipeline
        self.cuda = dev.type == "cuda"
        self.halo_rows = int(halo or self.HALO)
        self.strong = bool(conf.get("strong"))
        # weak (C2..C4): every rank holds its own tape of `rows` rows (its own seed) - N tapes of a collection decoded side by side is
        # what the shards of a longer tape look like; strong (C5): ONE tape (same seed everywhere), rank r holds plan_shards()[r].
        self.tape = tape or make_base_tape(seed=1000 + (0 if self.strong else rank), target_rows=int(base_rows), kind=conf["kind"], noise_mv=conf.get("noise_mv"))
        hdr = self.tape.spec.header()
        base = torch.from_numpy(self.tape.rows).to(dev)
        self.copies = max(1, int(round(total_rows / base.shape[0])))
        if self.strong:
            n_tape = self.copies * int(base.shape[0])
            spans = shard.plan_shards(n_tape, world)
            lo, hi = spans[rank]
            idx0 = lo % base.shape[0]
            reps = (hi - lo + idx0) // base.shape[0] + 2
            own = base.repeat(reps, 1)[idx0: idx0 + (hi - lo)]
            self.row_base = lo
            self.lens = [b - a for a, b in spans]
        else:
            own = base.repeat(self.copies, 1)
            self.row_base = rank * int(own.shape[0])
            self.lens = [int(own.shape[0])] * world
        self.nrows = int(own.shape[0])
        # time shards: this rank owns `nrows` rows; the tail of the ONE buffer receives the first rows of the rank(s) behind it
        self.sr = shard.ShardRows(own, self.halo_rows if (world > 1 and rank < world - 1) else 0)
        del own, base
        parmsets = None
        if conf["nparmsets"] > len(frontend.DEFAULT_PARMSETS[hdr.mode]):
            # the reference ships 5 GCR sets (src/parmsets.c:104-110); a .parms file may hold more - the sweep is filled up to 8 with
            # variations of the window width, the rise threshold and the minimum peak, as such a file would
            extra = [(1.4, 0.20, 0.2, 0.5, 0, 0.0), (1.6, 0.14, 0.0, 0.5, 0, 0.0), (1.5, 0.10, 0.1, 0.5, 0, 0.0), (1.3, 0.25, 0.2, 0.5, 0, 0.0)]
            parmsets = (list(frontend.DEFAULT_PARMSETS[hdr.mode]) + extra)[: conf["nparmsets"]]
        self.cfg = frontend.FrontEndConfig.from_header(hdr, nparmsets=conf["nparmsets"], find_zeros=conf["find_zeros"], parmsets=parmsets)
        if os.environ.get("RT_BENCH_EVENT_CAP"):                         # (experiments: the event regions' share of a burst's rows; the library's default is 1 / 8)
            self.cfg.events_per_sample_cap = float(os.environ["RT_BENCH_EVENT_CAP"])
        elif conf.get("event_cap"):
            self.cfg.events_per_sample_cap = float(conf["event_cap"])
        if conf.get("fixed_floor"):
            self.cfg.screen_floor_height = float(conf["fixed_floor"])
        if os.environ.get("RT_BENCH_SCREEN_FLOOR"):                     # (experiments: the candidate screen's assumed lower bound of the learned peak height, volts)
            self.cfg.screen_floor_height = float(os.environ["RT_BENCH_SCREEN_FLOOR"])
        def _make_gpu():
            f = frontend.FrontEnd(self.cfg, device=str(dev))
            bpr = os.environ.get("RT_BENCH_BURSTS_PER_ROW") or conf.get("bursts_per_row")
            if bpr:                           # (the event arena for this many bursts instead of the worst case; an overflow would be flagged and is on the line)
                f.bursts_hint = max(1024, int(float(total_rows) * float(bpr)))
            return f
        make = _make_gpu if fe_factory is None else (lambda: fe_factory(self.cfg))
        self.make = make
        self.fe = make()
        # fragments of the resident rows (one when the workspace fits)
        wrows = int(window_rows or conf["window_rows"] or 0)
        if wrows and wrows < self.nrows:
            wrows = wrows // 1024 * 1024
            self.frags = [(a, min(self.nrows, a + wrows)) for a in range(0, self.nrows, wrows)]
        else:
            self.frags = [(0, self.nrows)]
        # Default: one stream, steps back to back (the per-kernel HIP-event times are then contention-free, which is what the
        # roofline line needs).  --pipeline alternates two front-end contexts (own HIP stream, workspace and outputs) so that
        # the latency-bound sequential pass of step i overlaps the dense pass of step i+1.
        if pipeline and self.cuda:
            nctx = max(2, int(os.environ.get("RT_BENCH_CONTEXTS", "2")))
            self.fes = [self.fe] + [make() for _ in range(nctx - 1)]
            self.streams = [torch.cuda.Stream(dev) for _ in range(nctx)]
        else:
            self.fes = [self.fe, self.fe]
            # (a stream of its own, not the legacy default stream: that one cannot be captured - rtfe_set_graphs would quietly launch directly)
            self.streams = [torch.cuda.Stream(dev)] * 2 if self.cuda else [None, None]
        if self.cuda:
            for f in set(self.fes): f.set_timing(True)
        self.kms = {k: 0.0 for k in self.fe.kernel_names()} if self.cuda else {}
        self.scans_since_read = 0

    def collect(self):
        """The HIP-event times of the scans since the last call, added to kms (synchronises those scans - called once behind the timed
        loop, so the steps run back to back)."""
        self.scans_since_read = 0
        if getattr(self, "graphs_on", False): return      # (no events between a graph's kernels)
        for f in set(self.fes):
            ms, _ = f.kernel_ms()
            for k in self.kms: self.kms[k] += ms[k]

    def scan_frag(self, f, s, a, b):
        last = b >= self.nrows
        ntot = self.nrows + self.sr.got
        end = ntot if last else min(ntot, b + self.halo_rows)
        rows = self.sr.buf[a:end]
        if not self.cuda: rows = rows.numpy()
        kw = dict(stream=s.cuda_stream) if self.cuda else {}
        return f.scan(rows, row_base=self.row_base + a, first_is_tape_start=(self.rank == 0 and a == 0), own_rows=b - a, **kw)

    def step(self, i, timed=False, each=None):
        import contextlib
        s, f = self.streams[i % len(self.streams)], self.fes[i % len(self.fes)]
        with (self.torch.cuda.stream(s) if self.cuda else contextlib.nullcontext()):
            # the seam halo is the only exchange: neighbour isend/irecv over RCCL (xGMI), no collective on the data path
            self.shard.exchange_halo(self.sr, self.halo_rows, self.rank, self.world, self.dist, lens=self.lens)
            res = None
            if self.conf.get("fresh") and self.cuda:      # a tape's first scan: the handle's screen as rtfe_create left it
                f.reset_floor(s.cuda_stream)
            for a, b in self.frags:
                res = self.scan_frag(f, s, a, b)
                if each is not None:
                    each(res)
                self.scans_since_read += 1
                if timed and self.cuda and self.scans_since_read >= 48:      # (the front end keeps 64 scans' events: collect before they wrap - C3 / C4 at many steps)
                    self.collect()
            return res


def measure(name, args, rank, world, dev, dist, steps, warmup, min_seconds=0.5, total_rows=None):
    """Times `steps` steps of configuration `name` (repeated until at least min_seconds are inside the timed region - the steps run back to
    back, barrier + synchronize on both sides) and returns (line fields, workload, last result).  Collective when world > 1."""
    import torch
    from readtape_amd import frontend
    conf = CONFIGS[name]
    strong = bool(conf.get("strong"))
    # overlap: two scan contexts (front end, workspace, outputs, HIP stream) take the steps in turn, so that step i + 1's dense pass (VALU-bound)
    # runs beside step i's record chains (latency-bound) - as the streaming reader runs a tape's windows (ingest.py).  Every step is still a whole
    # scan of the resident tape and all of them are inside the timed region.  The per-kernel times (roofline) come from a serial pass of their own.
    overlap = (bool(conf.get("overlap")) and not args.no_overlap) or args.pipeline
    wl = Workload(conf, rank, world, dev, dist, float(total_rows or conf["rows"]), args.base_rows, window_rows=args.window_rows, pipeline=overlap)
    cfg, fe, frags, kms, nrows = wl.cfg, wl.fe, wl.frags, wl.kms, wl.nrows
    torch.cuda.synchronize(dev)
    calibrated = None
    if conf.get("calibrate_floor") and wl.cuda:      # the first scan of a tape learns the peak heights; the scans behind it screen against half the smallest one
        st0 = fe.scan_stats(wl.step(0))
        if st0.get("min_learned_height"):
            calibrated = min(4.0, 0.5 * st0["min_learned_height"])
            for f in set(wl.fes): f.close()
            wl.cfg.screen_floor_height = calibrated
            wl.fe = wl.make(); wl.fes = [wl.fe] + [wl.make() if overlap else wl.fe for _ in range(len(wl.streams) - 1)]
            for f in set(wl.fes): f.set_timing(True)
            cfg, fe = wl.cfg, wl.fe
    for i in range(warmup):
        wl.step(i)
    torch.cuda.synchronize(dev)
    wl.collect()                                     # (the warm-up scans' events: discarded)
    for k in kms: kms[k] = 0.0
    # ---- the serial pass: one context, steps back to back on one stream - contention-free HIP-event spans for kernel_ms / roofline ----
    n_serial = max(3, min(steps, 10))
    if world > 1: dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(n_serial):
        wl.step(0, timed=True)                       # (always context 0)
    torch.cuda.synchronize(dev)
    serial_s = (time.perf_counter() - t0) / n_serial
    wl.collect()
    for k in kms: kms[k] /= n_serial
    kms_serial = dict(kms)
    graphs = (bool(getattr(args, "graphs", False)) or bool(conf.get("graphs"))) and not getattr(args, "no_graphs", False) and wl.cuda and not overlap      # (graphs of two contexts do not run beside each other: measured)
    if graphs and world > 1:                         # a stream capture beside RCCL's own threads and the halo's send / receive has never run anywhere: refused, not risked
        if rank == 0: print("bench.py: --graphs is refused with --gpus > 1 (HIP graph capture beside RCCL is untested); direct launches", file=sys.stderr)
        graphs = False
    # C5 - what --gpus N runs, a rank's share is a short scan - records no per-kernel events in its timed region (24 records per scan: 1.32 -> 1.21 ms on a 6.9e7-row
    # scan; kernel_ms from the serial pass above).  Graph replay (--graphs) buys another 0.6 % there: not the default - its capture beside RCCL's threads is untested
    no_events = (bool(getattr(args, "no_kernel_events", False)) or bool(conf.get("no_events"))) and wl.cuda
    if graphs or no_events:                          # the timed region replays captured HIP graphs: no events between the kernels (kernel_ms: the serial pass above)
        for f in set(wl.fes): f.set_timing(False); f.set_graphs(bool(graphs))
        wl.graphs_on = True
        for i in range(2 * len(wl.fes)): wl.step(i)  # (capture + instantiate: once per context, outside the timed region - as a tape's first window pays it)
        torch.cuda.synchronize(dev)
    done, dt = 0, 0.0
    while True:                                      # rounds of `steps` steps until the timed region is long enough for the driver's clock to see it
        if world > 1: dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            res = wl.step(done + i, timed=True)
        torch.cuda.synchronize(dev)
        if world > 1: dist.barrier()
        d = time.perf_counter() - t0
        wl.collect()                                 # the steps' HIP events, recorded on the stream as the scans ran
        if world > 1:
            t = torch.tensor([d], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        dt += d; done += steps
        if dt >= min_seconds or done >= 64 * steps: break
    if graphs or no_events:
        for f in set(wl.fes): f.set_graphs(False); f.set_timing(True)
        wl.graphs_on = False
        for k in kms: kms[k] = kms_serial[k]
    kms_timed = {k: (v - kms_serial[k]) / max(done, 1) for k, v in kms.items()}      # (under overlap: spans that share the device)
    for k in kms: kms[k] = kms_serial[k]
    # what the step produced (one more, untimed pass: every fragment's tables are fetched before the next one reuses the buffers)
    tally = dict(events=0, bursts=0, bad=0)

    def count(r):
        r.fetch(events=False)
        tally["events"] += int(r.counts.sum())
        tally["bursts"] += int(r.nbursts)
        tally["bad"] += int(((r.bursts["flags"] & ~np.uint32(frontend.F_EXACT_START | frontend.F_STATE_AT_END)) != 0).sum())
    res = wl.step(0, each=count)
    nevents, bad = tally["events"], tally["bad"]
    try:                                             # diagnostics of the last scan (what the chains left to the literal detector / the sample path)
        sst = fe.scan_stats(res)
    except Exception:
        sst = None
    if world > 1:
        tot = torch.tensor([nevents, bad, nrows], device=dev, dtype=torch.int64)
        dist.all_reduce(tot)
        nevents_all, bad, rows_all = int(tot[0].item()), int(tot[1].item()), int(tot[2].item())
    else:
        nevents_all, rows_all = nevents, nrows
    # The dominant span.  Every span is bracketed by its own pair of HIP events on the stream it runs on; on the peak path k_bursts' span
    # (quiet map -> bursts -> restart rows) runs on the handle's side stream BESIDE k_prep: its time is shared device time, not work of its
    # own, so it never names the dominant kernel.
    cand = {k: v for k, v in kms.items() if not (k == "k_bursts" and cfg.mode == frontend.NRZI and not conf["find_zeros"] and os.environ.get("RTFE_OVERLAP", "1") != "0")}
    dom = max(cand, key=cand.get)
    alg_bytes = 2 * cfg.ntrks * nrows + 16 * nevents        # 18 B per sample instant + 16 B per event
    achieved = alg_bytes / (kms[dom] * 1e-3) / 1e9 if kms[dom] > 0 else 0.0
    step_s = dt / done
    whole = alg_bytes / step_s / 1e9 if step_s > 0 else 0.0      # (this GPU's algorithmic bytes over the step's wall time)
    traffic = traffic_all = traffic_src = None
    try:                                         # HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/gpu_traffic.sh: the counters cannot be read from inside this process)
        own = os.path.join(ROOT, "profiles", f"pmc_{name}.json")
        pm = json.load(open(own if os.path.exists(own) else os.path.join(ROOT, "profiles", "pmc_latest.json")))
        if name == pm.get("config", "C2") and dom in pm and abs(pm["workload_rows"] - nrows) < 0.01 * nrows:      # (tools/gpu_traffic.sh wrote it)
            traffic = (pm[dom]["fetch_bytes"] + pm[dom]["write_bytes"]) * len(frags)      # (per scan in the file; a step = len(frags) scans)
            traffic_all = sum(v["fetch_bytes"] + v["write_bytes"] for v in pm.get("kernels", {}).values()) * len(frags) or None      # every kernel of the step
            traffic_src = f"profiles/{os.path.basename(own)} (round {pm.get('round')}): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command line, not this run"
    except Exception:
        pass
    own_rows_bytes = 2 * cfg.ntrks * nrows                    # what the dominant kernel streams if it writes no events itself (k_dseg: the chains' kernel writes them)
    fields = {
        "value": round(rows_all * done / dt / 1e6, 1), "ms_per_step": round(step_s * 1e3, 4), "timed_steps": done, "timed_seconds": round(dt, 3),
        "ms_per_step_serial": round(serial_s * 1e3, 4),
        "overlap": ("two scan contexts on two HIP streams take the steps in turn (step i + 1's dense pass beside step i's chains); kernel_ms / roofline from a serial pass of "
                    f"{n_serial} steps on one context") if overlap else "none: one context, steps back to back",
        "config": {"workload": conf["workload"], "rows_per_gpu": nrows, "rows_total": rows_all,
                   "bytes_per_gpu": nrows * cfg.ntrks * 2, "events_per_gpu": nevents, "events_total": nevents_all, "bursts": tally["bursts"],
                   "flagged_bursts": bad, "parmsets": conf["nparmsets"], "launches_per_step": len(frags), "last_scan_stats": sst, "screen_floor_height": calibrated,
                   "sharding": ("one tape, time shards (plan_shards), neighbour halo only" if strong else "time shards, neighbour halo only") if world > 1 else "none"},
        "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
        "kernel_ms_in_timed_region": {k: round(v, 4) for k, v in kms_timed.items()} if overlap and not graphs and not no_events else None,
        "graphs": ("rtfe_set_graphs: every timed step replays the HIP graph its context captured at its first scan of these buffers" if graphs else None),
        "kernel_events_in_timed_region": not (graphs or no_events),
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_all_kernels": traffic_all, "traffic_source": traffic_src, "algorithmic_bytes": alg_bytes,
                     "frac_rows_only": round(own_rows_bytes / (kms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kms[dom] > 0 else None,
                     "whole_step": {"achieved": round(whole, 1), "frac": round(whole / HBM_PEAK_GBS, 4),
                                    "what": "the same algorithmic bytes over the step's wall time (all kernels of a scan, launch gaps included)"}},
    }
    return fields, wl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None, help="BASELINE.json configs[1..4]; default: C2 on one GPU, C5 (one tape, time-sharded: strong scaling) on several")
    ap.add_argument("--rows", type=float, default=None, help="sample instants per GPU (C5: of the whole tape); default: the config's")
    ap.add_argument("--window-rows", type=float, default=None)
    ap.add_argument("--base-rows", type=float, default=5e6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the compact C3 / C4 / C5 lines the default single-GPU run adds to its line")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="the timed region repeats its --steps steps until it is at least this long")
    ap.add_argument("--no-overlap", action="store_true", help="one scan context, steps back to back, also for the configurations that overlap two by default")
    ap.add_argument("--no-graphs", action="store_true", help="direct launches (the default everywhere; kept for the experiment scripts)")
    ap.add_argument("--no-kernel-events", action="store_true", help="experiments: no per-kernel HIP events in the timed region (direct launches; kernel_ms from the serial pass)")
    ap.add_argument("--graphs", action="store_true", help="rtfe_set_graphs: the timed steps replay captured HIP graphs (no per-kernel events in the timed region; kernel_ms from the serial pass)")
    ap.add_argument("--pipeline", action="store_true", help="alternate two front-end contexts on two HIP streams (steps overlap; per-kernel times then include contention)")
    args = ap.parse_args()
    default_line = args.config is None and int(os.environ.get("WORLD_SIZE", "1")) == 1
    if args.config is None:
        args.config = "C2" if int(os.environ.get("WORLD_SIZE", "1")) == 1 else "C5"
    conf = CONFIGS[args.config]

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    dev = torch.device(f"cuda:{local}")
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    strong = bool(conf.get("strong"))

    fields, wl = measure(args.config, args, rank, world, dev, dist, args.steps, args.warmup, args.min_seconds, total_rows=args.rows)
    tape, fes, copies = wl.tape, wl.fes, wl.copies
    if rank == 0:
        line = {"metric": "Msamples/sec (all tracks) 9-trk TBIN", "value": fields["value"], "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": fields["ms_per_step"], "ms_per_step_serial": fields["ms_per_step_serial"], "overlap": fields["overlap"], "higher_is_better": True, "scaling": "strong" if strong else "weak",
                "vs_baseline": None, "dtype": "i16/f32", "data": "synthetic", "timed_steps": fields["timed_steps"], "timed_seconds": fields["timed_seconds"],
                "config": fields["config"], "kernel_ms": fields["kernel_ms"], "kernel_ms_in_timed_region": fields["kernel_ms_in_timed_region"], "graphs": fields["graphs"], "kernel_events_in_timed_region": fields["kernel_events_in_timed_region"], "roofline": fields["roofline"]}
    import gc

    def release(w):                                  # a workload's device memory: its scan contexts' buffers, its rows
        for f in set(w.fes): f.close()
        w.fes, w.fe, w.make, w.sr = [], None, None, None
    release(wl)
    del wl, fes
    gc.collect(); torch.cuda.empty_cache()
    # ---- the other BASELINE.json configurations, measured in this process (compact: value, ms per step, dominant kernel, fractions) ----
    if default_line and not args.no_other_configs:
        others = {}
        for name, st, wu in (("C3", 8, 2), ("C4", 1, 1), ("C5", 10, 2), ("G1", 2, 1), ("P1", 2, 1), ("C2f", 10, 2), ("M8", 5, 2), ("M8f", 5, 2), ("N1", 10, 2), ("N1f", 5, 2), ("N2", 2, 1)):
            try:
                f2, w2 = measure(name, args, rank, world, dev, dist, st, wu, args.min_seconds)
                others[name] = {"workload": f2["config"]["workload"], "value": f2["value"], "unit": "Msamples/s", "ms_per_step": f2["ms_per_step"], "ms_per_step_serial": f2["ms_per_step_serial"], "overlap": f2["overlap"][:40], "graphs": bool(f2["graphs"]), "kernel_events_in_timed_region": f2["kernel_events_in_timed_region"], "timed_steps": f2["timed_steps"],
                                "rows": f2["config"]["rows_total"], "events": f2["config"]["events_total"], "parmsets": f2["config"]["parmsets"], "flagged_bursts": f2["config"]["flagged_bursts"],
                                "launches_per_step": f2["config"]["launches_per_step"], "dominant_kernel": f2["roofline"]["kernel"], "dominant_kernel_ms": f2["kernel_ms"][f2["roofline"]["kernel"]],
                                "frac": f2["roofline"]["frac"], "frac_rows_only": f2["roofline"]["frac_rows_only"], "whole_step_frac": f2["roofline"]["whole_step"]["frac"],
                                "traffic": f2["roofline"]["traffic"], "traffic_all_kernels": f2["roofline"]["traffic_all_kernels"], "kernel_ms": {k: v for k, v in f2["kernel_ms"].items() if v > 0.02},
                                "last_scan_stats": {k: f2["config"]["last_scan_stats"].get(k) for k in ("bursts", "redone", "parallel", "sequential", "gave_up", "min_learned_height", "screen_floor_now", "screen_floor_used")} if f2["config"]["last_scan_stats"] else None,
                                "screen_floor_height": f2["config"]["screen_floor_height"]}
                release(w2)
                del w2
            except Exception as e:                    # the headline number must not depend on the other lines
                others[name] = {"error": repr(e)[:300]}
                e.__traceback__ = None               # (the frames of a failed measure() hold its workload: let the collector below have them)
            gc.collect(); torch.cuda.empty_cache()
        if rank == 0: line["other_configs"] = others
    if rank == 0:
        ncopies = max(1, min(copies, 4))
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(tape, ncopies, conf)
        if not args.no_e2e and world == 1:
            try:
                # (C2, the driver's line: a sample long enough for the reader's pipeline to fill - 16 copies, ~9e7 rows, 1.6 GB; the
                #  CPU port that checks its .tap needs ~13 s for it.  The slower formats keep the CPU baseline's sample.)
                line["e2e"] = e2e_line(tape, max(1, min(copies, 16)) if args.config == "C2" else ncopies, conf, dev)
            except Exception as e:                    # the headline number must not depend on the bounded end-to-end sample
                line["e2e"] = {"error": repr(e)[:300]}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
