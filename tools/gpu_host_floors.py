"""GPU box helper: the floors of the streaming reader's host side - page cache -> pinned memory with n threads (os.preadv), pinned -> device, device -> pinned."""
import os, sys, time, tempfile, threading
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
n = 151 << 20
buf = torch.empty(n, dtype=torch.uint8, pin_memory=True)
dev = torch.empty(n, dtype=torch.uint8, device="cuda:0")
with tempfile.TemporaryDirectory() as wd:
    path = os.path.join(wd, "f.bin")
    with open(path, "wb") as f: f.write(os.urandom(1 << 20) * (n >> 20) * 4)
    print("tmp fs:", os.popen("df -T %s | tail -1" % wd).read().strip())
    fd = os.open(path, os.O_RDONLY)
    arr = buf.numpy()
    def rd(a, b, off):
        view = memoryview(arr)[a:b]; done = 0
        while done < len(view):
            done += os.preadv(fd, [view[done:]], off + a + done)
    for th in (1, 2, 4, 8, 16, 32):
        ex = ThreadPoolExecutor(th)
        best = 1e9
        for rep in range(6):
            off = (rep % 4) * n
            step = -(-n // th)
            t0 = time.perf_counter()
            fs = [ex.submit(rd, a, min(n, a + step), off) for a in range(0, n, step)]
            for f in fs: f.result()
            best = min(best, time.perf_counter() - t0)
        print("read %2d threads: %.2f ms  %.1f GB/s" % (th, best * 1e3, n / best / 1e9))
        ex.shutdown()
    for name, fn in (("H2D", lambda: dev.copy_(buf, non_blocking=True)), ("D2H", lambda: buf.copy_(dev, non_blocking=True))):
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print("%s 151 MB: %.2f ms  %.1f GB/s" % (name, best * 1e3, n / best / 1e9))
    # both at once
    ex = ThreadPoolExecutor(8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev.copy_(buf, non_blocking=True)
    buf2 = torch.empty(n, dtype=torch.uint8, pin_memory=True); arr = buf2.numpy()
    t1 = time.perf_counter()
    step = -(-n // 8)
    fs = [ex.submit(rd, a, min(n, a + step), 0) for a in range(0, n, step)]
    for f in fs: f.result()
    t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print("read 8 threads beside an H2D: %.2f ms; H2D done after %.2f ms (pinned alloc %.2f ms)" % ((t2 - t1) * 1e3, (t3 - t0) * 1e3, (t1 - t0) * 1e3))
print("cpus", os.cpu_count(), os.popen("lscpu | grep -i 'model name\\|numa node(s)\\|socket'").read())
