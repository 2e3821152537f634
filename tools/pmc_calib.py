"""GPU box, under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`: kernels of known HBM traffic in the access pattern the front end's
streaming kernels use (16 bytes per lane, coalesced), 1 GiB each (past the 256 MiB Infinity Cache).  tools/pmc_json.py reads the
counters of these dispatches and derives the factors that turn the counters into bytes (MI355X_MICROARCH.md, HBM: calibrate on a
known byte count in your own access pattern)."""
import torch

n = 1 << 30
a = torch.empty(n // 4, dtype=torch.int32, device="cuda")
b = torch.empty_like(a)
for _ in range(3):
    a.fill_(7)                    # writes n bytes, reads none      (kernel name holds "FillFunctor")
torch.cuda.synchronize()
for _ in range(3):
    torch.bitwise_xor(a, 5, out=b)   # reads n bytes, writes n bytes   (kernel name holds "bitwise_xor" / "BitwiseXor")
torch.cuda.synchronize()
print("calibration bytes", n)
