"""GPU box, under rocprofv3 --pmc SQ_INSTS_VMEM_WR WRITE_SIZE: one crafted mode-3 batch (7/8 records of `short` bases, 1/8 of `long`)
with the given N fraction through the device API, 3 launches.  Expected vector stores per record: 7/8 + ceil(long/1024)/8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import circkit_amd
from circkit_amd import workloads as W
short, long_, nfrac = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
dev = torch.device("cuda", 0)
ctx = circkit_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N = 800_000
lens = torch.full((N,), short, dtype=torch.int64)
lens[::8] = long_
offs = torch.zeros(N + 1, dtype=torch.int64)
offs[1:] = torch.cumsum(lens, 0)
total = int(offs[-1])
d_off = offs.to(dev)
d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
ctx.synth_fill_device(45, 0, total, d_bytes)
if nfrac > 0:
    W.sprinkle_n(d_bytes, total, nfrac, 46, dev)
d_out = torch.empty_like(d_bytes)
for _ in range(3):
    ctx.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=d_out)
torch.cuda.synchronize()
print("records", N, "bytes", total, "mode", ctx.last_batch_mode(), "status", ctx.batch_status())
