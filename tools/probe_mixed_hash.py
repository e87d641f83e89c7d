"""GPU box: config 4's batch (1M records, 200 b-20 kb) with the XXH3 requested as well (what `circkit uniq --canonicalize` asks
for on contigs of mixed lengths): time per batch, bytes only vs bytes + hash."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import circkit_amd
from circkit_amd import workloads as W
dev = torch.device("cuda", 0)
ctx = circkit_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N = 1_000_000
offs = W.log_uniform_offsets(N, 45, 200, 20000)
total = int(offs[-1])
d_off = offs.to(dev)
d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
ctx.synth_fill_device(45, 0, total, d_bytes)
d_out = torch.empty_like(d_bytes)
d_hash = torch.empty(N, dtype=torch.int64, device=dev)
for name, kw in (("bytes", {}), ("bytes + xxh3", {"out_xxh3": d_hash}), ("xxh3 only", {"out_xxh3": d_hash, "nobytes": True})):
    nb = kw.pop("nobytes", False)
    for _ in range(5):
        ctx.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=None if nb else d_out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctx.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=None if nb else d_out, **kw)
    e1.record(); e1.synchronize()
    print("%-14s %.3f ms per batch" % (name, e0.elapsed_time(e1) / 10))
