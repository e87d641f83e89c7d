"""GPU box: the streaming kernel on 10M x 1 kb, bytes only vs bytes + fused XXH3 (no table), ms per batch -- for ablation builds
of the hash path (CIRCKIT_LIB=...; their hashes are wrong on purpose, only the time is of interest)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import circkit_amd
from circkit_amd import workloads as W
dev = torch.device("cuda", 0)
ctx = circkit_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N, L = 10_000_000, 1000
d_bytes, d_off = W.fixed_length(ctx, dev, N, L, 42, 0)
d_out = torch.empty(N * L + 64, dtype=torch.uint8, device=dev)
d_hash = torch.empty(N, dtype=torch.int64, device=dev)
res = []
for name, kw in (("bytes", {}), ("bytes+xxh3", {"out_xxh3": d_hash})):
    for _ in range(5):
        ctx.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=d_out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctx.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=d_out, **kw)
    e1.record(); e1.synchronize()
    res.append("%s %.3f" % (name, e0.elapsed_time(e1) / 10))
print(os.path.basename(os.environ.get("CIRCKIT_LIB", "in-tree")), " ".join(res), flush=True)
