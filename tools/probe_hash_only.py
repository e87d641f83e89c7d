import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, circkit_amd
from circkit_amd import workloads as W
dev = torch.device("cuda", 0)
ctx = circkit_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N, L = 10_000_000, 1000
d_bytes, d_off = W.fixed_length(ctx, dev, N, L, 42, 0)
d_out = torch.empty(N * L + 64, dtype=torch.uint8, device=dev)
d_hash = torch.empty(N, dtype=torch.int64, device=dev)
for name, kw in (("bytes", {"out_bytes": d_out}), ("bytes+xxh3", {"out_bytes": d_out, "out_xxh3": d_hash}), ("xxh3 only", {"out_xxh3": d_hash})):
    for _ in range(5):
        ctx.canonicalize_batch_device(d_bytes, d_off, N, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctx.canonicalize_batch_device(d_bytes, d_off, N, **kw)
    e1.record(); e1.synchronize()
    print("%-12s %.3f ms" % (name, e0.elapsed_time(e1) / 10), flush=True)
