"""GPU box: a batch of mostly short records with a few long ones in random order (95 % uniform 300..900 b, 5 % 3..8 kb):
nearly every 16-record group of the streaming kernel holds a long record and cannot be staged, so the short records
depend on the rescue pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from circkit_amd import api

ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N = 5_000_000
g = torch.Generator(device="cpu").manual_seed(7)
u = torch.rand(N, generator=g)
lens = torch.where(u < 0.95, torch.randint(300, 901, (N,), generator=g), torch.randint(3000, 8001, (N,), generator=g)).to(torch.int64)
offs = torch.zeros(N + 1, dtype=torch.int64)
offs[1:] = torch.cumsum(lens, 0)
total = int(offs[-1])
o = offs.cuda()
d = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
out = torch.empty_like(d)
ctx.synth_fill_device(46, 0, total, d)
for _ in range(3):
    ctx.canonicalize_batch_device(d, o, N, out_bytes=out)
torch.cuda.synchronize()
ms = []
for _ in range(5):
    ctx.canonicalize_batch_device(d, o, N, out_bytes=out); ctx.synchronize(); ms.append(ctx.last_kernel_ms())
print("short+long: %d records, %.2f Gbases: %.3f ms (min %.3f) = %.2f Tbases/s" % (N, total / 1e9, sum(ms) / len(ms), min(ms), total / (sum(ms) / len(ms)) / 1e9), flush=True)
