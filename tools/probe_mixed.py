"""GPU box: where does the general (LDS-tier) kernel's time go on BASELINE config 4 (1M records, 200 b - 20 kb)?
Times canonicalize (both strands, bytes), lmsr (forward strand only: no reverse scan, no strand compare) and the
index-only variants (no emit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from circkit_amd import api

ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N = 1_000_000
g = torch.Generator(device="cpu").manual_seed(45)
u = torch.rand(N, generator=g, dtype=torch.float64)
lens = torch.exp(np.log(200.0) + u * (np.log(20000.0) - np.log(200.0))).to(torch.int64)
offs = torch.zeros(N + 1, dtype=torch.int64)
offs[1:] = torch.cumsum(lens, 0)
total = int(offs[-1])
o = offs.cuda()
d = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
out = torch.empty_like(d)
idx = torch.empty(N, dtype=torch.int32, device="cuda")
ctx.synth_fill_device(45, 0, total, d)
def t(name, fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        fn(); ctx.synchronize(); ms.append(ctx.last_kernel_ms())
    print("%-28s %.3f ms (min %.3f)" % (name, sum(ms) / len(ms), min(ms)), flush=True)
t("canonicalize bytes", lambda: ctx.canonicalize_batch_device(d, o, N, out_bytes=out)) if not os.environ.get("ONLY_LMSR") else None
t("lmsr bytes (fwd only)", lambda: ctx.lmsr_batch_device(d, o, N, out_bytes=out))
t("canonicalize index only", lambda: ctx.canonicalize_batch_device(d, o, N, out_index=idx))
t("lmsr index only", lambda: ctx.lmsr_batch_device(d, o, N, out_index=idx))
