"""GPU box only: is the streaming kernel bound by memory or by instruction issue?  Times, on the headline input,
canonicalize (both strands), lmsr (forward strand only: about half the VALU work, same bytes) and index-only
(no output bytes: half the traffic, same VALU work)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from circkit_amd import api

n, L = 10_000_000, 1000
ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d = torch.empty(n * L + 64, dtype=torch.uint8, device="cuda")
o = torch.empty(n + 1, dtype=torch.int64, device="cuda")
out = torch.empty_like(d)
idx = torch.empty(n, dtype=torch.int32, device="cuda")
ctx.synth_fill_device(1234, 0, n * L, d)
ctx.fixed_offsets_device(0, L, n, o)
def t(name, fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        fn(); ctx.synchronize(); ms.append(ctx.last_kernel_ms())
    print("%-28s %.3f ms (min %.3f)" % (name, sum(ms) / len(ms), min(ms)), flush=True)
t("canonicalize bytes", lambda: ctx.canonicalize_batch_device(d, o, n, out_bytes=out))
t("lmsr bytes (fwd only)", lambda: ctx.lmsr_batch_device(d, o, n, out_bytes=out))
t("canonicalize index only", lambda: ctx.canonicalize_batch_device(d, o, n, out_index=idx))
t("lmsr index only", lambda: ctx.lmsr_batch_device(d, o, n, out_index=idx))
