import sys, os, numpy as np, torch, time
sys.path.insert(0, os.getcwd())
import circkit_amd
from oracle import oracle as O
ctx = circkit_amd.Context(0)
dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
L = 1000
for N in (100_000, 1_000_000, 10_000_000):
    x = torch.empty(N * L + 64, dtype=torch.uint8, device=dev)
    off = torch.empty(N + 1, dtype=torch.int64, device=dev)
    ctx.synth_fill_device(42, 0, N * L, x); ctx.fixed_offsets_device(0, L, N, off)
    c1 = torch.empty_like(x); ctx.canonicalize_batch_device(x, off, N, out_bytes=c1)
    torch.cuda.synchronize(); print("pass1 ok", N, flush=True)
    lut = torch.arange(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"): lut[a] = b
    y = torch.empty_like(x)
    chunk = min(N, 1_000_000)
    for s in range(0, N, chunk):
        v = x[s * L:(s + chunk) * L].view(chunk, L)
        y[s * L:(s + chunk) * L] = torch.roll(lut[v.flip(1).long()], shifts=137, dims=1).reshape(-1)
    torch.cuda.synchronize(); print("y built", flush=True)
    c3 = torch.empty_like(x); s3 = torch.empty(N, dtype=torch.uint8, device=dev)
    ctx.canonicalize_batch_device(y, off, N, out_bytes=c3, out_strand=s3)
    torch.cuda.synchronize(); print("pass3 ok", flush=True)
    print("equal:", torch.equal(c1[:N*L], c3[:N*L]), "status", ctx.batch_status(), flush=True)
    del x, y, c1, c3
