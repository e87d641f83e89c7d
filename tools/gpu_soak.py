"""GPU box: determinism soak.  The same batches are canonicalized many times; a 64-bit checksum of the output bytes
(and of the hash array) must never change.  Catches races in the hand-counted vmcnt / barrier protocol that a single
parity run could miss.  usage: python tools/gpu_soak.py [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from circkit_amd import api

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)


def checksum(t):
    v = t.view(torch.int64) if t.dtype != torch.int64 else t
    w = torch.arange(1, v.numel() + 1, device=dev, dtype=torch.int64)
    return int((v * (w * 0x9E3779B1 + 12345)).sum().item())


def soak(name, lens_fn, n, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    lens = lens_fn(n, g).to(torch.int64)
    offs = torch.zeros(n + 1, dtype=torch.int64)
    offs[1:] = torch.cumsum(lens, 0)
    total = int(offs[-1])
    pad = (-total) % 8 + 64
    o = offs.to(dev)
    d = torch.empty(total + pad, dtype=torch.uint8, device=dev)
    ctx.synth_fill_device(seed, 0, total, d)
    out = torch.zeros(total + pad, dtype=torch.uint8, device=dev)
    h = torch.zeros(n, dtype=torch.int64, device=dev)
    ref = None
    for r in range(reps):
        out.zero_(); h.zero_()
        ctx.canonicalize_batch_device(d, o, n, out_bytes=out, out_xxh3=h if r % 2 else None)
        ctx.synchronize()
        c = (checksum(out[:(total + pad) // 8 * 8]), checksum(h) if r % 2 else None)
        if ref is None:
            ref = [c[0], None]
        if ref[1] is None and c[1] is not None:
            ref[1] = c[1]
        assert c[0] == ref[0] and (c[1] is None or c[1] == ref[1]), (name, r, c, ref)
    print("%-28s %d records, %.2f Gbases, %d repeats: stable" % (name, n, total / 1e9, reps), flush=True)


soak("fixed 1000", lambda n, g: torch.full((n,), 1000), 4_000_000, 42)
soak("uniform 48..1008", lambda n, g: torch.randint(48, 1009, (n,), generator=g), 4_000_000, 43)
soak("uniform 1009..2032", lambda n, g: torch.randint(1009, 2033, (n,), generator=g), 2_000_000, 44)
soak("95% short + 5% long", lambda n, g: torch.where(torch.rand(n, generator=g) < 0.95, torch.randint(300, 901, (n,), generator=g),
                                                     torch.randint(3000, 8001, (n,), generator=g)), 2_000_000, 45)
soak("log-uniform 200..20000", lambda n, g: torch.exp(np.log(200.0) + torch.rand(n, generator=g, dtype=torch.float64) * np.log(100.0)).to(torch.int64), 500_000, 46)
soak("log-uniform 200..300000", lambda n, g: torch.exp(np.log(200.0) + torch.rand(n, generator=g, dtype=torch.float64) * np.log(1500.0)).to(torch.int64), 40_000, 47)
soak("fixed 120000 (small batch)", lambda n, g: torch.full((n,), 120_000), 550, 48)
print("soak ok")
