"""GPU box: the pair build with bytes (`uniq --canonicalize`'s hash build: 10 GB read, 10 GB written) reads 3.8-4.2 ms depending on the
process it runs in, its hash-only twin 3.4 everywhere.  Does the placement of the OUTPUT relative to the input decide?  The batch
call with canonical bytes + hashes and with hashes only, the destination shifted by a skew against one big allocation; the relative
offset (dst - src) is printed modulo 2 MB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import circkit_amd
from circkit_amd import workloads as W
dev = torch.device("cuda", 0)
ctx = circkit_amd.Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
N, L = 10_000_000, 1000
d_bytes, d_off = W.fixed_length(ctx, dev, N, L, 42, 0)
big = torch.empty(N * L + (64 << 20), dtype=torch.uint8, device=dev)
d_hash = torch.empty(N, dtype=torch.int64, device=dev)
print("src %#x dst %#x" % (d_bytes.data_ptr(), big.data_ptr()))
def timed(fn, reps=5):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream); e1.synchronize()
    return e0.elapsed_time(e1) / reps
ho = timed(lambda: ctx.canonicalize_batch_device(d_bytes, d_off, N, out_xxh3=d_hash))
print("hash only %.3f ms" % ho)
for skew in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 1 << 21, (2 << 20) + 8192, (3 << 20) + 8192 + 256, 1 << 22, 5 << 20, 0, (1 << 20) + 65536, 17 << 20, 33 << 20):
    dst = big[skew:]
    kb = timed(lambda: ctx.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=dst))
    kh = timed(lambda: ctx.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=dst, out_xxh3=d_hash))
    print("skew %9d  (dst - src) mod 2MB = %8d: bytes %.3f ms   bytes + xxh3 %.3f ms" % (skew, (dst.data_ptr() - d_bytes.data_ptr()) % (2 << 20), kb, kh), flush=True)
