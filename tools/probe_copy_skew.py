"""GPU box: does the relative placement of the input and the output buffer matter for a 10 GB -> 10 GB stream?  The library's
plain copy kernel (variant 0) with the destination shifted by a skew against a 2 MB-aligned base; and the headline kernel the same."""
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
print("src %#x dst %#x" % (d_bytes.data_ptr(), big.data_ptr()))
nb = (N * L) & ~15
def timed(fn, reps=5):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for skew in ((1 << 20), 0, (3 << 20) + 8192 + 256, 0, 4096, 0, (1 << 20) + 4096, 0, 1 << 21, 1 << 22, 5 << 20):
    dst = big[skew:]
    cp = timed(lambda: ctx.bench_copy_device(d_bytes, dst, nb, 0))
    kn = timed(lambda: ctx.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=dst))
    print("skew %9d: copy %.3f ms = %.0f GB/s   canonicalize %.3f ms" % (skew, cp, 2 * nb / cp / 1e6, kn), flush=True)
