"""GPU box: what do the SHORT records of BASELINE config 4 cost?  The batch (1M records, lengths ~ 1/L on 200 b .. 20 kb: 35 % of them
<= 1008 symbols, 4 % of the bytes) against the same batch without them (and the short ones alone): ms per batch, bytes only."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import circkit_amd
from circkit_amd import workloads as W
dev = torch.device("cuda", 0)
ctx = circkit_amd.Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
N = 1_000_000
offs = W.log_uniform_offsets(N, 45)
lens = offs[1:] - offs[:-1]
def timed(fn, reps=8):
    fn(); fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream); e1.synchronize()
    return e0.elapsed_time(e1) / reps
def batch(sel_lens):
    o = torch.zeros(sel_lens.numel() + 1, dtype=torch.int64)
    o[1:] = torch.cumsum(sel_lens, 0)
    total = int(o[-1])
    d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    ctx.synth_fill_device(45, 0, total, d_bytes)
    d_out = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    d_off = o.to(dev)
    n = sel_lens.numel()
    ms = timed(lambda: ctx.canonicalize_batch_device(d_bytes, d_off, n, out_bytes=d_out))
    assert ctx.batch_status() == 0
    return n, total, ms
for name, sel in (("all", lens), ("> 1008 only", lens[lens > 1008]), ("<= 1008 only", lens[lens <= 1008]),
                  ("> 1008, the short ones replaced by copies of the long", torch.cat([lens[lens > 1008], lens[lens > 1008][: int((lens <= 1008).sum())]]))):
    n, total, ms = batch(sel)
    print("%-56s %8d records %6.2f GB  %.3f ms  (%.2f TB/s on 2L + 8)" % (name, n, total / 1e9, ms, (2 * total + 8 * n) / ms / 1e9), flush=True)
