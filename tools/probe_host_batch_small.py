"""GPU box: one host-buffer call under CIRCKIT_DEBUG_TIMING=1 (the library prints where a call's time goes).
usage: python tools/probe_host_batch_small.py [MB] [hash]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, circkit_amd
from circkit_amd import workloads as W
dev = torch.device("cuda", 0)
ctx = circkit_amd.Context(0)
lib = circkit_amd.load_library()
L = 1000
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
with_hash = len(sys.argv) > 2
S = mb * (1 << 20) // L; nb = S * L
d_bytes, d_off = W.fixed_length(ctx, dev, S, L, 42, 0)
torch.cuda.synchronize()
h_off = d_off.cpu().numpy().astype(np.uint64)
h_hash = np.empty(S, dtype=np.uint64)
pin_in, pin_out = lib.circkit_host_alloc(nb + 64), lib.circkit_host_alloc(nb + 64)
torch.frombuffer((ctypes.c_uint8 * nb).from_address(pin_in), dtype=torch.uint8).copy_(d_bytes[:nb])
torch.cuda.synchronize()
for i in range(4):
    t0 = time.perf_counter()
    rc = lib.circkit_canonicalize_batch(ctx._h, pin_in, h_off.ctypes.data, S, pin_out, None, None, h_hash.ctypes.data if with_hash else None)
    print("call %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
