import ctypes, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch, circkit_amd
from circkit_amd import workloads as W
dev = torch.device("cuda", 0)
ctx = circkit_amd.Context(0)
lib = circkit_amd.load_library()
L = 1000; mb = 16
S = mb * (1 << 20) // L; nb = S * L
d_bytes, d_off = W.fixed_length(ctx, dev, S, L, 42, 0)
torch.cuda.synchronize()
h_off = d_off.cpu().numpy().astype(np.uint64)
pin_in, pin_out = lib.circkit_host_alloc(nb + 64), lib.circkit_host_alloc(nb + 64)
torch.frombuffer((ctypes.c_uint8 * nb).from_address(pin_in), dtype=torch.uint8).copy_(d_bytes[:nb])
torch.cuda.synchronize()
for i in range(4):
    t0 = time.perf_counter()
    rc = lib.circkit_canonicalize_batch(ctx._h, pin_in, h_off.ctypes.data, S, pin_out, None, None, None)
    print("call %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
