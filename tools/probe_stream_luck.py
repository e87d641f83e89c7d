"""GPU box: is the host-buffer call's full-duplex copy a matter of which streams exist?  K extra HIP streams are created (and
used once) before the ctx makes its copy streams; 1 GB calls are timed.  One process per K (the assignment is per process)."""
import ctypes, os, subprocess, sys, time
if len(sys.argv) == 1:
    for k in (0, 1, 2, 3, 4, 5, 6, 7, 8):
        subprocess.run([sys.executable, __file__, str(k)])
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, circkit_amd
from circkit_amd import workloads as W
K = int(sys.argv[1])
dev = torch.device("cuda", 0)
extra = [torch.cuda.Stream(device=dev) for _ in range(K)]
for st in extra:
    with torch.cuda.stream(st):
        torch.zeros(16, device=dev).add_(1)
torch.cuda.synchronize()
ctx = circkit_amd.Context(0)
if os.environ.get("PROBE_USER_STREAM"):          # the ctx on a stream of the caller's
    user = torch.cuda.Stream(device=dev)
    ctx.set_stream(user.cuda_stream)
lib = circkit_amd.load_library()
L = 1000; S = 1 << 20; nb = S * L
d_bytes, d_off = W.fixed_length(ctx, dev, S, L, 42, 0)
torch.cuda.synchronize()
h_off = d_off.cpu().numpy().astype(np.uint64)
pin_in, pin_out = lib.circkit_host_alloc(nb + 64), lib.circkit_host_alloc(nb + 64)
torch.frombuffer((ctypes.c_uint8 * nb).from_address(pin_in), dtype=torch.uint8).copy_(d_bytes[:nb])
torch.cuda.synchronize()
def call():
    assert lib.circkit_canonicalize_batch(ctx._h, pin_in, h_off.ctypes.data, S, pin_out, None, None, None) == 0
call()
t0 = time.perf_counter()
for _ in range(3):
    call()
print("%d extra streams: %.2f ms per 1 GB call" % (K, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
