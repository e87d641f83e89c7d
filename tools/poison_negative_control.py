"""Negative control of tests/test_gpu_poison.py (GPU box): run once with CIRCKIT_LIB = tests/libcirckit_hip_poison_break.so
(python -c 'from tests import poison; poison.build(negative_control=True)': the poison build with the streaming loop's vmcnt
waits removed) and once with tests/libcirckit_hip_poison.so.  Measured (r03): 1289 poisoned records against 0 at 8M x 1 kb;
0 against 0 at 200k records -- the guard needs a saturated memory system to see anything."""
import ctypes, os, sys
import torch
sys.path.insert(0, "/root/repo")
import circkit_amd
from circkit_amd import workloads as W
lib = circkit_amd.load_library()
lib.circkit_debug_poison_count.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
dev = torch.device("cuda", 0)
ctx = circkit_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n, L = 8_000_000, 1000
d_bytes, d_off = W.fixed_length(ctx, dev, n, L, 42, 0)
d_out = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
for _ in range(3):
    ctx.canonicalize_batch_device(d_bytes, d_off, n, out_bytes=d_out)
cnt = ctypes.c_uint32(0)
lib.circkit_debug_poison_count(ctx._h, ctypes.byref(cnt))
print("negative control: poison count with the waits removed =", cnt.value)
