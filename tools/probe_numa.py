"""GPU box: does the NUMA node of the page-locked buffers matter for the host-buffer entry point?  The calling thread is
pinned to one node's CPUs while it allocates (first touch decides where the pages live), then 1 GB calls are timed."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, circkit_amd
from circkit_amd import workloads as W

def cpus(node):
    out = []
    for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out

dev = torch.device("cuda", 0)
ctx = circkit_amd.Context(0)
lib = circkit_amd.load_library()
L = 1000; S = 1 << 20; nb = S * L
d_bytes, d_off = W.fixed_length(ctx, dev, S, L, 42, 0)
torch.cuda.synchronize()
h_off = d_off.cpu().numpy().astype(np.uint64)
all_cpus = os.sched_getaffinity(0)
for rep in range(2):
    for node in (0, 1):
        os.sched_setaffinity(0, cpus(node))
        time.sleep(0.01)
        pin_in, pin_out = lib.circkit_host_alloc(nb + 64), lib.circkit_host_alloc(nb + 64)
        torch.frombuffer((ctypes.c_uint8 * nb).from_address(pin_in), dtype=torch.uint8).copy_(d_bytes[:nb])
        torch.cuda.synchronize()
        for where in (node, 1 - node):
            os.sched_setaffinity(0, cpus(where))
            def call():
                rc = lib.circkit_canonicalize_batch(ctx._h, pin_in, h_off.ctypes.data, S, pin_out, None, None, None)
                assert rc == 0
            call()
            t0 = time.perf_counter()
            for _ in range(3):
                call()
            print("buffers allocated from node %d, calls from node %d: %.2f ms per 1 GB call" % (node, where, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
        lib.circkit_host_free(pin_in); lib.circkit_host_free(pin_out)
os.sched_setaffinity(0, all_cpus)
