"""GPU box: the host-buffer entry point per call at different batch sizes (the CLI hands it 64 MB chunks): ms per call with
page-locked payload buffers, offsets pageable / page-locked, parts forced to 1 or left to the library."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import circkit_amd
from circkit_amd import workloads as W

dev = torch.device("cuda", 0)
ctx = circkit_amd.Context(0)
lib = circkit_amd.load_library()
L = 1000
WITH_HASH = os.environ.get("PROBE_HASH") == "1"      # + the XXH3 of every record (what uniq asks for)
for mb in (16, 64, 256, 1024):
    S = mb * (1 << 20) // L
    nb = S * L
    d_bytes, d_off = W.fixed_length(ctx, dev, S, L, 42, 0)
    torch.cuda.synchronize()
    h_off = d_off.cpu().numpy().astype(np.uint64)
    h_hash = np.empty(S, dtype=np.uint64)
    pin_in, pin_out, pin_off = lib.circkit_host_alloc(nb + 64), lib.circkit_host_alloc(nb + 64), lib.circkit_host_alloc(8 * (S + 1))
    torch.frombuffer((ctypes.c_uint8 * nb).from_address(pin_in), dtype=torch.uint8).copy_(d_bytes[:nb])
    ctypes.memmove(pin_off, h_off.ctypes.data, 8 * (S + 1))
    torch.cuda.synchronize()
    for label, off_ptr, parts in (("offsets pageable", h_off.ctypes.data, None), ("offsets pinned", pin_off, None), ("offsets pinned, 1 part", pin_off, "1")):
        if parts:
            os.environ["CIRCKIT_HOST_BATCH_PARTS"] = parts
        else:
            os.environ.pop("CIRCKIT_HOST_BATCH_PARTS", None)
        def call():
            rc = lib.circkit_canonicalize_batch(ctx._h, pin_in, off_ptr, S, pin_out, None, None, h_hash.ctypes.data if WITH_HASH else None)
            assert rc == 0, rc
        call(); call()
        reps = max(3, 512 // mb)
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        dt = (time.perf_counter() - t0) / reps
        print("%5d MB  %-24s %.3f ms per call = %.1f GB/s in + out" % (mb, label, dt * 1e3, 2 * nb / dt / 1e9), flush=True)
    # pageable payload buffers (plain numpy arrays): the runtime stages every copy
    pg_in = np.empty(nb + 64, dtype=np.uint8); pg_out = np.empty(nb + 64, dtype=np.uint8)
    ctypes.memmove(pg_in.ctypes.data, pin_in, nb)
    os.environ.pop("CIRCKIT_HOST_BATCH_PARTS", None)
    def callp():
        rc = lib.circkit_canonicalize_batch(ctx._h, pg_in.ctypes.data, h_off.ctypes.data, S, pg_out.ctypes.data, None, None, None)
        assert rc == 0, rc
    callp(); callp()
    reps = max(3, 512 // mb)
    t0 = time.perf_counter()
    for _ in range(reps):
        callp()
    dt = (time.perf_counter() - t0) / reps
    print("%5d MB  %-24s %.3f ms per call = %.1f GB/s in + out" % (mb, "pageable payload", dt * 1e3, 2 * nb / dt / 1e9), flush=True)
    for p in (pin_in, pin_out, pin_off):
        lib.circkit_host_free(p)
    del d_bytes, d_off
