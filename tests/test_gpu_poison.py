"""GPU guard for the streaming kernel's hand-counted `s_waitcnt vmcnt(N)` protocol (canon_stream.h): the CK_DEBUG_POISON build
of the library overwrites every LDS ring image with a poison pattern right before its DMA is re-issued and counts the records
whose staged chunk still reads the pattern -- i.e. were consumed before their DMA had landed.  The CPU emulator cannot see this
(its DMA is a synchronous memcpy); a miscount would otherwise only show as a rare wrong answer under timing pressure."""
import ctypes
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
import circkit_amd
from circkit_amd import workloads as W
lib = circkit_amd.load_library()
assert "poison" in circkit_amd.api.LIB_PATH
lib.circkit_debug_poison_count.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
dev = torch.device("cuda", 0)
ctx = circkit_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ref = None
# Sizes that saturate the memory system: the guard's negative control (tools/poison_negative_control.py: the same build with
# the waits REMOVED) counts ~1300 poisoned records at 8M x 1 kb -- and none at 200k, where every DMA lands in time anyway.
for name, n, L, nfrac, outs in (("ROWS=1", 6_000_000, 1000, 0.0, "bytes"), ("ROWS=1 short records (bytes-only pair build)", 12_000_000, 400, 0.0, "bytes"), ("ROWS=1 + XXH3 (pair build)", 6_000_000, 1000, 0.0, "hash"),
                                ("ROWS=1 XXH3 only (pair build, no stores)", 6_000_000, 1000, 0.0, "hashonly"), ("ROWS=1 + XXH3 ALPHA", 4_000_000, 1000, 0.01, "hash"), ("ROWS=1 ALPHA", 4_000_000, 1000, 0.01, "bytes"),
                                ("ROWS=2", 3_000_000, 1500, 0.0, "bytes"), ("index/strand build", 3_000_000, 777, 0.0, "aux")):
    d_bytes, d_off = W.fixed_length(ctx, dev, n, L, 42, 0)
    if nfrac:
        W.sprinkle_n(d_bytes, n * L, nfrac, 46, dev)
    d_out = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
    d_hash = torch.empty(n, dtype=torch.int64, device=dev) if outs in ("hash", "hashonly") else None
    d_idx = torch.empty(n, dtype=torch.int32, device=dev) if outs == "aux" else None
    d_st = torch.empty(n, dtype=torch.uint8, device=dev) if outs == "aux" else None
    for _ in range(3):                                   # the first batch also warms the device-side build selection
        ctx.canonicalize_batch_device(d_bytes, d_off, n, out_bytes=None if outs == "hashonly" else d_out, out_index=d_idx, out_strand=d_st, out_xxh3=d_hash)
    assert ctx.batch_status() == 0
    cnt = ctypes.c_uint32(123)
    assert lib.circkit_debug_poison_count(ctx._h, ctypes.byref(cnt)) == 0
    print("poison %%s: %%d" %% (name, cnt.value))
    assert cnt.value == 0, name
    if outs == "hashonly":                               # same batch as the case before: same hashes
        torch.cuda.synchronize()
        assert torch.equal(d_hash, ref_hash), name
        del d_bytes, d_off, d_out, d_hash
        torch.cuda.empty_cache()
        continue
    if outs == "hash" and not nfrac:
        torch.cuda.synchronize()
        ref_hash = d_hash.clone()
    # idempotence as a sanity check that the poisoned build still computes (the canonical form of a canonical record is itself)
    d_out2 = torch.empty_like(d_out)
    ctx.canonicalize_batch_device(d_out, d_off, n, out_bytes=d_out2)
    torch.cuda.synchronize()
    assert torch.equal(d_out[:n * L], d_out2[:n * L]), name
    del d_bytes, d_off, d_out, d_out2, d_hash, d_idx, d_st
    torch.cuda.empty_cache()
print("POISON-OK")
"""


def test_no_record_is_packed_before_its_dma_has_landed():
    from tests import poison
    lib = poison.build()
    env = dict(os.environ)
    env["CIRCKIT_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and "POISON-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
