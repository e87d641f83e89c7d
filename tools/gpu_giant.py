"""GPU box: one very long record through the host API (global-scratch tier), timed and checked against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from circkit_amd import api
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
rng = np.random.default_rng(9)
data = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
offs = np.array([0, n], dtype=np.uint64)
ctx = api.Context(0)
ctx.canonicalize_batch(data[:1000], np.array([0, 1000], dtype=np.uint64))
t0 = time.time(); got = ctx.canonicalize_batch(data, offs, want_bytes=True, want_xxh3=True); t1 = time.time()
exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=1); t2 = time.time()
print("n = %d: gpu (host API, incl. PCIe) %.3f s, oracle (1 core) %.3f s, bytes equal %s, hash equal %s" %
      (n, t1 - t0, t2 - t1, bool(np.array_equal(got["bytes"], exp)), bool(got["xxh3"][0] == exp_h[0])), flush=True)
