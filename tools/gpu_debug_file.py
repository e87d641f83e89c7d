"""GPU box: canonicalize the records of a FASTA fixture through the C ABI and list the records that differ from the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from circkit_amd import api
from oracle import oracle as O

path = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/ref_examples/nim_cated/realistic_input.fasta"
seqs = [O.normalize(r[1])[0] for r in O.read_fasta(open(path, "rb").read())]
offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
offs[1:] = np.cumsum([len(s) for s in seqs])
data = np.frombuffer(b"".join(seqs), dtype=np.uint8)
ctx = api.Context(0)
res = ctx.canonicalize_batch(data, offs, want_bytes=True, want_index=True, want_strand=True)
out = res["bytes"]
bad = 0
for i, s in enumerate(seqs):
    a, b = int(offs[i]), int(offs[i + 1])
    c = O.canonicalize(s)
    g = bytes(out[a:b])
    if g != c:
        bad += 1
        d = [k for k in range(len(c)) if g[k] != c[k]]
        print("rec", i, "group", i // 8, "len", len(s), "off", a, "a16", a & 15, "ndiff", len(d), "first", d[:5], "last", d[-3:],
              "is_rotation", g in (c + c), "acgt_only", set(s) <= set(b"ACGT"))
print("records", len(seqs), "bad", bad)
res2 = ctx.canonicalize_batch(data, offs, want_bytes=True)
print("bytes-only run equal to first run:", bool((res2["bytes"] == out).all()))
