"""GPU box: small synthetic batches through the C ABI; prints how wrong records are wrong."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from circkit_amd import api
from oracle import oracle as O
import seqsets

ctx = api.Context(0)
rng = np.random.default_rng(5)
def run(name, lens):
    seqs = [bytes(rng.choice(list(b"ACGT"), size=L).astype(np.uint8)) for L in lens]
    data, offs = seqsets.pack(seqs)
    res = ctx.canonicalize_batch(data, offs, want_bytes=True, want_index=True, want_strand=True)
    out, idx, st = res["bytes"], res["index"], res["strand"]
    bad = 0
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        exp, est, eidx = seqsets.expected(O, s)
        g = bytes(out[a:b])
        if g != exp or int(idx[i]) != eidx or int(st[i]) != est:
            bad += 1
            if bad <= 6:
                rc = O.revcomp(s) if hasattr(O, "revcomp") else None
                where = "fwd-rot" if g in s + s else ("rc-rot" if rc and g in rc + rc else "garbage")
                print(" ", name, "rec", i, "len", len(s), "off&15", a & 15, "idx", int(idx[i]), "exp", eidx, "strand", int(st[i]), "exp", est, "bytes:", "ok" if g == exp else where)
    print(name, "records", len(seqs), "bad", bad, flush=True)
run("fixed1000 x64", [1000] * 64)
run("fixed100 x64", [100] * 64)
run("fixed96 x64 (aligned)", [96] * 64)
run("fixed97 x64", [97] * 64)
run("fixed1008 x64", [1008] * 64)
run("mixed 48..1008 x200", [int(x) for x in rng.integers(48, 1009, size=200)])
run("mixed 60..200 x200", [int(x) for x in rng.integers(60, 200, size=200)])
