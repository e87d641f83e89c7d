"""GPU box: large randomized parity run of the batch ABI against the oracle (bytes + XXH3, and the bytes-only /
index builds on a subsample).  Mixtures aimed at the streaming kernel's rare paths: planted duplicate minimal
16-mers, tandem repeats, reverse-complement palindromes (equal minimal keys on both strands), records around the
48 / 240 / 1008 limits, N / '-' sprinkled in, lengths 0..1300.
usage: python tools/gpu_fuzz.py [seed] [records] [profile]   (profile "long": most records 1009..2032 bases -- the
two-words-per-lane build of the streaming kernel; "nrich": half of the records carry N / '-' -- the batch's mode gets
MODE_ALPHA: 4-bit register routine in the streaming kernel and the rescue pass; "longn": records of 1..9 kb with a few
N -- the 2-bit-with-N-mask mode of the LDS tiers and its fallbacks; "prefixn": records of 0.4..6 kb without A except for
planted A-runs followed by N / G / T / C, on both strands -- an N inside the minimal window and near-ties around it;
"team": records of 15..700 kb -- the multi-wave team modes and their fallbacks; "leanties": records of 1.0..20 kb -- the mixed
kernel's lean routine and the cases it settles itself: the minimal 16-mer planted two to five times, a reverse-complement
palindrome around it (both strands own it), lengths a little above multiples of 1024 with the minimum at the record's start
or end (the periodic twin), whole-record palindromes and tandem repeats, half of the batches with 1 % N)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch first)
from circkit_amd import api
from oracle import oracle as O
from tests import seqsets

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
profile = sys.argv[3] if len(sys.argv) > 3 else "short"
rng = np.random.default_rng(seed)
LUT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN-", b"TGCAN-"):
    COMP[a] = b


def rand(n):
    return LUT[rng.integers(0, 4, size=n)]


seqs = []
t0 = time.time()
if profile == "prefixn":
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    for i in range(count):
        L = int(rng.integers(400, 6000))
        bg = bytearray(rng.choice(list(b"CGT"), size=L, p=[0.2, 0.4, 0.4]).astype(np.uint8).tobytes())
        run = int(rng.integers(3, 14))
        for sp in sorted(rng.choice(np.arange(20, L - 40, 30), size=int(rng.integers(1, 5)), replace=False)):
            motif = b"A" * run + bytes(rng.choice(list(b"NGTCN"), size=1).astype(np.uint8)) + bytes(rng.choice(list(b"ACGTN"), size=8, p=[.24, .24, .24, .24, .04]).astype(np.uint8))
            if rng.random() < 0.5:
                motif = motif.translate(comp)[::-1]
            bg[sp:sp + len(motif)] = motif
        seqs.append(bytes(bg))
if profile == "leanties":
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    with_n = seed % 2 == 1
    for i in range(count):
        m = rng.integers(0, 100)
        L = int(rng.integers(1009, 20400)) if m % 3 else int(rng.integers(1, 20)) * 1024 + int(rng.integers(-8, 40))
        bg = bytearray(rng.choice(list(b"CGT"), size=L).astype(np.uint8).tobytes()) if m < 70 else bytearray(rand(L).tobytes())
        if m < 25:                                                        # the minimal key several times
            for sp in rng.integers(0, L - 16, size=int(rng.integers(2, 6))):
                bg[sp:sp + 16] = b"A" * 16
        elif m < 40:                                                      # a palindromic core that both strands own
            half = b"A" * int(rng.integers(6, 10)) + bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 5))).astype(np.uint8))
            pal = half + half.translate(comp)[::-1]
            sp = int(rng.integers(0, L - len(pal)))
            bg[sp:sp + len(pal)] = pal
        elif m < 55:                                                      # the minimum at the very start / end (wraps)
            sp = int(rng.choice([0, 1, 5, 15, 16, L - 1, L - 2, L - 15, L - 16, L - 17]))
            for k in range(17):
                bg[(sp + k) % L] = ord("A")
        elif m < 60:
            half = rand(L // 2).tobytes()
            bg = bytearray(half + half.translate(comp)[::-1])
        elif m < 64:
            unit = rand(int(rng.integers(50, 900))).tobytes()
            bg = bytearray((unit * (L // len(unit) + 1))[:L])
        if with_n:
            for p_ in rng.integers(0, len(bg), size=max(1, len(bg) // 100)):
                bg[int(p_)] = ord("N")
        s_ = bytes(bg)
        seqs.append(s_ if m % 2 else s_.translate(comp)[::-1])
if profile == "team":
    # 15..700 kb: the team modes (four waves in tier A, sixteen in the last LDS stage) and what they leave to the general
    # routine -- a few N, tandem repeats, reverse-complement palindromes, gaps
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for i in range(count):
        k = rng.integers(0, 100)
        L = int(rng.integers(15_000, 130_000)) if k < 85 else int(rng.integers(130_000, 700_000))
        m = rng.integers(0, 100)
        if m < 8:
            unit = rand(int(rng.integers(300, 3000))).tobytes()
            s = (unit * (L // len(unit) + 1))[:L]
        elif m < 12:
            half = rand(L // 2).tobytes()
            s = half + half.translate(comp)[::-1]
        else:
            a = rand(L)
            if m < 45:
                for _ in range(int(rng.integers(1, 40))):
                    a[int(rng.integers(0, L))] = ord("N") if rng.random() < 0.9 else ord("-")
            s = a.tobytes()
        seqs.append(s)
for i in range(0 if profile in ("prefixn", "team", "leanties") else count):
    k = rng.integers(0, 100)
    if profile == "longn":
        L = int(rng.integers(1009, 9000)) if k < 90 else int(rng.integers(48, 1009))
    elif profile == "long" and k < 85:
        L = int(rng.integers(1009, 2033)) if k < 60 else int(rng.choice([1009, 1010, 1023, 1024, 1025, 1039, 1040, 1041, 1164, 1500, 1679, 2015, 2016, 2017, 2031, 2032, 2033, 2047, 2048, 2049]))
    elif k < 30: L = 1000
    elif k < 60: L = int(rng.integers(48, 1009))
    elif k < 70: L = int(rng.choice([0, 1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 239, 240, 241, 242, 255, 256, 257,
                                      991, 992, 993, 1006, 1007, 1008, 1009, 1010, 1023, 1024, 1025]))
    elif k < 80: L = int(rng.integers(1, 1301))
    else: L = int(rng.integers(60, 400))
    s = rand(L)
    m = rng.integers(0, 100)
    if L >= 64:
        if m < 8:                                  # the minimal 16-mer planted twice (or more)
            reps = int(rng.integers(2, 4))
            w = int(rng.integers(8, 20))
            for _ in range(reps):
                p = int(rng.integers(0, L - w))
                s[p:p + w] = ord("A")
        elif m < 14:                               # tandem repeat, period p (may or may not divide L)
            p = int(rng.integers(1, 40))
            s = np.resize(s[:p], L)
        elif m < 18:                               # reverse-complement palindrome: both strands give the same rotation set
            h = s[:L // 2]
            s = np.concatenate([h, COMP[h][::-1]])
        elif m < 20:                               # almost periodic: one mismatch at the end
            p = int(rng.integers(1, 17))
            s = np.resize(s[:p], L).copy()
            s[-1] = ord("T") if s[-1] != ord("T") else ord("G")
        elif m < (25 if profile in ("short", "long") else 75):        # a few N / '-' somewhere
            for _ in range(int(rng.integers(1, 4 if profile != "longn" else 30))):
                s[int(rng.integers(0, len(s)))] = ord("N") if rng.random() < 0.7 else ord("-")
        elif m < 27:
            s[:] = ord("A")
            s[int(rng.integers(0, len(s)))] = ord("C")
    seqs.append(s.tobytes())
data, offs = seqsets.pack(seqs)
print("generated %d records, %d bases in %.1f s" % (count, len(data), time.time() - t0), flush=True)
t0 = time.time()
exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=16)
print("oracle %.1f s" % (time.time() - t0), flush=True)
ctx = api.Context(0)
t0 = time.time()
got = ctx.canonicalize_batch(data, offs, want_bytes=True, want_xxh3=True)
lean = ctx.canonicalize_batch(data, offs, want_bytes=True)
full = ctx.canonicalize_batch(data, offs, want_bytes=True, want_index=True, want_strand=True, want_xxh3=True)
# hash-only through the device entry point: no output bytes, the XXH3 of records the fused hash does not cover is taken from
# (rotation, strand) views of the input (what `uniq` without --canonicalize asks for)
dev = torch.device("cuda", 0)
d_data = torch.from_numpy(np.concatenate([data, np.zeros(64, dtype=np.uint8)])).to(dev)
d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
d_hash = torch.zeros(count, dtype=torch.int64, device=dev)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.canonicalize_batch_device(d_data, d_offs, count, out_xxh3=d_hash)
torch.cuda.synchronize()
honly = {"bytes": exp, "xxh3": d_hash.cpu().numpy().view(np.uint64)}
# the same batch twice more through the device entry point, bytes + hashes: by now the library launches the kernels of the mode
# the batches before reported (launch_canon's mode guess) instead of every build
d_out = torch.zeros(len(data) + 64, dtype=torch.uint8, device=dev)
d_hash2 = torch.zeros(count, dtype=torch.int64, device=dev)
for _ in range(2):
    ctx.canonicalize_batch_device(d_data, d_offs, count, out_bytes=d_out, out_xxh3=d_hash2)
torch.cuda.synchronize()
guessed = {"bytes": d_out[:len(data)].cpu().numpy(), "xxh3": d_hash2.cpu().numpy().view(np.uint64)}
print("gpu (3 builds, host API; hash-only, device API) %.1f s" % (time.time() - t0), flush=True)
bad = 0
for name, r in (("bytes+xxh3", got), ("bytes", lean), ("bytes+xxh3+index+strand", full), ("xxh3 only (views)", honly), ("bytes+xxh3, device API under the mode guess", guessed)):
    if not np.array_equal(r["bytes"], exp):
        for i in range(count):
            a, b = int(offs[i]), int(offs[i + 1])
            if r["bytes"][a:b].tobytes() != exp[a:b].tobytes():
                bad += 1
                if bad < 6:
                    print("MISMATCH", name, "rec", i, "len", b - a, seqs[i][:60])
    if r["xxh3"] is not None and not np.array_equal(r["xxh3"], exp_h):
        w = np.nonzero(r["xxh3"] != exp_h)[0]
        bad += len(w)
        print("HASH MISMATCH", name, len(w), "records, first", int(w[0]), "len", len(seqs[int(w[0])]))
# index / strand against the record-loop restatement on a subsample (python-level oracle calls are slow)
for i in rng.choice(count, size=min(count, 20000), replace=False):
    s = seqs[int(i)]
    _, st, idx = seqsets.expected(O, s)
    if int(full["strand"][i]) != st or (s and int(full["index"][i]) != idx):
        bad += 1
        if bad < 12:
            print("INDEX/STRAND MISMATCH rec", int(i), "len", len(s), "got", int(full["index"][i]), int(full["strand"][i]), "exp", idx, st)
print("seed %d: %d records, mismatches: %d" % (seed, count, bad), flush=True)
sys.exit(1 if bad else 0)
