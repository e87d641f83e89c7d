"""Shared input sets for the parity tests (oracle vs emulator on CPU, oracle vs HIP on the GPU)."""
import random

import numpy as np


def pack(seqs):
    lens = [len(s) for s in seqs]
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    data = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    return data, offs


def rand_seq(rng, n, alpha=b"ACGT"):
    return bytes(rng.choice(alpha) for _ in range(n))


def revcomp_acgt(s):
    return s.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]


def adversarial(seed=11):
    """SURVEY.md 8d adversarial set + boundary lengths of the packed-word machinery."""
    rng = random.Random(seed)
    S = [b"", b"A", b"T", b"AC", b"CA", b"ATGCA", b"AAA", b"ATT", b"TAA", b"banana", b"TGCA", b"GCAT",
         b"AATCAATTTCCTCCATCACCTAGTTTATGTAGAAACGCTGCTA", b"TCCTCCATCACCTAGTTTATGTAGAAACGCTGCTAAATCAATT"]
    for n in (1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 100, 255, 256, 257, 1000, 1008, 1023,
              1024, 1025, 1040, 2047, 2048, 2049, 3000):
        S.append(b"A" * n)                                   # all-A
        S.append(b"A" * (n - 1) + b"T")                      # A...AT
        S.append(b"T" + b"A" * (n - 1))                      # TA...A
        S.append(b"T" * n)                                   # max key everywhere
        S.append(rand_seq(rng, n))
        S.append(rand_seq(rng, n, b"ACGTN"))                 # 4-bit mode
        S.append(rand_seq(rng, n, b"ACGTN-"))
        S.append(rand_seq(rng, n, b"AC"))                    # low complexity
        if n <= 300:
            S.append(bytes(rng.randint(0x20, 0x7E) for _ in range(n)))   # arbitrary ASCII: 8-bit mode
            S.append(rand_seq(rng, n, b"ACGTRYKMacgtn"))
    for p in (1, 2, 3, 5, 7, 16, 17, 48, 50, 64, 100):        # period-p repeats, p | n and p does not divide n
        unit = rand_seq(rng, p)
        for reps in (2, 3, 10, 21):
            S.append(unit * reps)
            S.append((unit * reps)[:-1] if len(unit * reps) > 1 else unit)
            t = bytearray(unit * reps)
            t[-1] = ord("T") if t[-1] != ord("T") else ord("A")          # one mismatch at the end
            S.append(bytes(t))
            S.append(unit * reps + b"N")
    for n in (50, 64, 200, 1000):                            # reverse palindromes: fwd == rc
        h = rand_seq(rng, n // 2)
        S.append(h + revcomp_acgt(h))
    for _ in range(20):                                      # rotations / strand flips of one record
        base = rand_seq(rng, rng.randint(48, 400))
        k = rng.randrange(len(base))
        S.append(base[k:] + base[:k])
        S.append(revcomp_acgt(base))
    # poly-A runs inside random sequence, tandem repeats embedded in unique flanks
    for _ in range(20):
        a = rand_seq(rng, rng.randint(30, 500))
        S.append(a + b"A" * rng.randint(17, 80) + rand_seq(rng, rng.randint(30, 500)) + b"A" * rng.randint(17, 80))
        u = rand_seq(rng, rng.randint(1, 20))
        S.append(a + u * rng.randint(3, 30) + rand_seq(rng, 40) + u * rng.randint(3, 30))
    return S


def random_mixed(seed, count, lo, hi, alpha=b"ACGT"):
    rng = random.Random(seed)
    return [rand_seq(rng, rng.randint(lo, hi), alpha) for _ in range(count)]


def expected(O, s):
    """(canonical bytes, strand, reference-visible index) from the oracle."""
    l = O.lmsr(s)
    rc = O.lmsr(O.revcomp(l))
    fwd = l < rc
    idx = O.lmsr_index(s) if fwd else O.lmsr_index(O.revcomp(l))
    return (l if fwd else rc), (0 if fwd else 1), idx
