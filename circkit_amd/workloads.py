"""Synthetic BASELINE workloads, generated ON the device -- shared by bench.py and the full-size
tests so that both see the same bytes.  Host-side plumbing only (torch for device memory and index arithmetic, the
ctx for the counter-based base generator); nothing here is on the measured path.

    config 2 / 5  fixed_length()            N x L uniform ACGT, seed 42, base generator keyed by the global base index
    config 3      plant_duplicates()        second half = rotated / reverse-complemented copies of the first, shuffled
    config 4      log_uniform_offsets()     P(L) ~ 1/L on [200, 20000], seed 45
    1 % N         sprinkle_n()              every base replaced by N with the given probability
"""
import math

import torch


def fixed_length(ctx, dev, n_records, length, seed=42, first_record=0):
    """(d_bytes[+64], d_offsets) of records [first_record, first_record + n_records) of the job's seed stream."""
    total = n_records * length
    d_off = torch.empty(n_records + 1, dtype=torch.int64, device=dev)
    ctx.fixed_offsets_device(0, length, n_records, d_off)
    d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    ctx.synth_fill_device(seed, first_record * length, total, d_bytes)
    return d_bytes, d_off


def log_uniform_offsets(n_records, seed=45, lo=200, hi=20000):
    """CPU int64 offsets [n_records + 1] with lengths ~ 1/L on [lo, hi] (a truncated Zipf with exponent 1)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    u = torch.rand(n_records, generator=g, dtype=torch.float64)
    lens = torch.exp(math.log(lo) + u * (math.log(hi) - math.log(lo))).to(torch.int64)
    offs = torch.zeros(n_records + 1, dtype=torch.int64)
    offs[1:] = torch.cumsum(lens, 0)
    return offs


def sprinkle_n(d_bytes, total, frac, seed, dev):
    gen = torch.Generator(device=dev).manual_seed(seed)
    for c0 in range(0, total, 1 << 28):
        m = min(1 << 28, total - c0)
        d_bytes[c0:c0 + m][torch.rand(m, generator=gen, device=dev) < frac] = 0x4E


def complement_lut(dev):
    lut = torch.arange(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        lut[a] = b
    return lut


def plant_duplicates(d_bytes, n_records, length, dev, seed_dup=43, seed_shuffle=44):
    """config 3, in place: records [n/2, n) become uniformly chosen records of [0, n/2), each rotated by a uniform k
    and reverse-complemented with p = 0.5; then all records are shuffled.  Expected distinct canonical forms: n - n/2."""
    half = n_records // 2
    gen = torch.Generator(device=dev).manual_seed(seed_dup)
    lut = complement_lut(dev)
    view = d_bytes[:n_records * length].view(n_records, length)
    col = torch.arange(length, device=dev)
    for s0 in range(half, n_records, 500_000):
        m = min(500_000, n_records - s0)
        src = torch.randint(0, half, (m,), generator=gen, device=dev)
        k = torch.randint(0, length, (m, 1), generator=gen, device=dev)
        rows = torch.gather(view[src], 1, (col.unsqueeze(0) + k) % length)
        flip = torch.rand(m, generator=gen, device=dev) < 0.5
        rows[flip] = lut[rows[flip].flip(1).long()]
        view[s0:s0 + m] = rows
    perm = torch.randperm(n_records, generator=torch.Generator(device=dev).manual_seed(seed_shuffle), device=dev)
    for c0 in range(0, length, 100):                      # shuffle column block by column block (bounded temporaries)
        view[:, c0:c0 + 100] = view[perm, c0:c0 + 100]


def revcomp_rotate_csr(d_bytes, d_off, n_records, dev, shift=137, chunk=50_000):
    """Every record of a CSR batch reverse-complemented (ACGT; other bytes kept) and rotated by `shift` mod its length:
    the canonical form must not notice.  Returns a new payload tensor with the same offsets."""
    lut = complement_lut(dev)
    out = torch.empty_like(d_bytes)
    for r0 in range(0, n_records, chunk):
        r1 = min(n_records, r0 + chunk)
        o = d_off[r0:r1 + 1]
        lens = o[1:] - o[:-1]
        b0, b1 = int(o[0]), int(o[-1])
        if b1 == b0:
            continue
        rec = torch.repeat_interleave(torch.arange(r1 - r0, device=dev), lens)
        j = torch.arange(b0, b1, device=dev) - o[:-1][rec]
        ln = lens[rec]
        src = o[:-1][rec] + (ln - 1 - (j + shift) % ln)
        out[b0:b1] = lut[d_bytes[src].long()]
    return out
