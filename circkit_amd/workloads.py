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


# ---- config 3 over a whole multi-GPU job: duplicates whose original lives on ANOTHER rank -------------------------------
# The reference keeps the first record of each hash over the WHOLE input (src/uniq.rs:27,47-48).  A sharded job whose
# duplicates all sit in their original's shard cannot tell a working exchange from one that returns every rank's local
# answer, so here the second half of every rank's records are copies of base records drawn from ALL ranks' first halves.
# The base generator is counter-based (include/circkit.h, circkit_synth_fill_device): record g of the job is bases
# [g*L, (g+1)*L) of the seed stream, so any rank regenerates any other rank's base records locally -- no communication
# while building the workload, and every rank can also compute the job-wide expectation by itself.
def job_draws(n_records, length, world, rank, dev, seed_dup=43, seed_shuffle=44):
    """The planting decisions of `rank` (the same on whoever asks, given the same device type): for each of the n - n//2
    duplicates its source rank, source base record (< n//2), rotation and strand flip; and the shuffle of the shard."""
    half = n_records // 2
    d = n_records - half
    gen = torch.Generator(device=dev).manual_seed(seed_dup + rank)
    src_rank = torch.randint(0, world, (d,), generator=gen, device=dev)
    src_j = torch.randint(0, max(half, 1), (d,), generator=gen, device=dev)
    k = torch.randint(0, length, (d,), generator=gen, device=dev)
    flip = torch.rand(d, generator=gen, device=dev) < 0.5
    perm = torch.randperm(n_records, generator=torch.Generator(device=dev).manual_seed(seed_shuffle + rank), device=dev)
    return src_rank, src_j, k, flip, perm


def job_keys(n_records, length, world, rank, dev, seed_dup=43, seed_shuffle=44):
    """key[p] of position p of rank's shard = job-wide index of the base record it is a copy of (itself for a base record):
    two records of the job have the same canonical form iff their keys are equal (random 1 kb records do not collide)."""
    half = n_records // 2
    src_rank, src_j, _, _, perm = job_draws(n_records, length, world, rank, dev, seed_dup, seed_shuffle)
    pre = torch.empty(n_records, dtype=torch.int64, device=dev)
    pre[:half] = rank * n_records + torch.arange(half, device=dev)
    pre[half:] = src_rank * n_records + src_j
    return pre[perm]


def plant_job_duplicates(fill, d_bytes, n_records, length, dev, rank, world, seed=42, seed_dup=43, seed_shuffle=44, chunk=500_000):
    """In place, on a shard already filled with records [rank*n, (rank+1)*n) of the seed stream: records [n/2, n) become
    rotated / reverse-complemented copies of base records of ANY rank, then the shard is shuffled.
    fill(seed, first_base, n_bases) -> uint8 tensor on `dev` with those bases of the stream (the ctx's generator on a GPU,
    the checker's on the CPU tests)."""
    half = n_records // 2
    src_rank, src_j, k, flip, perm = job_draws(n_records, length, world, rank, dev, seed_dup, seed_shuffle)
    lut = complement_lut(dev)
    view = d_bytes[:n_records * length].view(n_records, length)
    col = torch.arange(length, device=dev)
    for q in range(world):
        mine = torch.nonzero(src_rank == q).flatten()
        if mine.numel() == 0:
            continue
        base = fill(seed, q * n_records * length, half * length)[:half * length].view(half, length)     # rank q's base records
        for c0 in range(0, mine.numel(), chunk):
            c = mine[c0:c0 + chunk]
            rows = torch.gather(base[src_j[c]], 1, (col.unsqueeze(0) + k[c].unsqueeze(1)) % length)
            f = flip[c]
            rows[f] = lut[rows[f].flip(1).long()]
            view[half + c] = rows
        del base
    for c0 in range(0, length, 100):
        view[:, c0:c0 + 100] = view[perm, c0:c0 + 100]


def job_first_seen(n_records, length, world, rank, dev, seed_dup=43, seed_shuffle=44):
    """Expected first-seen global index of every record of rank's shard over the WHOLE job (global index of position p
    of rank q = q*n + p), from the planting decisions alone; also the job's number of distinct records."""
    keys = [job_keys(n_records, length, world, q, dev, seed_dup, seed_shuffle) for q in range(world)]
    allk = torch.cat(keys)
    gidx = torch.arange(world * n_records, dtype=torch.int64, device=dev)
    first = torch.full((world * n_records,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
    first.scatter_reduce_(0, allk, gidx, "amin")
    distinct = int((first != torch.iinfo(torch.int64).max).sum())
    return first[keys[rank]], distinct, keys[rank]


def job_check(fs, keep, n_records, length, world, rank, dev, seed_dup=43, seed_shuffle=44):
    """A shard's first-seen indices and keep flags against job_first_seen(): (mismatching records, records of this shard whose
    first-seen record lives on another rank, distinct records of the job, the shard's keys).  bench.py exits non-zero on a
    mismatch; tests/test_uniq_sharded.py runs the same check under gloo, with broken exchanges as negative controls."""
    exp_fs, distinct, keys = job_first_seen(n_records, length, world, rank, dev, seed_dup, seed_shuffle)
    own = rank * n_records + torch.arange(n_records, dtype=torch.int64, device=dev)
    wrong = int((fs.view(torch.int64) != exp_fs).sum()) + int((keep.bool() != (exp_fs == own)).sum())
    cross = int(((exp_fs // n_records) != rank).sum())
    return wrong, cross, distinct, keys
