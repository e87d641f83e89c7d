"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the
CPU oracle on the same inputs, against the committed golden fixtures, and -- at BASELINE.json's full
sizes -- through size-independent properties (idempotence, rotation / strand invariance, hash agreement)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    import circkit_amd
    c = circkit_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    oracle.lib()
    return oracle


def _check(ctx, O, seqs, hashes=True):
    from tests import seqsets
    data, offs = seqsets.pack(seqs)
    got = ctx.canonicalize_batch(data, offs, want_bytes=True, want_index=True, want_strand=True, want_xxh3=hashes)
    exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
    if not np.array_equal(got["bytes"], exp):
        for i, s in enumerate(seqs):
            a, b = int(offs[i]), int(offs[i + 1])
            assert got["bytes"][a:b].tobytes() == exp[a:b].tobytes(), (i, len(s), s[:80])
    if hashes:
        assert np.array_equal(got["xxh3"], exp_h)
    for i, s in enumerate(seqs):
        _, st, idx = seqsets.expected(O, s)
        assert int(got["strand"][i]) == st, (i, s[:80])
        if s:
            assert int(got["index"][i]) == idx, (i, s[:80])


def test_reference_known_answers_single_record_api(ctx):
    ka = json.load(open(os.path.join(GOLDEN, "ref_known_answers.json")))
    for v in ka["lmsr_index"]:
        assert ctx.lmsr_index(v["in"].encode()) == v["out"]
    for v in ka["lmsr"]:
        assert ctx.lmsr(v["in"].encode()) == v["out"].encode()
    for v in ka["canonicalize"]:
        assert ctx.canonicalize(v["in"].encode()) == v["out"].encode()
    for v in ka["lmsr_idempotent"]:
        a = ctx.lmsr(v["in"].encode())
        assert ctx.lmsr(a) == a
    for v in ka["canonicalize_equal_pairs"]:
        assert ctx.lmsr(v["a"].encode()) == ctx.lmsr(v["b"].encode())
        assert ctx.canonicalize(v["a"].encode()) == ctx.canonicalize(v["b"].encode())
    assert ctx.lmsr_index(b"") == 0 and ctx.lmsr(b"") == b"" and ctx.canonicalize(b"") == b""


def test_single_record_api_rejects_non_ascii(ctx):
    import circkit_amd
    with pytest.raises(circkit_amd.CirckitError) as e:
        ctx.canonicalize(b"AC\xffGT")
    assert e.value.code == -6


def test_xxh3_golden_vectors(ctx):
    vecs = json.load(open(os.path.join(GOLDEN, "xxh3_vectors.json")))["vectors"]
    for v in vecs:
        b = v["in"].encode() if "in" in v else v["in_latin1"].encode("latin-1")
        assert "%016x" % ctx.xxh3_64(b) == v["xxh3_64"], len(b)


def test_reference_fixture_files(ctx, O):
    """tests/canon_uniq.rs:33-89: canonicalize / uniq --canonicalize on the reference's in.fasta files."""
    import circkit_amd
    for d in ("simple", "multiple_sequences", "multiple_sequences_split_lines", "rna_input", "repeated",
              "compressed_input", "compressed_output"):
        recs = O.read_fasta(open(os.path.join(GOLDEN, "ref_examples", d, "in.fasta"), "rb").read())
        want = {O.record_id(h): s.replace(b"\n", b"") for h, s in
                O.read_fasta(open(os.path.join(GOLDEN, "ref_examples", d, "out.fasta"), "rb").read())}
        got, seen = {}, set()
        for head, raw in recs:
            canon = ctx.canonicalize(circkit_amd.normalize(raw)[0])
            if d != "repeated":
                got[O.record_id(head)] = canon
            h = ctx.xxh3_64(canon)
            if d == "repeated" and h not in seen:
                got[O.record_id(head)] = canon
            seen.add(h)
        assert got == want, d


def test_adversarial_set(ctx, O):
    from tests import seqsets
    _check(ctx, O, seqsets.adversarial())


def test_random_sets_all_alphabets(ctx, O):
    from tests import seqsets
    _check(ctx, O, seqsets.random_mixed(41, 4000, 1000, 1000))
    _check(ctx, O, seqsets.random_mixed(42, 3000, 1, 2500))
    _check(ctx, O, seqsets.random_mixed(43, 1500, 1, 2500, b"ACGTN"))
    _check(ctx, O, seqsets.random_mixed(44, 1500, 1, 1500, b"-ACGNT"))
    _check(ctx, O, seqsets.random_mixed(45, 500, 1, 600, bytes(range(0x21, 0x7F))))
    _check(ctx, O, seqsets.random_mixed(46, 2000, 1, 400, b"AC"))
    _check(ctx, O, seqsets.random_mixed(47, 1000, 1, 300, b"A"))


def test_arbitrary_byte_values(ctx, O):
    """Every byte value (lib API accepts any &[u8]; order = unsigned bytes): byte-wide path, no crash."""
    import random
    rng = random.Random(77)
    seqs = [bytes(rng.randrange(256) for _ in range(rng.choice([7, 100, 1000, 1500]))) for _ in range(600)]
    seqs += [bytes([0]) * 1000, bytes([255]) * 999, bytes([0, 255] * 500)]
    _check(ctx, O, seqs)


def test_realistic_fixture_records(ctx, O):
    """676 real records (30..1342 nt, some with N) from the reference's nim_cated fixture."""
    recs = O.read_fasta(open(os.path.join(GOLDEN, "ref_examples", "nim_cated", "realistic_input.fasta"), "rb").read())
    seqs = [O.normalize(s)[0] for _, s in recs]
    assert len(seqs) == 676
    _check(ctx, O, seqs)


def test_lds_tiers_long_records(ctx, O):
    """Records that overflow tier A's 4 KiB slice go to the 40 KiB and 160 KiB tiers (config 4's 20 kb tail)."""
    from tests import seqsets
    seqs = seqsets.random_mixed(48, 12, 6000, 20000) + seqsets.random_mixed(49, 4, 3000, 12000, b"ACGTN") + \
        seqsets.random_mixed(50, 3, 60000, 120000) + [b"ACGT" * 30000, b"A" * 100000] + \
        seqsets.random_mixed(51, 40, 100, 1000)
    _check(ctx, O, seqs)


def test_records_beyond_the_lds_tiers(ctx, O):
    """Records no LDS tier can hold (2-bit beyond ~640 kb -- ~420 kb when the minimal key ties and the candidate bitmask
    is needed --, 4-bit beyond ~100 kb, arbitrary bytes beyond ~70 kb) are taken by the batch's last kernel in global
    scratch, same code: host API, all outputs; mixed into a batch of ordinary records."""
    from tests import seqsets
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    big = [acgt[rng.integers(0, 4, 2_000_000)].tobytes(),                # 2 Mb, the record that used to be refused
           acgt[rng.integers(0, 4, 300_001)].tobytes(),
           np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, 200_000)].tobytes(),
           rng.integers(1, 128, 100_000).astype(np.uint8).tobytes(),      # arbitrary ASCII: byte mode
           (acgt[rng.integers(0, 4, 977)].tobytes() * 400)[:390_000],     # long tandem repeat: the duel path, period 977
           b"A" * 280_000,
           (acgt[rng.integers(0, 4, 1013)].tobytes() * 600)[:600_000]]    # its strand fits tier D, strand + bitmask does not: moves on to the scratch
    seqs = seqsets.random_mixed(81, 30, 48, 1008) + big[:3] + seqsets.random_mixed(82, 10, 2000, 9000) + big[3:] + [b"ACGT"]
    _check(ctx, O, seqs)


def test_team_mode_records_of_20_to_80_kb(ctx, O):
    """Records too long for one wave's 5 KiB slice of tier A (2-bit: 20.4 kb) and short enough for the four slices of a
    workgroup together (81 kb) are canonicalized by the workgroup's four waves as a team -- in canon_kernel<4> (all
    outputs, host API) and in canon_mixed_kernel (bytes only, mode decided on the device).  Lengths around the row and
    slice limits; a tandem repeat, a reverse-complement palindrome and records with N take the tiers behind."""
    import torch
    from tests import seqsets
    rng = np.random.default_rng(21)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    lens = [20417, 20448, 20449, 24575, 24576, 24577, 32768, 40000, 49151, 49152, 65536, 81855, 81856, 81857, 81888, 81889, 90000]
    seqs = [acgt[rng.integers(0, 4, L)].tobytes() for L in lens]
    half = acgt[rng.integers(0, 4, 15000)].tobytes()
    withn = bytearray(acgt[rng.integers(0, 4, 50000)].tobytes()); withn[33333] = ord("N")
    seqs += [(acgt[rng.integers(0, 4, 977)].tobytes() * 40)[:30000], half + half.translate(comp)[::-1], bytes(withn)]
    seqs = seqsets.random_mixed(91, 40, 200, 20000) + seqs + seqsets.random_mixed(92, 300, 48, 1008)
    _check(ctx, O, seqs)
    data, offs = seqsets.pack(seqs)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_bytes = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    exp, _ = O.canonicalize_batch(data, offs, True, False, threads=8)
    for _ in range(2):                                       # the second batch runs with the first one's launch hints
        d_out = torch.zeros_like(d_bytes)
        ctx.canonicalize_batch_device(d_bytes, d_off, len(seqs), out_bytes=d_out)
        assert ctx.batch_status() == 0
        assert np.array_equal(d_out.cpu().numpy(), exp)
    ctx.use_own_stream()


def test_four_bit_team_long_records_with_gaps_and_n_in_the_winning_window(ctx, O):
    """What the 2-bit modes refuse and one wave's slice cannot hold in the 4-bit mode (beyond ~8 kb in stage A): gaps, an N inside
    the winning window behind a prefix that other windows share.  The workgroup's waves take such a record as a team in the
    4-bit mode (canon_core.h canon_record_team<4>: stage A up to ~41 kb, stage C ~80 kb, the team stage beyond) when the tiers
    have real work -- the batch before said so -- and wave 0 alone otherwise: the same batch three times, all outputs through
    the host API first."""
    import torch
    from tests import seqsets
    rng = np.random.default_rng(2204)
    comp = bytes.maketrans(b"ACGTN-", b"TGCAN-")
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = []
    for L in [8200, 8704, 9000, 12287, 12288, 12289, 16384, 20000, 30000, 40900, 40928, 40929, 41000, 60000, 79000, 81000, 120000, 200000]:
        s = bytearray(acgt[rng.integers(0, 4, L)].tobytes())
        for p in rng.integers(0, L, size=L // 300):
            s[int(p)] = ord("N")
        s[int(rng.integers(0, L))] = ord("-")
        seqs.append(bytes(s))
    for case in range(60):                                       # near-ties of the prefix rule (tests/test_emu_kernel.py), 9..40 kb
        L = int(rng.integers(9000, 40000))
        bg = bytearray(np.frombuffer(b"CGT", dtype=np.uint8)[rng.choice(3, size=L, p=[0.2, 0.4, 0.4])].tobytes())
        run = int(rng.integers(9, 14))
        for sp in sorted(rng.choice(np.arange(20, L - 40, 30), size=int(rng.integers(2, 5)), replace=False)):
            motif = b"A" * run + bytes(np.frombuffer(b"NGTCN", dtype=np.uint8)[rng.integers(0, 5, 1)]) + acgt[rng.integers(0, 4, 6)].tobytes()
            if rng.random() < 0.5:
                motif = motif.translate(comp)[::-1]
            bg[int(sp):int(sp) + len(motif)] = motif
        seqs.append(bytes(bg))
    seqs = seqsets.random_mixed(93, 20, 200, 20000) + seqs + seqsets.random_mixed(94, 100, 48, 1008)
    _check(ctx, O, seqs)
    data, offs = seqsets.pack(seqs)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_bytes = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
    for k in range(3):
        d_out = torch.zeros_like(d_bytes)
        d_hash = torch.zeros(len(seqs), dtype=torch.int64, device=dev)
        ctx.canonicalize_batch_device(d_bytes, d_off, len(seqs), out_bytes=d_out, out_xxh3=d_hash if k == 2 else None)
        assert ctx.batch_status() == 0
        assert np.array_equal(d_out.cpu().numpy(), exp)
        if k == 2:
            assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64), exp_h)
    ctx.use_own_stream()


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 255, 257, 4095, 4096, 4097, 9000])
def test_small_batch_segment_geometry(ctx, O, n):
    """Small batches get more list segments than the streaming kernel has workgroups with work, and the stages that walk
    ALL records of a mode-3 batch (canon_mixed_kernel; the rescue pass for batches that also want hashes) deal them out
    all_seg_cap at a time: record counts around the limits (16 per streaming workgroup, 4096 segments), device API, bytes
    only and bytes + XXH3."""
    import torch
    from tests import seqsets
    seqs = seqsets.random_mixed(300 + n, n, 48, 6000)            # more than one in eight beyond 2032 bases: mode 3
    data, offs = seqsets.pack(seqs)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_bytes = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
    d_out = torch.zeros_like(d_bytes)
    ctx.canonicalize_batch_device(d_bytes, d_off, n, out_bytes=d_out)
    assert ctx.batch_status() == 0
    assert np.array_equal(d_out.cpu().numpy(), exp)
    d_out.zero_()
    d_hash = torch.zeros(n, dtype=torch.int64, device=dev)
    ctx.canonicalize_batch_device(d_bytes, d_off, n, out_bytes=d_out, out_xxh3=d_hash)
    assert ctx.batch_status() == 0
    assert np.array_equal(d_out.cpu().numpy(), exp)
    assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64), exp_h)
    ctx.use_own_stream()


def test_device_api_finishes_long_records_on_the_device(ctx, O):
    """Device API: the batch call enqueues everything, records beyond the LDS tiers included (global-scratch stages);
    any synchronisation with the stream is enough."""
    import torch
    from tests import seqsets
    rng = np.random.default_rng(4)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = seqsets.random_mixed(83, 20, 900, 1008) + [acgt[rng.integers(0, 4, 700_000)].tobytes()] + seqsets.random_mixed(84, 5, 100, 300)
    data, offs = seqsets.pack(seqs)
    n = len(seqs)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_bytes = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_out = torch.zeros_like(d_bytes)
    d_hash = torch.zeros(n, dtype=torch.int64, device=dev)
    ctx.canonicalize_batch_device(d_bytes, d_off, n, out_bytes=d_out, out_xxh3=d_hash)
    torch.cuda.synchronize()                         # the caller's own synchronisation, not the library's
    exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
    assert np.array_equal(d_out.cpu().numpy(), exp)
    assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64), exp_h)
    assert ctx.batch_status() == 0
    # hash-only batch (uniq without --canonicalize): canonical bytes live in the ctx's scratch
    d_hash.zero_()
    ctx.canonicalize_batch_device(d_bytes, d_off, n, out_xxh3=d_hash)
    ctx.synchronize()
    assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64), exp_h)
    ctx.use_own_stream()


def test_back_to_back_device_batches_with_long_records(ctx, O):
    """Two device batches enqueued one behind the other with NO synchronisation in between, the first holding records
    beyond the LDS tiers: both complete (an earlier design finished such records from the host at the next
    synchronisation point and lost them when a second batch was enqueued first)."""
    import torch
    from tests import seqsets
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    first = seqsets.random_mixed(85, 40, 500, 1008) + [acgt[rng.integers(0, 4, 400_000)].tobytes(),
                                                       np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, 150_000)].tobytes()]
    second = seqsets.random_mixed(86, 300, 48, 1008)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    bufs = []
    for seqs in (first, second):
        data, offs = seqsets.pack(seqs)
        d_bytes = torch.from_numpy(data).to(dev)
        bufs.append((data, offs, len(seqs), d_bytes, torch.from_numpy(offs.astype(np.int64)).to(dev), torch.zeros_like(d_bytes),
                     torch.zeros(len(seqs), dtype=torch.int64, device=dev)))
    torch.cuda.synchronize()
    for _, _, n, d_bytes, d_off, d_out, d_hash in bufs:
        ctx.canonicalize_batch_device(d_bytes, d_off, n, out_bytes=d_out, out_xxh3=d_hash)
    torch.cuda.synchronize()
    for data, offs, n, _, _, d_out, d_hash in bufs:
        exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
        assert np.array_equal(d_out.cpu().numpy(), exp)
        assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64), exp_h)
    assert ctx.batch_status() == 0
    ctx.use_own_stream()


def test_long_record_scratch_limit_is_reported(O):
    """A record beyond the long-record scratch is left untouched and counted by circkit_ctx_batch_status (device API;
    the host API sizes the scratch itself)."""
    import torch
    import circkit_amd
    from tests import seqsets
    c = circkit_amd.Context(0)
    c.set_long_record_scratch(1 << 20)               # 1 MiB: pure-ACGT records up to ~4.19 Mb
    rng = np.random.default_rng(6)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = seqsets.random_mixed(87, 20, 100, 1008) + [acgt[rng.integers(0, 4, 1_000_000)].tobytes(), acgt[rng.integers(0, 4, 4_500_000)].tobytes()]
    data, offs = seqsets.pack(seqs)
    dev = torch.device("cuda", 0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    d_bytes = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_out = torch.zeros_like(d_bytes)
    c.canonicalize_batch_device(d_bytes, d_off, len(seqs), out_bytes=d_out)
    assert c.batch_status() == 1
    exp, _ = O.canonicalize_batch(data, offs, True, False, threads=8)
    cut = int(offs[-2])
    assert np.array_equal(d_out.cpu().numpy()[:cut], exp[:cut])           # everything but the refused record
    got = c.canonicalize_batch(data, offs)                                # host API: grows the scratch
    assert np.array_equal(got["bytes"], exp)
    c.close()


def test_refused_record_in_a_hash_only_batch_is_a_clean_status(O):
    """a hash-only batch (views, no bytes) with a record no stage can take.  The ctx's view array still holds what
    an earlier batch left at that index -- here the rotation (~5.9M, reverse strand) of a 6 Mb record, beyond the length of the
    3 Mb record that is then refused (arbitrary bytes: two stored strands do not fit the 4 MiB scratch) -- and the xxh3 pass
    must not address the payload with it: status 1 (CIRCKIT_ERR_TOO_LONG), no fault, every other record's hash right."""
    import torch
    import circkit_amd
    from tests import seqsets
    c = circkit_amd.Context(0)
    c.set_long_record_scratch(4 << 20)               # pure-ACGT records up to ~16 Mb (strand + tie bitmask); byte-alphabet records ~2 Mb
    dev = torch.device("cuda", 0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(61)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    small = seqsets.random_mixed(88, 20, 100, 240)
    # minimal rotation of the reverse strand starts ~5.9 Mb into it: a run of T near the record's start
    big = acgt[rng.integers(0, 4, 6_000_000)].copy()
    big[100_000:100_040] = ord("T")
    odd = np.frombuffer(b"ACGTXYZ", dtype=np.uint8)[rng.integers(0, 7, 3_000_000)]
    for seqs, want in ((small + [big.tobytes()], 0), (small + [odd.tobytes()], 1)):
        data, offs = seqsets.pack(seqs)
        d_bytes = torch.from_numpy(data).to(dev)
        d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
        d_hash = torch.zeros(len(seqs), dtype=torch.int64, device=dev)
        c.canonicalize_batch_device(d_bytes, d_off, len(seqs), out_xxh3=d_hash)
        assert c.batch_status() == want
        _, exp_h = O.canonicalize_batch(data[:int(offs[-2])], offs[:-1], False, True, threads=8)
        assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64)[:-1], exp_h)
    c.close()


def test_zipf_mixed_lengths_config4_shape(ctx, O):
    """BASELINE config 4 shape at a size the oracle finishes in seconds: 3000 records, L ~ 1/L on [200, 20000]."""
    rng = np.random.default_rng(45)
    u = rng.random(3000)
    lens = np.exp(np.log(200) + u * (np.log(20000) - np.log(200))).astype(np.int64)
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    data = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(offs[-1]))]
    got = ctx.canonicalize_batch(data, offs, want_xxh3=True)
    exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
    assert np.array_equal(got["bytes"], exp) and np.array_equal(got["xxh3"], exp_h)
    # variant with ~1 % N
    data2 = data.copy()
    data2[rng.random(len(data2)) < 0.01] = ord("N")
    got = ctx.canonicalize_batch(data2, offs)
    exp, _ = O.canonicalize_batch(data2, offs, True, False, threads=8)
    assert np.array_equal(got["bytes"], exp)


def test_device_generator_matches_host_generator(ctx, O):
    import torch
    n = 1_000_003
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.synth_fill_device(42, 12345, n, d)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), O.synth_fill(42, 12345, n))
    ctx.use_own_stream()


def test_full_size_properties_10m_x_1kb(ctx, O):
    """BASELINE config 2 at full size (10M x 1 kb, device resident): idempotence, strand invariance and
    rotation invariance of the canonical form, plus byte parity with the oracle on a 20k-record slice."""
    import torch
    N, L = 10_000_000, 1000
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    x = torch.empty(N * L + 64, dtype=torch.uint8, device=dev)
    off = torch.empty(N + 1, dtype=torch.int64, device=dev)
    ctx.synth_fill_device(42, 0, N * L, x)
    ctx.fixed_offsets_device(0, L, N, off)
    c1 = torch.empty_like(x)
    strand = torch.empty(N, dtype=torch.uint8, device=dev)
    idx = torch.empty(N, dtype=torch.int32, device=dev)
    ctx.canonicalize_batch_device(x, off, N, out_bytes=c1, out_index=idx, out_strand=strand)
    assert ctx.batch_status() == 0
    # oracle parity on a slice
    S = 20000
    exp, exp_h = O.canonicalize_batch(x[:S * L].cpu().numpy(), np.arange(S + 1, dtype=np.uint64) * np.uint64(L), True, True, 8)
    assert np.array_equal(c1[:S * L].cpu().numpy(), exp)
    # the benchmarked build -- canonical bytes only: canon_stream_kernel<StreamCfg<16,2,1,1>,false,false> -- over all
    # 10M records against the build with every output, and with it against the oracle slice
    c0 = torch.full_like(x, 0x3F)
    ctx.canonicalize_batch_device(x, off, N, out_bytes=c0)
    assert ctx.batch_status() == 0
    assert ctx.last_batch_mode() == 1
    assert torch.equal(c0[:N * L], c1[:N * L])
    assert bool((c0[N * L:] == 0x3F).all())
    # ...and the uniq build (bytes + fused XXH3): same bytes, hashes = oracle's on the slice
    hs = torch.empty(N, dtype=torch.int64, device=dev)
    c0.fill_(0x3F)
    ctx.canonicalize_batch_device(x, off, N, out_bytes=c0, out_xxh3=hs)
    torch.cuda.synchronize()
    assert torch.equal(c0[:N * L], c1[:N * L])
    assert np.array_equal(hs[:S].cpu().numpy().astype(np.uint64), exp_h)
    del c0, hs
    # idempotence: canonical input -> identical output, rotation index 0 whenever the forward strand is returned
    c2 = torch.empty_like(x)
    s2 = torch.empty_like(strand)
    i2 = torch.empty_like(idx)
    ctx.canonicalize_batch_device(c1, off, N, out_bytes=c2, out_index=i2, out_strand=s2)
    torch.cuda.synchronize()
    assert torch.equal(c1[:N * L], c2[:N * L])
    assert int((i2[s2 == 0] != 0).sum().item()) == 0
    del c2, s2, i2
    # strand + rotation invariance: reverse-complement every record and rotate it by 137
    lut = torch.arange(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        lut[a] = b
    y = torch.empty_like(x)
    chunk = 1_000_000
    for s in range(0, N, chunk):
        v = x[s * L:(s + chunk) * L].view(chunk, L)
        y[s * L:(s + chunk) * L] = torch.roll(lut[v.flip(1).long()], shifts=137, dims=1).reshape(-1)
    c3 = torch.empty_like(x)
    s3 = torch.empty_like(strand)
    ctx.canonicalize_batch_device(y, off, N, out_bytes=c3, out_strand=s3)
    torch.cuda.synchronize()
    assert torch.equal(c1[:N * L], c3[:N * L])
    # a strict strand decision flips with the input strand (ties = reverse palindromes keep strand 1)
    assert int(((strand + s3) != 1).sum().item()) <= 5
    ctx.use_own_stream()


def _make_dups(rng, base_seqs, n_dups):
    from tests import seqsets
    out = list(base_seqs)
    for _ in range(n_dups):
        s = base_seqs[rng.randrange(len(base_seqs))]
        k = rng.randrange(len(s)) if s else 0
        d = s[k:] + s[:k]
        if rng.random() < 0.5:
            d = seqsets.revcomp_acgt(d)
        out.append(d)
    rng.shuffle(out)
    return out


def test_uniq_config3_shape(ctx, O):
    """BASELINE config 3 at a size the oracle finishes in seconds: base records + ~50 % rotational / strand
    duplicates, shuffled.  Hash-only device batch -> device hash table -> first-seen == src/uniq.rs semantics."""
    import random
    import torch
    from circkit_amd import uniq
    from tests import seqsets
    rng = random.Random(43)
    base = seqsets.random_mixed(43, 3000, 1000, 1000) + seqsets.random_mixed(44, 500, 30, 3000) + [b"", b"A", b"ACGTN" * 50]
    seqs = _make_dups(rng, base, 3500)
    data, offs = seqsets.pack(seqs)
    n = len(seqs)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_bytes = torch.zeros(len(data) + 64, dtype=torch.uint8, device=dev)
    d_bytes[:len(data)] = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_hash = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.canonicalize_batch_device(d_bytes, d_off, n, out_xxh3=d_hash)          # hash-only mode
    assert ctx.batch_status() == 0
    _, exp_h = O.canonicalize_batch(data, offs, False, True, threads=8)
    assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64), exp_h)
    fs, keep = uniq.first_seen(uniq.DeviceTable(ctx), d_hash, base_index=0)
    exp_fs = O.uniq_first_seen(exp_h)
    assert np.array_equal(fs.cpu().numpy().astype(np.uint64), exp_fs)
    assert int(keep.sum().item()) == len({O.canonicalize(s) for s in base})
    # streaming batches with a running base index fold into the same table
    ctx.uniq_reset(n)
    half = n // 2
    ctx.uniq_insert_device(d_hash[:half].contiguous(), half, 0)
    ctx.uniq_insert_device(d_hash[half:].contiguous(), n - half, half)
    out = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.uniq_lookup_device(d_hash, n, out)
    assert np.array_equal(out.cpu().numpy().astype(np.uint64), exp_fs)
    ctx.use_own_stream()


def test_uniq_repeated_fixture(ctx, O):
    """tests/examples/repeated: five records, one survivor (tests/canon_uniq.rs:45)."""
    import circkit_amd
    recs = O.read_fasta(open(os.path.join(GOLDEN, "ref_examples", "repeated", "in.fasta"), "rb").read())
    seqs = [circkit_amd.normalize(s)[0] for _, s in recs]
    from tests import seqsets
    data, offs = seqsets.pack(seqs)
    got = ctx.canonicalize_batch(data, offs, want_xxh3=True)
    assert len(set(got["xxh3"].tolist())) == 1
    assert got["bytes"][:8].tobytes() == b"AAAAAAAT"


def test_lmsr_and_xxh3_device_batches(ctx, O):
    """circkit_lmsr_batch_device (forward strand only, lib/src/canonicalize.rs:41-47) and circkit_xxh3_batch_device."""
    import torch
    from tests import seqsets
    seqs = seqsets.random_mixed(61, 300, 1, 1500) + seqsets.random_mixed(62, 100, 1, 600, b"ACGTN") + [b"banana", b"TAA", b""]
    data, offs = seqsets.pack(seqs)
    n = len(seqs)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_bytes = torch.zeros(len(data) + 64, dtype=torch.uint8, device=dev)
    d_bytes[:len(data)] = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_out = torch.zeros_like(d_bytes)
    d_idx = torch.empty(n, dtype=torch.int32, device=dev)
    d_hash = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.lmsr_batch_device(d_bytes, d_off, n, out_bytes=d_out, out_index=d_idx)
    ctx.xxh3_batch_device(d_out, d_off, n, d_hash)
    torch.cuda.synchronize()
    out, idx, hs = d_out.cpu().numpy(), d_idx.cpu().numpy(), d_hash.cpu().numpy().astype(np.uint64)
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        assert out[a:b].tobytes() == O.lmsr(s), (i, s[:60])
        assert int(hs[i]) == O.xxh3_64(O.lmsr(s))
        if s:
            assert int(idx[i]) == O.lmsr_index(s)
    ctx.use_own_stream()


def test_pinned_host_buffers(ctx, O):
    """circkit_host_alloc: page-locked batch buffers through the host API."""
    import ctypes
    import circkit_amd
    from tests import seqsets
    lib = circkit_amd.load_library()
    seqs = seqsets.random_mixed(63, 200, 900, 1100)
    data, offs = seqsets.pack(seqs)
    n, total = len(seqs), len(data)
    pin_in, pin_out = lib.circkit_host_alloc(total + 64), lib.circkit_host_alloc(total + 64)
    assert pin_in and pin_out
    ctypes.memmove(pin_in, data.ctypes.data, total)
    rc = lib.circkit_canonicalize_batch(ctx._h, pin_in, offs.ctypes.data, n, pin_out, None, None, None)
    assert rc == 0
    got = np.ctypeslib.as_array(ctypes.cast(pin_out, ctypes.POINTER(ctypes.c_uint8)), shape=(total,)).copy()
    exp, _ = O.canonicalize_batch(data, offs, True, False, threads=4)
    assert np.array_equal(got, exp)
    lib.circkit_host_free(pin_in)
    lib.circkit_host_free(pin_out)


def test_device_buffers_need_no_alignment_padding_or_zero_base(ctx, O):
    """The streaming kernel stages 16-byte-aligned chunks of whole record groups; the device API must still accept a
    payload at any byte alignment, with no slack behind the last record, and offsets that do not start at 0 -- and
    must never touch a byte outside [offsets[0], offsets[n]) (canaries on both sides, input and output)."""
    import torch
    from tests import seqsets
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    seqs = seqsets.random_mixed(71, 700, 48, 1008) + seqsets.random_mixed(72, 60, 1, 60) + [b"ACGTN" * 50, b"A" * 700]
    data, offs = seqsets.pack(seqs)
    n, total = len(seqs), len(data)
    exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
    for shift, lead in ((0, 0), (1, 0), (7, 5), (15, 33), (8, 1000)):
        # layout of the device allocation: [shift bytes][lead bytes of canary][payload][canary]
        buf = torch.full((shift + lead + total + 32,), 0x4E, dtype=torch.uint8, device=dev)      # 'N' canaries
        view = buf[shift:]
        view[lead:lead + total] = torch.from_numpy(data).to(dev)
        d_off = torch.from_numpy((offs + np.uint64(lead)).astype(np.int64)).to(dev)
        out = torch.full_like(buf, 0x3F)
        d_hash = torch.empty(n, dtype=torch.int64, device=dev)
        for want_hash in (False, True):
            out.fill_(0x3F)
            ctx.canonicalize_batch_device(view, d_off, n, out_bytes=out[shift:], out_xxh3=d_hash if want_hash else None)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            lo = shift + lead
            assert np.array_equal(got[lo:lo + total], exp), (shift, lead, want_hash)
            assert (got[:lo] == 0x3F).all() and (got[lo + total:] == 0x3F).all(), "wrote outside the batch"
            if want_hash:
                assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64), exp_h)
        assert (buf[:shift + lead] == 0x4E).all() and (buf[shift + lead + total:] == 0x4E).all()
    ctx.use_own_stream()


def test_randomized_rare_paths_large():
    """tools/gpu_fuzz.py: 150k generated records aimed at the streaming kernel's rare paths (planted duplicate minimal
    16-mers, tandem repeats, reverse-complement palindromes, lengths around every limit, N / '-'), all three builds of
    the kernel, bytes + XXH3 + index + strand against the oracle.  Run as its own process (it owns its context)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_fuzz.py"), "77", "150000"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout


def test_two_contexts_in_two_threads(O):
    """include/circkit.h: one host thread per ctx, contexts independent -- two threads, each with its own ctx and
    stream, running different batches at the same time, get their own results."""
    import threading
    import circkit_amd
    from tests import seqsets
    results, errors = {}, []

    def work(tag, seed):
        try:
            c = circkit_amd.Context(0)
            for rep in range(4):
                seqs = seqsets.random_mixed(seed + rep, 3000, 1, 1400) + seqsets.random_mixed(seed + 50 + rep, 200, 48, 1008, b"ACGTN")
                data, offs = seqsets.pack(seqs)
                got = c.canonicalize_batch(data, offs, want_bytes=True, want_xxh3=True)
                exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=2)
                assert np.array_equal(got["bytes"], exp) and np.array_equal(got["xxh3"], exp_h), (tag, rep)
            c.close()
            results[tag] = True
        except Exception as e:      # noqa: BLE001 -- reported below, on the main thread
            errors.append((tag, repr(e)))

    ts = [threading.Thread(target=work, args=("a", 500)), threading.Thread(target=work, args=("b", 700))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert results == {"a": True, "b": True}


def test_uniq_insert_pairs_device(ctx, O):
    """circkit_uniq_insert_pairs_device: explicit global indices (what a rank folds in after the hash-range
    all-to-all); same first-seen answers as inserting the keys in shard order."""
    import torch
    rng = np.random.default_rng(12)
    n = 200_000
    h = (rng.integers(0, 50_000, size=n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    h[rng.integers(0, n, 50)] = np.uint64(0xFFFFFFFFFFFFFFFF)        # the table's empty marker is a legal hash value
    idx = rng.permutation(n).astype(np.int64) + 1000                  # arbitrary, non-contiguous global indices
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_h = torch.from_numpy(h.astype(np.int64)).to(dev)
    d_i = torch.from_numpy(idx).to(dev)
    ctx.uniq_reset(n)
    half = n // 2
    ctx.uniq_insert_pairs_device(d_h[:half], d_i[:half], half)
    ctx.uniq_insert_pairs_device(d_h[half:], d_i[half:], n - half)
    out = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.uniq_lookup_device(d_h, n, out)
    torch.cuda.synchronize()
    best = {}
    for hh, ii in zip(h.tolist(), idx.tolist()):
        if hh not in best or ii < best[hh]:
            best[hh] = ii
    assert out.cpu().numpy().tolist() == [best[x] for x in h.tolist()]
    ctx.use_own_stream()


def test_uniq_resolve_device_answers_and_keeps_its_table_private(ctx, O):
    """circkit_uniq_resolve_device (one shard, one call): first-seen indices and keep flags against the oracle, with a
    base index, the empty-marker hash among the keys and every key several times; its table holds shard-local values in
    a layout of its own, so insert / lookup are refused until the next reset."""
    import torch
    import circkit_amd
    rng = np.random.default_rng(13)
    n, base = 300_000, 5_000_000_000
    h = (rng.integers(0, 60_000, size=n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    h[rng.integers(0, n, 40)] = np.uint64(0xFFFFFFFFFFFFFFFF)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_h = torch.from_numpy(h.astype(np.int64)).to(dev)
    fs = torch.empty(n, dtype=torch.int64, device=dev)
    keep = torch.empty(n, dtype=torch.bool, device=dev)
    for _ in range(2):                                              # twice: the second call starts from a used table
        ctx.uniq_resolve_device(d_h, n, base, fs, keep)
    ctx.uniq_status()
    exp = O.uniq_first_seen(h)
    assert np.array_equal(fs.cpu().numpy().astype(np.uint64), exp + np.uint64(base))
    assert np.array_equal(keep.cpu().numpy(), exp == np.arange(n, dtype=np.uint64))
    with pytest.raises(circkit_amd.CirckitError):
        ctx.uniq_lookup_device(d_h, n, fs)
    with pytest.raises(circkit_amd.CirckitError):
        ctx.uniq_insert_device(d_h, n, 0)
    ctx.uniq_reset(n)
    ctx.uniq_insert_device(d_h, n, base)
    ctx.uniq_lookup_device(d_h, n, fs)
    torch.cuda.synchronize()
    assert np.array_equal(fs.cpu().numpy().astype(np.uint64), exp + np.uint64(base))
    ctx.use_own_stream()


@pytest.mark.parametrize("world", [1, 3, 8, 64])
def test_uniq_exchange_kernels(ctx, O, world):
    """The device steps of the multi-GPU exchange, without the collectives: circkit_uniq_partition_device groups the
    (hash, global index) rows by owner rank -- the same owner as circkit_amd/uniq.py's _owner --, counts them and tells every
    record its row; rows folded and answered in row order (insert_rows / lookup_rows) and gathered back through the
    slots give the oracle's first-seen."""
    import torch
    from circkit_amd import uniq
    rng = np.random.default_rng(14 + world)
    n, base = 500_000, 7_000_000_000
    h = rng.integers(0, 100_000, size=n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    h[rng.integers(0, n, 30)] = np.uint64(0xFFFFFFFFFFFFFFFF)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_h = torch.from_numpy(h.astype(np.int64)).to(dev)
    table = uniq.DeviceTable(ctx)
    rows, counts, slot = table.partition(d_h, base, world)
    torch.cuda.synchronize()
    owner = uniq._owner(d_h, world)
    assert torch.equal(counts, torch.bincount(owner, minlength=world))
    r = rows.cpu().numpy(); sl = slot.cpu().numpy().astype(np.int64)
    assert len(np.unique(sl)) == n                                          # every record its own row
    assert np.array_equal(r[sl, 0].astype(np.uint64), h) and np.array_equal(r[sl, 1], base + np.arange(n))
    row_owner = uniq._owner(rows[:, 0].contiguous(), world).cpu().numpy()
    assert np.all(np.diff(row_owner) >= 0)                                  # owners in rank order
    table.reset(n)
    table.insert_rows(rows)
    answers = table.lookup_rows(rows)
    fs, keep = table.gather(answers, slot, base)
    table.check()
    exp = O.uniq_first_seen(h)
    assert np.array_equal(fs.cpu().numpy().astype(np.uint64), exp + np.uint64(base))
    assert np.array_equal(keep.cpu().numpy(), exp == np.arange(n, dtype=np.uint64))
    ctx.use_own_stream()


def test_records_of_1009_to_2032_bases_two_words_per_lane(ctx, O):
    """The ROWS == 2 build of the streaming kernel (records up to 2032 bases, two packed words per lane): chosen from the
    batch's lengths -- by the host for host buffers, on the device for device buffers (both builds are launched, the
    one the previous batch used with the full-size grid)."""
    import torch
    from tests import seqsets
    seqs = seqsets.random_mixed(101, 600, 1009, 2032) + seqsets.random_mixed(102, 30, 1009, 2032, b"ACGTN") + \
        seqsets.random_mixed(103, 100, 48, 1008) + seqsets.random_mixed(104, 30, 2033, 2600) + \
        [b"ACGT" * 400, b"A" * 1500, seqsets.random_mixed(105, 1, 1164, 1164)[0] * 1]          # (one record in 25 with an N: below the alphabet rule's one in 16)
    _check(ctx, O, seqs)                                            # host API, every output
    data, offs = seqsets.pack(seqs)
    n = len(seqs)
    exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_bytes = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_hash = torch.zeros(n, dtype=torch.int64, device=dev)
    for rep in range(3):                                            # 1st: small-grid fallback runs it; 2nd / 3rd: full grid
        d_out = torch.zeros_like(d_bytes)
        ctx.canonicalize_batch_device(d_bytes, d_off, n, out_bytes=d_out, out_xxh3=d_hash if rep == 2 else None)
        torch.cuda.synchronize()
        # (with the XXH3 a batch of two-word records is the mixed-length kernels': 8-11 % faster than the two-word build's
        # per-record hash)
        assert ctx.last_batch_mode() == (3 if rep == 2 else 2)
        assert np.array_equal(d_out.cpu().numpy(), exp), rep
        if rep == 2:
            assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64), exp_h)
    # the two-word build has no alphabet twin: a batch of such records in which N is common (one record in sixteen or more) is the
    # mixed-length N kernels' (mode 3) -- left to the two-word build its N records went to LDS stage A one by one (6M x 1.5 kb with
    # 1 % N: 11.9 ms; 6.3 through mode 3)
    seqs = seqsets.random_mixed(111, 400, 1009, 2032) + seqsets.random_mixed(112, 300, 1009, 2032, b"ACGTN") + seqsets.random_mixed(113, 60, 48, 1008, b"ACGTN")
    data, offs = seqsets.pack(seqs)
    n = len(seqs)
    exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
    d_bytes = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_hash = torch.zeros(n, dtype=torch.int64, device=dev)
    for rep in range(3):
        d_out = torch.zeros_like(d_bytes)
        ctx.canonicalize_batch_device(d_bytes, d_off, n, out_bytes=d_out, out_xxh3=d_hash if rep else None)
        torch.cuda.synchronize()
        assert ctx.last_batch_mode() == 3 and ctx.batch_status() == 0
        assert np.array_equal(d_out.cpu().numpy(), exp), rep
        if rep:
            assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64), exp_h)
    ctx.use_own_stream()


def test_randomized_rare_paths_long_profile():
    """tools/gpu_fuzz.py with most records in 1009..2032 bases."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_fuzz.py"), "78", "100000", "long"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout


def test_device_api_batch_full_of_long_records(ctx, O):
    """Mode 3 of launch_canon decided on the device: with one record in eight beyond 2032 bases the staged kernel is
    skipped and the rescue pass walks every record."""
    import torch
    from tests import seqsets
    seqs = seqsets.random_mixed(111, 400, 48, 1008) + seqsets.random_mixed(112, 150, 2500, 9000) + \
        seqsets.random_mixed(113, 60, 1009, 2032) + seqsets.random_mixed(114, 40, 1, 47) + seqsets.random_mixed(115, 40, 100, 900, b"ACGTN")
    rng = np.random.default_rng(116)
    seqs = [seqs[i] for i in rng.permutation(len(seqs))]
    data, offs = seqsets.pack(seqs)
    n = len(seqs)
    exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_bytes = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_hash = torch.zeros(n, dtype=torch.int64, device=dev)
    d_idx = torch.zeros(n, dtype=torch.int32, device=dev)
    for rep in range(3):
        d_out = torch.zeros_like(d_bytes)
        ctx.canonicalize_batch_device(d_bytes, d_off, n, out_bytes=d_out, out_xxh3=d_hash if rep else None,
                                      out_index=d_idx if rep == 2 else None)
        torch.cuda.synchronize()
        assert ctx.last_batch_mode() == 3
        assert np.array_equal(d_out.cpu().numpy(), exp), rep
        if rep:
            assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64), exp_h)
    ctx.use_own_stream()


@pytest.mark.parametrize("profile,count", [("nrich", "100000"), ("longn", "20000")])
def test_randomized_n_paths(profile, count):
    """tools/gpu_fuzz.py with N / '-' in most records: "nrich" makes the batch's mode carry MODE_ALPHA (4-bit register
    routine in the streaming kernel and in the rescue pass), "longn" drives the 2-bit-with-N-mask mode of the LDS tiers
    (records of 1..9 kb with up to 30 N) and its fallbacks to the 4-bit mode."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_fuzz.py"), "79", count, profile], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout


@pytest.mark.parametrize("seed", ["20", "21"])
def test_randomized_lean_routine_ties(seed):
    """tools/gpu_fuzz.py "leanties": records of 1..20 kb aimed at what the mixed kernel's lean routine settles itself -- a
    minimal 16-mer owned several times, a palindromic core both strands own, the minimum wrapped around the record's end at
    lengths just above multiples of 1024, whole-record palindromes, tandem repeats; odd seeds sprinkle 1 % N (the N build
    and its exact resolution among sharers).  All three builds of the host API against the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_fuzz.py"), seed, "6000", "leanties"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout


@pytest.mark.parametrize("with_n", [False, True])
def test_low_complexity_records_in_a_mixed_length_batch(ctx, O, with_n):
    """Mode-3 batches (the mixed-length kernels, bytes-only and with the XXH3, device API) full of what the pure builds' 8-symbol
    prefix scan ties on (round 4): homopolymer runs of 8..40 A / T planted one to six times per record, microsatellites
    (AC)n / (AAT)n, a run that wraps around the record's end -- records of 1.1..14 kb.  A record whose minimal prefix has more
    owners than the routine compares is scanned again with 16-symbol keys; whatever is still tied is stage A's.  Bytes and
    hashes against the oracle; with_n: 1 % N (the N builds keep 16-symbol keys)."""
    import torch
    rng = np.random.default_rng(404 + with_n)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = []
    for k in range(600):
        n = int(rng.integers(1100, 14000))
        s = acgt[rng.integers(1, 4, n)].copy() if k % 3 else acgt[rng.integers(0, 4, n)].copy()      # two thirds without A outside the plants
        for _ in range(int(rng.integers(1, 7))):
            run = int(rng.integers(8, 41))
            p = int(rng.integers(0, n))
            kind = k % 5
            for i in range(run):
                s[(p + i) % n] = (b"A"[0] if kind < 3 else b"T"[0]) if kind != 4 else b"AC"[i & 1]
        if k % 11 == 0:
            s[:12] = ord("A"); s[-9:] = ord("A")                            # a run across the record's end
        if k % 13 == 0:
            unit = b"AAT"
            s[100:100 + 60] = np.frombuffer(unit * 20, dtype=np.uint8)
        if with_n:
            s[rng.random(n) < 0.01] = ord("N")
        seqs.append(s.tobytes())
    seqs += [bytes(acgt[rng.integers(0, 4, int(rng.integers(60, 900)))]) for _ in range(150)]
    from tests import seqsets
    data, offs = seqsets.pack(seqs)
    exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_bytes = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    for want_hash in (False, True):
        d_out = torch.zeros_like(d_bytes)
        d_hash = torch.zeros(len(seqs), dtype=torch.int64, device=dev)
        ctx.canonicalize_batch_device(d_bytes, d_off, len(seqs), out_bytes=d_out, out_xxh3=d_hash if want_hash else None)
        assert ctx.batch_status() == 0 and ctx.last_batch_mode() == 3
        got = d_out.cpu().numpy()
        if not np.array_equal(got, exp):
            for i in range(len(seqs)):
                a, b = int(offs[i]), int(offs[i + 1])
                assert got[a:b].tobytes() == exp[a:b].tobytes(), (want_hash, i, len(seqs[i]))
        if want_hash:
            assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64), exp_h)
    ctx.use_own_stream()


def test_batches_of_short_records_bytes_only_pair_build(ctx, O):
    """MODE_SHORT (end of round 4): a batch whose records are mostly <= 800 symbols runs the bytes-only PAIR build of the streaming
    kernel (canon_pair.h: two records per wave) -- decided on the device from the count kernel's samples, guessed from the third
    batch on.  Lengths 48..600 with everything the routine refuses mixed in (shorter, N, gaps, periodic, palindromes), fixed
    lengths 100 / 200 / 300 / 500, and a batch of records near 1 kb behind them (the other build, after a wrong guess)."""
    import torch
    from tests import seqsets
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    batches = [seqsets.random_mixed(901, 20000, 48, 600) + seqsets.random_mixed(902, 300, 1, 47) + seqsets.random_mixed(903, 300, 48, 600, b"ACGTN-") +
               seqsets.adversarial()[:200]]
    for L in (100, 200, 300, 500):
        batches.append(seqsets.random_mixed(910 + L, 6000, L, L))
    batches.append(seqsets.random_mixed(905, 6000, 900, 1008))
    batches.append(batches[0])
    for seqs in batches:
        data, offs = seqsets.pack(seqs)
        exp, _ = O.canonicalize_batch(data, offs, True, False, threads=8)
        d_bytes = torch.from_numpy(np.concatenate([data, np.zeros(64, dtype=np.uint8)])).to(dev)
        d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
        for rep in range(3):
            d_out = torch.zeros_like(d_bytes)
            ctx.canonicalize_batch_device(d_bytes, d_off, len(seqs), out_bytes=d_out)
            torch.cuda.synchronize()
            assert ctx.batch_status() == 0 and ctx.last_batch_mode() == 1
            assert np.array_equal(d_out[:len(data)].cpu().numpy(), exp), (len(seqs), rep)
    ctx.use_own_stream()
