"""Runs the wave-level kernel source (circkit_amd/csrc/canon_core.h) through the CPU fiber emulator and
checks it against the oracle.  This validates the kernel LOGIC on a machine without a GPU; the GPU parity
tests (test_gpu_parity.py, -m gpu) check the gfx950 build itself."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import seqsets
from tests.emu import emu


def check(seqs, **kw):
    data, offs = seqsets.pack(seqs)
    out, idx, strand, _, status, ndef = emu.canonicalize_batch(data, offs, **kw)
    assert status == 0
    deferred = 0
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        exp, est, eidx = seqsets.expected(O, s)
        if strand[i] == 0xFF and len(s):      # deferred to a bigger LDS tier: untouched
            deferred += 1
            continue
        assert out[a:b].tobytes() == exp, (i, len(s), s[:80])
        assert int(strand[i]) == est, (i, s[:80])
        if len(s):
            assert int(idx[i]) == eidx, (i, s[:80])
    assert deferred == ndef
    return ndef


def test_adversarial_set():
    assert check(seqsets.adversarial(), slice_dw=4096) == 0


def test_random_acgt_1kb():
    check(seqsets.random_mixed(21, 150, 1000, 1000))


def test_random_lengths_all_alphabets():
    check(seqsets.random_mixed(22, 300, 1, 700))
    check(seqsets.random_mixed(23, 200, 1, 700, b"ACGTN"))
    check(seqsets.random_mixed(24, 200, 1, 500, b"-ACGNT"))
    check(seqsets.random_mixed(25, 100, 1, 300, bytes(range(0x21, 0x7F))))
    check(seqsets.random_mixed(26, 300, 1, 200, b"AC"))
    check(seqsets.random_mixed(27, 300, 1, 120, b"A"))


def test_long_records_multi_row():
    check(seqsets.random_mixed(28, 6, 2000, 9000), slice_dw=4096)
    check(seqsets.random_mixed(29, 4, 1500, 5000, b"ACGTN"), slice_dw=4096)


def test_deferral_to_bigger_tier():
    # a 120-dword slice: one wave takes the 2-bit strand of a record up to 1888 bases (no bitmask needed), the four waves
    # of the workgroup together (team mode) up to 7648; beyond that the record is deferred
    seqs = seqsets.random_mixed(30, 12, 100, 3000) + seqsets.random_mixed(35, 6, 8000, 12000) + seqsets.random_mixed(36, 4, 3000, 7600)
    n = check(seqs, slice_dw=120)
    assert n == 6


def test_tied_minimal_key_without_room_for_the_bitmask_moves_on():
    """The 2-bit tiers admit a record by its strand alone; only a tie of the minimal key needs the candidate bitmask.
    Tandem repeats (every key ties) of 1600 bases: the strand needs 102 dwords, strand + bitmask 153.  A 120-dword
    slice defers exactly them -- untouched -- and finishes the random records around them; 160 dwords take all.  (The
    tier's team pass then gives them to wave 0 alone with the workgroup's whole LDS: solo=False looks at the tier before.)"""
    rng = np.random.default_rng(33)
    rep = [bytes(rng.choice(list(b"ACGT"), size=k).astype(np.uint8)) * (1600 // k) for k in (5, 8, 20, 32)]
    rnd = seqsets.random_mixed(34, 6, 1500, 1800)
    seqs = [rnd[0], rep[0], rnd[1], rep[1], rep[2], rnd[2], rnd[3], rep[3], rnd[4], rnd[5]]
    assert check(seqs, slice_dw=120, solo=False) == len(rep)
    assert check(seqs, slice_dw=160, solo=False) == 0
    assert check(seqs, slice_dw=120) == 0                 # wave 0 alone with the workgroup's four slices takes them


def test_wave_count_independent():
    seqs = seqsets.random_mixed(31, 40, 48, 300)
    data, offs = seqsets.pack(seqs)
    a = emu.canonicalize_batch(data, offs, n_waves=1)
    b = emu.canonicalize_batch(data, offs, n_waves=7)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_xxh3_wave_matches_golden_vectors():
    import json, os
    vecs = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "xxh3_vectors.json")))["vectors"]
    for v in vecs:
        b = v["in"].encode() if "in" in v else v["in_latin1"].encode("latin-1")
        assert "%016x" % emu.xxh3_64(b) == v["xxh3_64"], len(b)


def test_lmsr_forward_only_flag():
    seqs = seqsets.random_mixed(33, 60, 1, 400) + [b"banana", b"TAA", b"AAA"]
    data, offs = seqsets.pack(seqs)
    out, idx, strand, _, status, ndef = emu.canonicalize_batch(data, offs, flags=1)
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        assert out[a:b].tobytes() == O.lmsr(s)
        assert int(idx[i]) == O.lmsr_index(s) and int(strand[i]) == 0


def test_streaming_kernel_takes_the_headline_records():
    """Pure-ACGT 1 kb records must be handled by the register-resident streaming kernel, not the LDS tier."""
    seqs = seqsets.random_mixed(35, 64, 1000, 1000)
    data, offs = seqsets.pack(seqs)
    for k, (wpb, rpw, _) in emu.STAGED_GEOMETRIES.items():
        emu.canonicalize_batch(data, offs, staged=k, slice_dw=4096)
        last_group = len(seqs) - (len(seqs) - 1) // (wpb * rpw) * (wpb * rpw)
        assert emu.last_fast_count == len(seqs) - last_group     # the batch's last group is left to the general kernel
    for staged in (1, 2):
        emu.canonicalize_batch(*seqsets.pack([b"ACGTN" * 200, b"A" * 500, b"ACGT" * 10, b"ACGT" * 300] * 4), staged=staged)
        assert emu.last_fast_count == 0     # N, repeats, too short, too long -> general kernel


def test_fused_xxh3_matches_oracle():
    """XXH3 computed inside the streaming kernel (records of 241..1008 bases) and by the xxh3 pass (the rest)."""
    seqs = seqsets.random_mixed(36, 120, 241, 1008) + seqsets.random_mixed(37, 40, 1, 300) + \
        seqsets.random_mixed(38, 20, 900, 1100) + [b"ACGT" * 250, b"ACGTN" * 100] + \
        [seqsets.rand_seq(__import__("random").Random(n), n) for n in (241, 255, 256, 257, 319, 320, 321, 1000, 1007, 1008)]
    data, offs = seqsets.pack(seqs)
    out, idx, strand, hs, status, ndef = emu.canonicalize_batch(data, offs, slice_dw=4096, want_hash=True)
    exp, exp_h = O.canonicalize_batch(data, offs, True, True)
    assert np.array_equal(out, exp)
    assert np.array_equal(hs, exp_h)
    assert emu.last_fused_hash_count >= 130     # the 241..1008-base pure-ACGT records with a unique minimum


@pytest.mark.parametrize("staged", sorted(emu.STAGED_GEOMETRIES))
def test_staged_streaming_geometries(staged):
    """canon_stream.h under every workgroup geometry: ragged lengths around the eligibility limits, non-ACGT and
    periodic records mixed in, batches that end inside a group, workgroups with unequal iteration counts."""
    rng = np.random.default_rng(900 + staged)
    for it in range(6):
        n = int(rng.integers(1, 150))
        seqs = []
        for i in range(n):
            mode = (it + i // 16) % 4
            if mode == 0: L = 1000
            elif mode == 1: L = int(rng.integers(48, 1009))
            elif mode == 2: L = int(rng.choice([0, 1, 15, 16, 17, 47, 48, 49, 240, 241, 1007, 1008, 1009, 1300]))
            else: L = int(rng.integers(900, 1009))
            s = bytes(rng.choice(list(b"ACGT" if rng.random() < 0.9 else b"ACGTN-"), size=L).astype(np.uint8))
            if rng.random() < 0.05 and L >= 8:
                p = int(rng.integers(1, 9))
                s = (s[:p] * (L // p + 1))[:L]
            seqs.append(s)
        data, offs = seqsets.pack(seqs)
        # the three builds of the kernel: bytes only, bytes + fused XXH3, everything (index / strand)
        for want_hash, want_aux in ((False, False), (True, False), (True, True)):
            out, idx, strand, h, status, _ = emu.canonicalize_batch(data, offs, want_hash=want_hash, staged=staged,
                                                                    want_aux=want_aux, n_waves=int(rng.integers(1, 13)))
            for i, s in enumerate(seqs):
                a, b = int(offs[i]), int(offs[i + 1])
                c, est, eidx = seqsets.expected(O, s)
                assert out[a:b].tobytes() == c, (it, i, len(s))
                if want_hash:
                    assert int(h[i]) == O.xxh3_64(c), (it, i, len(s))
                if want_aux and len(s):
                    assert (int(strand[i]), int(idx[i])) == (est, eidx), (it, i, len(s))


def test_staged_kernel_unaligned_payload_and_nonzero_first_offset():
    """Payload pointer at any byte alignment, offsets[0] != 0, non-ACGT bytes right in front of / behind the batch:
    groups whose first chunk would start before the payload are left to the general kernel, nothing outside the
    batch is written, results unchanged."""
    seqs = seqsets.random_mixed(950, 90, 48, 1008) + seqsets.random_mixed(951, 10, 1, 47)
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s)[0] for s in seqs]
    for shift, lead in ((0, 0), (1, 0), (7, 5), (15, 33), (8, 1000)):
        for want_aux in (False, True):
            out, idx, strand, h, status, _ = emu.canonicalize_batch(data, offs, want_hash=True, want_aux=want_aux,
                                                                    base_shift=shift, lead=lead, n_waves=8)
            for i, s in enumerate(seqs):
                a, b = int(offs[i]), int(offs[i + 1])
                assert out[a:b].tobytes() == want[i], (shift, lead, i, len(s))
                assert int(h[i]) == O.xxh3_64(want[i])


def test_two_words_per_lane_records_take_the_streaming_kernel():
    """Pure-ACGT records of 1009..2032 bases (two packed words per lane) stay in the streaming kernel; 2033 and beyond,
    and 1009+ with N, go to the general kernel; results equal the oracle either way."""
    seqs = seqsets.random_mixed(970, 40, 1009, 2032) + seqsets.random_mixed(971, 8, 2032, 2032) + seqsets.random_mixed(972, 8, 1009, 1009)
    for k in emu.TWO_ROW:
        check(seqs, staged=k)
        wpb, rpw, _ = emu.STAGED_GEOMETRIES[k]
        assert emu.last_fast_count == len(seqs) - ((len(seqs) - 1) % (wpb * rpw) + 1)      # all but the batch's last group
    data, offs = seqsets.pack(seqs)
    _, _, _, h, _, _ = emu.canonicalize_batch(data, offs, want_hash=True, want_aux=False, staged=emu.TWO_ROW[0])
    assert emu.last_fused_hash_count == emu.last_fast_count > 0                            # XXH3 fused (two blocks + scramble)
    assert [int(x) for x in h] == [O.xxh3_64(O.canonicalize(s)) for s in seqs]
    check(seqs, staged=1, slice_dw=4096)
    assert emu.last_fast_count == 0                                                        # the one-word build leaves them
    seqs = seqsets.random_mixed(973, 20, 2033, 2100) + seqsets.random_mixed(974, 12, 1009, 2032, b"ACGTN")
    check(seqs, staged=emu.TWO_ROW[0], slice_dw=4096)
    assert emu.last_fast_count == 0


def test_rescue_pass_takes_short_records_of_unstaged_groups():
    """One long record per group of 16 pushes every group's span past its LDS image; the short pure-ACGT records around
    it must still be handled by the register routine (rescue pass), not by the LDS tiers -- and stay correct."""
    rng = np.random.default_rng(980)
    seqs = []
    for g in range(6):
        block = seqsets.random_mixed(981 + g, 15, 100, 1008)
        block.insert(int(rng.integers(0, 16)), seqsets.random_mixed(990 + g, 1, 30000, 30000)[0])
        seqs += block
    seqs += seqsets.random_mixed(999, 5, 100, 900, b"ACGTN")            # 4-bit register path, unless index / strand are asked for
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s) for s in seqs]
    for want_hash, want_aux in ((False, False), (True, False), (True, True)):
        out, idx, strand, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=want_hash, want_aux=want_aux, staged=1,
                                                                   slice_dw=8192, n_waves=24, alpha=True)
        assert emu.last_fast_count == 0                                  # no group could be staged
        assert emu.last_rescued_count == 6 * 15 + (0 if want_aux else 5)
        for i, s in enumerate(seqs):
            a, b = int(offs[i]), int(offs[i + 1])
            assert out[a:b].tobytes() == want[i][0], (i, len(s))
            if want_hash:
                assert int(h[i]) == O.xxh3_64(want[i][0])
            if want_aux:
                assert (int(strand[i]), int(idx[i])) == (want[i][1], want[i][2])


def test_mode_3_rescue_pass_takes_every_record():
    """Batches in which one record in eight is longer than 2032 bases skip the staged kernel (launch_canon, mode 3): the
    rescue pass walks all records (virtual list segments) and hands what it cannot take to the tiers."""
    seqs = seqsets.random_mixed(1001, 70, 48, 1008) + seqsets.random_mixed(1002, 20, 3000, 9000) + \
        seqsets.random_mixed(1003, 10, 1, 47) + seqsets.random_mixed(1004, 10, 100, 900, b"ACGTN") + [b"", b"ACGT" * 100]
    rng = np.random.default_rng(1005)
    seqs = [seqs[i] for i in rng.permutation(len(seqs))]
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s) for s in seqs]
    for want_hash, want_aux in ((False, False), (True, False), (True, True)):
        out, idx, strand, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=want_hash, want_aux=want_aux, staged=0,
                                                                   slice_dw=4096, n_waves=12)
        assert emu.last_rescued_count >= 65                          # the eligible short ones (minus rare ties)
        for i, s in enumerate(seqs):
            a, b = int(offs[i]), int(offs[i + 1])
            assert out[a:b].tobytes() == want[i][0], (i, len(s))
            if want_hash:
                assert int(h[i]) == O.xxh3_64(want[i][0])
            if want_aux and len(s):
                assert (int(strand[i]), int(idx[i])) == (want[i][1], want[i][2])


def test_four_bit_register_path_takes_n_and_gap_records():
    """Records over {-,A,C,G,N,T} of 48..1008 symbols: two 4-bit words per lane in registers (fast_canonw<4>) instead of
    the LDS tiers -- every length class of the periodic extension, sparse and dense N, '-' (sorts below A), runs."""
    rng = np.random.default_rng(1100)
    seqs = []
    for L in list(range(48, 80)) + [127, 128, 129, 255, 256, 257, 511, 512, 513, 999, 1000, 1001, 1006, 1007, 1008]:
        s = bytearray(seqsets.random_mixed(1101 + L, 1, L, L)[0])
        for p in rng.integers(0, L, size=max(1, L // 100)):
            s[int(p)] = ord("N")
        seqs.append(bytes(s))
    seqs += seqsets.random_mixed(1102, 60, 48, 1008, b"ACGTN") + seqsets.random_mixed(1103, 40, 48, 1008, b"ACGTN-")
    seqs += [b"N" * 100, b"ACGTN" * 60, b"-" * 50 + b"A" * 50, b"A" * 999 + b"N", b"N" + b"T" * 600, b"ACGT" * 100 + b"-"]
    one_n = bytearray(seqsets.random_mixed(1104, 1, 1000, 1000)[0])
    one_n[500] = ord("N")
    seqs.append(bytes(one_n))
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s) for s in seqs]
    for staged, want_hash in ((0, False), (0, True), (1, False), (1, True)):
        out, idx, strand, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=want_hash, want_aux=False, staged=staged,
                                                                   slice_dw=4096, n_waves=12)     # alpha: decided by the content rule
        # all but periodic records / ties on the 8-symbol key -- by the rescue pass, or (staged) by the streaming kernel's ALPHA build
        assert emu.last_rescued_count + (emu.last_fast_count if staged else 0) >= len(seqs) - 12
        if staged:
            assert emu.last_fast_count >= len(seqs) - 16 - 12            # everything but the batch's last group
        for i, s in enumerate(seqs):
            a, b = int(offs[i]), int(offs[i + 1])
            assert out[a:b].tobytes() == want[i][0], (i, len(s), s[:40])
            if want_hash:
                assert int(h[i]) == O.xxh3_64(want[i][0])


def test_two_bit_mode_with_n_mask_in_the_lds_tiers():
    """Long records with a few N: 2-bit words + a mask strand (canon_record_mode2n); N packs as G, so the cases where
    that matters must fall back to the 4-bit mode and still come out right: an N inside the minimal window, minimal
    windows that differ only by N vs G, ties in the packed space, '-' (sorts below A), N-only stretches."""
    rng = np.random.default_rng(1200)
    seqs = []
    for L in (1100, 1500, 2047, 2048, 2049, 3000, 4097, 6000):
        for frac in (0.001, 0.01, 0.05):
            s = bytearray(seqsets.random_mixed(1201 + L, 1, L, L)[0])
            for p in rng.integers(0, L, size=max(1, int(L * frac))):
                s[int(p)] = ord("N")
            seqs.append(bytes(s))
    body = seqsets.random_mixed(1202, 1, 2000, 2000)[0].replace(b"AAAA", b"ACAC")
    seqs += [b"AAAAAAAANAAAAAAAAAAA" + body,                       # the N sits in what packs as the minimal window
             b"AAAAAAAAAAAAAAAAAAAAG" + body + b"AAAAAAAAAAAAAAAAAAAAN" + body,     # ...GAAAA vs ...NAAAA: N > G decides
             b"AAAAAAAAAAAAAAAAAAAAN" + body + b"AAAAAAAAAAAAAAAAAAAAG" + body,
             b"N" * 1500, b"ACGTN" * 400, b"A" * 1200 + b"N", b"N" + b"T" * 1300,
             b"-" + body, body[:1000] + b"-N" + body[1000:]]
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s) for s in seqs]
    for want_aux, slice_dw in ((False, 1023), (True, 1023), (False, 9980)):
        out, idx, strand, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=want_aux, staged=0, slice_dw=slice_dw,
                                                                   n_waves=8, alpha=False)
        for i, s in enumerate(seqs):
            if strand[i] == 0xFF and want_aux:          # deferred: does not fit this slice
                continue
            a, b = int(offs[i]), int(offs[i + 1])
            if want_aux or slice_dw == 9980:
                assert out[a:b].tobytes() == want[i][0], (i, len(s), s[:30])
            if want_aux:
                assert (int(strand[i]), int(idx[i])) == (want[i][1], want[i][2]), (i, len(s))


def test_two_bit_mode_with_n_bitmask_is_the_mode_that_runs():
    """A 400-dword slice holds the 4-bit strands of records up to ~1.4 kb only, the 2-bit strand + N bitmask up to
    ~4.2 kb: records of 1.5..4 kb with a few N that come out right there took canon_record_mode2n.  Lengths around the
    multiples of 16 and 32 (tail word, bitmask extension), both strands winning, rotation index and strand checked."""
    rng = np.random.default_rng(1300)
    seqs = []
    for L in list(range(1536, 1536 + 34)) + [2048, 2049, 2063, 2064, 3000, 3999, 4000]:
        s = bytearray(seqsets.random_mixed(1301 + L, 1, L, L)[0])
        for p in rng.integers(0, L, size=max(1, L // 150)):
            s[int(p)] = ord("N")
        seqs.append(bytes(s))
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s) for s in seqs]
    out, idx, strand, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=True, staged=0, slice_dw=400, n_waves=8, alpha=False,
                                                               solo=False)
    done = 0
    for i, s in enumerate(seqs):
        if strand[i] == 0xFF:                       # an N inside the minimal window, or a tie: left to a tier with room for 4 bits
            continue
        done += 1
        a, b = int(offs[i]), int(offs[i + 1])
        assert out[a:b].tobytes() == want[i][0], (i, len(s))
        assert (int(strand[i]), int(idx[i])) == (want[i][1], want[i][2]), (i, len(s))
    assert done >= len(seqs) * 3 // 4, (done, len(seqs))
    assert {int(strand[i]) for i in range(len(seqs))} >= {0, 1}


def test_n_inside_the_minimal_window_prefix_rule():
    """canon_record_mode2n keeps a winner whose window holds an N when its packed prefix up to that N is unique (forward:
    j + 1 symbols, reverse: j symbols).  Planted near-ties around that rule, on both strands: the same A-run followed by
    N / G / T / C at the deciding offset, forward and as reverse complements, in a background that cannot compete."""
    rng = np.random.default_rng(1400)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    seqs = []
    for case in range(400):
        L = int(rng.integers(400, 900))
        bg = bytearray(rng.choice(list(b"CGT"), size=L, p=[0.2, 0.4, 0.4]).astype(np.uint8).tobytes())     # no A: the planted runs own the minimum
        run = int(rng.integers(3, 13))
        k = int(rng.integers(2, 5))
        spots = sorted(rng.choice(np.arange(20, L - 40, 30), size=k, replace=False))
        for sp in spots:
            tail = bytes(rng.choice(list(b"NGTCN"), size=1).astype(np.uint8)) + bytes(rng.choice(list(b"ACGT"), size=6).astype(np.uint8))
            motif = b"A" * run + tail
            if rng.random() < 0.5:
                motif = motif.translate(comp)[::-1]                      # the run shows up on the reverse strand
            bg[sp:sp + len(motif)] = motif
        seqs.append(bytes(bg))
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s) for s in seqs]
    # 100 dwords: room for the 2-bit strand + N bitmask of all of them, for the 4-bit strands of none -- what comes out
    # here came out of canon_record_mode2n; what it refuses is deferred (and checked with a slice that takes everything)
    out, idx, strand, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=True, staged=0, slice_dw=100, n_waves=8, alpha=False,
                                                               solo=False)
    assert status == 0
    kept_with_n_in_window = 0
    for i, s in enumerate(seqs):
        if strand[i] == 0xFF:
            continue
        a, b = int(offs[i]), int(offs[i + 1])
        assert out[a:b].tobytes() == want[i][0], (i, s)
        assert (int(strand[i]), int(idx[i])) == (want[i][1], want[i][2]), (i, s)
        kept_with_n_in_window += b"N" in want[i][0][:16]
    assert ndef < len(seqs) * 3 // 4 and kept_with_n_in_window >= 20, (ndef, kept_with_n_in_window)
    out, idx, strand, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=True, staged=0, slice_dw=400, n_waves=8, alpha=False)
    assert status == 0 and ndef == 0
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        assert out[a:b].tobytes() == want[i][0], (i, s)
        assert (int(strand[i]), int(idx[i])) == (want[i][1], want[i][2]), (i, s)


def test_team_mode_four_waves_one_record():
    """Records too long for one wave's slice (here 120 dwords: 1888 bases) and short enough for the workgroup's four
    slices together (7648) are canonicalized by the four waves as a team: rows dealt in turn, minimal key / owners /
    position joined through three LDS words.  Lengths around the row (1024 bases) and word boundaries, both strands,
    rotation index and strand; a reverse-complement palindrome (equal minimal keys on both strands) and a minimal
    16-mer with a few owners (twice, a homopolymer run, an imperfect repeat) are settled by the team, a true tandem
    repeat is left untouched for the tiers behind."""
    rng = np.random.default_rng(1500)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    seqs = []
    for L in [1889, 1890, 1904, 1905, 2047, 2048, 2049, 3071, 3072, 3073, 4095, 4096, 4097, 4112, 5000, 6143, 6144, 6145, 7000, 7647, 7648]:
        seqs.append(seqsets.random_mixed(1501 + L, 1, L, L)[0])
    unit = bytes(rng.choice(list(b"ACGT"), size=977).astype(np.uint8))
    half = seqsets.random_mixed(1502, 1, 1500, 1500)[0]
    withn = bytearray(seqsets.random_mixed(1503, 1, 4000, 4000)[0]); withn[2500] = ord("N")
    seqs.append(bytes(withn))                          # one N: the N-mask team (canon_record_team2n) has it
    seqs.append(half + half.translate(comp)[::-1])     # equal minimal keys on the two strands: settled by a full comparison
    twice = bytearray(seqsets.random_mixed(1504, 1, 5000, 5000)[0].replace(b"AAAA", b"ACAC"))
    twice[99:120] = b"C" + b"A" * 16 + b"CGTC"; twice[2999:3020] = b"T" + b"A" * 16 + b"CGTG"      # the minimal 16-mer twice: the smaller rotation
    seqs.append(bytes(twice))
    seqs.append((unit * 5)[:4000])                     # four or five owners of the minimal key, no period: the smallest rotation
    seqs.append(b"C" + b"A" * 19 + seqsets.random_mixed(1505, 1, 3000, 3000)[0].replace(b"AAAA", b"ACAC"))     # a homopolymer run: a row of owners
    odd = [unit * 4]                                   # a true period (3908 = 4 x 977): the general routine's
    seqs += odd
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s) for s in seqs]
    out, idx, strand, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=True, staged=0, slice_dw=120, n_waves=8, alpha=False,
                                                               solo=False)
    assert status == 0 and ndef == len(odd)
    strands = set()
    for i, s in enumerate(seqs[:-len(odd)]):
        a, b = int(offs[i]), int(offs[i + 1])
        assert strand[i] != 0xFF, (i, len(s))
        assert out[a:b].tobytes() == want[i][0], (i, len(s))
        assert (int(strand[i]), int(idx[i])) == (want[i][1], want[i][2]), (i, len(s))
        strands.add(int(strand[i]))
    assert strands == {0, 1}
    for i in range(len(seqs) - len(odd), len(seqs)):
        assert strand[i] == 0xFF                       # untouched
    # and with room for everything in one wave's slice the same records take the one-wave path: same answers
    out2, idx2, strand2, _, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=True, staged=0, slice_dw=4096, n_waves=8, alpha=False)
    assert status == 0 and ndef == 0
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        assert out2[a:b].tobytes() == want[i][0] and (int(strand2[i]), int(idx2[i])) == (want[i][1], want[i][2]), (i, len(s))


def test_team_mode_with_n_bitmask():
    """The N-mask team: records with a few N that are too long for one wave's slice (120 dwords: strand + bitmask of
    ~1.2 kb) and fit the workgroup's four slices (~5 kb).  Random N at 0.1-2 %, lengths around the row boundaries, and
    the planted near-ties of the prefix rule (an N inside the minimal window) on both strands."""
    rng = np.random.default_rng(1600)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    seqs = []
    for L in [1300, 1536, 2047, 2048, 2049, 2064, 3000, 3071, 3072, 3073, 4000, 4096, 4097, 5000, 5072]:
        for frac in (0.001, 0.02):
            s = bytearray(seqsets.random_mixed(1601 + L, 1, L, L)[0])
            for p in rng.integers(0, L, size=max(1, int(L * frac))):
                s[int(p)] = ord("N")
            seqs.append(bytes(s))
    for case in range(120):
        L = int(rng.integers(1400, 5000))
        bg = bytearray(rng.choice(list(b"CGT"), size=L, p=[0.2, 0.4, 0.4]).astype(np.uint8).tobytes())
        run = int(rng.integers(3, 13))
        for sp in sorted(rng.choice(np.arange(20, L - 40, 30), size=int(rng.integers(2, 5)), replace=False)):
            motif = b"A" * run + bytes(rng.choice(list(b"NGTCN"), size=1).astype(np.uint8)) + bytes(rng.choice(list(b"ACGT"), size=6).astype(np.uint8))
            if rng.random() < 0.5:
                motif = motif.translate(comp)[::-1]
            bg[sp:sp + len(motif)] = motif
        seqs.append(bytes(bg))
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s) for s in seqs]
    out, idx, strand, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=True, staged=0, slice_dw=120, n_waves=8, alpha=False,
                                                               solo=False)
    assert status == 0
    done = kept_with_n_in_window = 0
    for i, s in enumerate(seqs):
        if strand[i] == 0xFF:
            continue
        done += 1
        a, b = int(offs[i]), int(offs[i + 1])
        assert out[a:b].tobytes() == want[i][0], (i, len(s))
        assert (int(strand[i]), int(idx[i])) == (want[i][1], want[i][2]), (i, len(s))
        kept_with_n_in_window += b"N" in want[i][0][:16]
    assert done >= len(seqs) // 2 and kept_with_n_in_window >= 3, (done, len(seqs), kept_with_n_in_window)


def test_team_mode_four_bit():
    """The 4-bit team (round 4): what the 2-bit teams refuse -- gaps, an N inside the winning window behind a prefix that other
    windows share -- and the one-wave 4-bit mode has no room for (slice of 120 dwords: strand + candidate bitmask of ~750
    symbols; the four slices together hold a 4-bit strand of ~3.8 kb).  Without the wave-0-alone fallback (solo=False) a record
    with a gap can only have been done by this team.  Lengths around the row (512 symbols) and word (8) boundaries, both strands,
    a minimal 8-mer with a few owners (team_settle), rotation index and strand as the reference sees them."""
    rng = np.random.default_rng(1700)
    comp = bytes.maketrans(b"ACGTN-", b"TGCAN-")
    seqs = []
    for L in [760, 800, 1023, 1024, 1025, 1031, 1032, 1033, 1536, 2047, 2048, 2049, 3000, 3583, 3584, 3585, 3800, 3816]:
        s = bytearray(seqsets.random_mixed(1701 + L, 1, L, L)[0])
        for p in rng.integers(0, L, size=max(1, L // 200)):
            s[int(p)] = ord("-") if rng.random() < 0.5 else ord("N")
        s[int(rng.integers(0, L))] = ord("-")                    # at least one gap: never 2-bit material
        seqs.append(bytes(s))
    n_gap = len(seqs)
    for case in range(60):                                       # the planted near-ties of the prefix rule: 2n team or this one
        L = int(rng.integers(800, 3800))
        bg = bytearray(rng.choice(list(b"CGT"), size=L, p=[0.2, 0.4, 0.4]).astype(np.uint8).tobytes())
        run = int(rng.integers(2, 8))
        for sp in sorted(rng.choice(np.arange(20, L - 40, 30), size=int(rng.integers(2, 5)), replace=False)):
            motif = b"A" * run + bytes(rng.choice(list(b"NGTCN"), size=1).astype(np.uint8)) + bytes(rng.choice(list(b"ACGT"), size=6).astype(np.uint8))
            if rng.random() < 0.5:
                motif = motif.translate(comp)[::-1]
            bg[sp:sp + len(motif)] = motif
        seqs.append(bytes(bg))
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s) for s in seqs]
    out, idx, strand, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=True, staged=0, slice_dw=120, n_waves=8, alpha=False,
                                                               solo=False)
    assert status == 0
    done_gap = done = 0
    strands = set()
    for i, s in enumerate(seqs):
        if strand[i] == 0xFF:
            continue
        done += 1
        done_gap += i < n_gap
        a, b = int(offs[i]), int(offs[i + 1])
        assert out[a:b].tobytes() == want[i][0], (i, len(s))
        assert (int(strand[i]), int(idx[i])) == (want[i][1], want[i][2]), (i, len(s))
        strands.add(int(strand[i]))
    # (a tie of the minimal 8-mer beyond what the team settles -- a period, more than eight owners -- stays for wave 0 alone)
    assert done_gap == n_gap and done >= len(seqs) - 2 and strands == {0, 1}, (done_gap, n_gap, done, len(seqs))
    # with the fallback: everything, same answers
    out2, idx2, strand2, _, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=True, staged=0, slice_dw=120, n_waves=8, alpha=False)
    assert status == 0 and ndef == 0
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        assert out2[a:b].tobytes() == want[i][0] and (int(strand2[i]), int(idx2[i])) == (want[i][1], want[i][2]), (i, len(s))


@pytest.mark.parametrize("staged", [1, 3])
def test_register_routine_with_n_mask(staged):
    """The streaming kernel's ALPHA build, bytes only: records of 48..1008 bases with a few N go through fast_canon's N-mask
    variant (N packed as G, the reverse strand shows C, a mask word beside the strand word) -- every length class of the
    periodic extension, both strands, N next to the minimal window; planted A-runs with an N / G / T / C behind them put
    the N among the deciding symbols (the routine must refuse and the 4-bit routine answer)."""
    rng = np.random.default_rng(1700 + staged)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    seqs = []
    for L in list(range(48, 81)) + list(range(990, 1009)) + [127, 128, 129, 255, 256, 257, 511, 512, 513]:
        s = bytearray(seqsets.random_mixed(1701 + L, 1, L, L)[0])
        for p in rng.integers(0, L, size=max(1, L // 100)):
            s[int(p)] = ord("N")
        seqs.append(bytes(s))
    for case in range(150):
        L = int(rng.integers(200, 1009))
        bg = bytearray(rng.choice(list(b"CGT"), size=L, p=[0.2, 0.4, 0.4]).astype(np.uint8).tobytes())
        run = int(rng.integers(2, 15))
        for sp in sorted(rng.choice(np.arange(10, L - 30, 25), size=int(rng.integers(1, 4)), replace=False)):
            motif = b"A" * run + bytes(rng.choice(list(b"NGTCN"), size=1).astype(np.uint8)) + bytes(rng.choice(list(b"ACGTN"), size=5, p=[.23, .23, .23, .23, .08]).astype(np.uint8))
            if rng.random() < 0.5:
                motif = motif.translate(comp)[::-1]
            bg[sp:sp + len(motif)] = motif
        seqs.append(bytes(bg))
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s) for s in seqs]
    out, idx, strand, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=False, staged=staged, slice_dw=4096, n_waves=12,
                                                               alpha=True)
    assert status == 0 and ndef == 0
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        assert out[a:b].tobytes() == want[i][0], (i, len(s), s[:60])
    assert emu.last_fast_count >= len(seqs) - 70        # the planted near-ties and the batch's last group take the passes behind


# ---- canon_mixed_kernel's lean routines (canon_mixed.h): mode-3 batches that want bytes only --------------------------
def _sprinkle(rng, s, frac, ch=ord("N")):
    b = bytearray(s)
    for i in range(len(b)):
        if rng.random() < frac:
            b[i] = ch
    return bytes(b)


def _mixed_batch(seed, with_n):
    import random
    rng = random.Random(seed)
    seqs = seqsets.random_mixed(seed + 1, 40, 48, 1008) + seqsets.random_mixed(seed + 2, 30, 1009, 9000) + \
        seqsets.random_mixed(seed + 3, 6, 1, 47) + [b"", b"ACGT" * 400, b"A" * 3000]
    for n in (1100, 2500, 4097):                                          # long reverse-complement palindromes, rotations, repeats
        h = seqsets.rand_seq(rng, n // 2)
        seqs.append(h + seqsets.revcomp_acgt(h))
        base = seqsets.rand_seq(rng, n)
        k = rng.randrange(n)
        seqs += [base, base[k:] + base[:k], seqsets.revcomp_acgt(base)]
        u = seqsets.rand_seq(rng, 37)
        seqs.append(seqsets.rand_seq(rng, 600) + u * 40 + seqsets.rand_seq(rng, 500))
        seqs.append(seqsets.rand_seq(rng, 700) + b"A" * 40 + seqsets.rand_seq(rng, 900) + b"A" * 40)        # the minimal key twice
    if with_n:
        seqs = [_sprinkle(rng, s, 0.01) if i % 5 else s for i, s in enumerate(seqs)]
        seqs += [_sprinkle(rng, seqsets.rand_seq(rng, 3000), 0.02, ord("-")), _sprinkle(rng, seqsets.rand_seq(rng, 500), 0.3)]
    rng.shuffle(seqs)
    return seqs


@pytest.mark.parametrize("with_n", [False, True])
@pytest.mark.parametrize("base_shift,lead", [(0, 0), (5, 0), (0, 7), (11, 13), (15, 1)])
def test_mixed_kernel_lean_routines(with_n, base_shift, lead):
    """canon_mixed_kernel<NM> as launch_canon runs it for a mode-3 batch without hash / index outputs: short records
    through the register routine from memory, longer ones through the lean LDS routine (aligned chunks at every payload
    alignment, one scan loop for both strands), the rest -- ties, palindromes, gaps, records beyond the slice -- through the
    list into stage A.  Bytes against the oracle for every record; most records must be the lean routines' own."""
    seqs = _mixed_batch(4000 + base_shift + 16 * lead, with_n)
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s)[0] for s in seqs]
    out, _, _, _, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=False, staged=0, slice_dw=1280, n_waves=12,
                                                        alpha=with_n, mixed=True, base_shift=base_shift, lead=lead)
    assert status == 0 and ndef == 0
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        assert out[a:b].tobytes() == want[i], (i, len(s), s[:60])
    assert emu.last_rescued_count >= (60 if not with_n else 50)          # of ~100 records


def test_mixed_kernel_prefix_rule_and_extension_edges():
    """An N planted inside or right behind the minimal window of long records (the prefix rule's count over both strands),
    minimal windows that straddle the record's end / start (the periodic extension on both sides), at all sixteen
    alignments of the record in its first chunk."""
    import random
    rng = random.Random(77)
    seqs = []
    for k in range(48):
        n = rng.randint(1100, 2600)
        s = bytearray(seqsets.rand_seq(rng, n, b"CGT"))
        pos = rng.choice([0, 1, 7, n - 1, n - 9, n - 16, n - 17, n // 2])
        for i in range(18):
            s[(pos + i) % n] = ord("A")                                   # the minimal key, wrapping around the end for some
        if k % 3 == 0:
            s[(pos + rng.randint(3, 22)) % n] = ord("N")                 # ...with an N inside or just behind it
        if k % 3 == 1:
            s = bytearray(seqsets.revcomp_acgt(bytes(s).replace(b"N", b"A")))   # the reverse strand wins
            s[rng.randrange(n)] = ord("N")
        seqs.append(bytes(s) + seqsets.rand_seq(rng, k % 16, b"G"))       # shifts the next record's alignment
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s)[0] for s in seqs]
    for alpha in (True, False):
        out, _, _, _, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=False, staged=0, slice_dw=1280, n_waves=8,
                                                            alpha=alpha, mixed=True)
        assert status == 0 and ndef == 0
        for i, s in enumerate(seqs):
            a, b = int(offs[i]), int(offs[i + 1])
            assert out[a:b].tobytes() == want[i], (alpha, i, len(s))


def test_mixed_kernel_settles_a_tied_minimal_key():
    """The minimal 16-symbol key owned by several positions (pure build): the lean routine compares their rotations itself
    -- one such record in a million-record batch used to be a one-wave pass of stage A behind everything else.  Planted:
    the same minimal 16-mer two to five times, followed by different symbols (a run of 17 or 18 A as well: two or three
    owners side by side); periodic records still move on."""
    import random
    rng = random.Random(91)
    seqs = []
    for k in range(40):
        n_runs = rng.randint(2, 5)
        run = b"A" * (16 if k % 5 else rng.randint(17, 18))
        parts = []
        for _ in range(n_runs):
            parts.append(run + seqsets.rand_seq(rng, rng.randint(200, 900), b"CGT"))
        s = b"".join(parts)
        if k % 4 == 0:
            s = seqsets.revcomp_acgt(s)                                   # the reverse strand wins
        seqs.append(s + seqsets.rand_seq(rng, k % 16, b"G"))
    seqs += [seqsets.rand_seq(rng, 700, b"CGT") * 3, b"ACGT" * 500]       # periods: stage A's
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s)[0] for s in seqs]
    out, _, _, _, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=False, staged=0, slice_dw=1280, n_waves=8,
                                                        alpha=False, mixed=True)
    assert status == 0 and ndef == 0
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        assert out[a:b].tobytes() == want[i], (i, len(s))
    # most tied records are the lean routine's own (not: those of <= 1008 symbols -- the register routine leaves ties to stage
    # A --, two owners in words of the same lane, the two periodic records)
    assert emu.last_rescued_count >= 25


@pytest.mark.parametrize("with_n", [False, True])
def test_mixed_kernel_winner_seen_twice_by_one_lane(with_n):
    """Records a little longer than a multiple of 1024 symbols: the winning window near the record's start is seen again, as
    its periodic twin, in the last words -- which then belong to the SAME lane as word 0 or 1 (64 words per row).  The
    lean routine keeps the lane's second owner word; found on the GPU as ~100 records per million of config 4 that went to
    stage A for nothing (lengths 1021..1034, 2047, 4104, 5131 ...)."""
    import random
    rng = random.Random(123)
    seqs = []
    for n in list(range(1010, 1075, 3)) + list(range(2040, 2100, 5)) + [3073, 3080, 4104, 5131]:
        s = bytearray(seqsets.rand_seq(rng, n, b"CGT"))
        pos = rng.choice([0, 1, 2, 9, 15, 16, 17, n - 1, n - 2, n - 15, n - 16, n - 17, n - 31])
        for i in range(16):
            s[(pos + i) % n] = ord("A")
        if with_n:
            s[(pos + 40) % n] = ord("N")
        seqs.append((seqsets.revcomp_acgt(bytes(s).replace(b"N", b"C")) if n % 2 else bytes(s)) + seqsets.rand_seq(rng, n % 16, b"G"))
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s)[0] for s in seqs]
    out, _, _, _, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=False, staged=0, slice_dw=1368, n_waves=8,
                                                        alpha=with_n, mixed=True)
    assert status == 0 and ndef == 0
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        assert out[a:b].tobytes() == want[i], (i, len(s))
    if not with_n:
        assert emu.last_rescued_count >= len(seqs) - 1                    # (all but the batch's... none: the last record has its own build)


def test_mixed_kernel_minimal_key_inside_a_reverse_complement_palindrome():
    """A minimal 16-mer inside a reverse-complement palindrome of 18 or more (AAAAAATAGCTATTTTTT) is owned by BOTH strands:
    equal minimal keys, the two rotations are compared in full by the lean routine itself (seen on the GPU: ~6 records per
    million of config 4).  Whole-record palindromes included (equal rotations: the reverse strand's bytes, identical)."""
    import random
    rng = random.Random(321)
    seqs = []
    for k in range(30):
        half = b"A" * rng.randint(6, 9) + seqsets.rand_seq(rng, rng.randint(1, 4), b"ACGT")
        pal = half + seqsets.revcomp_acgt(half)                          # self reverse-complementary, starts with the A run
        body = seqsets.rand_seq(rng, rng.randint(1100, 4000), b"CGT")
        s = body[:len(body) // 3] + pal + body[len(body) // 3:]
        seqs.append(s + seqsets.rand_seq(rng, k % 16, b"G"))
    for n in (1200, 2600):
        h = seqsets.rand_seq(rng, n // 2)
        seqs.append(h + seqsets.revcomp_acgt(h))
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s)[0] for s in seqs]
    out, _, _, _, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=False, staged=0, slice_dw=1280, n_waves=8,
                                                        alpha=False, mixed=True)
    assert status == 0 and ndef == 0
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        assert out[a:b].tobytes() == want[i], (i, len(s))
    assert emu.last_rescued_count >= 28


@pytest.mark.parametrize("staged,alpha", [(1, False), (1, True), (0, False), (0, True), (10, False)])
def test_hash_only_batches_hash_a_view_of_the_input(staged, alpha):
    """`uniq` without --canonicalize (src/uniq.rs:45,55-60) wants the XXH3 of the canonical form and no bytes: records whose
    hash is not fused into the register routine (<= 240 symbols, N / gap records, everything the LDS tiers take) leave strand +
    rotation, and the xxh3 pass hashes that VIEW of the input -- no canonical bytes are written, no scratch is needed (the
    round-2 path sized one from the device's last offset, with a host synchronisation)."""
    seqs = seqsets.random_mixed(7001, 40, 48, 1008) + seqsets.random_mixed(7002, 30, 1, 260) + seqsets.random_mixed(7003, 12, 1009, 2500) + \
        seqsets.random_mixed(7004, 20, 30, 900, b"ACGTN") + seqsets.random_mixed(7005, 8, 10, 400, b"-ACGNT") + \
        seqsets.random_mixed(7006, 6, 1, 300, bytes(range(0x21, 0x7F))) + seqsets.adversarial()[:60] + [b"", b"ACGTRYKMacgtn" * 30]
    rng = np.random.default_rng(7007)
    seqs = [seqs[i] for i in rng.permutation(len(seqs))]
    data, offs = seqsets.pack(seqs)
    out, _, _, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=True, want_aux=False, staged=staged, slice_dw=4096, n_waves=8,
                                                        alpha=alpha, hash_only=True)
    assert status == 0 and ndef == 0
    assert (out == 0x3F).all()                                            # nothing was written as bytes
    for i, s in enumerate(seqs):
        assert int(h[i]) == O.xxh3_64(seqsets.expected(O, s)[0]), (i, len(s), s[:60])


def test_mixed_kernel_n_inside_the_minimal_window_is_resolved_among_the_sharers():
    """N build: an N inside the packed minimal key's window, other rotations sharing the symbols in front of it -- the
    true minimum is one of the sharers, found by lean_resolve_n (exact ranks A C G N T over 32 symbols) instead of a trip
    through stage A's 4-bit mode.  Planted A-runs with an N at every offset, a second shorter A-run as the sharer, both
    strands, plus N-rich random and two-letter records."""
    import random
    rng = random.Random(2025)
    seqs = [_sprinkle(rng, seqsets.rand_seq(rng, rng.randint(1009, 2600)), rng.choice([0.01, 0.03, 0.08])) for _ in range(40)]
    seqs += [_sprinkle(rng, seqsets.rand_seq(rng, rng.randint(1009, 2200), b"AC"), 0.02) for _ in range(6)]
    for k in range(40):
        n = rng.randint(1100, 2400)
        s = bytearray(seqsets.rand_seq(rng, n, b"CGT"))
        pos, pos2 = rng.randrange(n), rng.randrange(n)
        for i in range(rng.randint(8, 16)):
            s[(pos + i) % n] = ord("A")
        s[(pos + k % 13) % n] = ord("N")
        for i in range(rng.randint(5, 10)):
            s[(pos2 + i) % n] = ord("A")
        s = bytes(s)
        seqs.append(s if k % 2 else seqsets.revcomp_acgt(s.replace(b"N", b"X")).replace(b"X", b"N"))
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s)[0] for s in seqs]
    out, _, _, _, status, ndef = emu.canonicalize_batch(data, offs, want_hash=False, want_aux=False, staged=0, slice_dw=1368, n_waves=8,
                                                        alpha=True, mixed=True, base_shift=3)
    assert status == 0 and ndef == 0
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        assert out[a:b].tobytes() == want[i], (i, len(s))
    assert emu.last_rescued_count >= len(seqs) - 8


@pytest.mark.parametrize("with_n,hash_only", [(False, False), (False, True), (True, False), (True, True)])
def test_mixed_kernel_with_fused_xxh3(with_n, hash_only):
    """canon_mixed_h_kernel / canon_mixed_nh_kernel (`circkit uniq` on records of mixed lengths): the lean routine's output
    loop accumulates XXH3 block by block (a row of 64 chunks = one 1024-byte block, scramble behind every full one), for
    records of more than 240 symbols, with N too (round 4); short N records, the short-input classes and everything stage A
    takes are hashed by the xxh3 pass -- from the bytes, or (hash_only: no bytes asked for) from their views.  Lengths around every block and
    stripe boundary."""
    import random
    rng = random.Random(515)
    seqs = _mixed_batch(5150, with_n)
    for n in (241, 255, 256, 257, 1009, 1023, 1024, 1025, 1087, 1088, 1089, 2047, 2048, 2049, 2111, 2112, 2113, 3072, 3073, 4095, 4096, 4097, 5121):
        seqs.append(seqsets.rand_seq(rng, n))
        seqs.append(seqsets.revcomp_acgt(seqsets.rand_seq(rng, n, b"CGT") + b"A" * 17))
    rng.shuffle(seqs)
    data, offs = seqsets.pack(seqs)
    want = [seqsets.expected(O, s)[0] for s in seqs]
    out, _, _, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=True, want_aux=False, staged=0, slice_dw=1596 if with_n else 1280, n_waves=12,
                                                        alpha=with_n, mixed=True, hash_only=hash_only, base_shift=7)
    assert status == 0 and ndef == 0
    for i, s in enumerate(seqs):
        a, b = int(offs[i]), int(offs[i + 1])
        if not hash_only:
            assert out[a:b].tobytes() == want[i], (i, len(s))
        assert int(h[i]) == O.xxh3_64(want[i]), (i, len(s))
    if hash_only:
        assert (out == 0x3F).all()
    # the long records' hashes are the lean routine's own -- since round 4 with N as well (patched in registers ahead of the hash)
    print("fused", with_n, emu.last_fused_hash_count)
    assert emu.last_fused_hash_count >= (60 if not with_n else 45)


def _host_taper(n, G):
    """launch_canon's sizing of the walking stages' segments (circkit_hip.hip): (all_cap, taper_seg0, taper_log2)."""
    all_cap = -(-n // G)
    seg0 = log2 = 0
    if G >= 4096 and all_cap >= 16:
        while (2 << log2) <= G // 16:
            log2 += 1
        gen = 1 << log2
        seg0 = G - 3 * gen
        while seg0 * all_cap + gen * ((all_cap >> 1) + (all_cap >> 2) + (all_cap >> 3)) < n:
            all_cap += 1
    return all_cap, seg0, log2


def test_tapered_segments_tile_the_batch():
    """seg_records: equal segments, then three generations of smaller ones -- every record in exactly one segment."""
    import ctypes
    lib = ctypes.CDLL(emu.build())
    lib.emu_seg_cover.restype = ctypes.c_int64
    lib.emu_seg_cover.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    cases = [(1_000_000, 32768), (10_000_000, 32768), (999_983, 32768), (70_000, 4096), (65_536, 4096), (131_071, 8191),
             (5_000_000, 20_000), (524_288, 32768), (524_289, 32768), (100, 100), (50_000, 4095)]
    for n, G in cases:
        cap, seg0, log2 = _host_taper(n, G)
        used = lib.emu_seg_cover(n, G, cap, seg0, log2)
        assert used > 0, (n, G, cap, seg0, log2)
        if log2:
            gen = 1 << log2
            assert seg0 + 3 * gen == G and cap >> 3 >= 1
    # no taper = the plain mapping
    assert lib.emu_seg_cover(1000, 10, 100, 0, 0) == 10
    assert lib.emu_seg_cover(1001, 10, 100, 0, 0) == -1


@pytest.mark.parametrize("staged", [14, 15])
@pytest.mark.parametrize("hash_only", [False, True])
def test_pair_build_two_records_per_wave(staged, hash_only):
    """canon_pair.h (the fused-XXH3 streaming build for pure-ACGT batches): record A in lanes 0..31, record B in lanes 32..63 of one
    wave.  The halves must not see each other: every combination of a record the routine takes with a partner it does not
    (too short, too long, an N, a gap, a periodic record, a tied 8-symbol minimum, a reverse-complement palindrome), lengths on
    and off the 16-symbol grid in either half (the periodic extension's r == 0 / r != 0 forms), the limits 48 and 1008, XXH3's
    short-input classes (<= 240: left to the xxh3 pass, through a view when no bytes are written), every alignment of the
    record in its first chunk."""
    import random
    rng = random.Random(4100 + staged)
    R = lambda n, al=b"ACGT": seqsets.rand_seq(rng, n, al)
    pal = R(40)
    pal = pal + bytes(O.revcomp(pal))                                                 # its own reverse complement
    odd = [b"", R(7), R(47), R(1009), R(1500), R(3000), R(500)[:250] + b"N" + R(249), R(300) + b"-" + R(300), b"ACGT" * 200, b"A" * 777,
           (R(31) * 40)[:900], pal * 8, b"T" * 48, R(100, b"AC"), bytes(range(0x30, 0x7B)) * 5]
    good = [R(n) for n in (48, 49, 63, 64, 65, 240, 241, 255, 256, 257, 511, 512, 513, 527, 528, 529, 992, 1000, 1007, 1008)]
    seqs = []
    for g in good:
        o = rng.choice(odd)
        seqs += [g, o] if rng.random() < 0.5 else [o, g]
    for _ in range(60):
        seqs += [R(rng.randint(48, 1008)), R(rng.randint(48, 1008))]
    seqs += [R(1000) for _ in range(40)]                                               # (equal lengths: the lane constants are reused)
    seqs += [R(rng.choice([48, 64, 1008])) for _ in range(16)]
    data, offs = seqsets.pack(seqs)
    for base_shift in (0, 5):
        out, _, _, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=True, want_aux=False, staged=staged, slice_dw=4096, n_waves=8,
                                                            hash_only=hash_only, base_shift=base_shift)
        assert status == 0 and ndef == 0
        n_pair = sum(1 for s in seqs[:len(seqs) // (8 if staged == 14 else 16) * (8 if staged == 14 else 16)] if 48 <= len(s) <= 1008 and set(s) <= set(b"ACGT"))
        # (not all of them: the periodic and tied ones are refused, a group with a 3 kb record in it is not staged at all)
        assert emu.last_fast_count >= n_pair * 3 // 4
        for i, s in enumerate(seqs):
            c = seqsets.expected(O, s)[0]
            if not hash_only:
                assert out[int(offs[i]):int(offs[i + 1])].tobytes() == c, (i, len(s), s[:60])
            assert int(h[i]) == O.xxh3_64(c), (i, len(s), s[:60])
        if hash_only:
            assert (out == 0x3F).all()


def test_pair_build_every_length():
    """canon_pair.h's lane constants (periodic extension, reverse-strand windows, output cells, XXH3 stripes) are functions of the
    two record lengths: every length 48..1008 once in either half, next to a partner of another length, bytes and hashes."""
    import random
    rng = random.Random(5150)
    lens = list(range(48, 1009))
    rng.shuffle(lens)
    seqs = []
    for i, n in enumerate(lens):
        a, b = seqsets.rand_seq(rng, n), seqsets.rand_seq(rng, rng.randint(48, 1008))
        seqs += [a, b] if i % 2 else [b, a]
    seqs += [seqsets.rand_seq(rng, 500)] * 0 + [seqsets.rand_seq(rng, 700) for _ in range(16)]     # (the last group is never staged)
    data, offs = seqsets.pack(seqs)
    out, _, _, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=True, want_aux=False, staged=15, slice_dw=4096, n_waves=8)
    assert status == 0 and ndef == 0
    assert emu.last_fast_count >= len(seqs) - 16 - 60                                  # (tied 8- and 16-symbol minima are rare)
    for i, s in enumerate(seqs):
        c = seqsets.expected(O, s)[0]
        assert out[int(offs[i]):int(offs[i + 1])].tobytes() == c, (i, len(s))
        assert int(h[i]) == O.xxh3_64(c), (i, len(s))


@pytest.mark.parametrize("staged", [1, 13, 16])
@pytest.mark.parametrize("hash_only", [False, True])
def test_streaming_n_build_fuses_the_hash_of_records_with_n(staged, hash_only):
    """`uniq` on a batch with N (MODE_ALPHA): the N-mask variant of the register routine leaves the XXH3 of a record WITH N to the
    workgroup's merger like any other -- row sums of the patched cells, and the last stripe as patched bytes (group_hash_put_bytes:
    the strand in the slot would decode to G / C where the record holds N).  Before, every such record took the 4-bit routine and a
    pass of the xxh3 kernel over its bytes.  N in the last 64 bytes, in the first cell, next to the wrap; records without N in the
    same build; the ones the variant refuses (an N among the deciding symbols, a gap) still come out right through the 4-bit
    routine and the xxh3 pass."""
    import random
    rng = random.Random(6100 + staged)
    seqs = []
    for i in range(120):
        n = rng.choice([1000, 1000, 1000, 241, 256, 300, 777, 1008, rng.randint(241, 1008)])
        s = bytearray(seqsets.rand_seq(rng, n))
        for _ in range(rng.choice([0, 1, 1, 2, 5, 10, 30])):
            s[rng.randrange(n)] = ord("N")
        if i % 7 == 0:
            for p in (n - 1, n - 17, n - 64, 0, 15, 16):
                s[p] = ord("N")
        if i % 23 == 0:
            s[rng.randrange(n)] = ord("-")
        seqs.append(bytes(s))
    seqs += [seqsets.rand_seq(rng, 1000) for _ in range(8)] + [b"N" * 500, b"ACGTN" * 100]
    data, offs = seqsets.pack(seqs)
    out, _, _, h, status, ndef = emu.canonicalize_batch(data, offs, want_hash=True, want_aux=False, staged=staged, slice_dw=4096, n_waves=8,
                                                        alpha=True, hash_only=hash_only)
    assert status == 0 and ndef == 0
    for i, s in enumerate(seqs):
        c = seqsets.expected(O, s)[0]
        if not hash_only:
            assert out[int(offs[i]):int(offs[i + 1])].tobytes() == c, (i, len(s))
        assert int(h[i]) == O.xxh3_64(c), (i, len(s))
    with_n = sum(1 for s in seqs[:len(seqs) // 16 * 16 - 16] if b"N" in s and b"-" not in s)
    assert emu.last_fused_hash_count >= with_n * 2 // 3, (emu.last_fused_hash_count, with_n)      # (an N among the deciding symbols: one record in ten or so)
