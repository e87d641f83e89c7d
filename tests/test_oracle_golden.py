"""Pins the CPU oracle (oracle/) against every known answer and fixture the reference's own
tests hold for the canonicalize/uniq path.  CPU only."""
import json
import os
import random

import numpy as np
import pytest

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KA = json.load(open(os.path.join(GOLDEN, "ref_known_answers.json")))


@pytest.mark.parametrize("v", KA["lmsr_index"], ids=lambda v: v["in"])
def test_lmsr_index_known_answers(v):
    assert O.lmsr_index(v["in"].encode()) == v["out"]


@pytest.mark.parametrize("v", KA["lmsr"], ids=lambda v: v["in"])
def test_lmsr_known_answers(v):
    assert O.lmsr(v["in"].encode()) == v["out"].encode()


def test_lmsr_idempotent_vector():
    for v in KA["lmsr_idempotent"]:
        a = O.lmsr(v["in"].encode())
        assert O.lmsr(a) == a


@pytest.mark.parametrize("v", KA["canonicalize"], ids=lambda v: v["in"])
def test_canonicalize_known_answers(v):
    assert O.canonicalize(v["in"].encode()) == v["out"].encode()


def test_canonicalize_real_monomer_pair():
    for v in KA["canonicalize_equal_pairs"]:
        a, b = v["a"].encode(), v["b"].encode()
        assert O.lmsr(a) == O.lmsr(b)
        assert O.canonicalize(a) == O.canonicalize(b)
        assert O.canonicalize(a) == b"AAACGCTGCTAAATCAATTTCCTCCATCACCTAGTTTATGTAG"


def test_empty():
    assert O.lmsr_index(b"") == 0
    assert O.lmsr(b"") == b""
    assert O.canonicalize(b"") == b""


# -- the reference's proptests (lib/src/canonicalize.rs:216-231) restated with a seeded RNG --------
def test_lmsr_index_matches_naive_printable_ascii():
    rng = random.Random(1)
    for _ in range(3000):
        n = rng.randint(1, 100)
        s = bytes(rng.randint(0x20, 0x7E) for _ in range(n))
        assert O.lmsr_index(s) == O.lmsr_index_simple(s)


def test_lmsr_index_matches_naive_small_alphabets():
    rng = random.Random(2)
    for alpha in (b"A", b"AC", b"ACG", b"ACGT", b"-ACGNT"):
        for _ in range(1500):
            n = rng.randint(1, 60)
            s = bytes(rng.choice(alpha) for _ in range(n))
            assert O.lmsr_index(s) == O.lmsr_index_simple(s), s
    # periodic and near-periodic
    for _ in range(1500):
        p = rng.randint(1, 12)
        unit = bytes(rng.choice(b"ACGT") for _ in range(p))
        s = bytearray(unit * rng.randint(1, 12))
        if rng.random() < 0.5 and s:
            s[rng.randrange(len(s))] = rng.choice(b"ACGT")
        assert O.lmsr_index(bytes(s)) == O.lmsr_index_simple(bytes(s)), s


def test_quadratic_cost_model_gives_the_same_answers():
    """ck_oracle_lmsr_index_nth / canonicalize_batch_nth (bench.py's reference_faithful_quadratic leg): the reference's
    chars().nth() access pattern -- same answers as the byte-indexed transcription, only slower."""
    import numpy as np
    from tests import seqsets
    rng = random.Random(7)
    seqs = []
    for alpha in (b"AC", b"ACGT", b"-ACGNT"):
        for _ in range(400):
            seqs.append(bytes(rng.choice(alpha) for _ in range(rng.randint(1, 80))))
    seqs += [b"", b"A", b"ATGCA", b"banana", b"ACGT" * 20]
    for s in seqs:
        assert O.lmsr_index_nth(s) == O.lmsr_index(s), s
    data, offs = seqsets.pack(seqs)
    exp, _ = O.canonicalize_batch(data, offs, True, False, threads=2)
    assert np.array_equal(O.canonicalize_batch_nth(data, offs, threads=2), exp)


def test_idempotence_properties():
    rng = random.Random(3)
    for _ in range(2000):
        s = bytes(rng.randint(0x20, 0x7E) for _ in range(rng.randint(1, 100)))
        assert O.lmsr(O.lmsr(s)) == O.lmsr(s)
        d = bytes(rng.choice(b"ATGC") for _ in range(rng.randint(1, 100)))
        assert O.canonicalize(O.canonicalize(d)) == O.canonicalize(d)


def test_rotation_and_strand_invariance():
    rng = random.Random(4)
    for _ in range(500):
        d = bytes(rng.choice(b"ACGTN-") for _ in range(rng.randint(1, 200)))
        k = rng.randrange(len(d))
        rot = d[k:] + d[:k]
        assert O.canonicalize(rot) == O.canonicalize(d)
        assert O.canonicalize(O.revcomp(d)) == O.canonicalize(d)


# -- third-party pieces ----------------------------------------------------------------------------
def test_revcomp_pinned_cases():
    assert O.revcomp(b"TTATG") == b"CATAA"
    assert O.lmsr(O.revcomp(b"TTATG")) == b"AACAT"          # multiple_sequences fixture
    assert O.revcomp(b"ACGTN-") == b"-NACGT"
    assert O.revcomp(b"acgtn") == b"nacgt"
    assert O.revcomp(b"YRWSKMDVHB") == b"VDBHKMSWYR"


def test_normalize_recalled_doc_examples():
    assert O.normalize(b"ACGTU") == (b"ACGTT", True)
    assert O.normalize(b"acgtu") == (b"ACGTT", True)
    assert O.normalize(b"N.N-N~N N") == (b"N-N-N-NN", True)
    assert O.normalize(b"BDHVRYSWKM") == (b"NNNNNNNNNN", True)
    assert O.normalize(b"ACGTN-") == (b"ACGTN-", False)
    assert O.normalize(b"AC\nGT\r\n") == (b"ACGT", True)


def test_xxh3_vectors():
    vecs = json.load(open(os.path.join(GOLDEN, "xxh3_vectors.json")))["vectors"]
    assert len(vecs) > 250
    for v in vecs:
        b = v["in"].encode() if "in" in v else v["in_latin1"].encode("latin-1")
        assert "%016x" % O.xxh3_64(b) == v["xxh3_64"], len(b)
    # the three values quoted in SURVEY.md 8c
    assert O.xxh3_64(b"") == 0x2d06800538d394c2
    assert O.xxh3_64(b"AAAAAAAT") == 0x420cfc80456509d5
    assert O.xxh3_64(b"A" * 1000) == 0xb58eaaea0d13a6fa


# -- the reference's CLI fixtures (tests/canon_uniq.rs:33-89) --------------------------------------
def _id_seq_map(data):
    """tests/common.rs:33-88: id -> sequence with line breaks removed (bio::io::fasta reader)."""
    return {O.record_id(h): s.replace(b"\n", b"").replace(b"\r", b"") for h, s in O.read_fasta(data)}


def _fixture(d, name):
    return open(os.path.join(GOLDEN, "ref_examples", d, name), "rb").read()


@pytest.mark.parametrize("d", ["simple", "multiple_sequences", "multiple_sequences_split_lines", "rna_input",
                               "compressed_input", "compressed_output"])
def test_cli_canonicalize_fixtures(d):
    got = O.cli_canonicalize(_fixture(d, "in.fasta"))
    assert _id_seq_map(got) == _id_seq_map(_fixture(d, "out.fasta"))


@pytest.mark.parametrize("d", ["simple", "multiple_sequences", "multiple_sequences_split_lines", "rna_input",
                               "repeated", "compressed_input", "compressed_output"])
def test_cli_uniq_canonicalize_fixtures(d):
    got, _ = O.cli_uniq(_fixture(d, "in.fasta"), canonical_out=True)
    assert _id_seq_map(got) == _id_seq_map(_fixture(d, "out.fasta"))


def test_cli_literal_output_text():
    # tests/canon_uniq.rs:24-29
    assert b">seq1\nAATGC" in O.cli_canonicalize(b">seq1\nATGCA")
    # header description is kept (simple/in.fasta), sequence on one line, trailing newline
    assert O.cli_canonicalize(_fixture("simple", "in.fasta")) == b">first sequence\nAAAAAAAT\n"


def test_cli_uniq_repeated_table_and_raw_output():
    data = _fixture("repeated", "in.fasta")
    fa, table = O.cli_uniq(data, canonical_out=False)
    assert fa == b">seq1\nAAAAAAAT\n"
    assert table == b"id,duplicate_id\nseq1,seq2\nseq1,rna\nseq1,lowercase\nseq1,opposite_polarity\n"


def test_batch_matches_single_and_threads():
    rng = np.random.default_rng(7)
    lens = rng.integers(0, 300, size=200)
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    data = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(offs[-1]))]
    out1, h1 = O.canonicalize_batch(data, offs, True, True, threads=1)
    out4, h4 = O.canonicalize_batch(data, offs, True, True, threads=4)
    assert np.array_equal(out1, out4) and np.array_equal(h1, h4)
    for i in range(len(lens)):
        a, b = int(offs[i]), int(offs[i + 1])
        c = O.canonicalize(data[a:b].tobytes())
        assert out1[a:b].tobytes() == c
        assert int(h1[i]) == O.xxh3_64(c)
    _, h_only = O.canonicalize_batch(data, offs, False, True, threads=2)
    assert np.array_equal(h_only, h1)


def test_uniq_first_seen():
    h = np.array([5, 7, 5, 9, 7, 5, 0, 0], dtype=np.uint64)
    assert O.uniq_first_seen(h).tolist() == [0, 1, 0, 3, 1, 0, 6, 6]


def test_synth_fill_is_counter_based():
    a = O.synth_fill(42, 0, 5000)
    b = O.synth_fill(42, 1234, 1000)
    assert np.array_equal(a[1234:2234], b)
    assert set(a.tobytes()) == set(b"ACGT")
    assert not np.array_equal(O.synth_fill(43, 0, 100), a[:100])
