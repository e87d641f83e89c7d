"""CPU-only: the C++ FASTA -> CSR packer (host logic behind circkit_fasta_parse) against the checker's
restatement of seq_io's record semantics and needletail's normalize."""
import os
import random

import numpy as np
import pytest

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_examples")


@pytest.fixture(scope="module")
def api():
    import __graft_entry__ as g
    g.build()
    from circkit_amd import api
    return api


def _check(api, text):
    recs, data, offs, consumed = api.fasta_parse(text)
    exp = O.read_fasta(text)
    assert recs == exp
    assert consumed == len(text)
    for i, (_, s) in enumerate(exp):
        assert data[int(offs[i]):int(offs[i + 1])].tobytes() == O.normalize(s)[0]


def test_reference_fixtures(api):
    files = [os.path.join(GOLDEN, d, "in.fasta") for d in
             ("simple", "multiple_sequences", "multiple_sequences_split_lines", "rna_input", "repeated",
              "compressed_input", "compressed_output")]
    files += [os.path.join(GOLDEN, "test.fasta"), os.path.join(GOLDEN, "nim_cated", "realistic_input.fasta")]
    for f in files:
        _check(api, open(f, "rb").read())


def test_edge_cases(api):
    for t in (b"", b">a", b">a\n", b">a\nACGT", b">a\nACGT\n", b">a b c\nAC\nGT\n>b\n\n>c\nA\n",
              b"\n\n>a\nAC\n", b">a\r\nAC\r\nGT\r\n>b\r\nTT\r\n", b">a\nAC>GT\n>b\nA\n", b">\nACGT\n"):
        _check(api, t)
    with pytest.raises(ValueError):
        api.fasta_parse(b"ACGT\n>a\nAC\n")


def test_random_texts_and_chunked_parsing(api):
    rng = random.Random(5)
    for _ in range(200):
        recs = []
        for _ in range(rng.randint(1, 12)):
            head = bytes(rng.choice(b"abcXYZ 01_|") for _ in range(rng.randint(0, 12)))
            seq = bytes(rng.choice(b"ACGTacgtuUNn-.~RYKM \t") for _ in range(rng.randint(0, 150)))
            w = rng.choice([0, 7, 60])
            if w:
                seq = b"\n".join(seq[i:i + w] for i in range(0, len(seq), w))
            recs.append(b">" + head + b"\n" + seq)
        text = b"\n".join(recs) + (b"\n" if rng.random() < 0.5 else b"")
        _check(api, text)
        # streamed in two chunks: the first call must stop at a record start
        cut = rng.randrange(len(text) + 1)
        r1, d1, o1, c1 = api.fasta_parse(text[:cut], True, False)
        assert c1 <= cut and (c1 == 0 or c1 == len(text[:cut]) or text[c1:c1 + 1] == b">" or c1 == cut)
        r2, d2, o2, c2 = api.fasta_parse(text[c1:], c1 == 0, True)
        assert r1 + r2 == O.read_fasta(text)
