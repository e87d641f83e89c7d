"""Sanitizer targets of the CPU builds.  The GPU pool offers no device sanitizers, so:
  * the C++ host packer (circkit_amd/csrc/fasta_host.cpp, the code behind circkit_fasta_parse and the CLI's parser threads)
    is built with g++ -fsanitize=address,undefined and fed the texts of tests/test_fasta_host.py -- fixtures, edge cases,
    200 random texts whole and cut in two -- each in a heap block of exactly its size; results are compared with the oracle;
    every whole text is ALSO parsed the way the CLI's parser pool does it since round 4 -- in 1, 2, 3, 5 and 16 sub-ranges,
    a thread per sub-range for the parse and again for the placement into one CSR -- and must give the identical batch;
  * the CPU emulator of the kernel source (tests/emu) is ALWAYS built with -fsanitize=undefined,bounds without recovery, so
    every run of tests/test_emu_kernel.py is a UBSan run; here only that the instrumentation is really in the library."""
import os
import random
import struct
import subprocess

import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden", "ref_examples")
SAN = os.path.join(HERE, "san")
BIN = os.path.join(SAN, "fasta_san")


@pytest.fixture(scope="module")
def fasta_san():
    srcs = [os.path.join(SAN, "fasta_san_main.cpp"), os.path.join(ROOT, "circkit_amd", "csrc", "fasta_host.cpp")]
    deps = srcs + [os.path.join(ROOT, "circkit_amd", "csrc", "fasta_host.h"), os.path.join(ROOT, "include", "circkit.h")]
    if not os.path.exists(BIN) or any(os.path.getmtime(d) > os.path.getmtime(BIN) for d in deps):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                               "-fno-omit-frame-pointer", "-pthread", "-o", BIN] + srcs)
    return BIN


@pytest.fixture(scope="module")
def fasta_tsan():
    """the same driver under ThreadSanitizer: the sub-range jobs of a chunk run as threads"""
    out = os.path.join(SAN, "fasta_tsan")
    srcs = [os.path.join(SAN, "fasta_san_main.cpp"), os.path.join(ROOT, "circkit_amd", "csrc", "fasta_host.cpp")]
    deps = srcs + [os.path.join(ROOT, "circkit_amd", "csrc", "fasta_host.h"), os.path.join(ROOT, "include", "circkit.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-o", out] + srcs)
    return out


def texts():
    cases = []
    for d in ("simple", "multiple_sequences", "multiple_sequences_split_lines", "rna_input", "repeated", "compressed_input", "compressed_output"):
        cases.append(open(os.path.join(GOLDEN, d, "in.fasta"), "rb").read())
    cases.append(open(os.path.join(GOLDEN, "test.fasta"), "rb").read())
    cases.append(open(os.path.join(GOLDEN, "nim_cated", "realistic_input.fasta"), "rb").read())
    cases += [b"", b">a", b">a\n", b">a\nACGT", b">a\nACGT\n", b">a b c\nAC\nGT\n>b\n\n>c\nA\n", b"\n\n>a\nAC\n",
              b">a\r\nAC\r\nGT\r\n>b\r\nTT\r\n", b">a\nAC>GT\n>b\nA\n", b">\nACGT\n", b">", b"\n", b">\n>", b">a\n>", b"\r\n\r\n"]
    # long runs of ACGTN (the packer's sixteen-bytes-at-a-time path) with one stranger at every offset of the first blocks, line
    # breaks at every width around the block size, and tails of every length behind the last full block
    rng = random.Random(6)
    pure = lambda n: bytes(rng.choice(b"ACGTN") for _ in range(n))
    for off in range(0, 35):
        for odd in (b"a", b"u", b"\n", b" ", b"-", b"R", b"\r\n"):
            cases.append(b">x\n" + pure(off) + odd + pure(70 - off) + b"\n>y z\n" + pure(off + 16))
    for w in (15, 16, 17, 31, 32, 33, 60):
        seq = pure(200)
        cases.append(b">w\n" + b"\n".join(seq[i:i + w] for i in range(0, len(seq), w)) + b"\n")
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
        cases.append(b"\n".join(recs) + (b"\n" if rng.random() < 0.5 else b""))
    return cases


def run_cases(binary, cases, tmp_path):
    """cases: [(first_chunk, final_chunk, text)] -> [(rc, consumed, [(head, raw)], [normalized])]"""
    path = tmp_path / "cases.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(cases)))
        for first, final, t in cases:
            f.write(struct.pack("<BBI", int(first), int(final), len(t)) + t)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([binary, str(path)], capture_output=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stderr == b"", r.stderr.decode(errors="replace")[-4000:]
    out, pos, res = r.stdout, 0, []
    for first, final, t in cases:
        rc, consumed, n = struct.unpack_from("<iQQ", out, pos)
        pos += 20
        spans = [struct.unpack_from("<QQQQ", out, pos + 32 * i) for i in range(n)]
        pos += 32 * n
        recs, norm = [], []
        if rc == 0:
            offs = struct.unpack_from("<%dQ" % (n + 1), out, pos)
            pos += 8 * (n + 1)
            payload = out[pos:pos + offs[n]]
            pos += offs[n]
            recs = [(t[a:a + b], t[c:c + d]) for a, b, c, d in spans]
            norm = [payload[offs[i]:offs[i + 1]] for i in range(n)]
        res.append((rc, consumed, recs, norm))
    assert pos == len(out)
    return res


def test_fasta_packer_under_asan_and_ubsan(fasta_san, tmp_path):
    whole = [(True, True, t) for t in texts()]
    for (_, _, t), (rc, consumed, recs, norm) in zip(whole, run_cases(fasta_san, whole, tmp_path)):
        exp = O.read_fasta(t)
        assert rc == 0 and consumed == len(t) and recs == exp
        assert norm == [O.normalize(s)[0] for _, s in exp]
    # streamed in two chunks: the first call stops at a record start, the second takes the rest
    rng = random.Random(7)
    cuts = [(t, rng.randrange(len(t) + 1)) for _, _, t in whole if t]
    first = run_cases(fasta_san, [(True, False, t[:c]) for t, c in cuts], tmp_path)
    second = run_cases(fasta_san, [(f[1] == 0, True, t[f[1]:]) for (t, c), f in zip(cuts, first)], tmp_path)
    for (t, c), a, b in zip(cuts, first, second):
        assert a[0] == 0 and b[0] == 0 and a[1] <= c
        assert a[2] + b[2] == O.read_fasta(t)
        assert a[3] + b[3] == [O.normalize(s)[0] for _, s in O.read_fasta(t)]
    # a text that does not start with '>' is an error, not a crash
    (rc, _, _, _), = run_cases(fasta_san, [(True, True, b"ACGT\n>a\nAC\n")], tmp_path)
    assert rc != 0


def test_sub_range_parse_under_tsan(fasta_tsan, tmp_path):
    """The chunk -> sub-ranges -> one CSR path of the CLI's parser pool with a thread per sub-range, under ThreadSanitizer:
    no data race between the parse jobs, nor between the placement jobs (they write disjoint parts of the chunk's batch), and
    the same batch as the single call for every split (the driver exits 3 otherwise)."""
    path = tmp_path / "cases.bin"
    cases = texts()[::3]
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(cases)))
        for t in cases:
            f.write(struct.pack("<BBI", 1, 1, len(t)) + t)
    r = subprocess.run([fasta_tsan, str(path)], capture_output=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and r.stderr == b"", r.stderr.decode(errors="replace")[-3000:]


def test_emulator_is_built_with_ubsan():
    from tests.emu import emu
    so = emu.build()
    assert "-fsanitize=undefined,bounds" in emu.SANITIZE
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    symbols = subprocess.run(["nm", "-D", so], capture_output=True, text=True).stdout
    assert "libubsan" in needed and "__ubsan_handle" in symbols
