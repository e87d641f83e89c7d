"""rotate / cat / decat: host-only subcommands of the CLI, so these run without a GPU.
Checked three ways: the oracle restatement against the reference's fixtures (compared the way the reference's own
tests do: id -> sequence, line breaks ignored, tests/common.rs:32-84), the CLI byte-for-byte against the oracle, and
the flag / error behaviour of src/rotate.rs and src/commands.rs."""
import os
import subprocess

import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_examples")
BIN = os.path.join(ROOT, "circkit_amd", "circkit")

CASES = [("rotate_5", ["rotate", "--bases", "5"], dict(bases=5)),
         ("rotate_minus_5", ["rotate", "--bases", "-5"], dict(bases=-5)),
         ("rotate_0.25", ["rotate", "--percent", "0.25"], dict(percent=0.25)),
         ("rotate_0.5", ["rotate", "--percent", "0.5"], dict(percent=0.5)),
         ("cat", ["cat"], None),
         ("decat", ["decat"], None)]


@pytest.fixture(scope="module", autouse=True)
def cli_binary():
    from circkit_amd import build
    build.build_all()
    assert os.path.exists(BIN)


def by_id(data):
    return {O.record_id(h): O.full_seq(s) for h, s in O.read_fasta(data)}


def oracle_out(args, kw, data):
    if args[0] == "rotate":
        return O.cli_rotate(data, **kw)
    return O.cli_cat(data) if args[0] == "cat" else O.cli_decat(data)


def run(*args, stdin=None):
    return subprocess.run([BIN, *args], input=stdin, capture_output=True, timeout=60)


@pytest.mark.parametrize("name,args,kw", CASES)
def test_oracle_matches_the_reference_fixture(name, args, kw):
    data = open(os.path.join(GOLDEN, name, "in.fasta"), "rb").read()
    want = open(os.path.join(GOLDEN, name, "out.fasta"), "rb").read()
    assert by_id(oracle_out(args, kw, data)) == by_id(want)


@pytest.mark.parametrize("name,args,kw", CASES)
def test_cli_matches_oracle_and_fixture(name, args, kw, tmp_path):
    src = os.path.join(GOLDEN, name, "in.fasta")
    data = open(src, "rb").read()
    out = tmp_path / "out.fasta"
    r = run(*args, src, "-o", str(out))
    assert (r.returncode, r.stdout, r.stderr) == (0, b"", b"")         # tests/common.rs:17-21
    got = out.read_bytes()
    assert got == oracle_out(args, kw, data)
    assert by_id(got) == by_id(open(os.path.join(GOLDEN, name, "out.fasta"), "rb").read())
    # stdin -> stdout gives the same bytes
    r = run(*args, stdin=data)
    assert r.returncode == 0 and r.stdout == got


def test_rotate_semantics():
    fa = b">a x\nACGT\nAC\n>b\nTTTTG\r\n"
    for kw, args in ((dict(bases=1), ["-b", "1"]), (dict(bases=-1), ["-b", "-1"]), (dict(bases=6), ["--bases=6"]),
                     (dict(bases=13), ["-b", "13"]), (dict(bases=-13), ["-b", "-13"]), (dict(percent=0.34), ["-p", "0.34"]),
                     (dict(percent=1.5), ["--percent=1.5"])):
        r = run("rotate", *args, stdin=fa)
        assert r.returncode == 0 and r.stdout == O.cli_rotate(fa, **kw), (kw, r.stdout)
    assert O.cli_rotate(fa, bases=1) == b">a x\nCACGTA\n>b\nGTTTT\n"       # right rotation: the last base comes first
    assert O.cli_rotate(fa, bases=-1) == b">a x\nCGTACA\n>b\nTTTGT\n"


def test_rotate_errors():
    r = run("rotate", "-b", "0", stdin=b">a\nACGT\n")
    assert r.returncode == 1 and r.stdout == b"" and b"Rotation by 0 is not allowed" in r.stderr      # src/rotate.rs:20-22
    r = run("rotate", "-p", "0", stdin=b">a\nACGT\n")
    assert r.returncode == 1 and b"Rotation by 0 is not allowed" in r.stderr
    with pytest.raises(ValueError):
        O.cli_rotate(b">a\nACGT\n", bases=0)
    r = run("rotate", "-b", "1", "-p", "0.5", stdin=b">a\nACGT\n")
    assert r.returncode == 2 and b"cannot be used with" in r.stderr                                     # src/commands.rs:164,170
    r = run("rotate", stdin=b">a\nACGT\n")
    assert r.returncode == 101 and b"Must provide either --bases or --percent" in r.stderr             # src/rotate.rs:29 (panic)
    r = run("rotate", "-b", "x", stdin=b">a\nACGT\n")
    assert r.returncode == 2


def test_cat_decat_edge_cases():
    for fa in (b"", b">only header", b">e\n\n>odd\nACGTA\n>two lines\nAC\nGT", b"no header\nACGT\n"):
        r = run("cat", stdin=fa)
        assert r.returncode == 0 and r.stdout == O.cli_cat(fa), fa
        r = run("decat", stdin=fa)
        assert r.returncode == 0 and r.stdout == O.cli_decat(fa), fa
    assert O.cli_cat(b">odd\nACGTA\n") == b">odd\nACGTAACGTA\n"
    assert O.cli_decat(b">odd\nACGTA\n") == b">odd\nAC\n"
    assert O.cli_decat(O.cli_cat(b">x\nAC\nGT\n")) == b">x\nACGT\n"


def test_unknown_subcommand_and_flags():
    assert run("monomerize").returncode == 2
    assert run("cat", "--threads", "2", stdin=b">a\nA\n").returncode == 2        # cat takes no --threads (src/commands.rs:92-99)


def test_zstd_streams_without_the_zstd_binary(tmp_path):
    """.zst in and out go through libzstd.so.1 (dlopen) in a forked filter, level 1 like src/utils.rs:58-60; the
    reference's compressed fixture decodes, our output decodes with an independent zstd (pyarrow's)."""
    import numpy as np
    import pyarrow as pa
    src = os.path.join(GOLDEN, "compressed_input", "in.fasta.zst")
    plain = open(os.path.join(GOLDEN, "compressed_input", "in.fasta"), "rb").read()
    r = run("cat", src)
    assert r.returncode == 0 and r.stdout == O.cli_cat(plain)
    # a multi-megabyte stream both ways
    rng = np.random.default_rng(2)
    big = b"".join(b">r%d\n" % i + bytes(rng.choice(list(b"ACGT"), size=5000).astype(np.uint8)) + b"\n" for i in range(600))
    (tmp_path / "big.fasta").write_bytes(big)
    out = tmp_path / "big_cat.fasta.zst"
    r = run("cat", str(tmp_path / "big.fasta"), "-o", str(out))
    assert (r.returncode, r.stdout, r.stderr) == (0, b"", b"")
    raw = out.read_bytes()
    assert raw[:4] == b"\x28\xb5\x2f\xfd" and len(raw) < len(big)
    assert pa.input_stream(pa.BufferReader(raw), compression="zstd").read() == O.cli_cat(big)
    r = run("decat", str(out))                                  # and back in through the sniffer
    assert r.returncode == 0 and r.stdout == big
    # a truncated stream is an error, not a silent short read
    (tmp_path / "cut.fasta.zst").write_bytes(raw[:len(raw) // 2])
    r = run("cat", str(tmp_path / "cut.fasta.zst"))
    assert r.returncode != 0


@pytest.mark.parametrize("ext", ["gz", "bz2", "xz", "zst"])
def test_compressed_stdin_is_sniffed(ext, tmp_path):
    """src/utils.rs:21-24: niffler wraps stdin too, so `circkit cat < in.fasta.gz` and `... | circkit cat` both
    decode.  Redirected file (seekable: rewound) and pipe (a feeder child replays the sniffed bytes); a 3 MB stream as
    well, and short plain inputs whose first bytes the sniffer consumed."""
    import gzip, bz2, lzma
    import numpy as np
    src = os.path.join(GOLDEN, "compressed_input", "in.fasta." + ext)
    plain = open(os.path.join(GOLDEN, "compressed_input", "in.fasta"), "rb").read()
    want = O.cli_cat(plain)
    with open(src, "rb") as f:                                   # redirected file
        r = subprocess.run([BIN, "cat"], stdin=f, capture_output=True, timeout=60)
    assert (r.returncode, r.stdout) == (0, want), r.stderr
    r = run("cat", stdin=open(src, "rb").read())                 # pipe
    assert (r.returncode, r.stdout) == (0, want), r.stderr
    rng = np.random.default_rng(5)
    big = b"".join(b">r%d\n" % i + bytes(rng.choice(list(b"ACGT"), size=5000).astype(np.uint8)) + b"\n" for i in range(600))
    if ext == "zst":
        import pyarrow as pa
        sink = pa.BufferOutputStream()
        with pa.CompressedOutputStream(sink, "zstd") as z:
            z.write(big)
        packed = sink.getvalue().to_pybytes()
    else:
        packed = {"gz": gzip.compress, "bz2": bz2.compress, "xz": lzma.compress}[ext](big)
    r = run("cat", stdin=packed)
    assert (r.returncode, r.stdout) == (0, O.cli_cat(big)), r.stderr[:300]


def test_plain_stdin_keeps_the_sniffed_bytes():
    for data in (b">a\nA\n", b">ab\nACGT\n>c\nTT", b">x\n"):
        r = run("cat", stdin=data)
        assert (r.returncode, r.stdout) == (0, O.cli_cat(data)), (data, r.stderr)


# ---- clap's accepted argument forms (src/commands.rs:6-14,93-180) on the host-only arms; the GPU arms share the parser
# (tests/test_cli_gpu.py runs the same forms through canonicalize / uniq on the GPU box) ----
CAT_IN = os.path.join(GOLDEN, "cat", "in.fasta")


@pytest.mark.parametrize("form", [
    lambda out: ["cat", CAT_IN, "-o" + out],                      # -oFILE
    lambda out: ["cat", CAT_IN, "-o=" + out],                     # -o=FILE
    lambda out: ["cat", CAT_IN, "--output=" + out],
    lambda out: ["cat", "--output", out, CAT_IN],
    lambda out: ["cat", "-o", out, "--", CAT_IN],                 # `--` ends the options
    lambda out: ["-v", "cat", CAT_IN, "-o", out],                 # clap_verbosity_flag: global, before ...
    lambda out: ["cat", "-q", CAT_IN, "-o", out],                 # ... or after the subcommand
    lambda out: ["-vv", "cat", "-qo" + out, CAT_IN],              # repeated, and combined with a valued short
    lambda out: ["--verbose", "cat", "--quiet", CAT_IN, "-o", out],
])
def test_clap_forms_on_a_host_arm(form, tmp_path):
    out = str(tmp_path / "o.fasta")
    r = run(*form(out))
    assert (r.returncode, r.stdout, r.stderr) == (0, b"", b""), r.stderr
    assert open(out, "rb").read() == O.cli_cat(open(CAT_IN, "rb").read())


@pytest.mark.parametrize("args", [["rotate", "-b5"], ["rotate", "-b=5"], ["rotate", "--bases=5"], ["rotate", "-vb", "5"], ["rotate", "-qb5"]])
def test_clap_forms_of_a_valued_short(args, tmp_path):
    src = os.path.join(GOLDEN, "rotate_5", "in.fasta")
    r = run(*args, src)
    assert r.returncode == 0 and r.stdout == O.cli_rotate(open(src, "rb").read(), bases=5)


@pytest.mark.parametrize("args,needle", [
    (["cat", "-x"], b"unexpected argument '-x'"),
    (["cat", "--frobnicate"], b"unexpected argument '--frobnicate'"),
    (["cat", "a.fa", "b.fa"], b"unexpected argument 'b.fa'"),
    (["cat", "-o"], b"requires a value"),
    (["cat", "--quiet=3"], b"unexpected value"),
    (["canonicalize", "-tx", "a.fa"], b"invalid value 'x' for '--threads <THREADS>'"),
    (["uniq", "-ctfour", "a.fa"], b"invalid value 'four'"),
    (["cat", "-c"], b"unexpected argument '-c'"),                # -c belongs to uniq only
    (["rotate", "-b", "1", "-p", "0.5"], b"cannot be used with"),
])
def test_clap_errors_exit_2(args, needle):
    r = run(*args)
    assert r.returncode == 2 and r.stdout == b"" and needle in r.stderr, r.stderr


def test_version_and_help():
    for a in (["--version"], ["-V"]):
        r = run(*a)
        assert r.returncode == 0 and r.stdout.startswith(b"circkit ")
    r = run("canonicalize", "--help")
    assert r.returncode == 0 and b"USAGE" in r.stdout
