"""GPU: the `circkit` binary (C++ host + C ABI + HIP kernels) against the reference's CLI tests
(tests/canon_uniq.rs, tests/compression.rs) and its fixture files."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_examples")
BIN = os.path.join(ROOT, "circkit_amd", "circkit")


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(BIN)


def run(*args, stdin=None):
    return subprocess.run([BIN] + list(args), input=stdin, capture_output=True, timeout=120)


def id_seq_map(data):
    """tests/common.rs:33-88 (bio::io::fasta): id -> sequence with line breaks removed."""
    from oracle import oracle as O
    return {O.record_id(h): s.replace(b"\n", b"").replace(b"\r", b"") for h, s in O.read_fasta(data)}


def fixture(d, name):
    return os.path.join(GOLDEN, d, name)


def test_file_doesnt_exist():
    r = run("canonicalize", "test/file/doesnt/exist")          # tests/canon_uniq.rs:8-17
    assert r.returncode != 0
    assert b"No such file or directory" in r.stderr


def test_simple_fasta_file_to_stdout(tmp_path):
    f = tmp_path / "simple.fasta"                               # tests/canon_uniq.rs:19-31
    f.write_bytes(b">seq1\nATGCA")
    r = run("canonicalize", str(f))
    assert r.returncode == 0 and b">seq1\nAATGC" in r.stdout


def test_stdin_to_stdout():
    r = run("canonicalize", stdin=b">a desc\nTT\nATG\n>b\nAATGGA")
    assert r.returncode == 0 and r.stdout == b">a desc\nAACAT\n>b\nAAATGG\n" and r.stderr == b""


@pytest.mark.parametrize("command,directory", [
    ("canonicalize", "simple"), ("uniq", "simple"),
    ("canonicalize", "multiple_sequences"), ("uniq", "multiple_sequences"),
    ("canonicalize", "multiple_sequences_split_lines"), ("uniq", "multiple_sequences_split_lines"),
    ("canonicalize", "rna_input"), ("uniq", "rna_input"),
    ("uniq", "repeated")])
@pytest.mark.parametrize("threads", ["1", "4"])
def test_fasta_files(command, directory, threads, tmp_path):
    out = tmp_path / "out.fasta"                                # tests/canon_uniq.rs:33-89
    args = [command, fixture(directory, "in.fasta"), "--threads", threads, "-o", str(out)]
    if command == "uniq":
        args.append("--canonicalize")
    r = run(*args)
    assert r.returncode == 0 and r.stdout == b"" and r.stderr == b""
    assert id_seq_map(out.read_bytes()) == id_seq_map(open(fixture(directory, "out.fasta"), "rb").read())


@pytest.mark.parametrize("command", ["canonicalize", "uniq"])
@pytest.mark.parametrize("ext,tool", [("gz", "gzip"), ("bz2", "bzip2"), ("xz", "xz"), ("zst", "zstd")])
def test_compressed_output(command, ext, tool, tmp_path):
    if tool != "zstd" and not shutil.which(tool):               # tests/compression.rs:8-69; zstd goes through libzstd
        pytest.skip("%s is not installed on this box" % tool)
    out = tmp_path / ("out.fasta." + ext)
    args = [command, fixture("compressed_output", "in.fasta"), "-o", str(out)] + (["--canonicalize"] if command == "uniq" else [])
    r = run(*args)
    assert r.returncode == 0 and r.stdout == b"" and r.stderr == b""
    if tool == "zstd":
        import pyarrow as pa
        raw = out.read_bytes()
        assert raw[:4] == b"\x28\xb5\x2f\xfd"
        (tmp_path / "out.fasta").write_bytes(pa.input_stream(pa.BufferReader(raw), compression="zstd").read())
    else:
        subprocess.check_call([tool, "-d", str(out)])
    assert id_seq_map((tmp_path / "out.fasta").read_bytes()) == id_seq_map(open(fixture("compressed_output", "out.fasta"), "rb").read())


@pytest.mark.parametrize("command", ["canonicalize", "uniq"])
@pytest.mark.parametrize("ext,tool", [("gz", "gzip"), ("bz2", "bzip2"), ("xz", "xz"), ("zst", "zstd")])
def test_compressed_input(command, ext, tool, tmp_path):
    if tool != "zstd" and not shutil.which(tool):               # tests/compression.rs:71-119; zstd goes through libzstd
        pytest.skip("%s is not installed on this box" % tool)
    out = tmp_path / "out.fasta"
    args = [command, fixture("compressed_input", "in.fasta." + ext), "-o", str(out)] + (["--canonicalize"] if command == "uniq" else [])
    r = run(*args)
    assert r.returncode == 0 and r.stdout == b"" and r.stderr == b""
    assert id_seq_map(out.read_bytes()) == id_seq_map(open(fixture("compressed_input", "out.fasta"), "rb").read())


@pytest.mark.parametrize("ext,tool", [("gz", "gzip"), ("bz2", "bzip2"), ("xz", "xz"), ("zst", "zstd")])
def test_compressed_stdin(ext, tool, tmp_path):
    """src/utils.rs:21-24: stdin goes through niffler too -- `circkit canonicalize < in.fasta.gz` and a pipe."""
    if tool != "zstd" and not shutil.which(tool):
        pytest.skip("%s is not installed on this box" % tool)
    want = id_seq_map(open(fixture("compressed_input", "out.fasta"), "rb").read())
    src = fixture("compressed_input", "in.fasta." + ext)
    with open(src, "rb") as f:
        r = subprocess.run([BIN, "canonicalize"], stdin=f, capture_output=True, timeout=120)
    assert r.returncode == 0 and id_seq_map(r.stdout) == want, r.stderr
    r = run("canonicalize", stdin=open(src, "rb").read())
    assert r.returncode == 0 and id_seq_map(r.stdout) == want, r.stderr


def test_uniq_table_and_raw_output(tmp_path):
    """src/uniq.rs:50-70: without --canonicalize the kept record prints its RAW bytes; duplicates become table rows."""
    from oracle import oracle as O
    data = open(fixture("repeated", "in.fasta"), "rb").read()
    for ext, delim in (("csv", b","), ("tsv", b"\t")):
        out, table = tmp_path / ("o." + ext + ".fasta"), tmp_path / ("t." + ext)
        r = run("uniq", fixture("repeated", "in.fasta"), "-o", str(out), "--table", str(table))
        assert r.returncode == 0 and r.stderr == b""
        exp_fa, exp_table = O.cli_uniq(data, canonical_out=False, delimiter=delim)
        assert out.read_bytes() == exp_fa
        assert table.read_bytes() == exp_table
    # no duplicates -> empty table file (src/commands.rs:142)
    out, table = tmp_path / "o2.fasta", tmp_path / "t2.csv"
    r = run("uniq", fixture("multiple_sequences", "in.fasta"), "-o", str(out), "--table", str(table))
    assert r.returncode == 0 and table.read_bytes() == b""


def test_whole_file_outputs_match_the_record_loop_restatement(tmp_path):
    """Byte-for-byte output files on a realistic input (676 records) for canonicalize and both uniq modes."""
    from oracle import oracle as O
    src = os.path.join(GOLDEN, "nim_cated", "realistic_input.fasta")
    data = open(src, "rb").read()
    out = tmp_path / "c.fasta"
    assert run("canonicalize", src, "-o", str(out)).returncode == 0
    assert out.read_bytes() == O.cli_canonicalize(data)
    # duplicate the file onto itself with rotated copies so uniq has work to do
    recs = O.read_fasta(data)
    twice = data + b"".join(b">dup_" + h + b"\n" + s[7:] + s[:7] + b"\n" for h, s in recs[:200])
    f2 = tmp_path / "twice.fasta"
    f2.write_bytes(twice)
    for canon in (True, False):
        o, t = tmp_path / "u.fasta", tmp_path / "u.csv"
        args = ["uniq", str(f2), "-o", str(o), "--table", str(t)] + (["-c"] if canon else [])
        assert run(*args).returncode == 0
        exp_fa, exp_t = O.cli_uniq(twice, canonical_out=canon)
        assert o.read_bytes() == exp_fa
        assert t.read_bytes() == exp_t


def test_not_fasta_is_an_error(tmp_path):
    f = tmp_path / "x.txt"
    f.write_bytes(b"ACGT\n")
    r = run("canonicalize", str(f))
    assert r.returncode != 0 and b"Error" in r.stderr


def test_large_output_file_vs_stream_vs_oracle(tmp_path):
    """30k records (two line-wrapped, some with N, duplicates for uniq): a regular output file goes out as whole chunks
    assembled by a pool of threads and written in order by one (circkit_cli.cpp, stage 4), stdout as one ordered writev
    stream -- both must equal the record-loop restatement byte for byte, for canonicalize and for both uniq modes;
    appending (`>>`) must append."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    recs = []
    for i in range(30000):
        L = int(rng.integers(1, 1400)) if i % 3 else 1000
        s = bytes(rng.choice(list(b"ACGT" if i % 17 else b"ACGTN"), size=L).astype(np.uint8))
        if i % 5 == 0 and i:
            s = recs[int(rng.integers(0, len(recs)))][1]            # duplicate (same orientation is enough here)
        recs.append((b"r%d some text" % i, s))
    wrap = lambda s: b"\n".join(s[k:k + 70] for k in range(0, len(s), 70))
    data = b"".join(b">" + h + b"\n" + (wrap(s) if i % 2 else s) + b"\n" for i, (h, s) in enumerate(recs))
    src = tmp_path / "in.fasta"
    src.write_bytes(data)
    want_c = O.cli_canonicalize(data)
    for args, want in ((["canonicalize"], want_c), (["uniq"], O.cli_uniq(data)[0]), (["uniq", "-c"], O.cli_uniq(data, True)[0])):
        out = tmp_path / "out.fasta"
        r = run(*args, str(src), "-o", str(out))
        assert r.returncode == 0, r.stderr
        assert out.read_bytes() == want, args
        r = run(*args, str(src))
        assert r.returncode == 0 and r.stdout == want, args
    # shell append: positioned writes are not used on O_APPEND descriptors
    out = tmp_path / "app.fasta"
    out.write_bytes(b">existing\nAC\n")
    with open(out, "ab") as f:
        r = subprocess.run([BIN, "canonicalize", str(src)], stdout=f, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0
    assert out.read_bytes() == b">existing\nAC\n" + want_c
    # plain redirect into a file that already has content at the current offset: continue from there
    with open(out, "wb") as f:
        f.write(b">first\nGG\n")
        f.flush()
        r = subprocess.run([BIN, "canonicalize", str(src)], stdout=f, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0
    assert out.read_bytes() == b">first\nGG\n" + want_c


def test_record_longer_than_a_pipeline_chunk(tmp_path):
    """One 70 Mb record (line-wrapped) between ordinary ones: longer than the CLI's 64 MiB text chunk (the reader
    grows the chunk) and far beyond the LDS tiers (finished in global scratch) -- mapped-file and stdin input."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(8)
    big = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 70_000_000)]
    lines = big.reshape(-1, 100)
    wrapped = np.concatenate([lines, np.full((lines.shape[0], 1), 10, dtype=np.uint8)], axis=1).tobytes()
    small = [bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(30, 1200))).astype(np.uint8)) for _ in range(50)]
    data = b"".join(b">s%d\n" % i + s + b"\n" for i, s in enumerate(small[:25])) + b">big one\n" + wrapped + \
        b"".join(b">t%d\n" % i + s + b"\n" for i, s in enumerate(small[25:]))
    src = tmp_path / "in.fasta"
    src.write_bytes(data)
    want = b"".join(b">s%d\n" % i + O.canonicalize(s) + b"\n" for i, s in enumerate(small[:25])) + \
        b">big one\n" + O.canonicalize(big.tobytes()) + b"\n" + \
        b"".join(b">t%d\n" % i + O.canonicalize(s) + b"\n" for i, s in enumerate(small[25:]))
    out = tmp_path / "out.fasta"
    r = run("canonicalize", str(src), "-o", str(out))
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == want
    r = subprocess.run([BIN, "uniq", "-c"], input=data, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == want                                  # all records distinct: uniq -c prints the same file


@pytest.mark.parametrize("slots,chunk_mb,sub_kb", [("2", "1", None), ("3", "2", "64"), ("32", "1", "1"), ("6", "64", None), ("6", "64", "16"), ("4", "4", "300")])
def test_chunk_ring_geometry(tmp_path, slots, chunk_mb, sub_kb):
    """CIRCKIT_CLI_SLOTS x CIRCKIT_CLI_CHUNK_MB (the ring of chunks in flight: reader -> parsers -> device -> emit -> the
    writer thread): a 9 MB input with records that straddle chunk ends -- among them one longer than a chunk -- gives the
    same bytes for every geometry, for canonicalize into a file, uniq into a pipe and uniq --table.  sub_kb (CIRCKIT_CLI_SUB_KB):
    the size of the sub-ranges a chunk is parsed in by several threads before they are placed into one CSR (round 4) -- down to
    sub-ranges smaller than a record, where most of them come out empty."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    recs = []
    for i in range(9000):
        L = 2_500_000 if i == 4000 else int(rng.integers(1, 2000))
        s = bytes(rng.choice(list(b"ACGT"), size=L).astype(np.uint8))
        if i % 7 == 0 and i:
            s = recs[int(rng.integers(0, len(recs)))][1]
        recs.append((b"r%d" % i, s))
    data = b"".join(b">" + h + b"\n" + s + b"\n" for h, s in recs)
    src = tmp_path / "in.fasta"
    src.write_bytes(data)
    env = dict(os.environ, CIRCKIT_CLI_SLOTS=slots, CIRCKIT_CLI_CHUNK_MB=chunk_mb)
    if sub_kb:
        env["CIRCKIT_CLI_SUB_KB"] = sub_kb
    out = tmp_path / "out.fasta"
    r = subprocess.run([BIN, "canonicalize", str(src), "-o", str(out)], capture_output=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == O.cli_canonicalize(data)
    want_u, want_t = O.cli_uniq(data, True, delimiter=b"\t")
    table = tmp_path / "dups.tsv"
    r = subprocess.run([BIN, "uniq", "-c", str(src), "--table", str(table)], capture_output=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout == want_u
    assert table.read_bytes() == want_t


# ---- clap's accepted argument forms on the two GPU arms (src/commands.rs:6-14,112-149) ----
@pytest.mark.parametrize("form", [
    lambda i, o: ["uniq", i, "-ct4", "-o" + o],                  # combined shorts with a trailing valued one; -oFILE
    lambda i, o: ["uniq", "-c", "-t=4", "-o=" + o, i],
    lambda i, o: ["uniq", "--canon", "--threads=2", "--output=" + o, "--", i],
    lambda i, o: ["-q", "uniq", "--norm", "-vt2", i, "-o", o],   # global -v / -q on either side of the subcommand
    lambda i, o: ["uniq", "-qqc", i, "--output", o],
])
def test_clap_forms_uniq(form, tmp_path):
    out = tmp_path / "out.fasta"
    r = run(*form(fixture("repeated", "in.fasta"), str(out)))
    assert r.returncode == 0 and r.stdout == b"" and r.stderr == b"", r.stderr
    assert id_seq_map(out.read_bytes()) == id_seq_map(open(fixture("repeated", "out.fasta"), "rb").read())


@pytest.mark.parametrize("args", [["canonicalize", "-q"], ["-v", "canonicalize", "-t1"], ["canonicalize", "--quiet", "-t", "2"], ["-vvv", "canonicalize"]])
def test_clap_forms_canonicalize(args):
    r = run(*args, fixture("multiple_sequences", "in.fasta"))
    assert r.returncode == 0 and r.stderr == b""
    assert id_seq_map(r.stdout) == id_seq_map(open(fixture("multiple_sequences", "out.fasta"), "rb").read())


def test_supervisor_and_worker(tmp_path):
    """The GPU arms run in a worker process; the process the caller started is a supervisor that leaves with the worker's status
    as soon as the worker reports that all output is written and closed (the worker's 768 MB of pinned buffers, its mapping of
    the input and its GPU context are then taken apart in the background: 0.3 s of a 5 GB run's wall time).  Same bytes and
    same exit status with CIRCKIT_CLI_NO_SUPERVISOR=1 (one process); an error exit of the worker is the supervisor's; a
    consumer on a pipe gets all of the output and its end of file; SIGTERM to the supervisor ends the worker."""
    import signal
    import time
    src = fixture("multiple_sequences", "in.fasta")
    out_a, out_b = tmp_path / "a.fasta", tmp_path / "b.fasta"
    assert run("canonicalize", src, "-o", str(out_a)).returncode == 0
    r = subprocess.run([BIN, "canonicalize", src, "-o", str(out_b)], capture_output=True, timeout=120, env=dict(os.environ, CIRCKIT_CLI_NO_SUPERVISOR="1"))
    assert r.returncode == 0 and out_a.read_bytes() == out_b.read_bytes() and out_a.stat().st_size > 0
    for env in (dict(os.environ), dict(os.environ, CIRCKIT_CLI_NO_SUPERVISOR="1")):
        bad = tmp_path / "not.fasta"
        bad.write_bytes(b"this is not FASTA\n")
        r = subprocess.run([BIN, "uniq", str(bad)], capture_output=True, timeout=120, env=env)
        assert r.returncode == 1 and r.stderr                                            # (the worker's die())
    # a pipe consumer: complete output, and `cat` ends (the worker closes its stdout before it reports)
    p = subprocess.run("%s canonicalize %s | cat" % (BIN, src), shell=True, capture_output=True, timeout=60)
    assert p.returncode == 0 and p.stdout == out_a.read_bytes()
    # signals are passed on: a run reading an endless stdin is ended by SIGTERM to the process we started
    proc = subprocess.Popen([BIN, "canonicalize"], stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    proc.stdin.write(b">r\nACGTACGTAC\n")
    proc.stdin.flush()
    time.sleep(1.0)
    proc.send_signal(signal.SIGTERM)
    assert proc.wait(timeout=30) in (-signal.SIGTERM, 128 + signal.SIGTERM)
    proc.stdin.close()
