"""ctypes wrapper over oracle/libcirckit_oracle.so + a Python restatement of the record loop.

TEST INFRASTRUCTURE ONLY (see circkit_oracle.c header): imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by circkit_amd/.

Reference call sites restated here:
  src/canonicalize.rs:17-46  record loop + output format  -> cli_canonicalize()
  src/uniq.rs:24-83          dedup loop, table rows        -> cli_uniq()
  seq_io 0.3.2 FASTA reader semantics  -> read_fasta()
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcirckit_oracle.so")


def build(force=False):
    """Compile the C restatement with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "circkit_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libcirckit_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u8p, u64p = ctypes.c_void_p, ctypes.c_void_p
        L.ck_oracle_lmsr_index.restype = ctypes.c_size_t
        L.ck_oracle_lmsr_index.argtypes = [u8p, ctypes.c_size_t]
        L.ck_oracle_lmsr_index_simple.restype = ctypes.c_size_t
        L.ck_oracle_lmsr_index_simple.argtypes = [u8p, ctypes.c_size_t]
        L.ck_oracle_lmsr.argtypes = [u8p, ctypes.c_size_t, u8p]
        L.ck_oracle_revcomp.argtypes = [u8p, ctypes.c_size_t, u8p]
        L.ck_oracle_canonicalize.argtypes = [u8p, ctypes.c_size_t, u8p]
        L.ck_oracle_normalize.restype = ctypes.c_size_t
        L.ck_oracle_normalize.argtypes = [u8p, ctypes.c_size_t, u8p, ctypes.POINTER(ctypes.c_int)]
        L.ck_oracle_xxh3_64.restype = ctypes.c_uint64
        L.ck_oracle_xxh3_64.argtypes = [u8p, ctypes.c_size_t]
        L.ck_oracle_canonicalize_batch.argtypes = [u8p, u64p, ctypes.c_uint64, u8p, u64p, ctypes.c_int]
        L.ck_oracle_canonicalize_batch_nth.argtypes = [u8p, u64p, ctypes.c_uint64, u8p, ctypes.c_int]
        L.ck_oracle_lmsr_index_nth.restype = ctypes.c_size_t
        L.ck_oracle_lmsr_index_nth.argtypes = [u8p, ctypes.c_size_t]
        L.ck_oracle_uniq_first_seen.argtypes = [u64p, ctypes.c_uint64, u64p]
        L.ck_oracle_synth_fill.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, u8p]
        L.ck_oracle_complement.restype = ctypes.c_uint8
        L.ck_oracle_complement.argtypes = [ctypes.c_uint8]
        _lib = L
    return _lib


def _buf(b):
    b = bytes(b)
    return ctypes.create_string_buffer(b, len(b) if len(b) else 1), len(b)


def lmsr_index(s):
    b, n = _buf(s)
    return lib().ck_oracle_lmsr_index(ctypes.addressof(b), n)


def lmsr_index_simple(s):
    b, n = _buf(s)
    return lib().ck_oracle_lmsr_index_simple(ctypes.addressof(b), n)


def _map(fn, s):
    b, n = _buf(s)
    out = ctypes.create_string_buffer(n if n else 1)
    fn(ctypes.addressof(b), n, ctypes.addressof(out))
    return out.raw[:n]


def lmsr(s):
    return _map(lib().ck_oracle_lmsr, s)


def revcomp(s):
    return _map(lib().ck_oracle_revcomp, s)


def canonicalize(s):
    return _map(lib().ck_oracle_canonicalize, s)


def normalize(s):
    """Returns (bytes, changed) -- `changed` False is the reference's None."""
    b, n = _buf(s)
    out = ctypes.create_string_buffer(n if n else 1)
    ch = ctypes.c_int(0)
    m = lib().ck_oracle_normalize(ctypes.addressof(b), n, ctypes.addressof(out), ctypes.byref(ch))
    return out.raw[:m], bool(ch.value)


def xxh3_64(s):
    b, n = _buf(s)
    return int(lib().ck_oracle_xxh3_64(ctypes.addressof(b), n))


def lmsr_index_nth(s):
    """lmsr_index with the reference's chars().nth() access cost (same answer, quadratic time)."""
    b, n = _buf(s)
    return lib().ck_oracle_lmsr_index_nth(ctypes.addressof(b), n)


def canonicalize_batch_nth(bytes_arr, offsets, threads=1):
    """canonicalize_batch through the quadratic cost model of the reference as written; returns out_bytes."""
    bytes_arr = np.ascontiguousarray(bytes_arr, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    out = np.empty(max(len(bytes_arr), 1), dtype=np.uint8)
    lib().ck_oracle_canonicalize_batch_nth(bytes_arr.ctypes.data if len(bytes_arr) else None, offsets.ctypes.data, len(offsets) - 1,
                                           out.ctypes.data, int(threads))
    return out[:len(bytes_arr)]


def canonicalize_batch(bytes_arr, offsets, want_bytes=True, want_hash=False, threads=1):
    """bytes_arr: uint8 ndarray; offsets: uint64 ndarray [n+1]. Returns (out_bytes|None, hashes|None)."""
    bytes_arr = np.ascontiguousarray(bytes_arr, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    out = np.empty(max(len(bytes_arr), 1), dtype=np.uint8) if want_bytes else None
    hs = np.empty(max(n, 1), dtype=np.uint64) if want_hash else None
    lib().ck_oracle_canonicalize_batch(
        bytes_arr.ctypes.data if len(bytes_arr) else None, offsets.ctypes.data, n,
        out.ctypes.data if out is not None else None,
        hs.ctypes.data if hs is not None else None, int(threads))
    return (out[:len(bytes_arr)] if out is not None else None,
            hs[:n] if hs is not None else None)


def uniq_first_seen(hashes):
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    fs = np.empty(max(len(hashes), 1), dtype=np.uint64)
    lib().ck_oracle_uniq_first_seen(hashes.ctypes.data, len(hashes), fs.ctypes.data)
    return fs[:len(hashes)]


def synth_fill(seed, first_base, n_bases):
    out = np.empty(max(n_bases, 1), dtype=np.uint8)
    lib().ck_oracle_synth_fill(seed, first_base, n_bases, out.ctypes.data)
    return out[:n_bases]


# ---------------------------------------------------------------------------------------------
# Record loop restatement (small inputs; plain Python).
# ---------------------------------------------------------------------------------------------
def read_fasta(data):
    """seq_io-style records: [(head, raw_seq)] where head is the header line minus '>' and line
    terminator, raw_seq is every byte between the header line and the next record start,
    interior line breaks kept, final line terminator dropped."""
    if not data:
        return []
    # seq_io skips leading empty lines before the first '>' [recalled]; anything else is an error
    i = 0
    while i < len(data) and data[i:i + 1] in (b"\n", b"\r"):
        i += 1
    if i >= len(data):
        return []
    if data[i:i + 1] != b">":
        raise ValueError("FASTA parse error: expected '>' at record start")
    recs = []
    pos = i
    n = len(data)
    while pos < n:
        eol = data.find(b"\n", pos)
        if eol < 0:
            head = data[pos + 1:]
            recs.append((head.rstrip(b"\r"), b""))
            break
        head = data[pos + 1:eol]
        if head.endswith(b"\r"):
            head = head[:-1]
        nxt = data.find(b"\n>", eol)
        if nxt < 0:
            seq = data[eol + 1:]
            pos = n
            if seq.endswith(b"\n"):
                seq = seq[:-1]
            if seq.endswith(b"\r"):
                seq = seq[:-1]
        else:
            seq = data[eol + 1:nxt]
            if seq.endswith(b"\r"):
                seq = seq[:-1]
            pos = nxt + 1
        recs.append((head, seq))
    return recs


def record_id(head):
    """seq_io Record::id(): head up to the first space (src/uniq.rs:48,67)."""
    return head.split(b" ", 1)[0]


def cli_canonicalize(data):
    """src/canonicalize.rs:17-46 on an in-memory FASTA; returns the output file bytes."""
    out = []
    for head, seq in read_fasta(data):
        norm, _ = normalize(seq)
        out.append(b">" + head + b"\n" + canonicalize(norm) + b"\n")
    return b"".join(out)


def full_seq(raw):
    """seq_io Record::full_seq(): the record's sequence lines joined, line terminators (\\n, \\r\\n) removed."""
    return b"".join(l[:-1] if l.endswith(b"\r") else l for l in raw.split(b"\n"))


def _read_until_error(data):
    """`while let Some(Ok(record)) = reader.next()`: a parse error ends the loop silently."""
    try:
        return read_fasta(data)
    except ValueError:
        return []


def cli_rotate(data, bases=None, percent=None):
    """src/rotate.rs:9-50 on an in-memory FASTA; returns the output file bytes."""
    if bases == 0 or percent == 0.0:
        raise ValueError("Rotation by 0 is not allowed")                    # :20-22
    out = []
    for head, raw in _read_until_error(data):
        seq = full_seq(raw)
        start = int(math.floor(len(seq) * percent)) if percent is not None else bases      # :26-29
        at = len(seq) - (start % len(seq)) if start >= 0 else (-start) % len(seq)          # :37-40 (ZeroDivisionError = the panic)
        out.append(b">" + head + b"\n" + seq[at:] + seq[:at] + b"\n")
    return b"".join(out)


def cli_cat(data):
    """src/concatenate.rs:10-32: every sequence written twice in a row."""
    return b"".join(b">" + h + b"\n" + full_seq(r) * 2 + b"\n" for h, r in _read_until_error(data))


def cli_decat(data):
    """src/concatenate.rs:34-54: the first half (len / 2, rounded down) of every sequence."""
    return b"".join(b">" + h + b"\n" + full_seq(r)[:len(full_seq(r)) // 2] + b"\n" for h, r in _read_until_error(data))


def cli_uniq(data, canonical_out=False, delimiter=b","):
    """src/uniq.rs:24-83; returns (fasta_bytes, table_bytes)."""
    seen = {}
    out, rows = [], []
    for head, seq in read_fasta(data):
        norm, _ = normalize(seq)
        canon = canonicalize(norm)
        h = xxh3_64(canon)
        rid = record_id(head)
        if h not in seen:
            seen[h] = rid
            out.append(b">" + head + b"\n" + (canon if canonical_out else seq) + b"\n")
        else:
            rows.append(seen[h] + delimiter + rid + b"\n")
    table = (b"id" + delimiter + b"duplicate_id\n" + b"".join(rows)) if rows else b""
    return b"".join(out), table
