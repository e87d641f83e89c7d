"""Host-side mirror of the reference's lib-crate API over the C ABI (include/circkit.h).

    circkit::lmsr_index(&[u8]) -> usize      lib/src/canonicalize.rs:5    -> lmsr_index(b)
    circkit::lmsr(&[u8]) -> Vec<u8>          lib/src/canonicalize.rs:41   -> lmsr(b)
    circkit::canonicalize(&[u8]) -> Vec<u8>  lib/src/canonicalize.rs:54   -> canonicalize(b)
    xxhash_rust::xxh3::xxh3_64               call site src/uniq.rs:45     -> xxh3_64(b)

Everything computes on the GPU through libcirckit_hip.so; there is no CPU fallback -- importing works
without a GPU (so the ABI can be inspected), creating a Context does not.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CIRCKIT_LIB: another build of the same library (tools/build_variant.sh: A/B timing, the poison-check build of tests/) --
# selected here instead of being copied over the in-tree file, which build.py's mtime check would then take for current
LIB_PATH = os.environ.get("CIRCKIT_LIB") or os.path.join(_HERE, "libcirckit_hip.so")

OK = 0
ERRORS = {-1: "INVALID_ARG", -2: "NO_DEVICE", -3: "HIP", -4: "TOO_LONG", -5: "OOM", -6: "NOT_ASCII"}

# every symbol include/circkit.h declares, with ctypes signatures
_vp, _u64, _u32, _i, _sz = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_size_t
SIGNATURES = {
    "circkit_ctx_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "circkit_ctx_destroy": (_i, [_vp]),
    "circkit_last_error": (ctypes.c_char_p, [_vp]),
    "circkit_ctx_set_stream": (_i, [_vp, _vp]),
    "circkit_ctx_use_own_stream": (_i, [_vp]),
    "circkit_ctx_synchronize": (_i, [_vp]),
    "circkit_ctx_last_kernel_ms": (_i, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "circkit_ctx_batch_status": (_i, [_vp, ctypes.POINTER(_u32)]),
    "circkit_ctx_set_long_record_scratch": (_i, [_vp, _u64]),
    "circkit_ctx_last_batch_mode": (_i, [_vp, ctypes.POINTER(_u32)]),
    "circkit_canonicalize_batch_device": (_i, [_vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    "circkit_lmsr_batch_device": (_i, [_vp, _vp, _vp, _u64, _vp, _vp]),
    "circkit_xxh3_batch_device": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "circkit_canonicalize_batch": (_i, [_vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    "circkit_host_alloc": (_vp, [_sz]),
    "circkit_host_free": (None, [_vp]),
    "circkit_lmsr_index": (_i, [_vp, _vp, _sz, ctypes.POINTER(_sz)]),
    "circkit_lmsr": (_i, [_vp, _vp, _sz, _vp]),
    "circkit_canonicalize": (_i, [_vp, _vp, _sz, _vp]),
    "circkit_xxh3_64": (_i, [_vp, _vp, _sz, ctypes.POINTER(_u64)]),
    "circkit_uniq_reset": (_i, [_vp, _u64]),
    "circkit_uniq_insert_device": (_i, [_vp, _vp, _u64, _u64]),
    "circkit_uniq_insert_pairs_device": (_i, [_vp, _vp, _vp, _u64]),
    "circkit_uniq_partition_device": (_i, [_vp, _vp, _u64, _u64, ctypes.c_uint32, _vp, _vp, _vp]),
    "circkit_uniq_insert_rows_device": (_i, [_vp, _vp, _u64]),
    "circkit_uniq_lookup_rows_device": (_i, [_vp, _vp, _u64, _vp]),
    "circkit_uniq_gather_device": (_i, [_vp, _vp, _vp, _u64, _u64, _vp, _vp]),
    "circkit_uniq_lookup_device": (_i, [_vp, _vp, _u64, _vp]),
    "circkit_uniq_status": (_i, [_vp, ctypes.POINTER(_u32)]),
    "circkit_uniq_resolve_device": (_i, [_vp, _vp, _u64, _u64, _vp, _vp]),
    "circkit_uniq_first_seen": (_i, [_vp, _vp, _u64, _u64, _vp]),
    "circkit_fasta_parse": (_i, [_vp, _sz, _i, _i, ctypes.POINTER(_vp), ctypes.POINTER(_sz)]),
    "circkit_fasta_error": (ctypes.c_char_p, [_vp]),
    "circkit_fasta_n_records": (_u64, [_vp]),
    "circkit_fasta_bytes": (_vp, [_vp]),
    "circkit_fasta_offsets": (_vp, [_vp]),
    "circkit_fasta_record": (_i, [_vp, _u64] + [ctypes.POINTER(_sz)] * 4),
    "circkit_fasta_free": (None, [_vp]),
    "circkit_synth_fill_device": (_i, [_vp, _u64, _u64, _u64, _vp]),
    "circkit_fixed_offsets_device": (_i, [_vp, _u64, _u64, _u64, _vp]),
    "circkit_bench_copy_device": (_i, [_vp, _vp, _vp, _u64, _u32]),
    "circkit_normalize": (_sz, [_vp, _sz, _vp, ctypes.POINTER(_i)]),
    "circkit_version": (ctypes.c_char_p, []),
}

_lib = None


class CirckitError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("circkit error %s (%d): %s" % (ERRORS.get(code, "?"), code, msg))
        self.code = code


def load_library():
    """Loads the HIP shared library; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Loading
        # torch first makes our NEEDED libamdhip64.so.7 bind to that copy instead of pulling in /opt/rocm's
        # as a second runtime (two runtimes in one process cannot both own the GPU).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)      # AttributeError here = header/library mismatch
            except AttributeError:
                if os.environ.get("CIRCKIT_LIB"):    # an explicitly chosen variant (e.g. an older round's build in an A/B run) may lack newer symbols
                    import warnings
                    warnings.warn("CIRCKIT_LIB=%s lacks %s (a stale variant build?): calls of it will fail" % (LIB_PATH, name))
                    continue
                raise
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _ptr(x):
    """Device or host address of a torch tensor / numpy array / None."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return x.ctypes.data


class Context:
    """One GPU's worth of the hot path (circkit_ctx)."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = _vp()
        rc = self._lib.circkit_ctx_create(int(device), ctypes.byref(h))
        self._h = h
        if rc != OK:
            msg = self._lib.circkit_last_error(h).decode() if h else "no usable HIP device %d" % device
            if h:
                self._lib.circkit_ctx_destroy(h)
                self._h = None
            raise CirckitError(rc, msg)
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.circkit_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise CirckitError(rc, self._lib.circkit_last_error(self._h).decode())

    # -- stream / timing ------------------------------------------------------------------------
    def set_stream(self, stream_handle):
        """Run on this hipStream_t (int handle; 0/None = HIP's default stream), e.g.
        torch.cuda.current_stream().cuda_stream so the work is ordered with torch's."""
        self._check(self._lib.circkit_ctx_set_stream(self._h, stream_handle or None))

    def use_own_stream(self):
        self._check(self._lib.circkit_ctx_use_own_stream(self._h))

    def synchronize(self):
        self._check(self._lib.circkit_ctx_synchronize(self._h))

    def last_kernel_ms(self):
        ms = ctypes.c_float(0)
        self._check(self._lib.circkit_ctx_last_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def batch_status(self):
        n = _u32(0)
        rc = self._lib.circkit_ctx_batch_status(self._h, ctypes.byref(n))
        if rc not in (OK, -4):
            self._check(rc)
        return n.value

    def set_long_record_scratch(self, nbytes):
        self._check(self._lib.circkit_ctx_set_long_record_scratch(self._h, int(nbytes)))

    def last_batch_mode(self):
        m = _u32(0)
        self._check(self._lib.circkit_ctx_last_batch_mode(self._h, ctypes.byref(m)))
        return m.value

    # -- device-resident batches (torch tensors on this ctx's GPU) -------------------------------
    def canonicalize_batch_device(self, d_bytes, d_offsets, n_records, out_bytes=None, out_index=None,
                                  out_strand=None, out_xxh3=None):
        self._check(self._lib.circkit_canonicalize_batch_device(
            self._h, _ptr(d_bytes), _ptr(d_offsets), int(n_records), _ptr(out_bytes), _ptr(out_index),
            _ptr(out_strand), _ptr(out_xxh3)))

    def lmsr_batch_device(self, d_bytes, d_offsets, n_records, out_bytes=None, out_index=None):
        self._check(self._lib.circkit_lmsr_batch_device(self._h, _ptr(d_bytes), _ptr(d_offsets), int(n_records),
                                                        _ptr(out_bytes), _ptr(out_index)))

    def xxh3_batch_device(self, d_bytes, d_offsets, n_records, out_hash):
        self._check(self._lib.circkit_xxh3_batch_device(self._h, _ptr(d_bytes), _ptr(d_offsets), int(n_records),
                                                        _ptr(out_hash)))

    def synth_fill_device(self, seed, first_base, n_bases, d_bytes):
        self._check(self._lib.circkit_synth_fill_device(self._h, int(seed), int(first_base), int(n_bases), _ptr(d_bytes)))

    def fixed_offsets_device(self, base, record_len, n_records, d_offsets):
        self._check(self._lib.circkit_fixed_offsets_device(self._h, int(base), int(record_len), int(n_records),
                                                           _ptr(d_offsets)))

    def bench_copy_device(self, d_src, d_dst, nbytes, variant=0):
        self._check(self._lib.circkit_bench_copy_device(self._h, _ptr(d_src), _ptr(d_dst), int(nbytes), int(variant)))

    def uniq_reset(self, expected_keys):
        self._check(self._lib.circkit_uniq_reset(self._h, int(expected_keys)))

    def uniq_insert_device(self, d_hash, n, base_index=0):
        self._check(self._lib.circkit_uniq_insert_device(self._h, _ptr(d_hash), int(n), int(base_index)))

    def uniq_insert_pairs_device(self, d_hash, d_index, n):
        self._check(self._lib.circkit_uniq_insert_pairs_device(self._h, _ptr(d_hash), _ptr(d_index), int(n)))

    def uniq_partition_device(self, d_hash, n, base_index, world, d_rows, d_counts, d_slot):
        self._check(self._lib.circkit_uniq_partition_device(self._h, _ptr(d_hash), int(n), int(base_index), int(world), _ptr(d_rows),
                                                            _ptr(d_counts), _ptr(d_slot)))

    def uniq_insert_rows_device(self, d_rows, n):
        self._check(self._lib.circkit_uniq_insert_rows_device(self._h, _ptr(d_rows), int(n)))

    def uniq_lookup_rows_device(self, d_rows, n, d_answers):
        self._check(self._lib.circkit_uniq_lookup_rows_device(self._h, _ptr(d_rows), int(n), _ptr(d_answers)))

    def uniq_gather_device(self, d_answers, d_slot, n, base_index, d_first_seen, d_keep=None):
        self._check(self._lib.circkit_uniq_gather_device(self._h, _ptr(d_answers), _ptr(d_slot), int(n), int(base_index), _ptr(d_first_seen),
                                                         _ptr(d_keep)))

    def uniq_lookup_device(self, d_hash, n, d_first_seen):
        self._check(self._lib.circkit_uniq_lookup_device(self._h, _ptr(d_hash), int(n), _ptr(d_first_seen)))

    def uniq_resolve_device(self, d_hash, n, base_index, d_first_seen, d_keep=None):
        self._check(self._lib.circkit_uniq_resolve_device(self._h, _ptr(d_hash), int(n), int(base_index), _ptr(d_first_seen), _ptr(d_keep)))

    def uniq_first_seen(self, hashes, base_index=0):
        """Host-buffer streaming form (the CLI's batch loop): folds the batch into the ctx table, grown on demand, and
        returns first_seen (numpy uint64)."""
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        out = np.empty(max(len(hashes), 1), dtype=np.uint64)
        self._check(self._lib.circkit_uniq_first_seen(self._h, _ptr(hashes), len(hashes), int(base_index), _ptr(out)))
        return out[:len(hashes)]

    def uniq_status(self):
        """Waits for the queued table work; raises CirckitError (OOM) if keys found no slot."""
        self._check(self._lib.circkit_uniq_status(self._h, None))

    # -- host batches (numpy) -------------------------------------------------------------------
    def canonicalize_batch(self, data, offsets, want_bytes=True, want_index=False, want_strand=False,
                           want_xxh3=False):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        out = np.empty(max(len(data), 1), dtype=np.uint8) if want_bytes else None
        idx = np.empty(max(n, 1), dtype=np.uint32) if want_index else None
        st = np.empty(max(n, 1), dtype=np.uint8) if want_strand else None
        hs = np.empty(max(n, 1), dtype=np.uint64) if want_xxh3 else None
        self._check(self._lib.circkit_canonicalize_batch(self._h, _ptr(data) if len(data) else None, _ptr(offsets), n,
                                                         _ptr(out), _ptr(idx), _ptr(st), _ptr(hs)))
        return {"bytes": out[:len(data)] if out is not None else None,
                "index": idx[:n] if idx is not None else None,
                "strand": st[:n] if st is not None else None,
                "xxh3": hs[:n] if hs is not None else None}

    # -- single record: the lib-crate API --------------------------------------------------------
    def _single(self, fn, s):
        s = bytes(s)
        buf = ctypes.create_string_buffer(s, max(len(s), 1))
        out = ctypes.create_string_buffer(max(len(s), 1))
        self._check(fn(self._h, ctypes.addressof(buf), len(s), ctypes.addressof(out)))
        return out.raw[:len(s)]

    def lmsr_index(self, s):
        s = bytes(s)
        buf = ctypes.create_string_buffer(s, max(len(s), 1))
        r = _sz(0)
        self._check(self._lib.circkit_lmsr_index(self._h, ctypes.addressof(buf), len(s), ctypes.byref(r)))
        return r.value

    def lmsr(self, s):
        return self._single(self._lib.circkit_lmsr, s)

    def canonicalize(self, s):
        return self._single(self._lib.circkit_canonicalize, s)

    def xxh3_64(self, s):
        s = bytes(s)
        buf = ctypes.create_string_buffer(s, max(len(s), 1))
        r = _u64(0)
        self._check(self._lib.circkit_xxh3_64(self._h, ctypes.addressof(buf), len(s), ctypes.byref(r)))
        return r.value


def normalize(s):
    """needletail::sequence::normalize(seq, false): returns (bytes, changed)."""
    lib = load_library()
    s = bytes(s)
    buf = ctypes.create_string_buffer(s, max(len(s), 1))
    out = ctypes.create_string_buffer(max(len(s), 1))
    ch = _i(0)
    m = lib.circkit_normalize(ctypes.addressof(buf), len(s), ctypes.addressof(out), ctypes.byref(ch))
    return out.raw[:m], bool(ch.value)


_default = None


def default_context():
    global _default
    if _default is None:
        _default = Context(0)
    return _default


def lmsr_index(s):
    return default_context().lmsr_index(s)


def lmsr(s):
    return default_context().lmsr(s)


def canonicalize(s):
    return default_context().canonicalize(s)


def xxh3_64(s):
    return default_context().xxh3_64(s)


def fasta_parse(text, first_chunk=True, final_chunk=True):
    """FASTA text -> (records, normalized_bytes, offsets, consumed); records = [(head, raw_seq)] with seq_io
    semantics.  Host logic (circkit_fasta_parse); raises ValueError on a format error."""
    lib = load_library()
    text = bytes(text)
    buf = ctypes.create_string_buffer(text, max(len(text), 1))
    h = _vp()
    consumed = _sz(0)
    rc = lib.circkit_fasta_parse(ctypes.addressof(buf), len(text), int(first_chunk), int(final_chunk), ctypes.byref(h),
                                 ctypes.byref(consumed))
    try:
        if rc != OK:
            raise ValueError(lib.circkit_fasta_error(h).decode() if h else "fasta parse failed")
        n = lib.circkit_fasta_n_records(h)
        offs = np.ctypeslib.as_array(ctypes.cast(lib.circkit_fasta_offsets(h), ctypes.POINTER(ctypes.c_uint64)),
                                     shape=(n + 1,)).copy()
        total = int(offs[-1])
        data = np.ctypeslib.as_array(ctypes.cast(lib.circkit_fasta_bytes(h), ctypes.POINTER(ctypes.c_uint8)),
                                     shape=(max(total, 1),)).copy()[:total]
        recs = []
        ho, hl, ro, rl = _sz(0), _sz(0), _sz(0), _sz(0)
        for i in range(n):
            lib.circkit_fasta_record(h, i, ctypes.byref(ho), ctypes.byref(hl), ctypes.byref(ro), ctypes.byref(rl))
            recs.append((text[ho.value:ho.value + hl.value], text[ro.value:ro.value + rl.value]))
        return recs, data, offs, consumed.value
    finally:
        if h:
            lib.circkit_fasta_free(h)
