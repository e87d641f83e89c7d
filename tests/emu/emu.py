"""Loads the CPU fiber emulator of the wave kernels (tests only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcanon_emu.so")
_CSRC = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "circkit_amd", "csrc")


SANITIZE = ["-fsanitize=undefined,bounds", "-fno-sanitize-recover=undefined,bounds"]


def build():
    srcs = [os.path.join(_HERE, "emu.cpp"), os.path.join(_HERE, "wave_prims_emu.h")] + \
        [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")]
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        # UBSan + bounds checks compiled in, no recovery: undefined shifts, misaligned or out-of-bounds accesses in the kernel
        # source abort the test process (sanitizers on the CPU build only -- the GPU pool offers none)
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared"] + SANITIZE + ["-o", _SO, os.path.join(_HERE, "emu.cpp")])
    return _SO


_lib = None
last_fast_count = 0
last_rescued_count = 0
last_fused_hash_count = 0


# `staged`: canon_stream.h's k-th workgroup geometry (WPB, RPW, NBUF), k >= 1; TWO_ROW = the ROWS == 2 builds
STAGED_GEOMETRIES = {1: (16, 1, 2), 2: (8, 2, 3), 3: (4, 2, 2), 4: (8, 2, 2), 5: (2, 2, 4), 6: (1, 2, 3), 7: (4, 1, 4), 8: (8, 1, 6), 9: (4, 2, 4),
                     10: (16, 1, 2), 11: (4, 2, 2), 12: (8, 1, 3), 13: (8, 1, 2), 14: (4, 2, 2), 15: (8, 2, 2), 16: (4, 1, 2), 17: (8, 1, 2)}       # 13: the hash build's geometry for batches with N, 14 / 15: the pair build's, 16: the bytes-only N build's (round 4)
TWO_ROW = (10, 11, 12, 17)


def alpha_rule(data, offsets):
    """launch_canon's content rule (MODE_ALPHA): of up to 4096 evenly spaced records, one in 16 or more holds a byte
    outside ACGT among its first 1008."""
    n = len(offsets) - 1
    if n == 0:
        return False
    nc = min(n, 4096)
    step = n // nc
    ok = np.zeros(256, dtype=bool)
    ok[list(b"ACGT")] = True
    bad = 0
    for k in range(nc):
        a, b = int(offsets[k * step]), int(offsets[k * step + 1])
        bad += not ok[data[a:min(b, a + 1008)]].all()
    return bad > 0 and bad * 16 >= nc


def canonicalize_batch(data, offsets, slice_dw=1024, n_waves=3, want_hash=False, flags=0, staged=1, want_aux=True,
                       base_shift=0, lead=0, alpha=None, solo=True, mixed=False, hash_only=False):
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.emu_canonicalize_batch.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_uint64] + [ctypes.c_void_p] * 4 + \
            [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    if alpha is None:
        alpha = alpha_rule(data, offsets)
    # base_shift: misalignment of the payload pointer; lead: offsets[0] (canary bytes in front of the first record)
    assert offsets[0] == 0
    offsets = offsets + np.uint64(lead)
    raw = np.full(64 + base_shift + lead + len(data) + 64, 0x4E, dtype=np.uint8)
    skew = (-raw.ctypes.data) % 64 + base_shift         # payload pointer = 64-byte boundary + base_shift
    pad = raw[skew:]
    pad[lead:lead + len(data)] = data
    raw_out = np.full(len(raw), 0x3F, dtype=np.uint8)
    out = raw_out[skew:]
    idx = np.full(max(n, 1), 0xFFFFFFFF, dtype=np.uint32)
    strand = np.full(max(n, 1), 0xFF, dtype=np.uint8)
    hs = np.zeros(max(n, 1), dtype=np.uint64)
    ndef = ctypes.c_uint32(0)
    nfast = ctypes.c_uint32(0)
    nfused = ctypes.c_uint32(0)
    nresc = ctypes.c_uint32(0)
    # want_aux=False: no rotation index / strand outputs -- the streaming kernel's leaner builds (see launch_canon)
    # hash_only: no canonical bytes anywhere (launch_canon with d_hash and without d_out): views + the xxh3 pass over them
    st = _lib.emu_canonicalize_batch(pad.ctypes.data, offsets.ctypes.data, n, None if hash_only else out.ctypes.data,
                                     idx.ctypes.data if want_aux else None, strand.ctypes.data if want_aux else None,
                                     hs.ctypes.data if want_hash else None,
                                     slice_dw, n_waves, ctypes.byref(ndef), flags, ctypes.byref(nfast), ctypes.byref(nfused), int(staged), ctypes.byref(nresc), int(bool(alpha)) | (0 if solo else 2) | (4 if mixed else 0))
    assert st >= 0, "emulator rejected the launch (unknown `staged` geometry?)"
    global last_fast_count, last_fused_hash_count, last_rescued_count
    last_rescued_count = nresc.value
    last_fast_count = nfast.value
    last_fused_hash_count = nfused.value
    assert (out[lead + len(data):] == 0x3F).all() and (raw_out[:skew + lead] == 0x3F).all(), "kernel wrote outside the batch"
    return out[lead:lead + len(data)], idx[:n], strand[:n], hs[:n], st, ndef.value


def xxh3_64(b):
    global _lib
    if _lib is None:
        canonicalize_batch(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint64))
    _lib.emu_xxh3_64.restype = ctypes.c_uint64
    _lib.emu_xxh3_64.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    buf = ctypes.create_string_buffer(bytes(b) + b"\0" * 16, len(b) + 16)
    return int(_lib.emu_xxh3_64(ctypes.addressof(buf), len(b)))
