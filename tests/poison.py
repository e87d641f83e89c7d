"""Builds the CK_DEBUG_POISON variant of the HIP library (TEST INFRASTRUCTURE ONLY; see circkit_amd/csrc/canon_stream.h
stream_poison): the same source with every LDS ring image overwritten by a poison pattern right before its DMA is re-issued,
and a counter of records that still read the pattern.  tests/libcirckit_hip_poison.so is git-ignored and travels to the GPU
box like the product library; __graft_entry__.build() builds it, the GPU test builds it when it is missing or stale."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "circkit_amd", "csrc")
LIB = os.path.join(HERE, "libcirckit_hip_poison.so")


def build(force=False, negative_control=False):
    """negative_control: the same build with the loop's vmcnt waits REMOVED (CK_DEBUG_POISON_BREAK) -- for
    tools/poison_negative_control.py, which shows that the guard does count when the protocol is broken."""
    lib = LIB.replace(".so", "_break.so") if negative_control else LIB
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "circkit.h")]
    if not force and os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
        return lib
    import sys
    sys.path.insert(0, ROOT)
    from circkit_amd.build import _hipcc
    hipcc = _hipcc()
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-inline-asm", "-DCK_DEBUG_POISON"] +
                          (["-DCK_DEBUG_POISON_BREAK"] if negative_control else []) +
                          ["-o", lib, os.path.join(CSRC, "circkit_hip.hip"), os.path.join(CSRC, "fasta_host.cpp")])
    return lib
