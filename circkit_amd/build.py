"""Builds libcirckit_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcirckit_hip.so")
CLI = os.path.join(HERE, "circkit")
HIP_SOURCES = ["circkit_hip.hip", "fasta_host.cpp"]
HOST_SOURCES = ["circkit_cli.cpp"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X build needs the ROCm toolchain")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _all_deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "circkit.h"))
    return deps


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> circkit_amd/libcirckit_hip.so (cross-compiles without a GPU)."""
    if not force and not _stale(LIB, _all_deps()):
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-inline-asm",
           "-o", LIB] + [os.path.join(CSRC, s) for s in HIP_SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return LIB


def build_cli(force=False):
    """The C++ host: FASTA streaming packer + `circkit canonicalize|uniq` CLI, linked against the C ABI."""
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    if not all(os.path.exists(s) for s in srcs):
        return None
    if not force and not _stale(CLI, _all_deps() + [LIB]):
        return CLI
    cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-o", CLI] + srcs + \
          ["-L" + HERE, "-lcirckit_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    subprocess.check_call(cmd)
    return CLI


def build_all(force=False, verbose=False):
    build_library(force, verbose)
    build_cli(force)
    return LIB
