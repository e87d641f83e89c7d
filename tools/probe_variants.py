"""GPU box: round-robin timing of variant builds of the library IN ONE PROCESS (each variant its own ctypes handle, ctx and
stream-ordered buffers), on the fixed-length batch (10M x 1 kb) and optionally the mixed-length one: ms per batch for bytes,
bytes + XXH3, XXH3 only.  tools/probe_variants.py tag1 tag2 ...  (circkit_amd/libcirckit_hip_<tag>.so; `base` = the in-tree build).
Variants may be experiment builds whose results are wrong on purpose (only the time is of interest)."""
import importlib.util
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # noqa: E402
import circkit_amd.api as base_api              # noqa: E402
from circkit_amd import workloads as W          # noqa: E402


def load(tag):
    if tag == "base":
        return base_api
    path = os.path.join(ROOT, "circkit_amd", "libcirckit_hip_%s.so" % tag)
    os.environ["CIRCKIT_LIB"] = path
    spec = importlib.util.spec_from_file_location("circkit_api_" + tag, base_api.__file__)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    del os.environ["CIRCKIT_LIB"]
    return m


def main():
    tags = [a for a in sys.argv[1:] if not a.startswith("--")]
    mixed = "--mixed" in sys.argv
    rounds = 5
    dev = torch.device("cuda", 0)
    ctxs = {}
    for t in tags:
        c = load(t).Context(0)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        ctxs[t] = c
    c0 = ctxs[tags[0]]
    if mixed:
        N = 1_000_000
        offs = W.log_uniform_offsets(N, 45)
        if "--long" in sys.argv:                 # the tiers' own records: 20..80 kb (stage A: one wave up to 20.4 kb, its team beyond)
            N = 40_000
            offs = W.log_uniform_offsets(N, 45, 20000, 80000)
        total = int(offs[-1])
        d_off = offs.to(dev)
        d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
        c0.synth_fill_device(45, 0, total, d_bytes)
        if "--n1" in sys.argv:
            W.sprinkle_n(d_bytes, total, 0.01, 46, dev)
    else:
        N, L = 10_000_000, 1000
        total = N * L
        d_bytes, d_off = W.fixed_length(c0, dev, N, L, 42, 0)
    d_out = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    d_hash = torch.empty(N, dtype=torch.int64, device=dev)
    modes = (("bytes", dict(out_bytes=d_out)), ("bytes+xxh3", dict(out_bytes=d_out, out_xxh3=d_hash)), ("xxh3", dict(out_xxh3=d_hash)))
    if "--bytes-only" in sys.argv:
        modes = modes[:1]
    res = {(t, m): [] for t in tags for m, _ in modes}
    for r in range(rounds + 1):
        for t in tags:
            for m, kw in modes:
                c = ctxs[t]
                for _ in range(3 if r == 0 else 1):
                    c.canonicalize_batch_device(d_bytes, d_off, N, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    c.canonicalize_batch_device(d_bytes, d_off, N, **kw)
                e1.record()
                e1.synchronize()
                if r:
                    res[(t, m)].append(e0.elapsed_time(e1) / 5)
    for t in tags:
        print("%-14s" % t + "  ".join("%s min %.3f med %.3f" % (m, min(res[(t, m)]), statistics.median(res[(t, m)])) for m, _ in modes), flush=True)


if __name__ == "__main__":
    main()
