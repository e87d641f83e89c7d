"""GPU box, experiment build only (tools/build_variant.sh tl -DCK_MIXED_TIMELINE; CIRCKIT_LIB=.../libcirckit_hip_tl.so):
where the time of canon_mixed_kernel goes on config 4's batch -- start and end of every workgroup on the 100 MHz wall clock.
Prints the kernel's span, the number of resident workgroups over time and what the last ones to finish were doing."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import circkit_amd
from circkit_amd import api, workloads as W

dev = torch.device("cuda", 0)
ctx = circkit_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N = 1_000_000
nfrac = float(os.environ.get("TL_NFRAC", "0"))
offs = W.log_uniform_offsets(N, 45, 200, 20000)
total = int(offs[-1])
d_off = offs.to(dev)
d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
ctx.synth_fill_device(45, 0, total, d_bytes)
if nfrac:
    W.sprinkle_n(d_bytes, total, nfrac, 46, dev)
d_out = torch.empty_like(d_bytes)
for _ in range(8):          # (the host learns the batch's mode from the batches before: the full grid comes after a few)
    ctx.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=d_out)
    torch.cuda.synchronize()
lib = ctx._lib
lib.circkit_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
nwg = int(os.environ.get("TL_NWG", "65536"))
buf = np.zeros((nwg, 3), dtype=np.uint64)
rc = lib.circkit_debug_timeline(ctx._h, buf.ctypes.data, nwg)
assert rc == 0, rc
used = buf[:, 1] > 0
t = buf[used]
n = len(t)
t0 = int(t[:, 0].min())
start = (t[:, 0].astype(np.int64) - t0) / 100.0          # microseconds
end = (t[:, 1].astype(np.int64) - t0) / 100.0
span = end.max()
print("workgroups %d  span %.1f us  (first start 0, last start %.1f, first end %.1f)" % (n, span, start.max(), end.min()))
dur = end - start
print("workgroup duration: min %.1f  median %.1f  p90 %.1f  p99 %.1f  max %.1f us" % (dur.min(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
# bytes per workgroup (segment s = records [s * cap, (s + 1) * cap)); cap from the count of workgroups
cap = -(-N // n)
o = offs.numpy()
seg_bytes = np.array([o[min(N, (s + 1) * cap)] - o[min(N, s * cap)] for s in range(n)])
seg_max = np.array([np.diff(o[min(N, s * cap):min(N, (s + 1) * cap) + 1]).max() if s * cap < N else 0 for s in range(n)])
print("segment cap %d records; bytes per segment: mean %.0f  max %.0f;  corr(duration, bytes) %.3f" % (cap, seg_bytes.mean(), seg_bytes.max(), np.corrcoef(dur, seg_bytes)[0, 1]))
# residency over time
bins = 40
edges = np.linspace(0, span, bins + 1)
res = [(np.minimum(end, edges[i + 1]) - np.maximum(start, edges[i])).clip(0).sum() / (edges[i + 1] - edges[i]) for i in range(bins)]
print("resident workgroups per %.0f us bin:" % (span / bins))
print("  " + " ".join("%d" % r for r in res))
order = np.argsort(-end)[:12]
print("last to finish: (wg, start, end, duration, bytes, longest record)")
for w in order:
    print("  %6d  %8.1f %8.1f %7.1f  %8d %6d" % (w, start[w], end[w], dur[w], seg_bytes[w], seg_max[w]))
# start order vs index
late = np.argsort(-start)[:5]
print("last to start:", [(int(w), round(float(start[w]), 1)) for w in late])
xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64) & 0xF
print("workgroups per XCC:", np.bincount(xcc, minlength=8).tolist())
busy = [dur[xcc == x].sum() for x in range(8)]
print("busy workgroup-us per XCC:", [int(b) for b in busy])
lastx = [end[xcc == x].max() if (xcc == x).any() else 0 for x in range(8)]
print("last end per XCC:", [round(float(b), 1) for b in lastx])
np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "timeline_raw.npy"), t)
hw = (t[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
cus = np.unique(key)
print("distinct CUs seen:", len(cus))
# concurrency per CU at mid-kernel
for frac in (0.25, 0.5, 0.75):
    tt = span * frac
    live = (start <= tt) & (end > tt)
    c = np.bincount(np.searchsorted(cus, key[live]), minlength=len(cus))
    print("t=%.0f us: resident per CU: min %d  mean %.2f  max %d   histogram %s" % (tt, c.min(), c.mean(), c.max(), np.bincount(c).tolist()))
