"""GPU box, a -DCK_DEBUG_DUMP build through CIRCKIT_LIB: what the mixed-length kernel of BASELINE config 4 (+ 1 % N with --n1)
hands to stage A -- length histogram of the deferred records, with and without the alphabet flag (CK_DUMP_HIST=1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CK_DUMP_HIST"] = "1"
import torch                                   # noqa: E402
import circkit_amd.api as api                   # noqa: E402
from circkit_amd import workloads as W          # noqa: E402

dev = torch.device("cuda", 0)
c = api.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
N = 1_000_000
offs = W.log_uniform_offsets(N, 45)
total = int(offs[-1])
d_off = offs.to(dev)
d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
c.synth_fill_device(45, 0, total, d_bytes)
if "--n1" in sys.argv:
    W.sprinkle_n(d_bytes, total, 0.01, 46, dev)
d_out = torch.empty(total + 64, dtype=torch.uint8, device=dev)
for _ in range(2):       # the second batch runs with the mode the first one found
    c.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=d_out)
    torch.cuda.synchronize()
