"""GPU box: the multi-GPU `uniq` merge of circkit_amd/uniq.py over RCCL (torch.distributed backend "nccl") with the
HIP table, rehearsed in a group of ONE rank on the box's one GPU: the collectives (all_to_all_single / all_gather), the
pair inserts and the lookups all run on the device and are compared with the oracle's first-seen.
usage: python tools/gpu_uniq_nccl.py partition|allgather [keys]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

exchange = sys.argv[1] if len(sys.argv) > 1 else "partition"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

import circkit_amd
from circkit_amd import uniq
from oracle import oracle as O

rng = np.random.default_rng(21)
h = rng.integers(0, n // 2, size=n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)     # ~43 % of the keys repeat
h[rng.integers(0, n, 40)] = np.uint64(0xFFFFFFFFFFFFFFFF)                                 # the table's empty marker is a legal hash
base = 123_456_789_000                                                                    # a shard deep inside a big job
exp = O.uniq_first_seen(h).astype(np.int64) + base
ctx = circkit_amd.Context(0)
table = uniq.DeviceTable(ctx)
d_h = torch.from_numpy(h.astype(np.int64)).to(dev)
side = torch.cuda.Stream()
for stream in (torch.cuda.current_stream(), side):          # the table follows torch's current stream
    with torch.cuda.stream(stream):
        for rep in range(2):
            fs, keep = uniq.first_seen(table, d_h, base_index=base, exchange=exchange, force_exchange=True)
            table.check()
            got = fs.cpu().numpy()
            assert np.array_equal(got, exp), "first-seen differs from the oracle (%s, %d mismatches)" % (exchange, int((got != exp).sum()))
            assert int(keep.sum().item()) == len(np.unique(h))
dist.barrier()
dist.destroy_process_group()
ctx.close()
print("first-seen matches the oracle: %d keys, %d distinct, exchange=%s over nccl" % (n, len(np.unique(h)), exchange))
