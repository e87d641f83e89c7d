"""`uniq` first-seen resolution: host-side mirror of the reference's main-thread closure.

Reference (src/uniq.rs:42-78): records are visited in input order; key = xxh3_64(canonical bytes) (:45);
`seen` is a HashMap<u64, String> with hash-only equality (:27,47); the first record of each key is kept,
later ones become (first_id, duplicate_id) table rows (:63-70).  Order-independent restatement used here:
record i is kept iff i is the SMALLEST global index carrying its hash.

Single GPU: the ctx's device hash table (circkit_uniq_insert_device / _lookup_device).
Several GPUs: records are sharded by contiguous global index ranges; each rank hashes its own shard (no
collective on the canonicalize path), then ONE exchange step merges the hash sets -- an all-gather of the
per-rank hash arrays (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU node) -- after which
every rank folds all (hash, global index) pairs into its table and reads off the winners for its own shard.
"""
import torch
import torch.distributed as dist


class DeviceTable:
    """The circkit_ctx hash table (HIP kernels) behind the two calls the merge needs."""

    def __init__(self, ctx):
        self.ctx = ctx

    def reset(self, expected_keys):
        self.ctx.uniq_reset(expected_keys)

    def insert(self, hashes, base_index):
        self.ctx.uniq_insert_device(hashes, hashes.numel(), base_index)

    def lookup(self, hashes):
        out = torch.empty_like(hashes)
        self.ctx.uniq_lookup_device(hashes, hashes.numel(), out)
        return out


def first_seen(table, hashes, base_index=0, group=None):
    """hashes: int64/uint64 tensor of this rank's shard (xxh3 of canonical records, input order);
    base_index: global index of this shard's record 0.  Returns (first_seen_global_index, keep_mask) for the
    shard.  With an initialised process group the hash sets of all ranks are merged first."""
    n = hashes.numel()
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        table.reset(n)
        table.insert(hashes, base_index)
        fs = table.lookup(hashes)
    else:
        # shard sizes and bases may differ: exchange them, pad to the largest shard, all-gather once
        meta = torch.tensor([n, base_index], dtype=torch.int64, device=hashes.device)
        metas = [torch.empty_like(meta) for _ in range(world)]
        dist.all_gather(metas, meta, group=group)
        sizes = [int(m[0]) for m in metas]
        bases = [int(m[1]) for m in metas]
        cap = max(sizes)
        padded = torch.zeros(cap, dtype=hashes.dtype, device=hashes.device)
        padded[:n] = hashes
        gathered = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(gathered, padded, group=group)
        table.reset(sum(sizes))
        for r in range(world):
            if sizes[r]:
                table.insert(gathered[r][:sizes[r]].contiguous(), bases[r])
        fs = table.lookup(hashes)
    idx = torch.arange(base_index, base_index + n, dtype=torch.int64, device=hashes.device)
    keep = fs.view(torch.int64) == idx
    return fs, keep
