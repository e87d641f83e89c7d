"""`uniq` first-seen resolution: host-side mirror of the reference's main-thread closure.

Reference (src/uniq.rs:42-78): records are visited in input order; key = xxh3_64(canonical bytes) (:45);
`seen` is a HashMap<u64, String> with hash-only equality (:27,47); the first record of each key is kept,
later ones become (first_id, duplicate_id) table rows (:63-70).  Order-independent restatement used here:
record i is kept iff i is the SMALLEST global index carrying its hash.

Single GPU: the ctx's device hash table (circkit_uniq_insert_device / _lookup_device).
Several GPUs: records are sharded by contiguous global index ranges; each rank hashes its own shard (no
collective on the canonicalize path), then ONE exchange step resolves the duplicates across ranks
(torch.distributed; backend "nccl" = RCCL over xGMI on the GPU node):

* exchange="partition" (default): the key space is cut into `world` hash ranges, one per rank.  Every rank sends
  each (hash, global index) pair to the owner of its range (all-to-all), the owner folds what it receives into its
  table -- 1/world of all keys -- and answers every pair with the smallest index it has seen for that hash (a
  second all-to-all, same shape backwards).  With the device table every step between the collectives is a kernel
  of the library (circkit_uniq_partition_device / _insert_rows_ / _lookup_rows_ / _gather_device); the torch-op
  version below it serves the CPU tests' stand-in tables.  Per GPU and 10M-record shard at world 8: 140 MB out + 140 MB in +
  70 MB of answers each way, one table insert and one lookup of 10M keys.
* exchange="allgather": every rank gathers all hash arrays (world x 80 MB in) and folds ALL keys into its own table
  (world x the insert work).  Kept for comparison and as the simpler reference of the two.

xGMI is point-to-point, so the all-to-all's 7 simultaneous peer transfers use 7 links at once, while a ring
all-gather is bound by one link per hop.
"""
import torch
import torch.distributed as dist


class DeviceTable:
    """The circkit_ctx hash table (HIP kernels) behind the calls the merge needs.

    Stream ordering: the table kernels read tensors that torch ops and RCCL collectives have just produced, so they
    must run on the stream those are ordered on.  Every call therefore (re)binds the ctx to torch's CURRENT stream
    of the ctx's device -- a blocking collective (`async_op=False`) makes that stream wait for RCCL's -- instead of
    the ctx's private non-blocking stream.  The rebind is a lasting side effect on the ctx; circkit_ctx_set_stream
    orders the stream it moves to behind everything the ctx queued on the one it leaves (an event, no host wait), so
    `ctx.canonicalize_batch_device(..., out_xxh3=h)` on the old stream followed by `first_seen(DeviceTable(ctx), h)` reads
    a finished `h`.  Binding the stream that is already bound costs nothing."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._bind()

    def _bind(self):
        self.ctx.set_stream(torch.cuda.current_stream(self.ctx.device).cuda_stream)

    def reset(self, expected_keys):
        self._bind()
        self.ctx.uniq_reset(expected_keys)

    def insert(self, hashes, base_index):
        self._bind()
        self.ctx.uniq_insert_device(hashes, hashes.numel(), base_index)

    def insert_pairs(self, hashes, indices):
        self._bind()
        self.ctx.uniq_insert_pairs_device(hashes, indices, hashes.numel())

    def lookup(self, hashes):
        self._bind()
        out = torch.empty_like(hashes)
        self.ctx.uniq_lookup_device(hashes, hashes.numel(), out)
        return out

    def resolve(self, hashes, base_index):
        """reset + insert + lookup + keep flags of one shard in one library call."""
        self._bind()
        fs = torch.empty_like(hashes)
        keep = torch.empty(hashes.numel(), dtype=torch.bool, device=hashes.device)
        self.ctx.uniq_resolve_device(hashes, hashes.numel(), base_index, fs, keep)
        return fs, keep

    # the exchange's device steps (kernels of the library instead of argsort / bincount / gather ops)
    def partition(self, hashes, base_index, world):
        """(rows [n, 2] = (hash, global index) grouped by owner rank, counts [world], slot [n])"""
        self._bind()
        n = hashes.numel()
        rows = torch.empty((n, 2), dtype=torch.int64, device=hashes.device)
        counts = torch.empty(world, dtype=torch.int64, device=hashes.device)
        slot = torch.empty(n, dtype=torch.int32, device=hashes.device)
        self.ctx.uniq_partition_device(hashes, n, base_index, world, rows, counts, slot)
        return rows, counts, slot

    def insert_rows(self, rows):
        self._bind()
        self.ctx.uniq_insert_rows_device(rows, rows.shape[0])

    def lookup_rows(self, rows):
        self._bind()
        out = torch.empty(rows.shape[0], dtype=torch.int64, device=rows.device)
        self.ctx.uniq_lookup_rows_device(rows, rows.shape[0], out)
        return out

    def gather(self, answers, slot, base_index):
        self._bind()
        n = slot.numel()
        fs = torch.empty(n, dtype=torch.int64, device=slot.device)
        keep = torch.empty(n, dtype=torch.bool, device=slot.device)
        self.ctx.uniq_gather_device(answers, slot, n, base_index, fs, keep)
        return fs, keep

    def check(self):
        """Waits for the queued table work; raises if the table overflowed."""
        self.ctx.uniq_status()


def _owner(hashes, world):
    """Rank that owns a key: a few well-mixed middle bits of the hash (XXH3 output is uniform), sign-safe on int64."""
    return ((hashes >> 20) & 0x7FFFFFFF) % world


def first_seen(table, hashes, base_index=0, group=None, exchange="partition", force_exchange=False):
    """hashes: int64/uint64 tensor of this rank's shard (xxh3 of canonical records, input order);
    base_index: global index of this shard's record 0.  Returns (first_seen_global_index, keep_mask) for the
    shard.  With an initialised process group duplicates are resolved across all ranks (see the module docstring).
    force_exchange: run the collectives even in a group of one rank (rehearses the RCCL path on a single GPU)."""
    n = hashes.numel()
    initialised = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if initialised else 1
    if world == 1 and not (force_exchange and initialised) and hasattr(table, "resolve"):
        return table.resolve(hashes, base_index)
    if exchange == "partition" and hasattr(table, "partition") and world <= 64 and (world > 1 or (force_exchange and initialised)):
        # the device table: every step between the collectives is a kernel of the library
        rows, send_counts, slot = table.partition(hashes, base_index, world)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=group)
        in_split, out_split = send_counts.tolist(), recv_counts.tolist()
        recv = torch.empty((sum(out_split), 2), dtype=torch.int64, device=hashes.device)
        dist.all_to_all_single(recv, rows, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
        table.reset(max(recv.shape[0], 1))
        if recv.shape[0]:
            table.insert_rows(recv)
        answers = table.lookup_rows(recv)
        back = torch.empty(n, dtype=torch.int64, device=hashes.device)
        dist.all_to_all_single(back, answers, output_split_sizes=in_split, input_split_sizes=out_split, group=group)
        return table.gather(back, slot, base_index)
    idx = torch.arange(base_index, base_index + n, dtype=torch.int64, device=hashes.device)
    if world == 1 and not (force_exchange and initialised):
        table.reset(n)
        table.insert(hashes, base_index)
        fs = table.lookup(hashes)
    elif exchange == "partition":
        h64 = hashes.view(torch.int64)
        owner = _owner(h64, world)
        order = torch.argsort(owner, stable=True)
        send_counts = torch.bincount(owner, minlength=world)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=group)
        in_split, out_split = send_counts.tolist(), recv_counts.tolist()
        # one message per peer: column 0 = hash, column 1 = global index
        send = torch.stack([h64[order], idx[order]], dim=1).contiguous()
        recv = torch.empty((sum(out_split), 2), dtype=torch.int64, device=hashes.device)
        dist.all_to_all_single(recv, send, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
        got_h, got_i = recv[:, 0].contiguous(), recv[:, 1].contiguous()
        table.reset(max(got_h.numel(), 1))
        if got_h.numel():
            table.insert_pairs(got_h, got_i)
        answers = table.lookup(got_h).view(torch.int64) if got_h.numel() else got_h
        back = torch.empty(n, dtype=torch.int64, device=hashes.device)
        dist.all_to_all_single(back, answers.contiguous(), output_split_sizes=in_split, input_split_sizes=out_split, group=group)
        fs = torch.empty(n, dtype=torch.int64, device=hashes.device)
        fs[order] = back
    elif exchange == "allgather":
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
    else:
        raise ValueError("exchange must be 'partition' or 'allgather'")
    keep = fs.view(torch.int64) == idx
    return fs, keep
