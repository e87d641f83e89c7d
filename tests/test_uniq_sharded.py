"""world_size-2 gloo test (CPU) of the multi-GPU uniq merge: shard -> all-gather hash sets -> first-seen.
The GPU hash table is replaced by the CPU checker's first-seen routine (test infrastructure) so the
exchange / global-index logic of circkit_amd/uniq.py runs without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleTable:
    def __init__(self):
        self.h, self.i = [], []

    def reset(self, expected):
        self.h, self.i = [], []

    def insert(self, hashes, base):
        self.h.append(hashes.numpy().astype(np.uint64))
        self.i.append(np.arange(base, base + hashes.numel(), dtype=np.uint64))

    def insert_pairs(self, hashes, indices):
        self.h.append(hashes.numpy().astype(np.uint64))
        self.i.append(indices.numpy().astype(np.uint64))

    def lookup(self, hashes):
        h = np.concatenate(self.h)
        i = np.concatenate(self.i)
        best = {}
        for hh, ii in zip(h.tolist(), i.tolist()):
            if hh not in best or ii < best[hh]:
                best[hh] = ii
        return torch.tensor([best[x] for x in hashes.numpy().astype(np.uint64).tolist()], dtype=torch.int64)


class RowsTable(OracleTable):
    """NumPy stand-in for the DEVICE table's exchange steps, to the contract of include/circkit.h (partition / insert_rows /
    lookup_rows / gather): with it uniq.first_seen takes the branch a multi-GPU job takes (circkit_amd/uniq.py, "the device
    table: every step between the collectives is a kernel of the library"), so split sizes, row order across peers and the
    slot gather run under a real peer.  Inside an owner's group the rows are deliberately NOT in record order (the kernel
    promises none): an exchange that relied on it would fail here."""

    def partition(self, hashes, base_index, world):
        from circkit_amd import uniq
        self.partitioned = True
        n = hashes.numel()
        h = hashes.view(torch.int64)
        owner = uniq._owner(h, world).numpy()
        rng = np.random.default_rng(base_index + 17)
        order = np.lexsort((rng.permutation(n), owner))              # owners in rank order, any order inside
        rows = torch.empty((n, 2), dtype=torch.int64)
        rows[:, 0] = h[torch.from_numpy(order)]
        rows[:, 1] = torch.from_numpy(order.astype(np.int64)) + base_index
        counts = torch.from_numpy(np.bincount(owner, minlength=world).astype(np.int64))
        slot = torch.empty(n, dtype=torch.int32)
        slot[torch.from_numpy(order)] = torch.arange(n, dtype=torch.int32)
        return rows, counts, slot

    def insert_rows(self, rows):
        self.insert_pairs(rows[:, 0].contiguous(), rows[:, 1].contiguous())

    def lookup_rows(self, rows):
        if rows.shape[0] == 0:
            return torch.empty(0, dtype=torch.int64)
        return self.lookup(rows[:, 0].contiguous())

    def gather(self, answers, slot, base_index):
        self.gathered = True
        fs = answers[slot.long()]
        return fs, fs == torch.arange(base_index, base_index + slot.numel(), dtype=torch.int64)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, hashes, cuts, q, exchange, rows_table=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from circkit_amd import uniq
    lo, hi = cuts[rank], cuts[rank + 1]
    table = RowsTable() if rows_table else OracleTable()
    fs, keep = uniq.first_seen(table, torch.from_numpy(hashes[lo:hi].astype(np.int64)), base_index=lo, exchange=exchange)
    if rows_table:
        assert table.partitioned and table.gathered            # the device-rows branch really ran
    q.put((rank, fs.numpy(), keep.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,world,rows_table", [("partition", 2, False), ("allgather", 2, False), ("partition", 3, False),
                                                       ("partition", 2, True), ("partition", 3, True), ("partition", 4, True),
                                                       ("partition", 8, True), ("allgather", 8, False)])
def test_sharded_first_seen_matches_single_process(exchange, world, rows_table):
    """Both exchange steps of circkit_amd/uniq.py -- the hash-range all-to-all (default) and the all-gather -- give
    the single-process first-seen result, with unequal shards (one of them empty at world 3, 4 and 8) and sign-bit hashes.
    rows_table: the branch the device table takes (partition -> all_to_all -> insert_rows / lookup_rows -> all_to_all ->
    gather), with the table's kernels replaced by RowsTable."""
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    n = 5000
    hashes = rng.integers(0, 1200, size=n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)   # ~4x duplicates, top bit set in half
    cuts = {2: [0, 1777, n], 3: [0, 1777, 1777, n], 4: [0, 5, 1777, 1777, n],
            8: [0, 5, 700, 700, 1777, 2500, 2501, 4100, n]}[world]          # unequal shards (8 = BASELINE config 5's world size)
    expect = O.uniq_first_seen(hashes).astype(np.int64)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, hashes, cuts, q, exchange, rows_table)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        r, fs, keep = q.get(timeout=120)
        got[r] = (fs, keep)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fs = np.concatenate([got[r][0] for r in range(world)])
    keep = np.concatenate([got[r][1] for r in range(world)])
    assert np.array_equal(fs, expect)
    assert np.array_equal(keep, expect == np.arange(n))
    assert keep.sum() == len(set(hashes.tolist()))


def test_single_process_first_seen():
    from circkit_amd import uniq
    from oracle import oracle as O
    h = np.array([5, 7, 5, 9, 7, 5, 0, 0], dtype=np.uint64)
    fs, keep = uniq.first_seen(OracleTable(), torch.from_numpy(h.astype(np.int64)), base_index=100)
    assert fs.tolist() == [100, 101, 100, 103, 101, 100, 106, 106]
    assert keep.tolist() == [True, True, False, True, False, False, True, False]
    assert (O.uniq_first_seen(h) + 100).tolist() == fs.tolist()
