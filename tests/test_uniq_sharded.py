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


# ---- the bench's multi-rank `uniq` workload (config 3 over a whole job) and its check, under gloo at world 8 ---------------
def _job_worker(rank, world, port, n, length, q, broken):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from circkit_amd import uniq, workloads as W
    from oracle import oracle as O
    dev = torch.device("cpu")
    fill = lambda seed, first, nb: torch.from_numpy(O.synth_fill(seed, first, nb).copy())
    shard = fill(42, rank * n * length, n * length)
    W.plant_job_duplicates(fill, shard, n, length, dev, rank, world)
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(length)
    _, h = O.canonicalize_batch(shard.numpy(), offs, False, True)
    hashes = torch.from_numpy(h.astype(np.int64))
    if broken == "local":
        # an "exchange" that returns every rank's own answer: what a job-wide check must catch (missing #1)
        fs, keep = _local_first_seen(hashes, rank * n)
    else:
        if broken == "back":
            # the answers' way back delivered out of order (circkit_amd/uniq.py, the second all-to-all of the rows branch)
            real, calls = dist.all_to_all_single, [0]

            def shuffled(out, inp, *a, **kw):
                r = real(out, inp, *a, **kw)
                calls[0] += 1
                if calls[0] == 3 and out.numel() > 1:
                    out.copy_(out.roll(1))
                return r
            uniq.dist.all_to_all_single = shuffled
        fs, keep = uniq.first_seen(RowsTable(), hashes, base_index=rank * n, exchange="partition")
    wrong, cross, distinct, _ = W.job_check(fs, keep, n, length, world, rank, dev)
    q.put((rank, wrong, cross, distinct, int(keep.sum())))
    dist.barrier()
    dist.destroy_process_group()


def _local_first_seen(hashes, base):
    from oracle import oracle as O
    fs = torch.from_numpy(O.uniq_first_seen(hashes.numpy().astype(np.uint64)).astype(np.int64)) + base
    return fs, fs == torch.arange(base, base + hashes.numel(), dtype=torch.int64)


@pytest.mark.parametrize("broken", [None, "local", "back"])
def test_job_wide_duplicates_at_world_8(broken):
    """bench.py --workload uniq --gpus 8 in small: every rank plants duplicates of base records of ALL ranks
    (workloads.plant_job_duplicates, the generator regenerates any rank's records locally), hashes its shard, runs the
    exchange, and checks every record against the job-wide expectation (workloads.job_check -- the function bench.py exits
    non-zero on).  Negative controls: an exchange that answers locally, and answers that come back out of order, must fail
    the check (src/uniq.rs:47-48: first record of each hash over the WHOLE input)."""
    world, n, length = 8, 240, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, world, port, n, length, q, broken)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    wrong = sum(r[1] for r in res)
    cross = sum(r[2] for r in res)
    kept = sum(r[4] for r in res)
    assert all(r[3] == world * (n // 2) for r in res)
    assert cross > world * n // 4                     # most duplicates' originals live on another rank
    if broken is None:
        assert wrong == 0 and kept == world * (n // 2)
    else:
        assert wrong > 0
        if broken == "local":
            assert kept != world * (n // 2)           # the kept count alone catches this one only because owners are cross-rank
