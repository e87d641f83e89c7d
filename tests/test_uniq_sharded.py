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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, hashes, cuts, q, exchange):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from circkit_amd import uniq
    lo, hi = cuts[rank], cuts[rank + 1]
    fs, keep = uniq.first_seen(OracleTable(), torch.from_numpy(hashes[lo:hi].astype(np.int64)), base_index=lo, exchange=exchange)
    q.put((rank, fs.numpy(), keep.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,world", [("partition", 2), ("allgather", 2), ("partition", 3)])
def test_sharded_first_seen_matches_single_process(exchange, world):
    """Both exchange steps of circkit_amd/uniq.py -- the hash-range all-to-all (default) and the all-gather -- give
    the single-process first-seen result, with unequal shards (one of them empty at world 3) and sign-bit hashes."""
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    n = 5000
    hashes = rng.integers(0, 1200, size=n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)   # ~4x duplicates, top bit set in half
    cuts = [0, 1777, n] if world == 2 else [0, 1777, 1777, n]                                  # unequal shards
    expect = O.uniq_first_seen(hashes).astype(np.int64)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, hashes, cuts, q, exchange)) for r in range(world)]
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
