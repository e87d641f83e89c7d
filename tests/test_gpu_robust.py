"""GPU tests of the ctx's failure and ordering rules (run with -m gpu): a failed table rehash stays an error until
circkit_uniq_reset, and circkit_ctx_set_stream orders the stream it moves to behind the one it leaves."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    oracle.lib()
    return oracle


def test_failed_rehash_is_an_error_until_reset(O):
    """circkit_uniq_first_seen grows its table by rehashing into a new one.  If that fails after the old table is gone
    (injected: CIRCKIT_TEST_FAIL_REHASH), the stream's earlier batches are lost -- the call fails, and so does every later
    one instead of quietly starting an empty table; circkit_uniq_reset starts over."""
    import circkit_amd
    ctx = circkit_amd.Context(0)
    rng = np.random.default_rng(5)
    h1 = rng.integers(0, 500, size=1000).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    h2 = rng.integers(0, 500, size=100_000).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    fs1 = ctx.uniq_first_seen(h1, 0)
    assert np.array_equal(fs1, O.uniq_first_seen(h1))
    os.environ["CIRCKIT_TEST_FAIL_REHASH"] = "1"
    try:
        with pytest.raises(circkit_amd.CirckitError) as e:
            ctx.uniq_first_seen(h2, 1000)                       # (1000 + 100000) keys: the 65536-slot table must grow
        assert "rehash failed" in str(e.value)
    finally:
        del os.environ["CIRCKIT_TEST_FAIL_REHASH"]
    for _ in range(2):                                          # the failure is not forgotten
        with pytest.raises(circkit_amd.CirckitError) as e:
            ctx.uniq_first_seen(h2, 1000)
        assert "lost in a failed rehash" in str(e.value)
    ctx.uniq_reset(1)
    both = np.concatenate([h1, h2])
    fs = np.concatenate([ctx.uniq_first_seen(h1, 0), ctx.uniq_first_seen(h2, 1000)])       # grows for real this time
    assert np.array_equal(fs, O.uniq_first_seen(both))
    ctx.close()


def test_set_stream_orders_the_new_stream_behind_the_old(O):
    """A fresh ctx runs on its own non-blocking stream.  A hash batch enqueued there and, with no synchronisation in
    between, uniq.first_seen through a DeviceTable -- which rebinds the ctx to torch's current stream -- must see finished
    hashes: circkit_ctx_set_stream makes the new stream wait for the old one's work."""
    import torch
    import circkit_amd
    from circkit_amd import uniq
    from circkit_amd import workloads as W
    dev = torch.device("cuda", 0)
    n, L = 2_000_000, 1000
    setup = circkit_amd.Context(0)
    setup.set_stream(torch.cuda.current_stream().cuda_stream)
    d_bytes, d_off = W.fixed_length(setup, dev, n, L, 42, 0)
    W.plant_duplicates(d_bytes, n, L, dev, 43, 44)
    torch.cuda.synchronize()
    ctx = circkit_amd.Context(0)                                  # on its own stream
    d_out = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
    results = []
    for trial in range(3):
        d_hash = torch.zeros(n, dtype=torch.int64, device=dev)    # zeros: an overtaking table kernel would see them
        torch.cuda.synchronize()
        ctx.use_own_stream()
        ctx.canonicalize_batch_device(d_bytes, d_off, n, out_bytes=d_out, out_xxh3=d_hash)          # ~1 ms of kernels, not waited for
        fs, keep = uniq.first_seen(uniq.DeviceTable(ctx), d_hash, base_index=0)                    # torch's current stream
        torch.cuda.synchronize()
        results.append((fs.cpu().numpy(), int(keep.sum().item()), d_hash.cpu().numpy()))
    for fs, kept, h in results:
        assert kept == n - n // 2
        assert np.array_equal(fs.astype(np.uint64), O.uniq_first_seen(h.astype(np.uint64)))
    setup.close()
    ctx.close()


def test_hash_only_device_batches_enqueue_without_a_look_at_the_device(O):
    """circkit_canonicalize_batch_device with d_out_xxh3 and WITHOUT d_out_bytes (`uniq` without --canonicalize,
    src/uniq.rs:45,55-60): two different batches enqueued back to back on the ctx stream, nothing waited for in between --
    the call neither synchronises nor allocates from a device-side size (it used to copy offsets[n] back to
    size a scratch for the canonical bytes).  Records whose hash is not fused (<= 240 symbols, N / gap records, long ones)
    are hashed through their view of the input.  Hashes against the oracle."""
    import torch
    import circkit_amd
    from tests import seqsets
    dev = torch.device("cuda", 0)
    ctx = circkit_amd.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    batches = []
    for seed in (1, 2):
        seqs = seqsets.random_mixed(100 * seed, 3000, 900, 1008) + seqsets.random_mixed(100 * seed + 1, 400, 1, 260) + \
            seqsets.random_mixed(100 * seed + 2, 300, 48, 1000, b"ACGTN") + seqsets.random_mixed(100 * seed + 3, 60, 1009, 30000) + \
            seqsets.random_mixed(100 * seed + 4, 40, 10, 500, b"-ACGNT") + seqsets.random_mixed(100 * seed + 5, 20, 1, 300, bytes(range(0x21, 0x7F))) + \
            [b"", b"ACGTRYKMacgtn" * 40, seqsets.random_mixed(100 * seed + 6, 1, 120_000, 120_000)[0]]
        rng = np.random.default_rng(seed)
        seqs = [seqs[i] for i in rng.permutation(len(seqs))]
        data, offs = seqsets.pack(seqs)
        d_bytes = torch.from_numpy(np.concatenate([data, np.zeros(64, np.uint8)])).to(dev)
        d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
        d_hash = torch.zeros(len(seqs), dtype=torch.int64, device=dev)
        batches.append((data, offs, d_bytes, d_off, d_hash, len(seqs)))
    torch.cuda.synchronize()
    for _, _, d_bytes, d_off, d_hash, n in batches:                       # back to back
        ctx.canonicalize_batch_device(d_bytes, d_off, n, out_xxh3=d_hash)
    assert ctx.batch_status() == 0
    for data, offs, _, _, d_hash, n in batches:
        _, exp_h = O.canonicalize_batch(data, offs, False, True, threads=8)
        got = d_hash.cpu().numpy().astype(np.uint64)
        bad = np.nonzero(got != exp_h)[0]
        assert len(bad) == 0, (len(bad), bad[:5], [int(offs[i + 1] - offs[i]) for i in bad[:5]])
    ctx.close()


@pytest.mark.parametrize("parts", ["1", "2", "3", "8", "16", "32"])
def test_host_batch_in_parts(O, parts):
    """circkit_canonicalize_batch moves a host batch through the device in parts (copy-in / kernels / copy-out of
    neighbouring parts overlap on the ctx's own copy streams; 16 MB and more per part by default, forced here on a small
    batch): every output of every part lands in its place -- bytes, index, strand, hash against the oracle --, with records of
    all alphabets and lengths so that the parts differ in what their kernels do, and a second call right behind the first."""
    import circkit_amd
    from tests import seqsets
    os.environ["CIRCKIT_HOST_BATCH_PARTS"] = parts
    try:
        ctx = circkit_amd.Context(0)
        seqs = seqsets.random_mixed(900, 700, 900, 1008) + seqsets.random_mixed(901, 60, 1009, 12000) + seqsets.random_mixed(902, 100, 1, 300) + \
            seqsets.random_mixed(903, 80, 48, 2000, b"ACGTN") + seqsets.random_mixed(904, 30, 10, 400, b"-ACGNT") + [b"", b"A" * 5000, b"ACGT" * 700]
        for rnd in range(2):
            rng = np.random.default_rng(rnd)
            batch = [seqs[i] for i in rng.permutation(len(seqs))]
            data, offs = seqsets.pack(batch)
            got = ctx.canonicalize_batch(data, offs, want_bytes=True, want_index=True, want_strand=True, want_xxh3=True)
            exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
            assert np.array_equal(got["bytes"], exp) and np.array_equal(got["xxh3"], exp_h)
            for i in rng.choice(len(batch), size=200, replace=False):
                _, st, idx = seqsets.expected(O, batch[int(i)])
                assert int(got["strand"][i]) == st and (not batch[int(i)] or int(got["index"][i]) == idx), int(i)
            lean = ctx.canonicalize_batch(data, offs, want_bytes=True)
            assert np.array_equal(lean["bytes"], exp)
        ctx.close()
    finally:
        del os.environ["CIRCKIT_HOST_BATCH_PARTS"]


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [0, 16, 5])
def test_host_batch_page_locked_output(O, shift):
    """Page-locked input and output buffers (circkit_host_alloc), five parts: the copy-in rides the ctx stream between the
    parts' kernels, the copy-out runs on the ctx's priority stream at the same time -- same bytes as the oracle's, at any
    alignment of the output pointer and of the part boundaries, and nothing written outside the payload."""
    import ctypes
    import circkit_amd
    from tests import seqsets
    os.environ["CIRCKIT_HOST_BATCH_PARTS"] = "5"
    try:
        ctx = circkit_amd.Context(0)
        lib = circkit_amd.load_library()
        seqs = seqsets.random_mixed(910, 900, 900, 1008) + seqsets.random_mixed(911, 80, 1009, 9000) + seqsets.random_mixed(912, 200, 1, 333) + [b"", b"ACGTA" * 777]
        data, offs = seqsets.pack(seqs)
        nb = len(data)
        exp, _ = O.canonicalize_batch(data, offs, True, False, threads=8)
        pin_in, pin_out = lib.circkit_host_alloc(nb + 64), lib.circkit_host_alloc(nb + 64 + 32)
        assert pin_in and pin_out
        try:
            ctypes.memmove(pin_in, data.ctypes.data, nb)
            ctypes.memset(pin_out, 0xEE, nb + 64 + 32)
            offs64 = np.ascontiguousarray(offs, dtype=np.uint64)
            for _ in range(2):
                rc = lib.circkit_canonicalize_batch(ctx._h, pin_in, offs64.ctypes.data, len(seqs), pin_out + shift, None, None, None)
                assert rc == 0
                got = np.frombuffer((ctypes.c_uint8 * (nb + 64 + 32)).from_address(pin_out), dtype=np.uint8)
                assert np.array_equal(got[shift:shift + nb], exp)
                assert (got[:shift] == 0xEE).all() and (got[shift + nb:] == 0xEE).all()       # nothing outside the payload
        finally:
            lib.circkit_host_free(pin_in)
            lib.circkit_host_free(pin_out)
        ctx.close()
    finally:
        del os.environ["CIRCKIT_HOST_BATCH_PARTS"]


@pytest.mark.gpu
@pytest.mark.parametrize("want_hash", [False, True, "aux", "hash-only"])
def test_mode_guess_follows_the_data(O, want_hash):
    """launch_canon takes the mode the last device batches reported as the next batch's mode (only that mode's kernels are
    launched) once two launches in a row have read the same report.  Runs of batches of one kind with a change of kind
    behind each run -- short records, mixed lengths, short records with N, 1.5 kb records, records near 1 kb (the bytes-only build
    for batches of short records against the one for the others) -- so that every change meets a
    WRONG guess first: the bytes (and hashes) are the oracle's all the same, and every batch reports its OWN mode."""
    import torch
    import circkit_amd
    from tests import seqsets
    dev = torch.device("cuda", 0)
    ctx = circkit_amd.Context(0)
    want_aux = want_hash == "aux"          # + rotation index and strand (the builds with every output: no mixed-length kernel)
    hash_only = want_hash == "hash-only"   # no canonical bytes: views + the xxh3 pass for what the fused hash does not cover
    want_hash = want_hash in (True, "hash-only")
    n = 3000
    kinds = [(seqsets.random_mixed(301, n, 300, 1008), 1),
             (seqsets.random_mixed(302, n // 4, 200, 9000) + seqsets.random_mixed(303, n // 4, 2100, 20000), 3),
             (seqsets.random_mixed(304, n, 200, 1008, b"ACGTACGTACGTACGTACGTN"), 1),
             (seqsets.random_mixed(305, n, 1400, 1900), 2),
             (seqsets.random_mixed(306, n // 4, 2500, 12000, b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTN"), 3),
             (seqsets.random_mixed(307, n, 850, 1008), 1)]          # mode 1 like kind 0, but not MODE_SHORT: the other bytes-only build
    packed = []
    for seqs, mode in kinds:
        data, offs = seqsets.pack(seqs)
        exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
        packed.append((torch.from_numpy(np.concatenate([data, np.zeros(64, dtype=np.uint8)])).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev),
                       len(seqs), len(data), exp, exp_h, mode))
    order = [0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 0, 4, 4, 4, 4, 1, 0, 5, 5, 5, 5, 0, 0, 0, 5]
    for wait in (True, False):          # with a look at the device after every batch, and enqueued back to back
        outs = []
        for k in order:
            d_bytes, d_off, cnt, nb, exp, exp_h, mode = packed[k]
            d_out = torch.zeros(nb + 64, dtype=torch.uint8, device=dev)
            d_hash = torch.zeros(cnt, dtype=torch.int64, device=dev) if want_hash else None
            d_idx = torch.zeros(cnt, dtype=torch.int32, device=dev) if want_aux else None
            d_strand = torch.zeros(cnt, dtype=torch.uint8, device=dev) if want_aux else None
            ctx.canonicalize_batch_device(d_bytes, d_off, cnt, out_bytes=None if hash_only else d_out, out_xxh3=d_hash, out_index=d_idx, out_strand=d_strand)
            if wait:
                # (a batch of two-word records whose XXH3 is wanted -- and no index / strand -- is the mixed-length kernels')
                assert ctx.last_batch_mode() == (3 if mode == 2 and want_hash and not want_aux else mode), (k, mode)
            outs.append((k, d_out, d_hash, d_idx, d_strand))
        assert ctx.batch_status() == 0
        for k, d_out, d_hash, d_idx, d_strand in outs:
            _, _, cnt, nb, exp, exp_h, _ = packed[k]
            if not hash_only:
                assert np.array_equal(d_out[:nb].cpu().numpy(), exp), k
            if want_hash:
                assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64), exp_h), k
            if want_aux:
                gi, gs = d_idx.cpu().numpy(), d_strand.cpu().numpy()
                for i in range(0, cnt, max(1, cnt // 40)):
                    _, st, idx = seqsets.expected(O, kinds[k][0][i])
                    assert int(gs[i]) == st and int(gi[i]) == idx, (k, i)
    ctx.close()


@pytest.mark.gpu
def test_host_batch_staging_grows_and_shrinks_with_the_batch(O):
    """The ctx's page-locked staging of offsets / hashes / indices / strands (host_batch) is sized by the largest batch seen:
    a small batch, a 40x larger one, a small one again through ONE ctx -- every per-record output lands in its place."""
    import circkit_amd
    from tests import seqsets
    ctx = circkit_amd.Context(0)
    for seed, count in ((401, 500), (402, 20000), (403, 300), (404, 21000), (405, 1)):
        seqs = seqsets.random_mixed(seed, count, 30, 700)
        data, offs = seqsets.pack(seqs)
        got = ctx.canonicalize_batch(data, offs, want_bytes=True, want_index=True, want_strand=True, want_xxh3=True)
        exp, exp_h = O.canonicalize_batch(data, offs, True, True, threads=8)
        assert np.array_equal(got["bytes"], exp) and np.array_equal(got["xxh3"], exp_h)
        for i in range(0, count, max(1, count // 50)):
            _, st, idx = seqsets.expected(O, seqs[i])
            assert int(got["strand"][i]) == st and int(got["index"][i]) == idx, (count, i)
    ctx.close()


def _expected_first_seen(h):
    """smallest index per value (numpy): order of first occurrence"""
    order = np.argsort(h, kind="stable")
    hs = h[order]
    first_of_group = np.r_[True, hs[1:] != hs[:-1]]
    group_first_idx = order[first_of_group]                     # stable sort: the first of each run is the smallest index
    group_id = np.cumsum(first_of_group) - 1
    fs = np.empty(len(h), dtype=np.int64)
    fs[order] = group_first_idx[group_id]
    return fs


@pytest.mark.parametrize("shape", ["random", "empty_key", "one_bucket", "all_equal", "small", "threshold", "beyond"])
def test_resolve_in_lds_buckets_and_its_fallback(shape):
    """circkit_uniq_resolve_device (round 4): shards of 2^19 keys and more are resolved in LDS-sized buckets (count matrix ->
    scans -> scatter of {hash, index} rows -> one workgroup per bucket); a bucket beyond 3072 keys sends the whole shard to the
    HBM table instead.  first_seen / keep against a NumPy restatement of src/uniq.rs:47-48 for: random keys with ~3x
    duplication, the table's EMPTY marker (~0) as a key, keys that all share their top bits (one bucket: fallback), all keys
    equal (fallback), and a shard below the threshold (the HBM table directly)."""
    import torch
    import circkit_amd
    rng = np.random.default_rng(77)
    n = {"small": 200_000, "threshold": 1 << 19, "beyond": 22_000_000}.get(shape, 3_000_000)       # (beyond 21M keys: the HBM table again)
    if shape == "all_equal":
        h = np.full(n, 0x1234567890ABCDEF, dtype=np.uint64)
    else:
        h = rng.integers(0, 1 << 63, size=n // 3, dtype=np.uint64)[rng.integers(0, n // 3, size=n)] * np.uint64(2) + np.uint64(1)
        if shape == "empty_key":
            h[rng.integers(0, n, size=1000)] = np.uint64(0xFFFFFFFFFFFFFFFF)
            h[5] = np.uint64(0xFFFFFFFFFFFFFFFF)
        if shape == "one_bucket":
            h = (h >> np.uint64(20)) | np.uint64(0xABCDE00000000000)
    exp = _expected_first_seen(h)
    c = circkit_amd.Context(0)
    dev = torch.device("cuda", 0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    d_h = torch.from_numpy(h.view(np.int64)).to(dev)
    for base in (0, 7_000_000_000):
        fs = torch.full((n,), -1, dtype=torch.int64, device=dev)
        keep = torch.full((n,), 7, dtype=torch.uint8, device=dev)
        c.uniq_resolve_device(d_h, n, base, fs, keep)
        c.uniq_status()
        got = fs.cpu().numpy()
        assert np.array_equal(got, exp + base), (shape, base, int((got != exp + base).sum()))
        assert np.array_equal(keep.cpu().numpy(), (exp == np.arange(n)).astype(np.uint8))
    # the streaming table still works behind a resolve (it refuses until reset, then takes a stream)
    c.uniq_reset(1000)
    out = c.uniq_first_seen(np.array([5, 6, 5], dtype=np.uint64), base_index=10)
    assert out.tolist() == [10, 11, 10]
    c.close()
