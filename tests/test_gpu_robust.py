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
