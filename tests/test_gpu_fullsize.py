"""GPU tests at BASELINE.json's full sizes (configs 3 and 4), of the per-batch build selection, and of the multi-GPU
`uniq` exchange over RCCL (backend nccl, rehearsed at world size 1 on the box's one GPU).  Run on a real MI355X:
pytest -m gpu.  Full-size checks use the size-independent properties of the canonical form (idempotence, rotation and
strand invariance, the expected number of distinct records) plus the oracle on a slice, as SURVEY.md 8(d) prescribes."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="module")
def ctx():
    import torch
    import circkit_amd
    c = circkit_amd.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def test_uniq_config3_full_size(ctx, O):
    """BASELINE configs[2] as bench.py generates it: 10M x 1 kb, half of the records rotated / reverse-complemented
    copies of the other half, shuffled.  Exactly 5,000,000 records survive; first-seen indices of the first 100k
    records equal the oracle's on the oracle's own hashes of that sample (SURVEY.md 8d cfg 3)."""
    import torch
    from circkit_amd import uniq, workloads as W
    N, L = 10_000_000, 1000
    dev = torch.device("cuda", 0)
    x, off = W.fixed_length(ctx, dev, N, L, 42, 0)
    W.plant_duplicates(x, N, L, dev, 43, 44)
    out = torch.empty_like(x)
    hs = torch.empty(N, dtype=torch.int64, device=dev)
    ctx.canonicalize_batch_device(x, off, N, out_bytes=out, out_xxh3=hs)
    assert ctx.batch_status() == 0
    table = uniq.DeviceTable(ctx)
    fs, keep = uniq.first_seen(table, hs, base_index=0)
    table.check()
    assert int(keep.sum().item()) == N - N // 2
    # a record is kept iff nothing before it has its hash, and every record points at a kept one with the same hash
    assert bool((fs <= torch.arange(N, device=dev)).all())
    assert bool(keep[fs].all()) and bool((hs[fs] == hs).all())
    S = 100_000
    h_off = np.arange(S + 1, dtype=np.uint64) * np.uint64(L)
    exp, exp_h = O.canonicalize_batch(x[:S * L].cpu().numpy(), h_off, True, True, threads=8)
    assert np.array_equal(out[:S * L].cpu().numpy(), exp)
    assert np.array_equal(hs[:S].cpu().numpy().astype(np.uint64), exp_h)
    assert np.array_equal(fs[:S].cpu().numpy().astype(np.uint64), O.uniq_first_seen(exp_h))
    # hash-only mode (uniq without --canonicalize) gives the same hashes without an output buffer
    hs2 = torch.empty_like(hs)
    ctx.canonicalize_batch_device(x, off, N, out_xxh3=hs2)
    torch.cuda.synchronize()
    assert torch.equal(hs, hs2)


@pytest.mark.parametrize("n_frac", [0.0, 0.01])
def test_mixed_config4_full_size(ctx, O, n_frac):
    """BASELINE configs[3] at its full 1M records (4.3 Gbases, lengths log-uniform on [200, 20000]), and its 1 % N
    variant: oracle bytes on a 5k-record slice; over all records idempotence and invariance under reverse
    complement + rotation."""
    import torch
    from circkit_amd import workloads as W
    N = 1_000_000
    dev = torch.device("cuda", 0)
    offs = W.log_uniform_offsets(N, 45)
    total = int(offs[-1])
    off = offs.to(dev)
    x = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    ctx.synth_fill_device(45, 0, total, x)
    if n_frac:
        W.sprinkle_n(x, total, n_frac, 46, dev)
    c1 = torch.full_like(x, 0x3F)
    ctx.canonicalize_batch_device(x, off, N, out_bytes=c1)
    assert ctx.batch_status() == 0
    assert ctx.last_batch_mode() == 3                       # a third of the records are longer than 2032 bases
    assert bool((c1[total:] == 0x3F).all())
    S = 5000
    h_off = offs[:S + 1].numpy().astype(np.uint64)
    nb = int(h_off[-1])
    exp, _ = O.canonicalize_batch(x[:nb].cpu().numpy(), h_off, True, False, threads=8)
    assert np.array_equal(c1[:nb].cpu().numpy(), exp)
    c2 = torch.empty_like(x)
    ctx.canonicalize_batch_device(c1, off, N, out_bytes=c2)
    torch.cuda.synchronize()
    assert torch.equal(c1[:total], c2[:total]), "not idempotent"
    del c2
    y = W.revcomp_rotate_csr(x, off, N, dev, shift=137)
    c3 = torch.empty_like(x)
    ctx.canonicalize_batch_device(y, off, N, out_bytes=c3)
    torch.cuda.synchronize()
    assert torch.equal(c1[:total], c3[:total]), "canonical form changed under reverse complement + rotation"
    del c3
    # `uniq` on the same batch (round 3: the mixed kernels' builds with the fused XXH3): hashes of bytes + hash and of the
    # hash-only call agree with each other over all records, with the oracle on the slice; the reverse-complemented /
    # rotated batch hashes to the same values (that is what makes them duplicates, src/uniq.rs:45-48)
    h1 = torch.zeros(N, dtype=torch.int64, device=dev)
    h2 = torch.zeros(N, dtype=torch.int64, device=dev)
    h3 = torch.zeros(N, dtype=torch.int64, device=dev)
    c4 = torch.full_like(x, 0x3F)
    ctx.canonicalize_batch_device(x, off, N, out_bytes=c4, out_xxh3=h1)
    ctx.canonicalize_batch_device(x, off, N, out_xxh3=h2)
    ctx.canonicalize_batch_device(y, off, N, out_xxh3=h3)
    assert ctx.batch_status() == 0
    assert torch.equal(c1[:total], c4[:total])
    assert torch.equal(h1, h2) and torch.equal(h1, h3)
    _, exp_h = O.canonicalize_batch(x[:nb].cpu().numpy(), h_off, False, True, threads=8)
    assert np.array_equal(h1[:S].cpu().numpy().astype(np.uint64), exp_h)


def test_one_percent_n_headline_shape(ctx, O):
    """10M x 1 kb with 1 % N (practically every record holds an N): oracle on a slice, idempotence over everything."""
    import torch
    from circkit_amd import workloads as W
    N, L = 10_000_000, 1000
    dev = torch.device("cuda", 0)
    x, off = W.fixed_length(ctx, dev, N, L, 42, 0)
    W.sprinkle_n(x, N * L, 0.01, 46, dev)
    c1 = torch.empty_like(x)
    hs = torch.empty(N, dtype=torch.int64, device=dev)
    ctx.canonicalize_batch_device(x, off, N, out_bytes=c1, out_xxh3=hs)
    assert ctx.batch_status() == 0
    S = 20000
    exp, exp_h = O.canonicalize_batch(x[:S * L].cpu().numpy(), np.arange(S + 1, dtype=np.uint64) * np.uint64(L), True, True, 8)
    assert np.array_equal(c1[:S * L].cpu().numpy(), exp)
    assert np.array_equal(hs[:S].cpu().numpy().astype(np.uint64), exp_h)
    c2 = torch.empty_like(x)
    ctx.canonicalize_batch_device(c1, off, N, out_bytes=c2)
    torch.cuda.synchronize()
    assert torch.equal(c1[:N * L], c2[:N * L])


def test_build_selection_follows_the_batch_not_the_buffer(ctx, O):
    """A streaming host reuses ONE offsets buffer: a batch of short records, then a batch of 1.5 kb records, then
    short ones again.  The mode is decided from each batch's own lengths (1, 2, 1), never remembered by pointer."""
    import torch
    from tests import seqsets
    dev = torch.device("cuda", 0)
    n = 4000
    batches = [seqsets.random_mixed(201, n, 300, 1008), seqsets.random_mixed(202, n, 1400, 1600), seqsets.random_mixed(203, n, 48, 900)]
    cap = max(sum(len(s) for s in b) for b in batches)
    d_bytes = torch.zeros(cap + 64, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)          # the one offsets buffer
    d_out = torch.zeros_like(d_bytes)
    for seqs, want_mode in zip(batches, (1, 2, 1)):
        data, offs = seqsets.pack(seqs)
        d_bytes[:len(data)] = torch.from_numpy(data).to(dev)
        d_off.copy_(torch.from_numpy(offs.astype(np.int64)))
        ctx.canonicalize_batch_device(d_bytes, d_off, n, out_bytes=d_out)
        assert ctx.last_batch_mode() == want_mode
        exp, _ = O.canonicalize_batch(data, offs, True, False, threads=8)
        assert np.array_equal(d_out[:len(data)].cpu().numpy(), exp)


def test_every_mode_and_alphabet_back_to_back_without_a_wait(O):
    """Batches of all six (mode, alphabet) kinds -- short / 1.5 kb / mixed lengths, pure and with 1 % N -- enqueued one
    behind the other on one stream, twice round, nothing waited for in between: every batch's kernels are chosen on the
    device from the batch itself (the builds that do not match return at once, the grid-size hint lags a batch behind), with
    bytes only and with bytes + XXH3.  Round 3 added the two mixed-length kernels to the set."""
    import random
    import torch
    import circkit_amd
    from tests import seqsets
    dev = torch.device("cuda", 0)
    ctx = circkit_amd.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = random.Random(9)

    def sprinkle(s):
        b = bytearray(s)
        for i in range(len(b)):
            if rng.random() < 0.01:
                b[i] = ord("N")
        return bytes(b)

    kinds = [seqsets.random_mixed(301, 6000, 300, 1008), seqsets.random_mixed(302, 4000, 1100, 1900), seqsets.random_mixed(303, 3000, 200, 20000)]
    kinds += [[sprinkle(s) for s in k] for k in kinds]
    jobs = []
    for rnd in range(2):
        for k, seqs in enumerate(kinds):
            data, offs = seqsets.pack(seqs)
            d_bytes = torch.from_numpy(np.concatenate([np.zeros(k % 16, np.uint8), data])).to(dev)[k % 16:]     # every payload alignment
            d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
            d_out = torch.zeros(len(data) + 64, dtype=torch.uint8, device=dev)
            d_hash = torch.zeros(len(seqs), dtype=torch.int64, device=dev) if (k + rnd) % 2 else None
            jobs.append((data, offs, d_bytes, d_off, d_out, d_hash, len(seqs)))
    torch.cuda.synchronize()
    for _, _, d_bytes, d_off, d_out, d_hash, n in jobs:
        ctx.canonicalize_batch_device(d_bytes, d_off, n, out_bytes=d_out, out_xxh3=d_hash)
    assert ctx.batch_status() == 0
    for j, (data, offs, _, _, d_out, d_hash, n) in enumerate(jobs):
        exp, exp_h = O.canonicalize_batch(data, offs, True, d_hash is not None, threads=8)
        assert np.array_equal(d_out[:len(data)].cpu().numpy(), exp), j
        if d_hash is not None:
            assert np.array_equal(d_hash.cpu().numpy().astype(np.uint64), exp_h), j
    ctx.close()


def _run(cmd, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("exchange", ["partition", "allgather"])
def test_uniq_exchange_over_rccl_world_1(exchange):
    """circkit_amd/uniq.py's multi-GPU path with the HIP table on a device and backend nccl (= RCCL): all_to_all_single
    / all_gather + circkit_uniq_insert_pairs_device + lookup, checked against the oracle's first-seen.  One rank (the
    box has one GPU); its own process because it owns a process group."""
    out = _run([sys.executable, os.path.join(ROOT, "tools", "gpu_uniq_nccl.py"), exchange])
    assert "first-seen matches the oracle" in out


def test_bench_uniq_runs_the_exchange_and_counts_globally():
    """bench.py --workload uniq with the RCCL path forced: the timed step calls uniq.first_seen over nccl and the line
    reports the job's unique count (bench.py itself fails if it is not records/2)."""
    import json
    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "uniq", "--records", "400000", "--steps", "2", "--warmup", "1"],
               env={"CIRCKIT_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29541"})
    line = json.loads(out.strip().splitlines()[-1])
    assert line["unique_records"] == 200000
    assert "RCCL" in line["config"]["parallelism"]
    assert line["cpu_baseline"]["gpu_output_matches"] is True


def test_bench_two_ranks_rehearsal_checks_first_seen_across_ranks():
    """bench.py --workload uniq at world 2 on this one GPU (both ranks on device 0, collectives over gloo: a rehearsal, RCCL
    refuses two ranks per device): every rank plants duplicates of BOTH ranks' base records, the all-gather exchange resolves
    first-seen across the ranks inside the timed step, and bench.py's own job-wide check passes -- every record of every shard
    against the planting decisions, an oracle slice per rank, records owned by the other rank present."""
    import json
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
                os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "uniq", "--exchange", "allgather", "--records", "200000",
                "--steps", "2", "--warmup", "1"], env={"CIRCKIT_BENCH_SHARE_GPU": "1", "CIRCKIT_BENCH_BACKEND": "gloo"})
    line = json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["unique_records"] == 200000
    chk = line["uniq_job_check"]
    assert chk["first_seen_mismatches_job"] == 0 and chk["records_owned_by_another_rank_job"] > 50000
    assert chk["oracle_slice"]


def test_bench_two_ranks_rehearsal_default_workload():
    """The line the driver's scaling run asks for -- `bench.py --gpus N`, default workload, started WITHOUT a launcher around it --
    rehearsed with two ranks on this one GPU over gloo: the parent starts its own ranks, every rank's shard is its own range of
    the job's seed stream, MAX-over-ranks timing, one JSON line from rank 0, clean exit of both ranks."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"CIRCKIT_BENCH_SHARE_GPU": "1", "CIRCKIT_BENCH_BACKEND": "gloo"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--records", "300000", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["records_total"] == 600000 and line["scaling"] == "weak"
    assert line["roofline"]["frac"] > 0 and "other_workloads" not in line and "cli" not in line


@pytest.mark.parametrize("workload", [["--workload", "uniq"], ["--workload", "mixed", "--with-hash"]])
def test_bench_hash_only_lines(workload):
    """bench.py --hash-only (`circkit uniq` without --canonicalize: no canonical bytes are written, SURVEY 8d's L + 16 bytes per
    record): the line prices the step on that formula, the hashes of the device path, of the host-buffer path and of the CPU
    oracle agree, and the unique count is right."""
    import json
    records = "300000" if workload[1] == "uniq" else "40000"
    out = _run([sys.executable, os.path.join(ROOT, "bench.py")] + workload + ["--hash-only", "--records", records, "--steps", "2", "--warmup", "1", "--no-copy"])
    line = json.loads(out.strip().splitlines()[-1])
    assert "hash-only" in line["metric"]
    n = int(records)
    total = line["roofline"]["algorithmic_bytes"] - 16 * n
    assert total == (n * 1000 if workload[1] == "uniq" else total) and total > 0
    assert line["end_to_end"]["matches_device_path"] is True
    assert line["cpu_baseline"]["gpu_output_matches"] is True
    if workload[1] == "uniq":
        assert line["unique_records"] == n // 2


def test_config5_rank7_shard(ctx, O):
    """BASELINE configs[4] as far as one GPU can run it: the shard rank 7 of `bench.py --gpus 8` takes -- 12,500,000 records x
    1 kb at global record index 87,500,000 (base offsets up to 10^11 in the counter-based generator, 12.5 GB in + 12.5 GB
    out in ONE batch).  Oracle bytes on a 20k slice whose INPUT is regenerated on the host at the same global base (so the
    device generator's 64-bit base arithmetic is pinned too), two slices deep inside the shard, idempotence and invariance
    under reverse complement + rotation over the whole shard (lib/src/canonicalize.rs:54-63, :124-132, :216-231)."""
    import torch
    from circkit_amd import workloads as W
    N, L, rank = 12_500_000, 1000, 7
    dev = torch.device("cuda", 0)
    x, off = W.fixed_length(ctx, dev, N, L, 42, first_record=rank * N)
    c1 = torch.full_like(x, 0x3F)
    ctx.canonicalize_batch_device(x, off, N, out_bytes=c1)
    assert ctx.batch_status() == 0
    assert ctx.last_batch_mode() == 1
    assert bool((c1[N * L:] == 0x3F).all())
    h_off = np.arange(20_001, dtype=np.uint64) * np.uint64(L)
    for r0 in (0, 6_250_000, N - 20_000):             # first, middle and last records of the shard
        host_in = O.synth_fill(42, (rank * N + r0) * L, 20_000 * L)
        assert np.array_equal(host_in, x[r0 * L:(r0 + 20_000) * L].cpu().numpy()), "device generator differs from the host's at the global base"
        exp, _ = O.canonicalize_batch(host_in, h_off, True, False, threads=8)
        assert np.array_equal(c1[r0 * L:(r0 + 20_000) * L].cpu().numpy(), exp)
    c2 = torch.empty_like(x)
    ctx.canonicalize_batch_device(c1, off, N, out_bytes=c2)
    assert ctx.batch_status() == 0
    assert torch.equal(c1[:N * L], c2[:N * L]), "not idempotent"
    del c2
    y = W.revcomp_rotate_csr(x, off, N, dev, shift=137)
    c3 = torch.empty_like(x)
    ctx.canonicalize_batch_device(y, off, N, out_bytes=c3)
    assert ctx.batch_status() == 0
    assert torch.equal(c1[:N * L], c3[:N * L]), "canonical form changed under reverse complement + rotation"


def test_job_wide_uniq_workload_on_one_gpu(ctx, O):
    """The multi-rank `uniq` workload of bench.py (workloads.plant_job_duplicates: a duplicate's original lives on ANY rank)
    with all 8 shards built and hashed one after the other on this GPU: the device hashes of the whole job, resolved by the
    oracle's first-seen map, equal the expectation bench.py checks every rank against (workloads.job_first_seen), most
    first-seen records live on another rank, and every shard's first records pass the oracle slice (bytes, XXH3, and canonical
    form == that of the base record its key names)."""
    import torch
    import bench
    from circkit_amd import workloads as W
    world, N, L = 8, 40_000, 1000
    dev = torch.device("cuda", 0)

    def fill(seed, first_base, n_bases):
        buf = torch.empty(n_bases + 64, dtype=torch.uint8, device=dev)
        ctx.synth_fill_device(seed, first_base, n_bases, buf)
        return buf
    job_h = []
    for rank in range(world):
        x, off = W.fixed_length(ctx, dev, N, L, 42, rank * N)
        W.plant_job_duplicates(fill, x, N, L, dev, rank, world)
        out = torch.empty_like(x)
        hs = torch.empty(N, dtype=torch.int64, device=dev)
        ctx.canonicalize_batch_device(x, off, N, out_bytes=out, out_xxh3=hs)
        assert ctx.batch_status() == 0
        job_h.append(hs.cpu().numpy().astype(np.uint64))
        keys = W.job_keys(N, L, world, rank, dev)

        class A:
            hash_only = False
        assert bench.job_oracle_slice(np, torch, A, N, L, rank, x, {"out": out, "hash": hs}, keys) is None
    fs = O.uniq_first_seen(np.concatenate(job_h)).astype(np.int64)
    cross = 0
    for rank in range(world):
        exp, distinct, _ = W.job_first_seen(N, L, world, rank, dev)
        assert distinct == world * (N // 2)
        assert np.array_equal(exp.cpu().numpy(), fs[rank * N:(rank + 1) * N])
        cross += int(((exp // N) != rank).sum())
    assert cross > world * N // 4
