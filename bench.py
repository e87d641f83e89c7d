#!/usr/bin/env python3
"""bench.py -- canonicalize throughput on synthetic FASTA payload, BASELINE.json's metric.

A "step" is one pass of the hot path (circkit_canonicalize_batch_device: both strands' least rotation,
select, emit canonical bytes) over one device-resident CSR batch.  Default workload = BASELINE.json
configs[1]: 10,000,000 records x 1,000 b, i.i.d. uniform ACGT, seed 42, generated ON the device.
Records shard across GPUs with no data-path collective (weak scaling: every rank owns --records
records); the only collectives are the timing barrier and a MAX over ranks.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def pmc_traffic(n_records, length):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes
    (profiles/traffic.json; bench.py cannot collect PMC counters itself).  None if the profile is for
    another workload."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if t.get("workload") == "canonicalize %d x %d" % (n_records, length):
            return t["traffic_bytes"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--records", type=int, default=10_000_000, help="records per GPU")
    ap.add_argument("--length", type=int, default=1000)
    ap.add_argument("--cpu-sample", type=int, default=3_000_000,
                    help="records timed on the host cores (rank 0, N=1): ~20 core-seconds of work at the default")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workload", default="canonicalize", choices=["canonicalize", "uniq", "mixed"],
                    help="canonicalize = BASELINE configs[1] (the headline metric); uniq = configs[2] (50 %% rotational/"
                         "strand duplicates, hash + first-seen); mixed = configs[3] (1M records, L ~ 1/L on [200, 20000])")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("CIRCKIT_BENCH_FORCE_DIST") == "1"   # the latter: rehearse the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import circkit_amd
    ctx = circkit_amd.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    N, L = args.records, args.length
    if args.workload == "mixed":
        # config 4: lengths with P(L) ~ 1/L on [200, 20000] (log-uniform), seed 45
        N = min(N, 1_000_000)
        g = torch.Generator(device="cpu").manual_seed(45 + rank)
        u = torch.rand(N, generator=g, dtype=torch.float64)
        lens = torch.exp(np.log(200.0) + u * (np.log(20000.0) - np.log(200.0))).to(torch.int64)
        offs = torch.zeros(N + 1, dtype=torch.int64)
        offs[1:] = torch.cumsum(lens, 0)
        total = int(offs[-1])
        d_off = offs.to(dev)
    else:
        total = N * L
        d_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
        ctx.fixed_offsets_device(0, L, N, d_off)
    d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    d_out = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    # global base index keeps every rank's shard distinct: record g of the job = bases [g*L, (g+1)*L)
    ctx.synth_fill_device(42 if args.workload != "mixed" else 45, rank * total, total, d_bytes)
    d_hash = d_fs = None
    if args.workload == "uniq":
        # config 3: the second half of the shard = uniformly chosen records of the first half, rotated by a
        # uniform k and reverse-complemented with p = 0.5; then the whole shard is shuffled (seeds 43/44)
        half = N // 2
        gen = torch.Generator(device=dev).manual_seed(43 + rank)
        lut = torch.arange(256, dtype=torch.uint8, device=dev)
        for a_, b_ in zip(b"ACGT", b"TGCA"):
            lut[a_] = b_
        view = d_bytes[:N * L].view(N, L)
        col = torch.arange(L, device=dev)
        for s0 in range(half, N, 500_000):
            m = min(500_000, N - s0)
            src = torch.randint(0, half, (m,), generator=gen, device=dev)
            k = torch.randint(0, L, (m, 1), generator=gen, device=dev)
            rows = torch.gather(view[src], 1, (col.unsqueeze(0) + k) % L)
            flip = torch.rand(m, generator=gen, device=dev) < 0.5
            rows[flip] = lut[rows[flip].flip(1).long()]
            view[s0:s0 + m] = rows
        perm = torch.randperm(N, generator=torch.Generator(device=dev).manual_seed(44 + rank), device=dev)
        for c0 in range(0, L, 100):                       # shuffle records column block by column block (bounded temp)
            view[:, c0:c0 + 100] = view[perm, c0:c0 + 100]
        d_hash = torch.empty(N, dtype=torch.int64, device=dev)
        d_fs = torch.empty(N, dtype=torch.int64, device=dev)
        del view, perm
    torch.cuda.synchronize()

    def step():
        if args.workload == "uniq":
            # canonical bytes + XXH3 + first-seen table: what `circkit uniq --canonicalize` computes per batch
            ctx.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=d_out, out_xxh3=d_hash)
            ctx.uniq_reset(N)
            ctx.uniq_insert_device(d_hash, N, rank * N)
            ctx.uniq_lookup_device(d_hash, N, d_fs)
        else:
            ctx.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=d_out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps      # HIP events on the launch stream
    unprocessed = ctx.batch_status()
    if unprocessed:
        raise SystemExit("%d records were not processed" % unprocessed)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    result = None
    if rank == 0:
        seq_per_s = world * N * args.steps / dt
        algo_bytes = 2 * total + 8 * N                 # read L + write L per record + one u64 offset
        if args.workload == "uniq":
            algo_bytes += 8 * N + 32 * N               # + u64 hash per record + table insert/lookup (key + value each)
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        result = {
            "metric": {"canonicalize": "canonicalize sequences/sec (10M x 1kb synthetic FASTA payload)",
                       "uniq": "uniq sequences/sec (10M x 1kb, ~50% rotational/strand duplicates)",
                       "mixed": "canonicalize sequences/sec (1M records, 200b-20kb log-uniform lengths)"}[args.workload],
            "value": seq_per_s,
            "unit": "sequences/s",
            "gbases_per_s": world * total * args.steps / dt / 1e9,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": {"canonicalize": "canonicalize, %d records x %d b per GPU, uniform ACGT seed 42, "
                                                    "device-resident CSR (BASELINE configs[1])" % (N, L),
                                    "uniq": "uniq --canonicalize, %d records x %d b per GPU, half are rotated / reverse-"
                                            "complemented copies, shuffled (BASELINE configs[2])" % (N, L),
                                    "mixed": "canonicalize, %d records, lengths log-uniform on [200, 20000], %d bases per "
                                             "GPU (BASELINE configs[3])" % (N, total)}[args.workload],
                       "records_per_gpu": N, "record_len": L,
                       "parallelism": "records sharded over %d GPU(s), no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": pmc_traffic(N, L) if args.workload == "canonicalize" else None,
                         "kernel": "canon_stream_kernel", "kernel_ms": kernel_ms, "algorithmic_bytes": algo_bytes},
        }
        if args.workload == "uniq":
            result["unique_records"] = int((d_fs == torch.arange(rank * N, rank * N + N, device=dev)).sum().item())
        if world == 1 and not args.no_cpu and args.workload == "canonicalize":
            from oracle import oracle as O
            S = min(args.cpu_sample, N)
            h_in = d_bytes[:S * L].cpu().numpy()
            h_off = np.arange(S + 1, dtype=np.uint64) * np.uint64(L)
            cores = min(len(os.sched_getaffinity(0)), 16)
            O.lib()
            c0 = time.perf_counter()
            h_out, _ = O.canonicalize_batch(h_in, h_off, True, False, threads=cores)
            cdt = time.perf_counter() - c0
            same = bool(np.array_equal(h_out, d_out[:S * L].cpu().numpy()))
            S1 = min(S, 100_000)                    # the same restatement on ONE core, smaller sample
            c1 = time.perf_counter()
            O.canonicalize_batch(h_in[:S1 * L], h_off[:S1 + 1], True, False, threads=1)
            one_core = S1 / (time.perf_counter() - c1)
            result["cpu_baseline"] = {
                "value": S / cdt, "unit": "sequences/s", "cores": cores, "host_cores_visible": len(os.sched_getaffinity(0)),
                "kind": "port", "one_core_value": one_core,
                "sample": "first %d records of the same device-generated batch; C restatement of the reference "
                          "path (linear-time byte-indexed Duval variant, faster than the reference's O(n^2) "
                          "chars().nth() loop), %d pthreads" % (S, cores),
                "gpu_output_matches": same,
            }
            if not same:
                print(json.dumps(result))
                raise SystemExit("GPU output differs from the CPU oracle on the sample")
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
