#!/usr/bin/env python3
"""bench.py -- canonicalize throughput on synthetic FASTA payload, BASELINE.json's metric.

A "step" is one pass of the hot path (circkit_canonicalize_batch_device: both strands' least rotation,
select, emit canonical bytes) over one device-resident CSR batch.  Default workload = BASELINE.json
configs[1]: 10,000,000 records x 1,000 b, i.i.d. uniform ACGT, seed 42, generated ON the device; with
--gpus 8 it is configs[4]: 100M records, 12.5M per GPU.  Records shard across GPUs by contiguous global
index ranges with no data-path collective (weak scaling); the only collectives of `canonicalize` are the
timing barrier and a MAX over ranks.  `--workload uniq` adds the one real exchange of the path: the
hash-range all-to-all of circkit_amd/uniq.py over RCCL, inside the timed step.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset) starts its own ranks: the parent -- which
never imports torch and never touches a GPU -- runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on this
very file as a child process, lets rank 0's JSON line through on stdout and exits with the launcher's code.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def pmc_traffic(workload):
    """(HBM bytes per step, source) from the committed rocprofv3 --pmc passes: bench.py cannot collect PMC counters
    itself, so the number is the one of profiles/traffic.json when that profile is of this very workload."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        for entry in t.get("entries", [t]):
            if entry.get("workload") == workload:
                return entry["traffic_bytes"], "profiles/traffic.json (%s; %s)" % (entry.get("source", "rocprofv3 --pmc, separate passes"),
                                                                                   entry.get("scope", "per launch"))
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def self_launch(n_ranks):
    """One process per GPU, started from here: the launcher is a CHILD (never exec: this process may already hold state the
    box forbids replacing), stdout / stderr are inherited so rank 0's JSON line is this command's output."""
    import signal
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the launcher and its ranks get a session of their own: a SIGTERM / SIGINT that reaches only this process (a driver's
    # timeout) is passed on to the whole group, and nothing is left behind holding GPU memory when this process goes
    child = subprocess.Popen(cmd, env=env, start_new_session=True)

    def forward(signum, _frame):
        try:
            os.killpg(child.pid, signum)
        except ProcessLookupError:
            pass
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, forward)
    try:
        return child.wait()
    finally:
        if child.poll() is None:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--records", type=int, default=None,
                    help="records per GPU (default: 10M; 12.5M at --gpus 8 = BASELINE configs[4]; 1M for --workload mixed)")
    ap.add_argument("--length", type=int, default=1000)
    ap.add_argument("--n-frac", type=float, default=0.0, help="fraction of bases replaced by N (SURVEY 8d: the 1 %% N variants)")
    ap.add_argument("--cpu-sample", type=int, default=None,
                    help="records timed on the host cores (rank 0, N=1); default: 100k records per usable core (~0.7 s)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive host-buffer leg (end_to_end)")
    ap.add_argument("--no-copy", action="store_true", help="skip the same-box plain-copy yardstick (roofline.copy_ceiling_gbps)")
    ap.add_argument("--no-cli", action="store_true", help="default run only: skip the `cli` block (the circkit binary on 5 GB of FASTA in /dev/shm)")
    ap.add_argument("--no-others", action="store_true",
                    help="default run (canonicalize, 1 GPU) only: skip the `other_workloads` block -- BASELINE configs[2] and [3] and their "
                         "variants, 5 steps each in the same process AFTER the headline's timed region (never inside it; `value` is untouched)")
    ap.add_argument("--exchange", default="partition", choices=["partition", "allgather"], help="uniq at --gpus > 1")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams (each with its own ctx and output buffers) the steps are dealt to round-robin, so that the "
                         "short kernels behind a batch's main kernel (rescue pass, LDS tiers, table) overlap the next batch's main "
                         "kernel instead of running alone on the chip (measured: headline 3.71 -> 3.57 ms per step, mixed 2.36 -> 2.23); "
                         "the default 1 = strictly one batch after the other, so that the per-step time IS the kernel chain's time "
                         "and agrees with a rocprofv3 kernel trace of the same command")
    ap.add_argument("--max-len", type=int, default=20000, help="mixed: upper end of the log-uniform lengths (BASELINE configs[3]: 20000)")
    ap.add_argument("--with-hash", action="store_true",
                    help="mixed: also the XXH3 of every record and the first-seen resolution (`circkit uniq --canonicalize` on records of mixed lengths)")
    ap.add_argument("--hash-only", action="store_true",
                    help="uniq / --with-hash: no canonical bytes are written (`circkit uniq` without --canonicalize, src/uniq.rs:55-60): "
                         "XXH3 of the canonical form + first-seen only; SURVEY 8d prices this mode at L + 16 bytes per record")
    ap.add_argument("--workload", default="canonicalize", choices=["canonicalize", "uniq", "mixed"],
                    help="canonicalize = BASELINE configs[1] (the headline metric); uniq = configs[2] (50 %% rotational/"
                         "strand duplicates, hash + first-seen); mixed = configs[3] (1M records, L ~ 1/L on [200, 20000])")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    force_dist = os.environ.get("CIRCKIT_BENCH_FORCE_DIST") == "1"   # rehearse the launcher + RCCL path on one GPU
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_dist):
        sys.exit(self_launch(args.gpus))           # before anything of torch / HIP is loaded in this process

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
    # CIRCKIT_BENCH_SHARE_GPU=1 + CIRCKIT_BENCH_BACKEND=gloo: a REHEARSAL of the multi-rank path on a one-GPU box -- every rank
    # on device 0, the collectives over gloo (which moves device tensors through the host; all_gather and all_reduce only, so
    # `--exchange allgather`).  RCCL refuses two ranks on one device; nothing measured this way is a result.
    share_gpu = os.environ.get("CIRCKIT_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("CIRCKIT_BENCH_BACKEND", "nccl")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints a version banner on STDOUT when its first communicator comes up; this command's stdout is the ONE JSON
        # line.  File descriptor 1 points at stderr until the communicator exists (a barrier forces it into being).
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    import circkit_amd
    from circkit_amd import uniq as U
    from circkit_amd import workloads as W
    ctx = circkit_amd.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    config5 = args.workload == "canonicalize" and world == 8 and args.records is None and args.length == 1000
    N = args.records if args.records is not None else (12_500_000 if config5 else 10_000_000)
    L = args.length
    if args.workload == "mixed":
        # config 4: lengths with P(L) ~ 1/L on [200, 20000] (log-uniform), seed 45
        N = args.records if args.records is not None else 1_000_000
        offs = W.log_uniform_offsets(N, 45 + rank, 200, args.max_len)
        total = int(offs[-1])
        d_off = offs.to(dev)
        d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
        ctx.synth_fill_device(45, rank * total, total, d_bytes)
    else:
        # global record index keeps every rank's shard distinct: record g of the job = bases [g*L, (g+1)*L) of seed 42
        total = N * L
        d_bytes, d_off = W.fixed_length(ctx, dev, N, L, 42, rank * N)
    if args.n_frac > 0:
        W.sprinkle_n(d_bytes, total, args.n_frac, 46 + rank, dev)
    if args.workload == "uniq" and world > 1:
        # config 3 over the whole job: a duplicate's original lives on ANY rank (7 of 8 on another one at world 8), so the
        # kept count and the first-seen indices are only right if the exchange resolves first-seen ACROSS ranks
        def fill(seed, first_base, n_bases):
            buf = torch.empty(n_bases + 64, dtype=torch.uint8, device=dev)
            ctx.synth_fill_device(seed, first_base, n_bases, buf)
            return buf
        W.plant_job_duplicates(fill, d_bytes, N, L, dev, rank, world)
    elif args.workload == "uniq":
        W.plant_duplicates(d_bytes, N, L, dev, 43 + rank, 44 + rank)          # config 3
    torch.cuda.synchronize()

    # one lane per stream: its own ctx (lists, table), torch stream and output buffers; the input is shared
    S = max(1, args.streams)
    if use_dist and args.workload == "uniq":
        S = 1                       # the exchange's collectives stay on one stream, in one order on every rank
    lanes = []
    for k in range(S):
        c = ctx if k == 0 else circkit_amd.Context(local_rank)
        st = stream if k == 0 else torch.cuda.Stream(device=dev)
        c.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            lane = {"ctx": c, "stream": st, "out": torch.empty(total + 64, dtype=torch.uint8, device=dev),
                    "hash": torch.empty(N, dtype=torch.int64, device=dev) if args.workload == "uniq" or args.with_hash else None, "fs": None, "keep": None}
            lane["table"] = U.DeviceTable(c)
        lanes.append(lane)
    d_out = lanes[0]["out"]
    state = lanes[0]
    counter = [0]

    def step():
        ln = lanes[counter[0] % S]
        counter[0] += 1
        with torch.cuda.stream(ln["stream"]):
            if args.workload == "uniq":
                # what `circkit uniq --canonicalize` computes per batch: canonical bytes + XXH3, then the first-seen
                # resolution -- the ctx table on one GPU, the hash-range exchange over RCCL on several
                ln["ctx"].canonicalize_batch_device(d_bytes, d_off, N, out_bytes=None if args.hash_only else ln["out"], out_xxh3=ln["hash"])
                ln["fs"], ln["keep"] = U.first_seen(ln["table"], ln["hash"], base_index=rank * N, exchange=args.exchange,
                                                    force_exchange=force_dist)
            elif args.with_hash:
                ln["ctx"].canonicalize_batch_device(d_bytes, d_off, N, out_bytes=None if args.hash_only else ln["out"], out_xxh3=ln["hash"])
                ln["fs"], ln["keep"] = U.first_seen(ln["table"], ln["hash"], base_index=rank * N, exchange=args.exchange, force_exchange=force_dist)
            else:
                ln["ctx"].canonicalize_batch_device(d_bytes, d_off, N, out_bytes=ln["out"])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in lanes]
    import gc
    gc.collect()
    gc.disable()            # (a collection inside the 40-100 ms of the timed region once cost a run 19 ms of wall time)
    t0 = time.perf_counter()
    ev0.record(stream)
    for ln in lanes[1:]:
        ln["stream"].wait_event(ev0)                    # every lane starts behind the start event
    for _ in range(args.steps):
        step()
    for ln, e in zip(lanes, ev1):
        e.record(ln["stream"])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    kernel_ms = max(ev0.elapsed_time(e) for e in ev1) / args.steps      # HIP events on the launch streams
    for ln in lanes:
        unprocessed = ln["ctx"].batch_status()
        if unprocessed:
            raise SystemExit("%d records were not processed" % unprocessed)
    unique_global = None
    if args.workload == "uniq":
        for ln in lanes:
            ln["table"].check()
        u = state["keep"].sum().to(torch.int64)
        if use_dist:
            dist.all_reduce(u)
        unique_global = int(u.item())
        if L >= 64 and unique_global != world * (N // 2):            # 1 kb random records collide with negligible probability
            raise SystemExit("uniq kept %d records, expected %d" % (unique_global, world * (N // 2)))
        if world > 1:
            # every record of every shard against the job-wide expectation (first-seen = smallest GLOBAL index with the same
            # base record, from the planting decisions every rank can recompute), then an oracle slice per rank that ties the
            # planting decisions to the bytes.  A rank that fails makes every rank exit non-zero.
            wrong, cross, distinct, keys = W.job_check(state["fs"], state["keep"], N, L, world, rank, dev)
            slice_err = job_oracle_slice(np, torch, args, N, L, rank, d_bytes, state, keys) if not args.no_cpu else None
            bad = torch.tensor([wrong + (1 if slice_err else 0), cross], dtype=torch.int64, device=dev)
            dist.all_reduce(bad)
            uniq_job_check = {"records_checked_per_rank": N, "first_seen_mismatches_job": int(bad[0]), "records_owned_by_another_rank_job": int(bad[1]),
                              "oracle_slice": "first %d records of every rank: canonical bytes + XXH3 vs the oracle, and each record's canonical form == "
                                              "the oracle's canonical form of the base record its key names (regenerated on the host)" % min(N, 20_000)
                                              if not args.no_cpu else None}
            if slice_err:
                print("rank %d: %s" % (rank, slice_err), file=sys.stderr)
            if int(bad[0]):
                raise SystemExit("uniq: %d first-seen / keep mismatches against the job-wide expectation (rank %d: %d)" % (int(bad[0]), rank, wrong))
            if distinct != world * (N // 2) or int(bad[1]) == 0:
                raise SystemExit("uniq workload is not the cross-rank one it should be (distinct %d, cross-rank owners %d)" % (distinct, int(bad[1])))
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        seq_per_s = world * N * args.steps / dt
        # read L + write L per record + one u64 offset; uniq: + the u64 hash per record.  The table's scattered
        # accesses (a 16-byte slot read + written by the insert, read by the lookup: 32 B per record) are NOT in SURVEY's
        # formula; the fraction with them is reported next to the judged one.
        algo_bytes = 2 * total + 8 * N
        table_bytes = 0
        if args.workload == "uniq" or args.with_hash:
            algo_bytes += 8 * N
            table_bytes = 32 * N
            if args.hash_only:
                algo_bytes = total + 16 * N           # "uniq hash-only mode (no canonical bytes written back): L + 16"
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        nvar = ", %g %% of the bases replaced by N" % (100 * args.n_frac) if args.n_frac > 0 else ""
        # the kernel chain the step really ran: which streaming build the batch's own mode picked, which table path
        two_row = 1008 < L <= 2032
        hashb = args.workload == "uniq"
        pair = hashb and not two_row and not nvar       # the builds with the fused XXH3: 8-wave workgroups; pure ACGT: two records per wave (canon_pair.h)
        stream_build = "canon_stream_kernel<StreamCfg<%d,2,%d,%d>,%s,false,false,%s>" % (
            8 if hashb and not two_row else 16, 2 if pair else 1,
            2 if two_row else 1, "true" if hashb else "false", "true" if nvar and L <= 1008 else "false")
        if args.workload == "canonicalize":
            metric = "canonicalize sequences/sec (%s x %d b synthetic FASTA payload)" % (
                "10M" if N == 10_000_000 else "100M over 8 GPUs" if config5 else "%d per GPU" % N, L)
            wl = "canonicalize, %d records x %d b per GPU%s, uniform ACGT seed 42%s, device-resident CSR (%s)" % (
                N, L, ", 100M records in total" if config5 else "", nvar,
                "BASELINE configs[4]" if config5 else "BASELINE configs[1]" if (N, L) == (10_000_000, 1000) and not nvar else "variant of BASELINE configs[1]")
            kernel = stream_build if not nvar else stream_build + " (+ canon_rescue_kernel<false,false,true> and canon_kernel<4> for its leftovers)"
            par = "records sharded over %d GPU(s) by contiguous index ranges, no data-path collective" % world
        elif args.workload == "uniq":
            metric = "uniq sequences/sec (%d x %d b per GPU, ~50%% rotational/strand duplicates)" % (N, L)
            wl = ("uniq --canonicalize, %d records x %d b per GPU, half are rotated / reverse-complemented copies, shuffled "
                  "(BASELINE configs[2])" % (N, L))
            if not use_dist:
                kernel = stream_build + " + uniq_resolve_insert_kernel + uniq_resolve_lookup_kernel"
            elif args.exchange == "partition":
                kernel = stream_build + " + uniq_partition_kernel + all_to_all x3 + uniq_insert_rows_kernel + uniq_lookup_rows_kernel + uniq_gather_kernel"
            else:
                kernel = stream_build + " + all_gather + uniq_insert_kernel x%d + uniq_lookup_kernel" % world
            par = ("records sharded over %d GPU(s); first-seen resolved by one hash-range all-to-all over RCCL (exchange=%s)"
                   % (world, args.exchange)) if use_dist else "1 GPU: the ctx hash table, no collective"
            if args.hash_only:
                metric = metric.replace("uniq sequences/sec", "uniq (hash-only) sequences/sec")
                wl = wl.replace("uniq --canonicalize,", "uniq without --canonicalize (no canonical bytes written: XXH3 of the canonical form + first-seen),")
                kernel += " (+ xxh3_kernel on (rotation, strand) views for records whose hash is not fused)"
        else:
            metric = "canonicalize sequences/sec (%d records, 200b-%dkb log-uniform lengths)" % (N, args.max_len // 1000)
            wl = "canonicalize, %d records, lengths log-uniform on [200, %d], %d bases per GPU%s (%s)" % (
                N, args.max_len, total, nvar, "BASELINE configs[3]" if (N, args.max_len) == (1_000_000, 20000) and not nvar else "variant of BASELINE configs[3]")
            build = {(False, False): "canon_mixed_kernel", (True, False): "canon_mixed_n_kernel", (False, True): "canon_mixed_h_kernel",
                     (True, True): "canon_mixed_nh_kernel"}[(bool(nvar), args.with_hash)]
            kernel = build + (" + uniq_resolve_insert_kernel + uniq_resolve_lookup_kernel" if args.with_hash else "") + " (whole step; per-kernel split in profiles/)"
            if args.with_hash:
                metric = "uniq --canonicalize sequences/sec (%d records, 200b-%dkb log-uniform lengths, all distinct)" % (N, args.max_len // 1000)
                wl = wl.replace("canonicalize,", "canonicalize + XXH3 + first-seen,")
                if args.hash_only:
                    metric = metric.replace("uniq --canonicalize", "uniq (hash-only)")
                    wl = wl.replace("canonicalize + XXH3 + first-seen,", "XXH3 of the canonical form + first-seen, no canonical bytes written,")
            par = "records sharded over %d GPU(s) by contiguous index ranges, no data-path collective" % world
        traffic, traffic_source = pmc_traffic("%s %d x %d" % (args.workload, N, L) + (" n%g" % args.n_frac if nvar else "") + (" hash" if args.with_hash else "") + (" hash-only" if args.hash_only else ""))
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": kernel, "kernel_ms": kernel_ms, "algorithmic_bytes": algo_bytes}
        if table_bytes:
            roofline["frac_incl_table_accesses"] = (algo_bytes + table_bytes) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
            roofline["table_access_bytes"] = table_bytes
        if not args.no_copy:
            roofline.update(copy_ceiling(torch, ctx, stream, d_bytes, d_out, total, achieved))
            # d_out holds the canonical records again (the checks below read it) -- the batch call alone: a whole step() would
            # enter the exchange's collectives on this rank only
            with torch.cuda.stream(lanes[0]["stream"]):
                lanes[0]["ctx"].canonicalize_batch_device(d_bytes, d_off, N, out_bytes=None if (args.hash_only and lanes[0]["hash"] is not None) else lanes[0]["out"],
                                                          out_xxh3=lanes[0]["hash"])
            torch.cuda.synchronize()
        result = {
            "metric": metric, "value": seq_per_s, "unit": "sequences/s",
            "gbases_per_s": world * total * args.steps / dt / 1e9,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl, "records_per_gpu": N, "records_total": world * N,
                       "record_len": L if args.workload != "mixed" else "200..%d (mean %d)" % (args.max_len, total // N),
                       "parallelism": par + ("; steps dealt round-robin to %d HIP streams (one ctx each)" % S if S > 1 else "")},
            "roofline": roofline,
            "unprocessed_records": "0 (circkit_ctx_batch_status after the last step of every lane; all steps run the same batch)",
        }
        if unique_global is not None:
            result["unique_records"] = unique_global
        if args.workload == "uniq" and world > 1:
            result["uniq_job_check"] = uniq_job_check
        args._state_hash = state["hash"]
        if world == 1 and not args.no_e2e:
            result["end_to_end"] = end_to_end(np, torch, circkit_amd, ctx, args, N, L, d_bytes, d_off, d_out)
        if world == 1 and not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(args, np, torch, N, L, total, d_bytes, d_off, d_out, state)
            if not result["cpu_baseline"]["gpu_output_matches"]:
                print(json.dumps(result))
                raise SystemExit("GPU output differs from the CPU oracle on the sample")
        if world == 1 and args.workload == "canonicalize" and not args.no_cli and not args.n_frac and S == 1 and L == 1000:
            try:
                result["cli"] = cli_block(np, torch, d_bytes, d_out, N, L)
            except Exception as e:      # noqa: BLE001  (an I/O problem of this side block must not cost the run its headline line)
                result["cli"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and args.workload == "canonicalize" and not args.no_others and not args.n_frac and S == 1:
            t_o = time.perf_counter()
            try:        # (a mismatch against the oracle raises SystemExit and ends the run non-zero; anything else is recorded)
                result["other_workloads"] = other_workloads(np, torch, circkit_amd, ctx, stream, dev, d_bytes, d_off, d_out, N, L, not args.no_cpu)
            except Exception as e:      # noqa: BLE001
                result["other_workloads"] = {"error": "%s: %s" % (type(e).__name__, e)}
            result["other_workloads"]["wall_s"] = time.perf_counter() - t_o
        print(json.dumps(result), flush=True)
    if use_dist:
        # every rank leaves together: rank 0's epilogue (the copy yardstick, the line) is still running when the others get here
        dist.barrier()
        dist.destroy_process_group()
    for ln in lanes:
        ln["ctx"].close()


def job_oracle_slice(np, torch, args, N, L, rank, d_bytes, state, keys):
    """Checker leg of a multi-rank `uniq` run (the oracle as the checker, on every rank, never timed): the first records of
    this rank's shard -- canonical bytes and XXH3 against the oracle, and the planting decisions against the bytes: the
    canonical form of record p must be the oracle's canonical form of the job's base record keys[p], regenerated on the
    host from the counter-based generator.  Returns an error text or None."""
    from oracle import oracle as O
    S = min(N, 20_000)
    h_off = np.arange(S + 1, dtype=np.uint64) * np.uint64(L)
    h_in = d_bytes[:S * L].cpu().numpy()
    exp, exp_h = O.canonicalize_batch(h_in, h_off, True, True, threads=min(8, len(os.sched_getaffinity(0))))
    if not args.hash_only and not np.array_equal(exp, state["out"][:S * L].cpu().numpy()):
        return "canonical bytes differ from the oracle on the slice"
    if not np.array_equal(exp_h.view(np.int64), state["hash"][:S].cpu().numpy()):
        return "XXH3 differs from the oracle on the slice"
    k = keys[:S].cpu().numpy()
    base = np.concatenate([O.synth_fill(42, int(g) * L, L) for g in k.tolist()])
    base_canon, _ = O.canonicalize_batch(base, h_off, True, False, threads=min(8, len(os.sched_getaffinity(0))))
    if not np.array_equal(base_canon, exp):
        return "a record's canonical form is not that of the base record its key names"
    return None


def cli_block(np, torch, d_bytes, d_out, N, L):
    """What a user of the drop-in runs: the `circkit` binary (C++ host: reader -> parser pool -> GPU through the C ABI -> emit ->
    writer) on a FASTA file of the headline batch's first records -- 5 GB in /dev/shm (1 GB when it has no room), headers `>r0000000` --, wall time of the
    whole process (HIP start-up and exit included) into /dev/null and into a /dev/shm file; the file is compared with the
    device path's canonical bytes (src/utils.rs:9-72, src/canonicalize.rs:31-44).  Never `value`."""
    import shutil
    exe = os.path.join(ROOT, "circkit_amd", "circkit")
    if not os.path.exists(exe) or not os.path.isdir("/dev/shm"):
        return {"error": "no circkit binary or no /dev/shm"}
    # 5M records = 5 GB when /dev/shm holds input + output + slack (the process's fixed costs -- loading the HIP libraries,
    # creating the context, exit: 0.5-0.6 s -- are more than half of a 1 GB run's wall time), else 1M
    free = shutil.disk_usage("/dev/shm").free
    R = min(N, 5_000_000) if free >= 4 * min(N, 5_000_000) * (L + 11) else min(N, 1_000_000)
    if free < 3 * R * (L + 11):
        return {"error": "not enough room in /dev/shm"}
    tag = "circkit_bench_%d" % os.getpid()
    src, dst = "/dev/shm/%s_in.fasta" % tag, "/dev/shm/%s_out.fasta" % tag
    BLK = 500_000                                                   # records per host block (0.5 GB at a time)

    def fasta(rows, first):
        m = rows.shape[0]
        out = np.empty((m, L + 11), dtype=np.uint8)
        out[:, 0] = ord(">"); out[:, 1] = ord("r"); out[:, 9] = 10; out[:, L + 10] = 10
        idx = np.arange(first, first + m)
        for d in range(7):
            out[:, 8 - d] = 48 + (idx // 10 ** d) % 10
        out[:, 10:L + 10] = rows
        return out
    try:
        with open(src, "wb") as f:
            for b in range(0, R, BLK):
                m = min(BLK, R - b)
                fasta(d_bytes[b * L:(b + m) * L].cpu().numpy().reshape(m, L), b).tofile(f)
        res = {"input": "%d records x %d b, %d bytes of FASTA in /dev/shm (the headline batch's first records)" % (R, L, R * (L + 11)),
               "threads": "default (the CPUs this process may use)"}
        env = dict(os.environ, CIRCKIT_CLI_TIMING="1")
        for sink, target, cmd in (("dev_null", "/dev/null", "canonicalize"), ("tmpfs_file", dst, "canonicalize"), ("uniq_dev_null", "/dev/null", "uniq")):
            t0 = time.perf_counter()
            r = subprocess.run([exe, cmd, src, "-o", target], capture_output=True, env=env)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": "circkit %s failed: %s" % (cmd, r.stderr.decode(errors="replace")[-300:])}
            res[sink] = {"wall_s": dt, "records_per_s": R / dt, "gbytes_per_s": R * (L + 11) / dt / 1e9}
            # the binary's own clock (CIRCKIT_CLI_TIMING): main() up to the last byte written, the HIP / context start-up inside it
            # (runs next to reading and parsing) and the pipeline behind it
            import re
            m_ = re.search(r"main\(\) to here ([0-9.]+) s;\s+HIP / ctx start-up ([0-9.]+) s, pipeline ([0-9.]+) s", r.stderr.decode(errors="replace"))
            if m_:
                res[sink].update({"main_s": float(m_.group(1)), "hip_startup_s": float(m_.group(2)), "pipeline_s": float(m_.group(3)),
                                  "pipeline_records_per_s": R / float(m_.group(3))})
        ok = os.path.getsize(dst) == R * (L + 11)
        with open(dst, "rb") as f:
            for b in range(0, R, BLK):
                if not ok:
                    break
                m = min(BLK, R - b)
                want = fasta(d_out[b * L:(b + m) * L].cpu().numpy().reshape(m, L), b)
                got = np.frombuffer(f.read(m * (L + 11)), dtype=np.uint8)
                ok = got.size == want.size and bool(np.array_equal(got, want.reshape(-1)))
        res["output_matches_device_path"] = bool(ok)
        res["note"] = ("wall time of the whole process, one run per sink: loading the HIP libraries, context start-up (0.1-0.3 s, next to reading "
                       "and parsing) and exit are inside -- 0.3-0.45 s of the wall lie outside main()'s pipeline whatever the input size")
        return res
    finally:
        for f in (src, dst):
            try:
                os.unlink(f)
            except OSError:
                pass


def other_workloads(np, torch, circkit_amd, ctx, stream, dev, d_bytes, d_off, d_out, N, L, check, steps=5):
    """The other BASELINE configs one GPU holds, on the driver's box: `uniq` (configs[2]) with and without canonical bytes,
    the mixed-length batch (configs[3]) plain, with 1 % N and with the XXH3 -- warm-up + `steps` timed steps each (HIP events on
    the launch stream), AFTER everything of the headline.  Per workload: ms per step, algorithmic bytes (2L + 8 per
    record, + 8 with the hash, L + 16 hash-only), fraction of the 8 TB/s peak, and the oracle (the checker) on a slice."""
    from circkit_amd import uniq as U
    from circkit_amd import workloads as W
    if check:
        from oracle import oracle as O
    table = U.DeviceTable(ctx)
    threads = min(16, len(os.sched_getaffinity(0)))
    res = {}

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(steps):
            fn()
        e1.record(stream)
        e1.synchronize()
        left = ctx.batch_status()
        if left:
            raise SystemExit("other_workloads: %d records were not processed" % left)
        return e0.elapsed_time(e1) / steps

    def entry(name, wl, ms, n, algo, ok, **more):
        res[name] = dict({"workload": wl, "ms_per_step": ms, "sequences_per_s": n / (ms * 1e-3), "algorithmic_bytes": algo,
                          "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "steps": steps, "oracle_slice_matches": ok}, **more)
        if ok is False:
            print(json.dumps({"other_workloads": res}))
            raise SystemExit("other_workloads: %s differs from the CPU oracle on the slice" % name)

    # ---- configs[2]: the headline's batch with half of it turned into rotated / reverse-complemented copies, shuffled
    total = N * L
    W.plant_duplicates(d_bytes, N, L, dev, 43, 44)
    d_hash = torch.empty(N, dtype=torch.int64, device=dev)
    st = {}

    def uniq_step(out):
        ctx.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=out, out_xxh3=d_hash)
        st["fs"], st["keep"] = U.first_seen(table, d_hash, base_index=0)
    S = min(N, 20_000)
    if check:
        h_off = np.arange(S + 1, dtype=np.uint64) * np.uint64(L)
        exp, exp_h = O.canonicalize_batch(d_bytes[:S * L].cpu().numpy(), h_off, True, True, threads=threads)
        exp_fs = O.uniq_first_seen(exp_h).astype(np.int64)
    for name, out in (("uniq", d_out), ("uniq_hash_only", None)):
        if out is None:
            d_hash.zero_()
        ms = timed(lambda: uniq_step(out))
        table.check()
        kept = int(st["keep"].sum())
        ok = None
        if check:
            ok = bool(np.array_equal(d_hash[:S].cpu().numpy(), exp_h.view(np.int64))) and \
                bool(np.array_equal(st["fs"][:S].cpu().numpy().astype(np.int64), exp_fs)) and \
                (out is None or bool(np.array_equal(out[:S * L].cpu().numpy(), exp))) and (L < 64 or kept == N - N // 2)
        entry(name, "uniq %s, %d x %d b, ~50 %% rotational/strand duplicates (BASELINE configs[2])%s" % (
            "--canonicalize" if out is not None else "without --canonicalize: no canonical bytes written", N, L, "" if (N, L) == (10_000_000, 1000) else " at this run's size"),
            ms, N, (2 * total + 16 * N) if out is not None else (total + 16 * N), ok, unique_records=kept,
            check="first %d records: XXH3 + first-seen%s vs the oracle; kept == N/2" % (S, " + canonical bytes" if out is not None else ""))
    del d_hash
    # ---- configs[3]: 1M records, lengths log-uniform on [200, 20000]; then with 1 % N; the plain one also with the XXH3.
    # A context of its own, as a job on such a batch would have: the first one's hash table has grown to the 10M-key batch
    # (grow-only) and a 1M-key batch would pay for clearing and probing all of it.
    ctx_uniq, ctx = ctx, circkit_amd.Context(dev.index if dev.index is not None else 0)
    ctx.set_stream(stream.cuda_stream)
    table = U.DeviceTable(ctx)
    M = 1_000_000
    offs = W.log_uniform_offsets(M, 45, 200, 20000)
    mtotal = int(offs[-1])
    m_off = offs.to(dev)
    m_in = d_bytes[:mtotal + 64] if d_bytes.numel() >= mtotal + 64 else torch.empty(mtotal + 64, dtype=torch.uint8, device=dev)
    m_out = d_out[:mtotal + 64] if d_out.numel() >= mtotal + 64 else torch.empty(mtotal + 64, dtype=torch.uint8, device=dev)
    ctx.synth_fill_device(45, 0, mtotal, m_in)
    m_hash = torch.empty(M, dtype=torch.int64, device=dev)
    SM = 4000
    mh_off = offs[:SM + 1].numpy().astype(np.uint64)
    mnb = int(mh_off[-1])

    def mixed_check(with_hash):
        if not check:
            return None
        e, eh = O.canonicalize_batch(m_in[:mnb].cpu().numpy(), mh_off, True, with_hash, threads=threads)
        ok = bool(np.array_equal(m_out[:mnb].cpu().numpy(), e))
        return ok and (not with_hash or bool(np.array_equal(m_hash[:SM].cpu().numpy(), eh.view(np.int64))))
    wl = "%d records, lengths log-uniform on [200, 20000], %d bases (BASELINE configs[3])" % (M, mtotal)
    ms = timed(lambda: ctx.canonicalize_batch_device(m_in, m_off, M, out_bytes=m_out))
    entry("mixed", "canonicalize, " + wl, ms, M, 2 * mtotal + 8 * M, mixed_check(False), check="first %d records: canonical bytes vs the oracle" % SM)

    def mixed_hash_step():
        ctx.canonicalize_batch_device(m_in, m_off, M, out_bytes=m_out, out_xxh3=m_hash)
        st["fs"], st["keep"] = U.first_seen(table, m_hash, base_index=0)
    ms = timed(mixed_hash_step)
    table.check()
    entry("mixed_with_hash", "uniq --canonicalize (canonicalize + XXH3 + first-seen), " + wl, ms, M, 2 * mtotal + 16 * M, mixed_check(True),
          unique_records=int(st["keep"].sum()), check="first %d records: canonical bytes + XXH3 vs the oracle" % SM)
    W.sprinkle_n(m_in, mtotal, 0.01, 46, dev)
    ms = timed(lambda: ctx.canonicalize_batch_device(m_in, m_off, M, out_bytes=m_out))
    entry("mixed_n1pct", "canonicalize, " + wl + ", 1 % of the bases replaced by N", ms, M, 2 * mtotal + 8 * M, mixed_check(False),
          check="first %d records: canonical bytes vs the oracle" % SM)
    torch.cuda.synchronize()
    ctx.close()
    ctx = ctx_uniq
    # ---- variants of the headline's shape (not BASELINE configs; round 4 found their slow corners): records of 300 b (the batch's
    # bytes cut anew -- the bytes-only build for batches of short records), then 1 % N in the 1 kb records: canonicalize and uniq
    del m_in, m_out, m_hash
    ctx.synth_fill_device(42, 0, total, d_bytes)
    table = U.DeviceTable(ctx)
    L3 = 300
    N3 = total // L3
    d_off3 = torch.arange(N3 + 1, dtype=torch.int64, device=dev) * L3
    ms = timed(lambda: ctx.canonicalize_batch_device(d_bytes, d_off3, N3, out_bytes=d_out))
    ok = None
    if check:
        S3 = 20_000
        e3, _ = O.canonicalize_batch(d_bytes[:S3 * L3].cpu().numpy(), np.arange(S3 + 1, dtype=np.uint64) * np.uint64(L3), True, False, threads=threads)
        ok = bool(np.array_equal(d_out[:S3 * L3].cpu().numpy(), e3))
    entry("canonicalize_300b", "canonicalize, %d x %d b (the headline batch's bytes cut into 300 b records)" % (N3, L3), ms, N3, 2 * N3 * L3 + 8 * N3, ok,
          check="first 20000 records: canonical bytes vs the oracle")
    del d_off3
    W.sprinkle_n(d_bytes, total, 0.01, 46, dev)
    if check:
        exp, exp_h = O.canonicalize_batch(d_bytes[:S * L].cpu().numpy(), h_off, True, True, threads=threads)
    ms = timed(lambda: ctx.canonicalize_batch_device(d_bytes, d_off, N, out_bytes=d_out))
    entry("canonicalize_n1pct", "canonicalize, %d x %d b, 1 %% of the bases replaced by N" % (N, L), ms, N, 2 * total + 8 * N,
          None if not check else bool(np.array_equal(d_out[:S * L].cpu().numpy(), exp)), check="first %d records: canonical bytes vs the oracle" % S)
    d_hash = torch.empty(N, dtype=torch.int64, device=dev)
    ms = timed(lambda: uniq_step(d_out))
    table.check()
    entry("uniq_n1pct", "uniq --canonicalize, %d x %d b, 1 %% of the bases replaced by N (no planted duplicates)" % (N, L), ms, N, 2 * total + 16 * N,
          None if not check else bool(np.array_equal(d_out[:S * L].cpu().numpy(), exp)) and bool(np.array_equal(d_hash[:S].cpu().numpy(), exp_h.view(np.int64))),
          unique_records=int(st["keep"].sum()), check="first %d records: canonical bytes + XXH3 vs the oracle" % S)
    del d_hash
    res["note"] = ("same process and box as the headline, after its timed region and checks; one HIP stream; each entry = 2 warm-up + %d timed "
                   "steps of the whole kernel chain of that workload (HIP events on the launch stream); frac = algorithmic_bytes / time / 8 TB/s" % steps)
    return res


def copy_ceiling(torch, ctx, stream, d_src, d_dst, total, achieved):
    """What THIS box copies, in this process, right behind the timed steps: the batch's payload (`total` bytes in, `total`
    out) through the library's plain copy kernels (circkit_bench_copy_device: four shapes + hipMemcpyAsync), best of them,
    HIP events on the launch stream.  Boxes of the pool differ by several percent; `frac_of_copy` is the figure that does not."""
    nb = total & ~15
    best, best_v, per_variant = None, None, {}
    # a plain copy's rate moves 3-5 % with how source and destination lie relative to each other (tools/probe_copy_skew.py): the
    # batch's own output buffer, and a scratch destination 2 MB + 8 KB out of step with the source; the best counts
    scratch = None
    try:
        scratch = torch.empty(nb + (8 << 20), dtype=torch.uint8, device=d_src.device)
        a0 = (-(scratch.data_ptr() - d_src.data_ptr())) % (2 << 20)           # scratch[a0:] is congruent to the source modulo 2 MB
        skewed = scratch[a0 + (2 << 20) + 8192:]
    except RuntimeError:
        skewed = None
    for v in range(5):
        for tag, dst in (("", d_dst), ("s", skewed)):
            if dst is None:
                continue
            ctx.bench_copy_device(d_src, dst, nb, v)                     # warm
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(3):
                ctx.bench_copy_device(d_src, dst, nb, v)
            e1.record(stream)
            e1.synchronize()
            ms = e0.elapsed_time(e1) / 3
            per_variant[str(v) + tag] = 2 * nb / (ms * 1e-3) / 1e9
            if best is None or ms < best:
                best, best_v = ms, str(v) + tag
    del scratch, skewed
    gbps = 2 * nb / (best * 1e-3) / 1e9
    return {"copy_ceiling_gbps": gbps, "frac_of_copy": achieved / gbps, "copy_ms": best,
            "copy_note": "same process, same box: %d B read + %d B written by a plain 16 B/lane copy kernel (best of 5 shapes x 2 placements of the destination -- 's' = 2 MB + 8 KB out of step with the source: variant %s); "
                         "GB/s per variant: %s" % (nb, nb, best_v, json.dumps({k: round(x) for k, x in per_variant.items()}))}


def want_out_rate_check(nb, dt, hash_only):
    """True when a host-buffer call moved its bytes at a rate that says the copy engines were shared with something else:
    below 65 GB/s for both directions together (quiet link: 80-85), or below 40 GB/s when only the input crosses (quiet: 50)."""
    rate = (1 if hash_only else 2) * nb / dt / 1e9
    return rate < (40.0 if hash_only else 65.0)


def end_to_end(np, torch, circkit_amd, ctx, args, N, L, d_bytes, d_off, d_out):
    """PCIe-inclusive rate (never `value`): the HOST-buffer entry point circkit_canonicalize_batch on page-locked
    buffers -- H2D of payload + offsets, the same kernels, D2H of the canonical bytes, one synchronisation -- on a sample
    of the batch (<= 1M records, <= 1 GB).  One call after the other: what one thread of a host program sees per call.
    Inside a call the library moves the batch in parts, so that the two directions of the link overlap (round 3)."""
    import ctypes
    lib = circkit_amd.load_library()
    S = min(N, 1_000_000)
    if args.workload == "mixed":
        S = min(S, 250_000)
    h_off_t = d_off[:S + 1].cpu()
    h_off = (h_off_t - h_off_t[0]).numpy().astype(np.uint64)
    nb = int(h_off[-1])
    first = int(h_off_t[0])
    pin_in, pin_out = lib.circkit_host_alloc(nb + 64), lib.circkit_host_alloc(nb + 64)
    if not pin_in or not pin_out:
        return {"error": "circkit_host_alloc failed"}
    h_in = torch.frombuffer((ctypes.c_uint8 * nb).from_address(pin_in), dtype=torch.uint8)
    h_in.copy_(d_bytes[first:first + nb])
    torch.cuda.synchronize()
    want_hash = args.workload == "uniq" or args.with_hash
    hash_only = want_hash and args.hash_only
    h_hash = np.empty(S, dtype=np.uint64) if want_hash else None
    state_hash = getattr(args, "_state_hash", None)

    def call():
        rc = lib.circkit_canonicalize_batch(ctx._h, pin_in, h_off.ctypes.data, S, None if hash_only else pin_out, None, None,
                                            h_hash.ctypes.data if want_hash else None)
        if rc:
            raise SystemExit("circkit_canonicalize_batch failed: %d" % rc)
    call()
    reps = 3
    # The copies are the runtime's DMA engines'.  For some seconds after a process that held tens of GB of device memory has
    # exited (a test run in front of this one, say) those engines are also busy with what the driver does to the memory it got
    # back, and the two directions of a call take turns instead of overlapping (tools/probe_after_big_process.sh: 40 ms per GB
    # instead of 24.8, for 3-10 s).  That is the box, not the path: measure again after a pause until the link is quiet (six
    # attempts at most), report every attempt.
    attempts = []
    for attempt in range(6):
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        dt = (time.perf_counter() - t0) / reps
        attempts.append(dt * 1e3)
        if not want_out_rate_check(nb, dt, hash_only):
            break
        time.sleep(3.0)
    dt = min(attempts) * 1e-3
    if hash_only:       # nothing came back but the hashes: they must be the device path's
        same = state_hash is not None and bool(np.array_equal(h_hash.view(np.int64), state_hash[:S].cpu().numpy()))
    else:
        got = torch.frombuffer((ctypes.c_uint8 * nb).from_address(pin_out), dtype=torch.uint8)
        same = bool(torch.equal(got, d_out[first:first + nb].cpu()))
    lib.circkit_host_free(pin_in)
    lib.circkit_host_free(pin_out)
    return {"value": S / dt, "unit": "sequences/s", "kind": "PCIe-inclusive, host-buffer API (circkit_canonicalize_batch), pinned buffers, one call after the other (inside a call: up to 16 parts, "
                                           "copy-in / kernels / copy-out of neighbouring parts overlap)",
            "sample": "first %d records (%d bases) of the batch" % (S, nb), "ms_per_call": dt * 1e3,
            "attempts_ms": [round(x, 3) for x in attempts], "first_attempt_value": S / (attempts[0] * 1e-3),
            "attempts_note": "one attempt = 3 calls; a slow attempt (copy engines shared with the driver's handling of device memory a "
                             "process before this one released) is repeated after 3 s; `value` is the best attempt, `first_attempt_value` the first",
            "h2d_plus_d2h_gbps": (1 if hash_only else 2) * nb / dt / 1e9, "pcie_ceiling_note": "PCIe Gen5 x16 ~63 GB/s per direction: <= 6.3e7 sequences/s at 1 kb with both directions fully overlapped",
            "matches_device_path": same}


def cpu_baseline(args, np, torch, N, L, total, d_bytes, d_off, d_out, state):
    """The oracle (C restatement of the reference path) on EVERY host core this process may use, on a bounded sample
    of the same device-generated batch, next to the same restatement on one core; the GPU's output on the sample is
    compared byte for byte (uniq: also the first-seen indices)."""
    from oracle import oracle as O
    visible = len(os.sched_getaffinity(0))
    quota = visible                 # the box's CPU share: a cgroup quota below the affinity mask makes extra threads fight for it
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, -(-int(q) // int(per)))
    except (OSError, ValueError):
        pass
    cores = min(visible, quota)
    S = min(N, args.cpu_sample if args.cpu_sample is not None else max(200_000, 100_000 * cores))
    if args.workload == "mixed":
        S = min(S, 100_000)                       # ~3.6 kb per record on average
        h_off = d_off[:S + 1].cpu().numpy().astype(np.uint64)
    else:
        h_off = np.arange(S + 1, dtype=np.uint64) * np.uint64(L)
    nb = int(h_off[-1])
    h_in = d_bytes[:nb].cpu().numpy()
    O.lib()
    want_hash = args.workload == "uniq" or args.with_hash
    c0 = time.perf_counter()
    h_out, h_hash = O.canonicalize_batch(h_in, h_off, True, want_hash, threads=cores)
    fs = O.uniq_first_seen(h_hash) if want_hash else None          # single thread, as the reference's main-thread closure
    cdt = time.perf_counter() - c0
    if want_hash and args.hash_only:       # no canonical bytes on the device in this mode: the hashes of the canonical forms instead
        same = bool(np.array_equal(h_hash.view(np.int64), state["hash"][:S].cpu().numpy()))
    else:
        same = bool(np.array_equal(h_out, d_out[:nb].cpu().numpy()))
    if want_hash:
        # first-seen over the whole batch restricted to the first S records = first-seen of the sample alone
        same = same and bool(np.array_equal(fs.astype(np.int64), state["fs"][:S].cpu().numpy().astype(np.int64)))
    S1 = min(S, 100_000 if args.workload != "mixed" else 20_000)    # the same restatement on ONE core, smaller sample
    c1 = time.perf_counter()
    _, hh = O.canonicalize_batch(h_in[:int(h_off[S1])], h_off[:S1 + 1], True, want_hash, threads=1)
    if want_hash:
        O.uniq_first_seen(hh)
    one_core = S1 / (time.perf_counter() - c1)
    # the reference AS WRITTEN indexes with chars().nth(): O(n^2) per record.  Cost model (oracle/circkit_oracle.c
    # ck_oracle_lmsr_index_nth) on a small subsample, all cores; same answers, checked
    SQ = min(S, max(64, int(10_000 * (1000.0 / L) ** 2)) if args.workload != "mixed" else 400)
    c2 = time.perf_counter()
    q_out = O.canonicalize_batch_nth(h_in[:int(h_off[SQ])], h_off[:SQ + 1], threads=cores)
    qdt = time.perf_counter() - c2
    q_same = bool(np.array_equal(q_out, h_out[:int(h_off[SQ])]))
    what = {"canonicalize": "normalize-free canonicalize", "uniq": "canonicalize + XXH3-64 on all cores, then the first-seen map on one "
            "thread (the reference's main-thread closure)", "mixed": "canonicalize"}[args.workload]
    return {
        "value": S / cdt, "unit": "sequences/s", "cores": cores, "host_cores_visible": visible, "cgroup_cpu_quota": quota,
        "kind": "port", "one_core_value": one_core,
        "sample": "first %d records (%d bases) of the same device-generated batch; C restatement of the reference path: %s "
                  "(linear-time byte-indexed Duval variant, faster than the reference's O(n^2) chars().nth() loop), %d pthreads = every core "
                  "this process may use (%d visible, cgroup quota %d); one_core_value on the first %d records" % (S, nb, what, cores, visible, quota, S1),
        "gpu_output_matches": same,
        "reference_faithful_quadratic": {
            "value": SQ / qdt, "unit": "sequences/s", "cores": cores, "sample": "first %d records" % SQ, "same_answers": q_same,
            "note": "COST MODEL, not the reference binary: the restatement with the reference's own access pattern -- "
                    "s.chars().nth(i) walks the text from its start at every access (lib/src/canonicalize.rs:17-27), O(n^2) per "
                    "record; how fast the real walk is depends on the rustc that built the binary"},
    }


if __name__ == "__main__":
    main()
