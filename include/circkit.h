/*
 * circkit.h -- C ABI of the MI355X (gfx950) drop-in for circkit's `canonicalize` / `uniq` hot path.
 *
 * The reference (Benjamin-Lee/circkit, Rust) has no FFI: the seam this library sits behind is the
 * lib-crate API plus the two closures the CLI hands to seq_io::parallel_fasta.  Each entry point below
 * names the reference interface it replaces (paths relative to the reference checkout); INTEGRATION.md
 * shows the `extern "C"` block a maintainer would add on the Rust side.
 *
 * Conventions
 *  - plain pointers and sizes only; caller allocates and owns every buffer; the library keeps no
 *    pointer after a call returns (device entry points: after the stream work completes).
 *  - return value 0 = CIRCKIT_OK, negative = error; no exceptions or panics cross the ABI;
 *    circkit_last_error(ctx) gives the message of the last failing call on that ctx.
 *  - one ctx per GPU, used by one host thread at a time; ctxs are independent (multi-GPU = one ctx,
 *    one process or thread, per device).
 *  - there is NO CPU fallback: every compute entry point runs the HIP kernels and fails with
 *    CIRCKIT_ERR_NO_DEVICE / CIRCKIT_ERR_HIP when it cannot.
 *  - CSR batches: bytes[offsets[i] .. offsets[i+1]) is record i; offsets has n_records + 1 entries,
 *    offsets[0] may be non-zero; records are what the reference's worker closure hands to
 *    circkit::canonicalize, i.e. already normalized (src/canonicalize.rs:24-29).  A record may be any
 *    byte string; records of pure ACGT take the 2-bit path, {-,A,C,G,N,T} the 4-bit path, everything
 *    else the byte-wide path (unsigned byte order, as the reference's slice comparison).
 */
#ifndef CIRCKIT_H
#define CIRCKIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CIRCKIT_OK 0
#define CIRCKIT_ERR_INVALID_ARG (-1)
#define CIRCKIT_ERR_NO_DEVICE (-2)   /* no HIP device / device index out of range */
#define CIRCKIT_ERR_HIP (-3)         /* a HIP runtime call or kernel failed */
#define CIRCKIT_ERR_TOO_LONG (-4)    /* a record of 2^31 symbols or more (cyclic positions are 32-bit) */
#define CIRCKIT_ERR_OOM (-5)
#define CIRCKIT_ERR_NOT_ASCII (-6)   /* single-record API only: mirrors the reference's from_utf8().unwrap() panic */

typedef struct circkit_ctx circkit_ctx;

/* ---- context ---------------------------------------------------------------------------------- */
/* Replaces: nothing in the reference (it is a single CPU process); one ctx ~ one worker pool
 * (`--threads`, src/commands.rs:120-123) bound to one GPU. */
int circkit_ctx_create(int device, circkit_ctx** out);
int circkit_ctx_destroy(circkit_ctx* ctx);
const char* circkit_last_error(const circkit_ctx* ctx);
/* Launch all subsequent work of this ctx on `hip_stream` (a hipStream_t; NULL = HIP's default stream).
 * A fresh ctx launches on a private non-blocking stream; circkit_ctx_use_own_stream returns to it.
 * Switching ORDERS the streams: the new stream waits (on the device, through an event -- the host does not block) for
 * everything the ctx has queued on the stream it leaves, so un-synchronised batches on either side of a switch cannot
 * overlap in the ctx's lists, counters and table, and a call after the switch may read what a call before it wrote.
 * The stream being left must still exist.  Binding the stream that is already bound is free. */
int circkit_ctx_set_stream(circkit_ctx* ctx, void* hip_stream);
int circkit_ctx_use_own_stream(circkit_ctx* ctx);
/* Block until all work queued by this ctx has finished. */
int circkit_ctx_synchronize(circkit_ctx* ctx);
/* Milliseconds the canonicalize kernels of the most recent *_batch_device call took on the GPU
 * (hipEvent pair recorded on the launch stream around the kernels; synchronizes on the stop event). */
int circkit_ctx_last_kernel_ms(circkit_ctx* ctx, float* ms);

/* ---- batch, device resident (the hot path) ---------------------------------------------------- */
/* Replaces, for a whole batch of records: the worker-closure body `circkit::canonicalize(&normalized)`
 * (src/canonicalize.rs:29, src/uniq.rs:40) = lib/src/canonicalize.rs:54-63 (lmsr x2 + revcomp + select),
 * and, when d_out_xxh3 is given, `xxh3_64(canonicalized)` (src/uniq.rs:45).
 * All pointers are DEVICE pointers; the call only enqueues work on the ctx stream.
 *   d_bytes      payload; must be readable for total_bytes (= offsets[n_records]) bytes
 *   d_offsets    uint64[n_records + 1]
 *   d_out_bytes  nullable; canonical sequences, same offsets as the input
 *   d_out_index  nullable; uint32[n_records]: lmsr_index(s) when the forward strand wins, else
 *                lmsr_index(revcomp(lmsr(s))) -- the two indices the reference computes (:43, :56)
 *   d_out_strand nullable; uint8[n_records]: 0 = lmsr(s) returned, 1 = lmsr(revcomp) returned (:58-62)
 *   d_out_xxh3   nullable; uint64[n_records]: XXH3-64 (seed 0) of the canonical sequence
 * n_records must be < 2^30 and every record shorter than 2^31 bytes.  No alignment or padding is required of
 * d_bytes / d_out_bytes, and offsets[0] need not be 0.
 * The call enqueues EVERYTHING the batch needs: once the stream has run past it the outputs are complete, whichever
 * way the caller synchronises, and the next batch may be enqueued straight behind it.  Records too long for the
 * on-chip tiers (pure ACGT beyond ~640 kb, other alphabets beyond ~70-320 kb) are taken by the batch's last two
 * kernels in a ctx-owned global-memory scratch (256 MiB unless circkit_ctx_set_long_record_scratch says otherwise:
 * pure ACGT up to ~1 Gb, arbitrary bytes up to ~126 MB); a record beyond that is left untouched and counted by
 * circkit_ctx_batch_status. */
int circkit_canonicalize_batch_device(circkit_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_offsets,
                                      uint64_t n_records, uint8_t* d_out_bytes, uint32_t* d_out_index,
                                      uint8_t* d_out_strand, uint64_t* d_out_xxh3);

/* lmsr() for a whole batch: forward strand only (lib/src/canonicalize.rs:41-47); d_out_index[i] =
 * lmsr_index(record i) (lib/src/canonicalize.rs:5). */
int circkit_lmsr_batch_device(circkit_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_offsets,
                              uint64_t n_records, uint8_t* d_out_bytes, uint32_t* d_out_index);
/* xxh3_64 (call site src/uniq.rs:45) of every record of a device-resident CSR batch. */
int circkit_xxh3_batch_device(circkit_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_offsets,
                              uint64_t n_records, uint64_t* d_out_xxh3);

/* Waits for the most recent batch and returns the number of its records that could not be processed (longer than
 * the long-record scratch allows, or 2^31 symbols or more; they were left untouched): non-zero makes the status
 * CIRCKIT_ERR_TOO_LONG. */
int circkit_ctx_batch_status(circkit_ctx* ctx, uint32_t* n_unprocessed);
/* Size in bytes of the global-memory scratch the device entry points give to records beyond the on-chip tiers (a
 * record of n symbols needs ~0.25 n bytes when pure ACGT -- 0.38 n if its minimal 16-mer repeats --, ~1.13 n for
 * {-,A,C,G,N,T}, ~2.13 n otherwise).  The host-buffer entry points see the lengths and grow the scratch by
 * themselves.  Synchronizes. */
int circkit_ctx_set_long_record_scratch(circkit_ctx* ctx, uint64_t bytes);
/* Which build of the streaming kernel the most recent device batch selected, from that batch's own lengths: 1 = one
 * packed word per lane (records up to 1008 b), 2 = two words (a quarter or more of the records in 1009..2032 b),
 * 3 = neither (an eighth or more of the records longer; the per-record passes take everything).  Diagnostic: every
 * mode computes the same results (and the library launches a batch's kernels for the mode the batches before it reported,
 * so a change of kind costs one or two slower batches, never a wrong answer).  Synchronizes. */
int circkit_ctx_last_batch_mode(circkit_ctx* ctx, uint32_t* mode);

/* ---- batch, host buffers ---------------------------------------------------------------------- */
/* Same contract with HOST pointers: copies the batch into ctx-owned device buffers (grow-only), runs
 * circkit_canonicalize_batch_device, copies the requested outputs back and returns when they are complete.
 * Batches of 32 MB and more go through the device in up to sixteen parts: part k + 1 is copied in (on the ctx stream, behind
 * part k's kernels) while part k - 1 is copied out (on a stream of the ctx's own), so with page-locked buffers both
 * directions of the link are busy at once (1 GB of 1 kb records: 23-24 ms per call instead of 38; 64 MB: 2.0 ms; 16 MB: 0.8).
 * The offsets and the per-record outputs cross through page-locked staging of the ctx, whatever memory the caller's are in.
 * Buffers from circkit_host_alloc (page-locked) are copied by DMA; pageable memory goes through the runtime's
 * staging.  Streaming hosts overlap this call with their own parsing / writing (see circkit_cli.cpp).
 * offsets[0] must be 0. */
int circkit_canonicalize_batch(circkit_ctx* ctx, const uint8_t* bytes, const uint64_t* offsets,
                               uint64_t n_records, uint8_t* out_bytes, uint32_t* out_index,
                               uint8_t* out_strand, uint64_t* out_xxh3);

/* Page-locked host memory for batch buffers: with bytes / out_bytes allocated here the host-buffer entry points
 * copy by plain DMA instead of through the runtime's pageable staging.  Needs a HIP device (NULL otherwise). */
void* circkit_host_alloc(size_t bytes);
void circkit_host_free(void* p);

/* ---- single record: 1:1 mirror of the lib crate (lib/src/lib.rs:1,3) ------------------------------ */
/* pub fn lmsr_index(x: &[u8]) -> usize            lib/src/canonicalize.rs:5  */
int circkit_lmsr_index(circkit_ctx* ctx, const uint8_t* s, size_t n, size_t* out_index);
/* pub fn lmsr(s: &[u8]) -> Vec<u8>                lib/src/canonicalize.rs:41  (out has n bytes) */
int circkit_lmsr(circkit_ctx* ctx, const uint8_t* s, size_t n, uint8_t* out);
/* pub fn canonicalize(s: &[u8]) -> Vec<u8>        lib/src/canonicalize.rs:54  (out has n bytes) */
int circkit_canonicalize(circkit_ctx* ctx, const uint8_t* s, size_t n, uint8_t* out);
/* xxhash_rust::xxh3::xxh3_64(bytes)               call site src/uniq.rs:45 */
int circkit_xxh3_64(circkit_ctx* ctx, const uint8_t* s, size_t n, uint64_t* out_hash);

/* ---- uniq: first-seen resolution on the device ------------------------------------------------ */
/* Replaces the `seen: HashMap<u64, String, NoHash>` logic of src/uniq.rs:27,47-48,66: for every record
 * i, d_first_seen[i] = the smallest global index whose hash equals d_hash[i] (hash-only equality, as in
 * the reference); record i is kept iff d_first_seen[i] == base_index + i.
 * `base_index` is the global index of record 0 of this batch (multi-GPU sharding / streaming batches).
 * The table persists in the ctx across calls until circkit_uniq_reset, so batches (and hash sets
 * gathered from other GPUs) can be folded in one after another:
 *   circkit_uniq_insert_device   folds (hash, global index) pairs into the table
 *   circkit_uniq_lookup_device   reads the winner for each hash */
int circkit_uniq_reset(circkit_ctx* ctx, uint64_t expected_keys);
int circkit_uniq_insert_device(circkit_ctx* ctx, const uint64_t* d_hash, uint64_t n, uint64_t base_index);
/* the same with an explicit global index per key: the keys a rank receives in the multi-GPU exchange (hash-range
 * partition, circkit_amd/uniq.py) are a subset of every other rank's shard, not a contiguous range */
int circkit_uniq_insert_pairs_device(circkit_ctx* ctx, const uint64_t* d_hash, const uint64_t* d_index, uint64_t n);
int circkit_uniq_lookup_device(circkit_ctx* ctx, const uint64_t* d_hash, uint64_t n, uint64_t* d_first_seen);
/* The device steps of the multi-GPU exchange (circkit_amd/uniq.py, exchange = "partition"; the reference is one process:
 * src/uniq.rs:27 has no counterpart).  The key space is cut into `world` (<= 64) ranges, a key belongs to rank
 * ((hash >> 20) & 0x7FFFFFFF) % world.
 *   partition    d_rows[n][2] = {hash, base_index + i} with the rows of one owner together, owners in rank order (what
 *                an all-to-all sends; any order inside an owner's group), d_counts[world] = rows per owner, d_slot[i] =
 *                row of record i.  n < 2^32 - 1.  d_rows -- here and in insert_rows / lookup_rows -- must be 16-byte
 *                aligned (a row is one 16-byte access).
 *   insert_rows  folds received rows into the table;  lookup_rows  d_answers[k] = smallest index seen for row k's hash
 *   gather       d_first_seen[i] = d_answers[d_slot[i]] (the answers come back in row order), d_keep[i] (nullable) =
 *                1 iff that is base_index + i */
int circkit_uniq_partition_device(circkit_ctx* ctx, const uint64_t* d_hash, uint64_t n, uint64_t base_index, uint32_t world,
                                  uint64_t* d_rows, uint64_t* d_counts, uint32_t* d_slot);
int circkit_uniq_insert_rows_device(circkit_ctx* ctx, const uint64_t* d_rows, uint64_t n);
int circkit_uniq_lookup_rows_device(circkit_ctx* ctx, const uint64_t* d_rows, uint64_t n, uint64_t* d_answers);
int circkit_uniq_gather_device(circkit_ctx* ctx, const uint64_t* d_answers, const uint32_t* d_slot, uint64_t n, uint64_t base_index,
                               uint64_t* d_first_seen, uint8_t* d_keep);
/* One shard in one call: reset (sized for n keys), insert with indices base_index .. base_index + n - 1, lookup, and
 * d_keep[i] (nullable, uint8) = 1 iff d_first_seen[i] == base_index + i -- the reference's per-record decision "emit,
 * or write a table row" (src/uniq.rs:47-62).  n < 2^32 - 1.  The table's contents are private to the call (it keeps
 * shard-local indices in a layout of its own): circkit_uniq_insert_* / _lookup_device refuse to touch it until the next
 * circkit_uniq_reset.  (Shards of 2^19 .. 21M keys are resolved in LDS-sized buckets with the table's memory as scratch --
 * same answers, no table contents at all afterwards; CIRCKIT_UNIQ_NO_BUCKETS=1 in the environment keeps the table path.) */
int circkit_uniq_resolve_device(circkit_ctx* ctx, const uint64_t* d_hash, uint64_t n, uint64_t base_index,
                                uint64_t* d_first_seen, uint8_t* d_keep);
/* reset / insert / lookup / resolve only enqueue work.  circkit_uniq_status waits for it and fails with CIRCKIT_ERR_OOM when
 * keys found no slot (more distinct keys than circkit_uniq_reset was told to expect); *n_overflowed (nullable) = how many. */
int circkit_uniq_status(circkit_ctx* ctx, uint32_t* n_overflowed);

/* Host-buffer form for streaming hosts (the CLI's batch loop): folds this batch's hashes (global indices
 * base_index .. base_index + n - 1) into the ctx table -- created and grown on demand, earlier batches kept --
 * and returns first_seen[i] for the batch.  Synchronizes.  If growing the table fails half-way (a failed rehash), the
 * earlier batches are lost: this call and every later one return CIRCKIT_ERR_HIP until circkit_uniq_reset. */
int circkit_uniq_first_seen(circkit_ctx* ctx, const uint64_t* hash, uint64_t n, uint64_t base_index,
                            uint64_t* first_seen);

/* ---- FASTA -> CSR packer (host logic, no GPU) --------------------------------------------------- */
/* Replaces seq_io 0.3.2's fasta::Reader record boundaries + the normalize step of the worker closure
 * (src/canonicalize.rs:14-27, src/uniq.rs:24-38).  Parses the complete records of text[0, n): header span,
 * raw sequence span (RefRecord::seq(): interior line breaks kept, final one dropped) and the normalized
 * bytes in CSR layout (padded with 64 zero bytes).  first_chunk: skip leading blank lines and require '>';
 * final_chunk = 0: the last record may be cut by the chunk end, parsing stops at its start and *consumed
 * says where to resume.  Returns CIRCKIT_ERR_INVALID_ARG on a format error (message: circkit_fasta_error). */
typedef struct circkit_fasta_batch circkit_fasta_batch;
int circkit_fasta_parse(const uint8_t* text, size_t n, int first_chunk, int final_chunk, circkit_fasta_batch** out,
                        size_t* consumed);
const char* circkit_fasta_error(const circkit_fasta_batch* b);
uint64_t circkit_fasta_n_records(const circkit_fasta_batch* b);
const uint8_t* circkit_fasta_bytes(const circkit_fasta_batch* b);
const uint64_t* circkit_fasta_offsets(const circkit_fasta_batch* b);
int circkit_fasta_record(const circkit_fasta_batch* b, uint64_t i, size_t* head_off, size_t* head_len, size_t* raw_off,
                         size_t* raw_len);
void circkit_fasta_free(circkit_fasta_batch* b);

/* ---- synthetic input on the device (bench / tests) ------------------------------ */
/* Fills d_bytes[0..n_bases) with uniform ACGT from the counter-based generator keyed by
 * (seed, first_base + i) -- the same bytes oracle/ck_oracle_synth_fill produces on the host. */
int circkit_synth_fill_device(circkit_ctx* ctx, uint64_t seed, uint64_t first_base, uint64_t n_bases,
                              uint8_t* d_bytes);
/* d_offsets[i] = base + i * record_len for i in 0..n_records (inclusive). */
int circkit_fixed_offsets_device(circkit_ctx* ctx, uint64_t base, uint64_t record_len, uint64_t n_records,
                                 uint64_t* d_offsets);

/* Measurement helper, no counterpart in the reference (the roofline is quoted against what this very box
 * copies): enqueues ONE plain device-to-device copy of `bytes` bytes (rounded down to a multiple of 16; both buffers
 * 16-byte aligned) on the ctx stream.  variant 0..3 = copy kernels of different shapes (16 B per lane; 4 / 8 / 4 / 2 loads
 * in flight per lane over 2048 / 2048 / 8192 / 65536 workgroups), 4 = hipMemcpyAsync.  bench.py times each and reports the
 * best as roofline.copy_ceiling_gbps. */
int circkit_bench_copy_device(circkit_ctx* ctx, const void* d_src, void* d_dst, uint64_t bytes, uint32_t variant);

/* ---- host-side normalisation used by the packer ------------------------------------------------ */
/* needletail::sequence::normalize(seq, false)      call sites src/canonicalize.rs:24, src/uniq.rs:35
 * Host logic of the CSR packer (strips line breaks while it computes offsets); returns the new length,
 * sets *changed to 0 when the reference would have returned None. */
size_t circkit_normalize(const uint8_t* s, size_t n, uint8_t* out, int* changed);

const char* circkit_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CIRCKIT_H */
