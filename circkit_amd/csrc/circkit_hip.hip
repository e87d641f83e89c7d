// circkit_hip.hip -- gfx950 kernels + the C ABI of include/circkit.h.
// Built by __graft_entry__.build():  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -> libcirckit_hip.so
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/circkit.h"
#include "canon_core.h"
#include "canon_fast.h"
#include "canon_stream.h"
#include "canon_mixed.h"
#include "fasta_host.h"
#ifndef CK_FAST_WPE
#define CK_FAST_WPE 8     // min waves per SIMD the streaming kernel is compiled for (two 16-wave workgroups per CU: <= 64 VGPRs)
#endif
#ifndef CK_STREAM_WPB
#define CK_STREAM_WPB 16     // waves per workgroup of the staged streaming kernel
#endif
#ifndef CK_STREAM_NBUF
#define CK_STREAM_NBUF 2     // LDS images per workgroup (NBUF-1 groups in flight); 2 x 32 KiB, two 16-wave workgroups per CU
#endif
#ifndef CK_FAST_BPC
#define CK_FAST_BPC 128   // workgroups launched per CU (4-6 resident; the rest queue: finer dynamic balance)
#endif
#include "xxh3_core.h"

namespace {

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// One wavefront per record, WPB wavefronts per workgroup, each with its own LDS slice; the last LDS dword is the
// workgroup's deferral counter.  Consumes the segmented list of the previous stage (or all records).
#ifndef CK_TIER_WPE
#define CK_TIER_WPE 7      // min waves per SIMD the LDS-tier kernels are compiled for: 72 VGPRs, no spills (measured: 1 -> 76 VGPRs / 6 waves +2.7 %, 8 -> spills +1.2 %)
#endif
// T4: with the 4-bit team in the team pass.  Inlined it costs this kernel five spilled vector registers -- on its own path, but
// a kernel with a scratch allocation is ~5 us slower to dispatch, twice in EVERY batch (stages A and C; measured on config 4
// + 1 % N: 2.068 -> 2.078 ms); as a __noinline__ call it takes the kernel from 72 to 87 registers.  So two builds: the one
// without it for batches whose predecessor left the tiers next to nothing (launch_canon's tiers_idle; their few long records
// with an N in the winning window go to wave 0 alone, as before round 4), the one with it for tiers that have real work.
// (With the list entries' second flag bit the allocator's outcome for the build with it is 72 registers without spills as well --
// an outcome five lines of unrelated code can change back: the two builds stay.)
template <int WPB, bool T4 = false>
__global__ __launch_bounds__(WPB * 64, CK_TIER_WPE) void canon_kernel(ck::CanonArgs a, uint32_t nvb, uint32_t* giants)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* blk_count = lds + WPB * a.slice_dw;
    uint32_t* lut = blk_count + 4;                   // 2-bit decode table shared by the workgroup's waves
    uint32_t* lutn = lut + ck::FAST_LUT_DW;          // ...and the 'G' -> 'N' patch table of the 2-bit-with-N-mask mode
    uint32_t live = ~0u;      // bit j: this workgroup's j-th virtual workgroup has input (those from the 32nd on are walked unseen)
    if (a.list && gridDim.x < nvb) {
        // walking grid (the previous batch left the tiers next to nothing): one parallel look at this workgroup's input
        // segments; if all are empty, zero the output counts and leave -- walking them one dependent load and two barriers
        // at a time made the five idle tiers ~80 us of every batch (headline 3.74 -> 3.70 ms, same box).  If a few are not
        // (the leftovers of a mixed-length batch with N: 244 records in 32768 segments), only those are walked: stage A of
        // that batch 77 -> 70 us (what remains is its longest record in the 4-bit mode, one wave)
        const uint32_t per = a.segs_per_block, mine = (nvb - blockIdx.x + gridDim.x - 1) / gridDim.x;
        if (threadIdx.x == 0) blk_count[1] = 0;
        __syncthreads();
        uint32_t any = 0;
        for (uint32_t t = threadIdx.x; t < mine * per; t += WPB * 64) {
            const uint32_t j = t / per, sgm = (blockIdx.x + j * gridDim.x) * per + t % per;
            if (sgm < a.in_nseg && a.list_count[sgm]) { any = 1; if (j < 32) atomicOr(blk_count + 1, 1u << j); }
        }
        if (!__syncthreads_or((int)any)) {
            if (a.defer_count)
                for (uint32_t t = threadIdx.x; t < mine; t += WPB * 64) a.defer_count[blockIdx.x + t * gridDim.x] = 0;
            return;
        }
        live = ck::uniform(blk_count[1]);            // (the team pass's use of this word is behind the loop's first barrier)
    }
    ck::fast_lut_init(lut, threadIdx.x, WPB * 64);
    ck::fast_lutn_init(lutn, threadIdx.x, WPB * 64);
    const uint32_t wib = ck::uniform(threadIdx.x >> 6);
    // `nvb` virtual workgroups (one output segment each) walked by a grid that is no bigger than what keeps the chip
    // busy: on the batches where these tiers have nothing to do (the headline) one workgroup per segment was 12-28 us of
    // pure dispatch per tier
    for (uint32_t vb = blockIdx.x, j = 0; vb < nvb; vb += gridDim.x, ++j) {
        if (j < 32 && !((live >> j) & 1u)) {
            if (threadIdx.x == 0 && a.defer_count) a.defer_count[vb] = 0;
            continue;
        }
        if (threadIdx.x == 0) *blk_count = 0;
        __syncthreads();
        ck::canon_wave_loop(a, lds + wib * a.slice_dw, lut, blk_count, vb, nvb, wib, WPB, lutn);
        __syncthreads();
        if (WPB > 1) ck::team_pass(a, lds, lut, lutn, blk_count, vb, wib, WPB, true, T4);      // records too long for one wave's slice: all waves together
        if (threadIdx.x == 0 && a.defer_count) a.defer_count[vb] = *blk_count;
        if (threadIdx.x == 0 && giants && *blk_count) atomicAdd(giants, *blk_count);     // last LDS tier: tell the global-scratch kernel there is work
    }
}

// The last LDS stage: one 16-wave workgroup with the CU's whole LDS.  A record the team can take -- its 2-bit strand (with a
// few N: + the bitmask) fits 157 KiB: pure ACGT up to ~640 kb -- is canonicalized by the sixteen waves together
// (canon_core.h team mode) where the one-wave tier this replaces ran ONE wave per CU; everything else (4-bit and byte
// modes, ties, reverse-complement palindromes) is wave 0's alone with the same LDS, as before, the others waiting at the
// barrier.  a.slice_dw = a sixteenth of the LDS dwords.
constexpr int TEAM_WAVES = 16;
// the input segments of virtual workgroup `block`, entry by entry, by all TEAM_WAVES waves of the workgroup; buf: total_dw
// dwords of LDS -- or of global scratch (canon_global_kernel): the waves of one workgroup share their CU's vector cache,
// the workgroup barriers between the team's steps order their accesses there as they do in LDS
__device__ __forceinline__ void team_stage_block(const ck::CanonArgs& a, uint32_t* buf, uint32_t total_dw, const uint32_t* lut, const uint32_t* lutn,
                                                 uint32_t* blk_count, uint32_t block, uint32_t wib)
{
    ck::CanonArgs solo = a;
    solo.slice_dw = total_dw;
    for (uint32_t sgm = block * a.segs_per_block; sgm < (block + 1) * a.segs_per_block && sgm < a.in_nseg; ++sgm) {
        const uint32_t count = a.list_count[sgm];
        const uint32_t* seg = a.list + (uint64_t)sgm * a.in_seg_cap;
        for (uint32_t i = 0; i < count; ++i) {
            const uint32_t entry = seg[i], rec = entry & ck::ENTRY_REC;
            const uint64_t len = a.offsets[rec + 1] - a.offsets[rec];
            bool done = false, no_2n = (entry & ck::ENTRY_NO_2N) != 0;
            if (len >= 48 && len < (1ull << 31)) {
                const uint32_t n = (uint32_t)len;
                int why = (entry & ck::ENTRY_NOT_ACGT) ? 1 : 3;
                if (why == 3 && ck::need_dw_strand2(n) <= total_dw) { why = ck::canon_record_team2(a, rec, buf, lut, blk_count + 1, wib, TEAM_WAVES); done = why == 0; }
                if (why == 1 && !(entry & ck::ENTRY_NO_2N) && ck::need_dw_2n(n) <= total_dw) {
                    done = ck::canon_record_team2n(a, rec, buf, lut, lutn, blk_count + 1, wib, TEAM_WAVES);
                    no_2n = true;
                }
                if (!done && why == 1 && ck::need_dw_strand4(n) <= total_dw) done = ck::canon_record_team4(a, rec, buf, blk_count + 1, wib, TEAM_WAVES) == 0;
            }
            if (!done) {
                // not the team's: wave 0 alone with the same buffer, the general routine; the others wait
                if (wib == 0) {
                    bool not_acgt = (entry & ck::ENTRY_NOT_ACGT) != 0 || no_2n;
                    if (!ck::canon_record(solo, rec, buf, lut, lutn, not_acgt, no_2n)) ck::defer_record(a, blk_count, block, rec, not_acgt, no_2n);
                }
                __syncthreads();
            }
        }
    }
}
__global__ __launch_bounds__(TEAM_WAVES * 64) void canon_team_kernel(ck::CanonArgs a, uint32_t nvb, uint32_t* giants)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t total_dw = TEAM_WAVES * a.slice_dw;
    uint32_t* blk_count = lds + total_dw;
    uint32_t* lut = blk_count + 4;
    uint32_t* lutn = lut + ck::FAST_LUT_DW;
    {
        // nothing in any of this workgroup's input segments (every ordinary batch): one parallel look, done
        const uint32_t per = a.segs_per_block, mine = blockIdx.x < nvb ? (nvb - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
        uint32_t any = 0;
        for (uint32_t t = threadIdx.x; t < mine * per; t += TEAM_WAVES * 64) {
            const uint32_t sgm = (blockIdx.x + (t / per) * gridDim.x) * per + t % per;
            if (sgm < a.in_nseg) any |= a.list_count[sgm];
        }
        if (!__syncthreads_or((int)any)) {
            for (uint32_t t = threadIdx.x; t < mine; t += TEAM_WAVES * 64) a.defer_count[blockIdx.x + t * gridDim.x] = 0;
            return;
        }
    }
    ck::fast_lut_init(lut, threadIdx.x, TEAM_WAVES * 64);
    ck::fast_lutn_init(lutn, threadIdx.x, TEAM_WAVES * 64);
    const uint32_t wib = ck::uniform(threadIdx.x >> 6);
    for (uint32_t vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
        if (threadIdx.x == 0) *blk_count = 0;
        __syncthreads();
        team_stage_block(a, lds, total_dw, lut, lutn, blk_count, vb, wib);
        __syncthreads();
        if (threadIdx.x == 0) {
            a.defer_count[vb] = *blk_count;
            if (giants && *blk_count) atomicAdd(giants, *blk_count);       // tell the global-scratch kernel there is work
        }
    }
}

// The end of the line: records no LDS tier can hold (2-bit beyond ~640 kb).  Same code, one wave per record, with the
// packed strands and the candidate bitmask in a slice of GLOBAL scratch instead of LDS.  The lanes of one wavefront
// hand data to each other through that memory; wave_sync()'s wavefront-scope fences are what the AMDGPU memory model
// asks for there (one wave, one L1, in-order vector memory), so canon_core.h runs unchanged.
// One launch, two phases.  Phase 1: every workgroup (one wave) takes its share of the input segments with a slice of the
// scratch; what still does not fit goes to its output segment.  Phase 2: the workgroup that finishes LAST (arrival
// ticket) takes all output segments with the WHOLE scratch -- by then nobody else uses it.  The hand-over of the
// lists between workgroups follows the agent-scope release / acquire recipe (cdna_hip_programming.md, Guideline 16):
// stores drained, release fence, ticket; the last arriver acquires before it reads.  a2 = the arguments of phase 2.
// Sixteen waves per workgroup: a record whose 2-bit strand (with a few N: + bitmask) fits the slice is the team's, as in
// canon_team_kernel, with the scratch slice in place of the LDS (a 100 Mb record: one wave took 0.29 s); the rest is wave 0's.
__global__ __launch_bounds__(TEAM_WAVES * 64) void canon_global_kernel(ck::CanonArgs a, ck::CanonArgs a2, uint32_t* scratch, uint32_t* ticket, const uint32_t* giants,
                                                                       const uint32_t* tiers_busy, uint32_t* hint_out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *hint_out = *tiers_busy ? 2u : 1u;     // pinned host word: 1 = the tiers idled, 2 = they worked
    if (*giants == 0) return;           // nothing came out of the last LDS tier (every ordinary batch): the launch costs ~3 us, not ~12
    __shared__ uint32_t blk_count[4], lut[ck::FAST_LUT_DW], lutn[ck::FAST_LUTN_DW], last;
    ck::fast_lut_init(lut, threadIdx.x, TEAM_WAVES * 64);
    ck::fast_lutn_init(lutn, threadIdx.x, TEAM_WAVES * 64);
    const uint32_t wib = ck::uniform(threadIdx.x >> 6);
    if (threadIdx.x == 0) blk_count[0] = 0;
    __syncthreads();
    team_stage_block(a, scratch + (size_t)blockIdx.x * a.slice_dw, a.slice_dw, lut, lutn, blk_count, blockIdx.x, wib);
    __syncthreads();
    if (threadIdx.x == 0) {
        a.defer_count[blockIdx.x] = blk_count[0];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t arrived = atomicAdd(ticket, 1u);
        last = arrived == gridDim.x - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *ticket = 0;                                    // ready for the next batch
        }
    }
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) blk_count[0] = 0;
    __syncthreads();
    team_stage_block(a2, scratch, a2.slice_dw, lut, lutn, blk_count, 0, wib);     // no output list: leftovers are counted in status[0]
}

// one record in global scratch (the host API's single-record calls)
__global__ __launch_bounds__(64) void canon_global_one_kernel(ck::CanonArgs a, uint32_t* scratch)
{
    __shared__ uint32_t blk_count, lut[ck::FAST_LUT_DW], lutn[ck::FAST_LUTN_DW];
    ck::fast_lut_init(lut, threadIdx.x, 64);
    ck::fast_lutn_init(lutn, threadIdx.x, 64);
    if (threadIdx.x == 0) blk_count = 0;
    __syncthreads();
    ck::canon_wave_loop(a, scratch, lut, &blk_count, blockIdx.x, gridDim.x, 0, 1, lutn);
}

// XXH3-64 of the listed records (the ones canon_global_kernel finished after the batch's own hash pass)
__global__ __launch_bounds__(64) void xxh3_list_kernel(const uint8_t* bytes, const uint64_t* offsets, const uint32_t* list, uint32_t count,
                                                      uint64_t* out)
{
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t r = list[i];
        const uint64_t off = offsets[r];
        const uint64_t h = ck::xxh3_64_wave(bytes + off, (uint32_t)(offsets[r + 1] - off));
        if (ck::lane_id() == 0) out[r] = h;
    }
}

// The streaming kernel with workgroup-staged input (canon_stream.h): a ring of images of record groups per
// workgroup, the decode table and the deferral counter.
#ifndef CK_STREAM_RPW
#define CK_STREAM_RPW 1      // records per wave per group
#endif
// Two builds of every streaming kernel: ROWS = 1 takes records of 48..1008 bases (one packed word per lane), ROWS = 2
// also 1009..2032 (two words per lane, 2 KiB of image per record); it is used when at least one record in four is
// such a record.  Carrying
// the two-word path makes the one-word path ~12 % longer, hence two builds.  Who decides: the host when it has the
// offsets (host-buffer API); otherwise the device, for every batch anew: stream_count_kernel counts the batch's
// lengths and every kernel derives the mode from the counters (batch_mode).  The call stays asynchronous: BOTH builds
// are launched, the one the previous batch used with a full grid, the other with a small one -- until the batches before
// have reported the same mode twice: then only that mode's build is launched (launch_canon, MODE_GUESS).
using StreamC = ck::StreamCfg<CK_STREAM_WPB, CK_STREAM_NBUF, CK_STREAM_RPW, 1>;
// (the two-word build in 8-wave workgroups, four per CU: 9 GB batches of 1.2 / 1.5 / 2 kb records 5.11 / 4.50 / 3.82 -> 4.98 / 4.32 / 3.76 ms)
#ifndef CK_STREAM_WPB_2
#define CK_STREAM_WPB_2 8
#endif
using StreamC2 = ck::StreamCfg<CK_STREAM_WPB_2, CK_STREAM_NBUF, CK_STREAM_RPW, 2>;
// The builds with the fused XXH3 (ROWS = 1) run 8-wave workgroups, four per CU, groups of 8 records (round 4): they are bound by
// instruction issue, and four independent barrier domains per CU leave fewer issue slots empty than two -- hash only -3 % on
// every box measured, bytes + hash -2 % .. 0.  The bytes-only builds are bound by the memory system and keep 16 waves (-1 % on one
// box, +3.5 % on another).
#ifndef CK_STREAM_WPB_HASH
#define CK_STREAM_WPB_HASH 4
#endif
using StreamCH = ck::StreamCfg<CK_STREAM_WPB_HASH, CK_STREAM_NBUF, CK_STREAM_RPW, 1>;
// ... and, for batches of pure ACGT, take TWO records per wave, one per half-wave (canon_pair.h): the scalar side of an iteration
// and the vector work that does not depend on the bytes are shared by two records.  CK_STREAM_WPB_PAIR waves, groups of twice as
// many records; CK_PAIR_WPE waves per SIMD (the register budget the build is compiled for).  CK_STREAM_WPB_PAIR 0: no such build.
// The bytes-only N build (MODE_ALPHA batches): 4 waves.  A record with an N among the symbols that decide (one in
// thirteen at 1 % N) takes the 4-bit routine behind the N-mask routine -- twice the work --, and the workgroup's barrier makes every
// wave wait for it: with 16 waves 72 % of the iterations hold such a record, with 8 waves 47 %, with 4 waves 27 % (10M x 1 kb, 1 % N, one
// box: 16 waves 4.90 ms, 8 waves 4.65, 4 waves 4.56, 2 waves 4.70).
#ifndef CK_STREAM_WPB_ALPHA
#define CK_STREAM_WPB_ALPHA 4
#endif
using StreamCA = ck::StreamCfg<CK_STREAM_WPB_ALPHA, CK_STREAM_NBUF, CK_STREAM_RPW, 1>;
#ifndef CK_STREAM_WPB_PAIR
#define CK_STREAM_WPB_PAIR 8
#endif
#ifndef CK_PAIR_WPE
#define CK_PAIR_WPE 6
#endif
#if CK_STREAM_WPB_PAIR
using StreamCHP = ck::StreamCfg<CK_STREAM_WPB_PAIR, CK_STREAM_NBUF, 2, 1>;
#else
using StreamCHP = StreamCH;
#endif
// The build with every output (index / strand / forward-only) needs ~100 SGPRs: at 16 waves per workgroup only one
// workgroup would fit a CU (measured: 5.1-5.7 ms instead of 3.4-4.0).  It keeps 4-wave workgroups (2 records per
// wave), where SGPRs only cost a seventh wave per SIMD.
using StreamCAux = ck::StreamCfg<4, 4, 2, 1>;
using StreamCAux2 = ck::StreamCfg<4, 2, 2, 2>;

// The ROWS = 2 build is used when at least 1 record in 4 is a 1009..2032-base one (it runs the shorter records 4-6 %
// slower, the longer ones 1.5x faster than LDS tier A; at 15 % -- BASELINE config 4 -- it measured 2 % slower overall).
// The decision is taken again for EVERY batch from the batch's own offsets (an earlier version remembered it per
// offsets pointer, which a host that reuses one offsets buffer defeats) -- from a SAMPLE of them: every mode computes
// the same results, the choice only has to be right about which is fastest, and reading all 80 MB of a 10M-record
// batch's offsets cost 16-21 us per batch.  Every 2^count_shift(n)-th record is looked at (<= 128k of them).
__host__ __device__ inline uint32_t count_shift(uint64_t n) { return n <= (1u << 17) ? 0u : (uint32_t)(63 - __builtin_clzll(n)) - 16u; }
__host__ __device__ inline uint64_t count_samples(uint64_t n) { return (n + (1ull << count_shift(n)) - 1) >> count_shift(n); }
// 1 / 2: which build of the streaming kernel; 3: neither -- with one record in eight longer than 2032 bases at most
// 12 % of the 16-record groups could be staged, and the rescue pass takes every record straight away
__host__ __device__ inline uint32_t stream_mode(uint64_t two, uint64_t lng, uint64_t n)
{
    return lng && lng * 8 >= n ? 3u : (two && two * 4 >= n ? 2u : 1u);
}
// The same kernel samples the CONTENT: up to CONTENT_SAMPLES evenly spaced records, a wave each, first 1008 bytes -- does
// the record hold a byte outside ACGT?  With one sampled record in 16 (or more) doing so, the batch's mode carries
// MODE_ALPHA and the rescue pass runs its build with the 4-bit register path (canon_stream.h); otherwise the lean build,
// which leaves the odd N-bearing record to the LDS tiers (carrying the 4-bit path costs the lean one 25-45 %, measured).
constexpr uint32_t CONTENT_SAMPLES = 4096, MODE_ALPHA = 4;
// MODE_SHORT (mode 1, no MODE_ALPHA): most of the batch's records are short -- the bytes-only streaming kernel then runs its pair
// build (two records per wave, canon_pair.h); records of up to SHORT_MAX_N symbols count as short, half of the samples make the mode
constexpr uint32_t MODE_SHORT = 8, SHORT_MAX_N = 800;
__host__ __device__ inline uint32_t alpha_mode(uint64_t bad, uint64_t sampled) { return bad && bad * 16 >= sampled ? MODE_ALPHA : 0u; }
// The batch's mode from the samples.  The two-word build of the streaming kernel has no alphabet twin -- its records with an N went
// to LDS stage A one by one (6M x 1.5 kb with 1 % N: 11.9 ms, 0.19 of peak) --, the mixed-length kernels have one: a batch of
// two-word records WITH N is theirs (mode 3).
// ... and so is one whose XXH3 is wanted (hashed: hashes and no index / strand): the two-word build finishes every record's hash by
// itself (fast_hash2: two blocks, a scramble between them), the mixed-length kernels with the XXH3 are 8-11 % faster on such batches
// (6M x 1.5 kb: 6.69 -> 6.12 ms; bytes only the two-word build wins, 4.45 against 4.74).
__host__ __device__ inline uint32_t batch_mode_of(uint64_t two, uint64_t lng, uint64_t n, uint64_t bad, uint64_t sampled, uint64_t shortc, bool hashed)
{
    const uint32_t m = stream_mode(two, lng, n), al = alpha_mode(bad, sampled);
    return (m == 2 && (al || hashed) ? 3u : m) | al | (m == 1 && !al && shortc * 2 >= n ? MODE_SHORT : 0u);
}
// ctl: [0] two-word records among the samples, [1] longer ones, [2] arrival ticket, [3] content samples with a byte
// outside ACGT -- all zero on entry and on exit; *mode receives stream_mode() | alpha_mode() of the samples (written by
// the workgroup that arrives last), counters[0] (records beyond the LDS tiers) and counters[3] (records nothing could
// take) are zeroed: nothing of this batch has touched them yet, nothing of the previous one is still running.
__global__ __launch_bounds__(1024) void stream_count_kernel(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offsets, uint64_t n, uint32_t* ctl,
                                                            uint32_t* mode, uint32_t* counters, uint32_t hashed)
{
    __shared__ uint32_t blk[4];
    if (threadIdx.x < 4) blk[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 3) { counters[0] = 0; counters[1] = 0; counters[3] = 0; counters[16] = 0; counters[17] = 0; }      // ([16], [17]: the mixed N builds' segment ticket -- self-zeroing, zeroed here as well)
    __syncthreads();
    const uint32_t shift = count_shift(n);
    const uint64_t ns = count_samples(n);
    uint32_t two = 0, lng = 0, sht = 0; // records of 1009..2032 bases / longer ones (no build can stage their group) / short ones
    for (uint64_t j = (uint64_t)blockIdx.x * 1024 + threadIdx.x; j < ns; j += (uint64_t)gridDim.x * 1024) {
        const uint64_t i = j << shift;
        const uint64_t len = offsets[i + 1] - offsets[i];
        two += len > ck::FAST_MAX_N && len <= ck::FAST2_MAX_N;
        lng += len > ck::FAST2_MAX_N;
        sht += len <= SHORT_MAX_N;
    }
    // content samples: record k * cstep, one wave each
    const uint64_t nc = n < CONTENT_SAMPLES ? n : CONTENT_SAMPLES, cstep = n / nc;
    uint32_t bad = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 6); k < nc; k += (uint64_t)gridDim.x * 16) {
        const uint64_t off = offsets[k * cstep], len = offsets[k * cstep + 1] - off;
        const uint32_t m = len < ck::FAST_MAX_N ? (uint32_t)len : ck::FAST_MAX_N, t = ck::lane_id();
        uint32_t miss = 0;
        if (m >= 16 && 16 * t < m) (void)ck::fast_pack(ck::load16(bytes + off + (16 * t + 16 <= m ? 16 * t : m - 16)), miss);
        bad += ck::ballot(miss != 0) != 0;
    }
    // one global atomic per workgroup and counter, and none for the common batch: thousands of waves adding to the
    // same two words serialise at ~10 ns each (measured: 188 us on BASELINE config 4 with one atomic per wave)
    if (ck::ballot(two | lng) != 0) {
        const uint64_t t2 = ck::wave_sum_u64(two), tl = ck::wave_sum_u64(lng);
        if (ck::lane_id() == 0 && t2) atomicAdd(&blk[0], (uint32_t)t2);
        if (ck::lane_id() == 0 && tl) atomicAdd(&blk[1], (uint32_t)tl);
    }
    if (ck::lane_id() == 0 && bad) atomicAdd(&blk[2], bad);
    if (ck::ballot(sht != 0) != 0) {                        // (none for the headline batch)
        const uint64_t ts = ck::wave_sum_u64(sht);
        if (ck::lane_id() == 0) atomicAdd(&blk[3], (uint32_t)ts);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blk[0]) atomicAdd(ctl, blk[0]);
        if (blk[1]) atomicAdd(ctl + 1, blk[1]);
        if (blk[2]) atomicAdd(ctl + 3, blk[2]);
        if (blk[3]) atomicAdd(ctl + 5, blk[3]);
        // arrival ticket: the adds above are device-scope atomics (performed at the memory side, in order behind this
        // wave's earlier ones once drained), the last arriver reads the sums with atomics as well
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(ctl + 2, 1u) == gridDim.x - 1) {
            const uint32_t t2 = atomicExch(ctl, 0u), tl = atomicExch(ctl + 1, 0u), tb = atomicExch(ctl + 3, 0u), ts = atomicExch(ctl + 5, 0u);
            *mode = batch_mode_of(t2, tl, ns, tb, nc, ts, hashed != 0);
            atomicExch(ctl + 2, 0u);
        }
    }
}
// every kernel of a batch takes the mode from the same word (or the host's answer): bits 0..1 = stream_mode, MODE_ALPHA
__device__ __forceinline__ uint32_t batch_mode(const uint32_t* __restrict__ mode, uint32_t host_mode)
{
    return host_mode ? host_mode & 15u : *mode;         // (bit 31: MODE_GUESS, see launch_canon)
}
// Rescue pass (canon_stream.h): the streaming kernel's leftovers that are eligible by themselves, one wave per record.
// A small persistent grid walks the list segments (a batch the streaming kernel handled completely leaves them
// empty: the pass then costs a microsecond, not the dispatch of one workgroup per segment); segment s of the input
// list yields segment s of the output list, so the tiers behind keep their geometry.
// ALPHA: the build with the 4-bit register path and the prefetching loop, for batches whose mode carries MODE_ALPHA
// (builds with index / strand outputs exist only as ALPHA = false: their N-bearing records take the LDS tiers).
template <bool HASH, bool AUX, bool ALPHA>
__global__ __launch_bounds__(256) void canon_rescue_kernel(ck::CanonArgs a, const uint32_t* __restrict__ mode_word, uint32_t host_mode, uint32_t* mode_out,
                                                           uint32_t* tiers_busy)
{
    const uint32_t mode = batch_mode(mode_word, host_mode);
    if (!AUX && ((mode & MODE_ALPHA) != 0) != ALPHA) return;            // the other build has this batch
    if (!AUX && (mode & 3) == 3) return;                                 // ... or a canon_mixed kernel (either alphabet, with or without the XXH3)
    if (blockIdx.x == 0 && threadIdx.x == 0) *mode_out = (host_mode >> 31) ? *mode_word : mode;     // for circkit_ctx_last_batch_mode() and the next batch's launch: the batch's OWN mode
    const bool all_records = (mode & 3) == 3;                           // the streaming kernel stood this batch out
    if (!all_records) {
        // the ordinary batch leaves this pass (next to) nothing: one parallel look at the workgroup's segments first
        const uint32_t mine = blockIdx.x < a.in_nseg ? (a.in_nseg - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
        uint32_t any = 0;
        for (uint32_t t = threadIdx.x; t < mine; t += 256) any |= a.list_count[blockIdx.x + t * gridDim.x];
        if (!__syncthreads_or((int)any)) {
            for (uint32_t t = threadIdx.x; t < mine; t += 256) a.defer_count[blockIdx.x + t * gridDim.x] = 0;
            return;
        }
    }
    __shared__ uint32_t lut[ck::FAST_LUT_DW], seg_count;
    ck::fast_lut_init(lut, threadIdx.x, 256);
    ck::RescueState<HASH, AUX> st;
    if (HASH) st.hc = ck::fast_hash_const();
    const uint32_t wib = ck::uniform(threadIdx.x >> 6);
    uint32_t passed_on = 0, walked = 0;          // (thread 0) records this workgroup hands to the tiers / segments it walked
    for (uint32_t sgm = blockIdx.x; sgm < a.in_nseg; sgm += gridDim.x) {
        if (threadIdx.x == 0) seg_count = 0;
        __syncthreads();
        ck::canon_rescue_segment<HASH, AUX, ALPHA>(a, lut, st, &seg_count, sgm, wib, 4, all_records);
        __syncthreads();
        if (threadIdx.x == 0) { a.defer_count[sgm] = seg_count; passed_on += seg_count; ++walked; }
    }
    // a hint for the NEXT batch's launch: do the LDS tiers have real work (more than one record per segment on
    // average)?  Idle tiers are launched with small grids (they walk the segments): a full-size grid that finds every
    // segment empty costs 12-17 us per tier.  Plain store, same value from whoever stores.
    if (threadIdx.x == 0 && passed_on > walked) *tiers_busy = 1;
}

// Batches of mixed lengths (mode 3: the streaming kernel stands them out): ONE kernel walks all records, four waves per
// workgroup with a stage-A slice each (canon_mixed.h).  A record of 48..1008 bases takes the register routine straight from
// memory, a longer one the LEAN LDS routine -- aligned chunks, one scan loop for both strands, the unique-minimum path only;
// the latency-bound short records and the bandwidth-bound long ones share the CUs.  Everything the lean routines do not
// decide (ties, palindromes, strangers in the alphabet, records beyond the slice) goes to segment s of the list stage A
// consumes, whose kernel carries the general routine and the team mode.  Segment s = records [s * all_seg_cap, (s + 1) *
// all_seg_cap).  NM: the build for batches whose mode carries MODE_ALPHA (N packed as G + a mask bit per symbol).
// Batches that also want the XXH3, the rotation index or the strand keep the rescue pass + stage A.
// amdgpu_num_vgpr(36): on gfx90a+ the attribute counts in the unified VGPR + AGPR file at twice the value -- 72 registers --
// and, unlike __launch_bounds__(256, 7), leaves the kernel all 102 scalar registers.  What the CU then holds is SIX of these
// workgroups, not seven: 106 SGPRs + the trap handler's 16 round to 128 of the SIMD's 800 (tools/probe_timeline.py stamps
// every workgroup: exactly 6 resident per CU; 7 with CK_MIXED_WAVES=7 = 94 SGPRs and 19 spills, 8 with =8 = 78 and 32).
// Measured on one box: 1.85 ms at six, 1.86 at seven, 1.98 at eight (smaller slices: more records for stage A) -- the
// kernel is bound by what a CU issues and moves per cycle, not by latency; the fewest spills win.
#ifdef CK_MIXED_TIMELINE
// Experiment builds only (tools/probe_timeline.py): start / end of every workgroup of the last mixed-length kernel on the
// 100 MHz wall clock, and the CU it ran on.
__device__ uint64_t g_timeline[3 * 65536];
#endif
// Waves per workgroup of the mixed-length kernels (a slice of LDS each; a workgroup takes one list segment -- ~30 records of config 4 --
// and leaves when its slowest wave does).  Measured on config 4, same box, 24 waves per CU in every case (tools/probe_variants.py):
// bytes only 1 / 2 / 4 / 8 waves 1.90 / 1.88 / 1.81 / 1.75 ms, with the XXH3 2.17 / 2.11 / 2.20 / 2.52.  The N builds' resident grid
// is sized for four.
#ifndef CK_MIXED_WPB_PURE
#define CK_MIXED_WPB_PURE 8
#endif
#ifndef CK_MIXED_WPB_HASH
#define CK_MIXED_WPB_HASH 2
#endif
#ifndef CK_MIXED_WPB_N
#define CK_MIXED_WPB_N 4
#endif
constexpr int mixed_wpb(bool nm, bool hash) { return nm ? CK_MIXED_WPB_N : (hash ? CK_MIXED_WPB_HASH : CK_MIXED_WPB_PURE); }
template <bool NM, bool HASH>
// Segments are handed out by a TICKET (round 4): a grid of as many workgroups as the chip holds at once, each taking the next
// segment from a global counter until the segments run out -- no workgroup hand-over between segments (round 3's timeline of one
// workgroup per segment: 1450 of 1536 slots filled in the steady state), the order stays the dispatch order the tapered tail is
// built for.  ticket[0] = next segment, ticket[1] = workgroups that are through; the last one through zeroes both, so every
// launch finds them zero (a kernel that returns at its mode check never touches them).
__device__ __forceinline__ void canon_mixed_body(const ck::CanonArgs& a, const uint32_t* __restrict__ mode_word, uint32_t host_mode, uint32_t* mode_out, uint32_t* tiers_busy,
                                                 uint32_t* ticket)
{
    constexpr int WPB = mixed_wpb(NM, HASH);
    const uint32_t mode = batch_mode(mode_word, host_mode);
    if ((mode & 3) != 3 || ((mode & MODE_ALPHA) != 0) != NM) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) *mode_out = (host_mode >> 31) ? *mode_word : mode;
#ifdef CK_MIXED_TIMELINE
    const uint64_t tl_start = wall_clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* blk_count = lds + WPB * a.slice_dw;
    uint32_t* lut = blk_count + 4;
    uint32_t* lutn = lut + ck::FAST_LUT_DW;                // N builds: the 'G' / 'C' -> 'N' patch tables (the output is patched in registers)
    uint32_t* htab = lutn + ck::LEAN_LUTN_DW;              // (HASH builds: the XXH3 per-pair constants)
    ck::fast_lut_init(lut, threadIdx.x, 64 * WPB);
    if (NM) ck::lean_lutn_init(lutn, threadIdx.x, 64 * WPB);
    if (HASH) ck::lean_hash_table_init(htab, threadIdx.x);
    ck::RescueState<HASH, false> st;
    if (HASH) {                                          // (only the lane's stripe secrets stay in registers: lean_hash_refill)
        st.hc.k0 = ck::xsec64(8 * (ck::lane_id() >> 2) + 16 * (ck::lane_id() & 3));
        st.hc.k1 = ck::xsec64(8 * (ck::lane_id() >> 2) + 16 * (ck::lane_id() & 3) + 8);
    }
    const uint32_t wib = ck::uniform(threadIdx.x >> 6);
    uint32_t* slice = lds + wib * a.slice_dw;
    const uint64_t payload_end = a.offsets[a.n_records];
    uint32_t passed_on = 0, walked = 0, static_sgm = blockIdx.x;
    for (;;) {
        // (ticket == nullptr: the static mapping -- workgroup b takes segments b, b + grid, ...; with one workgroup per segment
        // that is one segment each.  The pure builds keep it: handed out by ticket to a resident grid they measured 2.5 % SLOWER
        // on config 4, the N builds 2 % faster.)
        if (threadIdx.x == 0) { blk_count[0] = 0; blk_count[1] = ticket ? atomicAdd(ticket, 1u) : static_sgm; }
        static_sgm += gridDim.x;
        __syncthreads();
        const uint32_t sgm = blk_count[1];
        if (sgm >= a.in_nseg) break;                       // (every wave of every workgroup gets here: the segments run out)
        ck::canon_mixed_segment<NM, HASH>(a, slice, lut, st, blk_count, sgm, wib, WPB, payload_end, htab, lutn);
        __syncthreads();
        if (threadIdx.x == 0) { a.defer_count[sgm] = *blk_count; passed_on += *blk_count; ++walked; }
    }
    if (threadIdx.x == 0) {
        if (passed_on > walked) *tiers_busy = 1;
        if (ticket && atomicAdd(ticket + 1, 1u) == gridDim.x - 1) { atomicExch(ticket, 0u); atomicExch(ticket + 1, 0u); }
    }
#ifdef CK_MIXED_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x < 65536) {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_timeline[3 * blockIdx.x] = tl_start; g_timeline[3 * blockIdx.x + 1] = wall_clock64();
        g_timeline[3 * blockIdx.x + 2] = ((uint64_t)xcc << 32) | hwid;
    }
#endif
}
#ifndef CK_MIXED_NM_VGPR
#define CK_MIXED_NM_VGPR 48       // the N builds: five workgroups per CU by LDS (strand + N bits) = five waves per SIMD: 96 registers each
#endif
#ifndef CK_MIXED_H_VGPR
#define CK_MIXED_H_VGPR 40       // the builds with the fused XXH3: 80 registers, six waves per SIMD
#endif
#ifndef CK_MIXED_N_SLICE
#define CK_MIXED_N_SLICE 1904     // dwords per wave of the N builds: strand + N bits + candidates of a 20 kb record (1253 + 628 + 18); five workgroups per CU by LDS
#endif
#ifdef CK_MIXED_WAVES
#define CK_MIXED_ATTR(VGPR, NM, HASH) __launch_bounds__(64 * mixed_wpb(NM, HASH), CK_MIXED_WAVES)
#else
#define CK_MIXED_ATTR(VGPR, NM, HASH) __launch_bounds__(64 * mixed_wpb(NM, HASH)) __attribute__((amdgpu_num_vgpr(VGPR)))
#endif
#define CK_MIXED_KERNEL(NAME, NM, HASH, VGPR)                                                                                                  \
    __global__ CK_MIXED_ATTR(VGPR, NM, HASH) void NAME(ck::CanonArgs a, const uint32_t* __restrict__ mode_word,   \
                                                                                     uint32_t host_mode, uint32_t* mode_out, uint32_t* tiers_busy, uint32_t* ticket) \
    {                                                                                                                                          \
        canon_mixed_body<NM, HASH>(a, mode_word, host_mode, mode_out, tiers_busy, ticket);                                                     \
    }
CK_MIXED_KERNEL(canon_mixed_kernel, false, false, 36)
CK_MIXED_KERNEL(canon_mixed_n_kernel, true, false, CK_MIXED_NM_VGPR)
CK_MIXED_KERNEL(canon_mixed_h_kernel, false, true, CK_MIXED_H_VGPR)          // + XXH3 (uniq on batches of mixed lengths)
CK_MIXED_KERNEL(canon_mixed_nh_kernel, true, true, CK_MIXED_NM_VGPR)
#undef CK_MIXED_KERNEL

// GH: the fused XXH3 is finished per 16-record group by one wave (canon_fast.h group_hash_*): 18.2 KiB more LDS, which
// the ROWS = 2 build cannot afford next to its 64 KiB of images (two workgroups per CU are what matters most).
#ifndef CK_GROUP_HASH
#define CK_GROUP_HASH 1
#endif
// PERSIST = false: one workgroup per list segment (the build the host expects to match the batch, at full size).
// PERSIST = true: `nvb` virtual workgroups walked by a small grid -- the OTHER build, whose workgroups normally return
// at once (4-5 us instead of the ~40 us a full-size idle grid costs) and take the whole batch when the expectation was
// wrong.  Two instantiations because the loop over virtual workgroups costs the hot path scalar registers (measured:
// +2 % on the headline with one kernel for both).
// ALPHA (ROWS = 1, no index / strand outputs): the build for batches with MODE_ALPHA -- records with N or '-' take the
// 4-bit register routine inside the staged loop.  The builds without it answer for every other batch, MODE_ALPHA or not.
template <class StreamC, bool HASH, bool AUX, bool PERSIST, bool ALPHA = false>
__global__ __launch_bounds__(StreamC::WPB * 64, (HASH && !AUX && StreamC::ROWS == 1 && (ALPHA || StreamC::RPW == 2)) ? CK_PAIR_WPE : (StreamC::WPB >= 8 || (ALPHA && !HASH && !AUX)) ? CK_FAST_WPE : 4) void canon_stream_kernel(ck::CanonArgs a, const uint32_t* __restrict__ mode,
                                                                                                             uint32_t host_mode, uint32_t nvb)
{
    const uint32_t bm = batch_mode(mode, host_mode);
    if ((bm & 3) != (uint32_t)StreamC::ROWS) return;                             // the other build (or none) has this batch
    constexpr bool HAS_ALPHA_TWIN = StreamC::ROWS == 1 && !AUX;
    if (HAS_ALPHA_TWIN && ((bm & MODE_ALPHA) != 0) != ALPHA) return;
    if (HAS_ALPHA_TWIN && !ALPHA && !HASH && ((bm & MODE_SHORT) != 0) != (StreamC::RPW == 2)) return;       // bytes only: the pair build has the batches of short records
    constexpr bool GH = CK_GROUP_HASH && HASH && !AUX && StreamC::ROWS == 1 && (StreamC::RPW == 1 || !ALPHA) && StreamC::GROUP <= 16;
    constexpr bool PAIR = GH && StreamC::RPW == 2;                                // canon_pair.h: two records per wave
    constexpr bool PAIR_B = !HASH && !AUX && !ALPHA && StreamC::ROWS == 1 && StreamC::RPW == 2;    // the bytes-only pair build: MODE_SHORT batches
    __shared__ __attribute__((aligned(16))) uint32_t lds[StreamC::LDS_DW + (GH ? ck::gh_lds_dw<(int)StreamC::GROUP, PAIR>() : PAIR_B ? StreamC::GROUP * ck::PAIR_SCRATCH_DW : 0)];
    uint32_t* lut = lds + StreamC::NBUF * StreamC::BUF_DW;
    uint32_t* blk_count = lut + ck::FAST_LUT_DW;
    uint32_t* gh = lds + StreamC::LDS_DW;
    ck::fast_lut_init(lut, threadIdx.x, StreamC::WPB * 64);
    if (GH) ck::group_hash_init(gh + 2 * StreamC::GROUP * ck::GH_STRIDE_DW, threadIdx.x);
    if (PAIR) ck::group_hash_secret_init(gh + 2 * StreamC::GROUP * ck::GH_STRIDE_DW + ck::GH_CONST_DW, threadIdx.x);
    if (!PERSIST) {
        if (threadIdx.x == 0) *blk_count = 0;
        __syncthreads();
        ck::canon_stream_wave_loop<StreamC, HASH, AUX, GH, ALPHA>(a, lut, lds, blk_count, blockIdx.x, gridDim.x, gh);
        __syncthreads();
        if (threadIdx.x == 0) a.defer_count[blockIdx.x] = *blk_count;
        return;
    }
    for (uint32_t vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
        if (threadIdx.x == 0) *blk_count = 0;
        __syncthreads();
        ck::canon_stream_wave_loop<StreamC, HASH, AUX, GH, ALPHA>(a, lut, lds, blk_count, vb, nvb, gh);
        ck::vmem_wait<0>();                              // no DMA of this virtual workgroup may land in the next one's images
        __syncthreads();
        if (threadIdx.x == 0) a.defer_count[vb] = *blk_count;
    }
}

// XXH3-64 of each record of a CSR batch, one wavefront per record (see xxh3_core.h).  With `hashed` (the flags of
// the records the streaming kernel hashed itself) a wave takes 64 records at a time, one flag per lane, and only
// visits the ones still missing -- the pass over an all-hashed batch is one byte load per record.
// view (hash-only batches): `bytes` is the INPUT payload and record r's canonical form the rotation / reverse complement of it
// that view[r] names (CanonArgs::out_view, comp = the ctx's complement table) -- nothing was written out as bytes.
__global__ __launch_bounds__(256) void xxh3_kernel(const uint8_t* bytes, const uint64_t* offsets, uint64_t n_records,
                                                   uint64_t* out, const uint8_t* hashed, const uint32_t* view, const uint8_t* comp)
{
    const uint32_t wave = ck::uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * 4;
    const ck::XWaveConst xk = ck::xwave_const();
    // (views: the complement table in LDS -- a reverse-strand view looks every byte up, and the short records' lanes each their own)
    __shared__ uint8_t comp_lds[256];
    if (view) { comp_lds[threadIdx.x] = comp[threadIdx.x]; __syncthreads(); comp = comp_lds; }
    if (!hashed) {
        for (uint64_t r = wave; r < n_records; r += n_waves) {
            const uint64_t off = offsets[r];
            const uint64_t h = ck::xxh3_64_wave(bytes + off, (uint32_t)(offsets[r + 1] - off), xk);
            if (ck::lane_id() == 0) out[r] = h;
        }
        return;
    }
    // A wave looks at `span` consecutive records at a time: 512 (8 flags per lane) when the batch is big enough to keep every wave
    // busy that way, fewer -- a multiple of 64 -- for smaller batches: with 512 a million-record batch of mixed lengths whose
    // hashes are all still to do (the N build of the mixed kernel) kept a quarter of the waves busy, 2.85 ms for 4.3 GB.
    const uint64_t per_wave = (n_records + n_waves - 1) / n_waves;
    const uint32_t span = per_wave >= 512 ? 512u : (per_wave <= 64 ? 64u : (uint32_t)((per_wave + 63) / 64) * 64u);
    // up to 512 flags per look (8 per lane, one 8-byte load: the flag array is 8-byte aligned and padded): an all-hashed batch of
    // 10M records is 2-3 dependent loads per wave instead of 19 (28 -> ~8 us per batch)
    for (uint64_t big = (uint64_t)wave * span; big < n_records; big += (uint64_t)n_waves * span) {
      const uint64_t f0 = big + 8 * ck::lane_id();
      uint64_t flags8 = ~0ull;
      if (f0 < n_records && 8 * ck::lane_id() < span) {
          flags8 = *reinterpret_cast<const uint64_t*>(hashed + f0);
          if (n_records - f0 < 8) flags8 |= ~0ull << (8 * (n_records - f0));       // bytes past the last record
      }
      // a zero byte = a record still to hash
      if (ck::ballot(((flags8 - 0x0101010101010101ull) & ~flags8 & 0x8080808080808080ull) != 0) == 0) continue;
      for (uint64_t base = big; base < big + span && base < n_records; base += 64) {
        const uint64_t mine = base + ck::lane_id();
        const bool need = mine < n_records && !hashed[mine];
        uint64_t todo = ck::ballot(need);
        if (!todo) continue;
        // the chunk's offsets in one coalesced load (lane L: record base + L), handed out by v_readlane: no dependent
        // round trip in front of every record
        const uint64_t my_off = need ? offsets[mine] : 0;
        const uint32_t my_len = need ? (uint32_t)(offsets[mine + 1] - my_off) : 0u;
        // XXH3's short-input classes (<= 240 bytes) are ONE scalar recipe (xxh3_short): every lane hashes its own record instead of
        // the whole wave hashing one after the other -- a batch of short reads (20M x 200 b, none of whose hashes the kernels
        // in front fuse) took 58 ms with bytes, 89 ms from views; ~60 times fewer instructions this way
        const bool mine_short = need && my_len <= 240u;
        const uint64_t shorts = ck::ballot(mine_short);
        if (shorts) {
            if (mine_short) {
                const uint8_t* p = bytes + my_off;
                uint64_t h;
                if (view) {
                    const uint32_t v = view[mine], rot = v & 0x7FFFFFFFu;
                    h = ck::xxh3_short(ck::XView{ p, my_len, rot < my_len ? rot : 0u, (v >> 31) != 0, comp }, my_len);
                } else h = ck::xxh3_short(ck::XPlain{ p }, my_len);
                out[mine] = h;
            }
            todo &= ~shorts;
        }
        while (todo) {
            const uint32_t l = (uint32_t)ck::ffs64(todo);
            todo &= todo - 1;
            const uint64_t off = ((uint64_t)ck::readlane((uint32_t)(my_off >> 32), l) << 32) | ck::readlane((uint32_t)my_off, l);
            const uint32_t len = ck::readlane(my_len, l);
            const uint64_t h = view ? ck::xxh3_64_wave_view(bytes + off, len, view[base + l], comp, xk) : ck::xxh3_64_wave(bytes + off, len, xk);
            if (ck::lane_id() == 0) out[base + l] = h;
        }
      }
    }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

// 16 bases per thread, 16-byte stores; base g uses bits 2*(g&31) of splitmix64(seed*K + (g>>5)).
__global__ __launch_bounds__(256) void synth_fill_kernel(uint64_t seed, uint64_t first_base, uint64_t n_bases,
                                                         uint8_t* out)
{
    const uint64_t key = seed * 0xD1342543DE82EF95ULL;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 16;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n_bases; i += stride) {
        const uint64_t g0 = first_base + i;
        uint64_t wi = g0 >> 5, w = splitmix64(key + wi);
        uint32_t o[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const uint64_t g = g0 + b;
            if ((g >> 5) != wi) { wi = g >> 5; w = splitmix64(key + wi); }
            const uint32_t code = (uint32_t)(w >> (2 * (g & 31))) & 3u;
            o[b >> 2] |= ((0x54474341u >> (8 * code)) & 0xFFu) << (8 * (b & 3));
        }
        if (i + 16 <= n_bases) {
            ck::store16(out + i, ck::u32x4{ o[0], o[1], o[2], o[3] });
        } else {
            for (uint64_t b = 0; i + b < n_bases; ++b) out[i + b] = (uint8_t)(o[b >> 2] >> (8 * (b & 3)));
        }
    }
}

// Plain device-to-device copy, 16 bytes per lane, UNROLL loads in flight per lane: the same-process, same-box yardstick the
// bench line quotes next to the canonicalize kernel's rate (circkit_bench_copy_device; shapes from tools/microbench/ceiling_bench.hip).
typedef uint32_t copy_v4 __attribute__((ext_vector_type(4)));
template <int UNROLL>
__global__ __launch_bounds__(256) void bench_copy_kernel(const copy_v4* __restrict__ in, copy_v4* __restrict__ out, uint64_t n16)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256 * UNROLL;
    uint64_t i = (uint64_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    for (; i + 256 * (UNROLL - 1) < n16; i += stride) {
        copy_v4 r[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) r[u] = in[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) out[i + u * 256] = r[u];
    }
    for (; i < n16; i += 256) out[i] = in[i];                 // the last, partial tile of the workgroup that owns it
}

__global__ void fixed_offsets_kernel(uint64_t base, uint64_t len, uint64_t n, uint64_t* off)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) off[i] = base + i * len;
}

// uniq: open-addressing table keyed by the 64-bit hash, value = smallest global record index.  One 16-byte slot
// {key, value} per entry, so that claiming a key and folding its index touch ONE 64-byte sector (two arrays: two
// random DRAM accesses per key).  Slot [mask + 1] keeps the value of the key that equals the EMPTY marker.
constexpr uint64_t UNIQ_EMPTY = ~0ull;
struct __attribute__((aligned(16))) UniqSlot { unsigned long long key, val; };
__device__ __forceinline__ uint64_t uniq_slot(uint64_t h, uint64_t mask) { return (h ^ (h >> 29)) & mask; }
__device__ __forceinline__ UniqSlot uniq_peek(const UniqSlot* p)
{
    typedef unsigned long long v2 __attribute__((ext_vector_type(2)));
    const v2 v = *reinterpret_cast<const v2*>(p);          // one global_load_dwordx4
    return UniqSlot{ v.x, v.y };
}
// Folds (h, v) into the table.  The scattered device-scope atomics are the limiter (~20 G/s chip-wide, measured:
// 10M keys x CAS + min = 1.08 ms), so a plain 16-byte read goes first: a key that is already there with a smaller
// value -- the later copies of a duplicated record, since threads take records roughly in index order -- needs no
// atomic at all, a present key one, a new key two.  The read may be stale; it is only trusted where staleness cannot
// hurt: a key, once written, never changes, and a value only decreases.
__device__ __forceinline__ bool uniq_fold(UniqSlot* t, uint64_t mask, uint64_t h, unsigned long long v)
{
    if (h == UNIQ_EMPTY) { atomicMin(&t[mask + 1].val, v); return true; }
    uint64_t s = uniq_slot(h, mask), probes = 0;
    for (;;) {
        const UniqSlot cur = uniq_peek(t + s);
        if (cur.key == h) {
            if (cur.val > v) atomicMin(&t[s].val, v);
            return true;
        }
        if (cur.key == UNIQ_EMPTY) {
            const unsigned long long old = atomicCAS(&t[s].key, (unsigned long long)UNIQ_EMPTY, (unsigned long long)h);
            if (old == UNIQ_EMPTY || old == h) { atomicMin(&t[s].val, v); return true; }
        }
        s = (s + 1) & mask;
        if (++probes > mask) return false;                  // table full
    }
}

// value of key i: index[i] when given (pairs gathered from other ranks), else base + i (a shard in input order)
__global__ __launch_bounds__(256) void uniq_insert_kernel(const uint64_t* __restrict__ hash, const uint64_t* __restrict__ index, uint64_t n, uint64_t base,
                                                          UniqSlot* t, uint64_t mask, uint32_t* status)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (!uniq_fold(t, mask, hash[i], index ? index[i] : base + i)) atomicAdd(status, 1u);
}

// keep (nullable): keep[i] = 1 iff record base + i is the first with its hash -- the reference's "emit or table row"
// decision (src/uniq.rs:47-62) for a shard whose record i has global index base + i
__global__ __launch_bounds__(256) void uniq_lookup_kernel(const uint64_t* __restrict__ hash, uint64_t n, const UniqSlot* __restrict__ t, uint64_t mask,
                                                          uint64_t* first_seen, uint8_t* keep, uint64_t base)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t h = hash[i];
        uint64_t r = UNIQ_EMPTY;
        if (h == UNIQ_EMPTY) {
            r = t[mask + 1].val;
        } else {
            uint64_t s = uniq_slot(h, mask), probes = 0;
            for (;;) {
                const UniqSlot cur = uniq_peek(t + s);
                if (cur.key == h) { r = cur.val; break; }
                if (cur.key == UNIQ_EMPTY || ++probes > mask) break;
                s = (s + 1) & mask;
            }
        }
        first_seen[i] = r;
        if (keep) keep[i] = r == base + i;
    }
}

// circkit_uniq_resolve_device's own pair of kernels: one shard, one insert pass, one lookup pass, LOCAL indices
// (i < 2^32 - 1: 0xFFFFFFFF is the "nothing yet" value of either half).  The value word is split: its high half receives the index of the record that CLAIMED the key -- a plain
// store, the claimer needs no second atomic --, its low half the smallest index among the records that found the key
// present (32-bit atomicMin, skipped when the peek already shows a smaller index in either half).  first-seen =
// min(low, high).  With threads taking records roughly in index order a key costs ONE atomic (the CAS) instead of two:
// 10M records, half of them duplicates: 507 -> ~350 us (the 207 us of peeks stay).
__device__ __forceinline__ bool uniq_fold_local(UniqSlot* t, uint64_t mask, uint64_t h, uint32_t i)
{
    if (h == UNIQ_EMPTY) { atomicMin(reinterpret_cast<unsigned int*>(&t[mask + 1].val), i); return true; }
    uint64_t s = uniq_slot(h, mask), probes = 0;
    for (;;) {
        const UniqSlot cur = uniq_peek(t + s);
        unsigned int* halves = reinterpret_cast<unsigned int*>(&t[s].val);      // [0] low, [1] high (little endian)
        if (cur.key == h) {
            const uint32_t lo = (uint32_t)cur.val, hi = (uint32_t)(cur.val >> 32);
            if ((lo < hi ? lo : hi) > i) atomicMin(halves, i);
            return true;
        }
        if (cur.key == UNIQ_EMPTY) {
            const unsigned long long old = atomicCAS(&t[s].key, (unsigned long long)UNIQ_EMPTY, (unsigned long long)h);
            if (old == UNIQ_EMPTY) { halves[1] = i; return true; }             // ours: nobody else writes this half
            if (old == h) { atomicMin(halves, i); return true; }
        }
        s = (s + 1) & mask;
        if (++probes > mask) return false;                  // table full
    }
}
__global__ __launch_bounds__(256) void uniq_resolve_insert_kernel(const uint64_t* __restrict__ hash, uint64_t n, UniqSlot* t, uint64_t mask, uint32_t* status,
                                                                  const uint32_t* only_if = nullptr)
{
    if (only_if && *only_if == 0) return;                   // (the bucketed resolve's fallback: runs only when a bucket was too big)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (!uniq_fold_local(t, mask, hash[i], (uint32_t)i)) atomicAdd(status, 1u);
}
__global__ __launch_bounds__(256) void uniq_resolve_lookup_kernel(const uint64_t* __restrict__ hash, uint64_t n, const UniqSlot* __restrict__ t, uint64_t mask,
                                                                  uint64_t* first_seen, uint8_t* keep, uint64_t base, const uint32_t* only_if = nullptr)
{
    if (only_if && *only_if == 0) return;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t h = hash[i];
        uint64_t v = UNIQ_EMPTY;
        if (h == UNIQ_EMPTY) {
            v = t[mask + 1].val;
        } else {
            uint64_t s = uniq_slot(h, mask), probes = 0;
            for (;;) {
                const UniqSlot cur = uniq_peek(t + s);
                if (cur.key == h) { v = cur.val; break; }
                if (cur.key == UNIQ_EMPTY || ++probes > mask) break;
                s = (s + 1) & mask;
            }
        }
        const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32), r = lo < hi ? lo : hi;
        first_seen[i] = r == 0xFFFFFFFFu ? UNIQ_EMPTY : base + r;           // (not found: only after a table overflow)
        if (keep) keep[i] = r == (uint32_t)i;
    }
}

// moves every (key, smallest index) entry of an old table into a bigger one
__global__ __launch_bounds__(256) void uniq_rehash_kernel(const UniqSlot* __restrict__ old, uint64_t oslots, UniqSlot* t, uint64_t mask)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < oslots; i += stride) {
        const UniqSlot e = uniq_peek(old + i);
        if (i + 1 == oslots) { if (e.val != UNIQ_EMPTY) atomicMin(&t[mask + 1].val, e.val); continue; }   // the EMPTY-key slot
        if (e.key != UNIQ_EMPTY) (void)uniq_fold(t, mask, e.key, e.val);
    }
}

// ---- multi-GPU exchange (circkit_amd/uniq.py, exchange="partition"): the key space is cut into `world` ranges, a key
// belongs to rank ((h >> 20) & 0x7FFFFFFF) % world (well-mixed middle bits of the XXH3 output).
constexpr uint32_t UNIQ_MAX_WORLD = 64;
__device__ __forceinline__ uint32_t uniq_owner(uint64_t h, uint32_t world) { return (uint32_t)((h >> 20) & 0x7FFFFFFFu) % world; }
// pass 1: keys per owner (one global atomic per workgroup and owner)
__global__ __launch_bounds__(256) void uniq_partition_count_kernel(const uint64_t* __restrict__ hash, uint64_t n, uint32_t world, unsigned long long* counts)
{
    __shared__ uint32_t hist[UNIQ_MAX_WORLD];
    if (threadIdx.x < UNIQ_MAX_WORLD) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) atomicAdd(&hist[uniq_owner(hash[i], world)], 1u);
    __syncthreads();
    if (threadIdx.x < world && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}
// pass 2: rows[(pos)] = {hash, base + i} with the rows of one owner together (owners in rank order, any order inside),
// slot[i] = pos.  Each workgroup takes chunks of 256 * PER keys: a histogram of the chunk in LDS (the LDS atomic's return value
// is the key's rank inside its owner's share of the chunk), one global atomic per owner to reserve the chunk's rows, then every
// key its place.  PER = 4 with the keys RE-READ in the second phase (they are in the vector cache) instead of 16 with owner and
// rank of every key kept in registers: 131 VGPRs / 3 waves per SIMD -> under 40 / 8; the owners' base
// offsets are a prefix sum of the 64 counts done once per workgroup, not a loop per chunk.
__global__ __launch_bounds__(256) void uniq_partition_scatter_kernel(const uint64_t* __restrict__ hash, uint64_t n, uint64_t base, uint32_t world,
                                                                     const unsigned long long* __restrict__ counts, unsigned long long* cursor,
                                                                     uint64_t* rows, uint32_t* slot)
{
    __shared__ uint32_t hist[UNIQ_MAX_WORLD];
    __shared__ unsigned long long first[UNIQ_MAX_WORLD], start[UNIQ_MAX_WORLD];
    constexpr uint32_t PER = 4;
    const uint64_t chunk = 256ull * PER;
    if (threadIdx.x < UNIQ_MAX_WORLD) {                       // (one wave) exclusive prefix sum of the owners' totals
        unsigned long long v = threadIdx.x < world ? counts[threadIdx.x] : 0ull, incl = v;
        for (uint32_t d = 1; d < UNIQ_MAX_WORLD; d <<= 1) {
            const uint32_t lo = ck::shfl((uint32_t)incl, threadIdx.x - d), hi = ck::shfl((uint32_t)(incl >> 32), threadIdx.x - d);
            if (threadIdx.x >= d) incl += ((unsigned long long)hi << 32) | lo;
        }
        first[threadIdx.x] = incl - v;
    }
    for (uint64_t c0 = (uint64_t)blockIdx.x * chunk; c0 < n; c0 += (uint64_t)gridDim.x * chunk) {
        if (threadIdx.x < UNIQ_MAX_WORLD) hist[threadIdx.x] = 0;
        __syncthreads();
        uint32_t rank[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint64_t i = c0 + k * 256 + threadIdx.x;
            rank[k] = i < n ? atomicAdd(&hist[uniq_owner(hash[i], world)], 1u) : 0u;
        }
        __syncthreads();
        if (threadIdx.x < world)
            start[threadIdx.x] = first[threadIdx.x] + (hist[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (unsigned long long)hist[threadIdx.x]) : 0ull);
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint64_t i = c0 + k * 256 + threadIdx.x;
            if (i < n) {
                const uint64_t h = hash[i];
                const uint64_t pos = start[uniq_owner(h, world)] + rank[k];
                typedef unsigned long long v2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<v2*>(rows + 2 * pos) = v2{ h, base + i };
                slot[i] = (uint32_t)pos;
            }
        }
        __syncthreads();
    }
}
// rows of {hash, global index}: fold them into the table / answer each with the smallest index seen for its hash
__global__ __launch_bounds__(256) void uniq_insert_rows_kernel(const uint64_t* __restrict__ rows, uint64_t n, UniqSlot* t, uint64_t mask, uint32_t* status)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const UniqSlot r = uniq_peek(reinterpret_cast<const UniqSlot*>(rows) + i);
        if (!uniq_fold(t, mask, r.key, r.val)) atomicAdd(status, 1u);
    }
}
__global__ __launch_bounds__(256) void uniq_lookup_rows_kernel(const uint64_t* __restrict__ rows, uint64_t n, const UniqSlot* __restrict__ t, uint64_t mask,
                                                               uint64_t* answers)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t h = rows[2 * i];
        uint64_t r = UNIQ_EMPTY;
        if (h == UNIQ_EMPTY) {
            r = t[mask + 1].val;
        } else {
            uint64_t s = uniq_slot(h, mask), probes = 0;
            for (;;) {
                const UniqSlot cur = uniq_peek(t + s);
                if (cur.key == h) { r = cur.val; break; }
                if (cur.key == UNIQ_EMPTY || ++probes > mask) break;
                s = (s + 1) & mask;
            }
        }
        answers[i] = r;
    }
}
// first_seen[i] = answers[slot[i]] (the answers arrive in row order)
__global__ __launch_bounds__(256) void uniq_gather_kernel(const uint64_t* __restrict__ answers, const uint32_t* __restrict__ slot, uint64_t n, uint64_t base,
                                                          uint64_t* first_seen, uint8_t* keep)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t r = answers[slot[i]];
        first_seen[i] = r;
        if (keep) keep[i] = r == base + i;
    }
}

// every slot = {EMPTY, EMPTY}: 16 bytes per thread and trip; also zeroes the overflow counter
__global__ __launch_bounds__(256) void uniq_clear_kernel(UniqSlot* t, uint64_t slots, uint32_t* overflow, const uint32_t* only_if = nullptr)
{
    if (only_if && *only_if == 0) return;
    typedef unsigned long long v2 __attribute__((ext_vector_type(2)));
    if (overflow && blockIdx.x == 0 && threadIdx.x == 0) *overflow = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += stride)
        *reinterpret_cast<v2*>(t + i) = v2{ UNIQ_EMPTY, UNIQ_EMPTY };
}

// ---- circkit_uniq_resolve_device in LDS-sized buckets (round 4) --------------------------------------------------------------
// The open-addressing table in HBM costs every key a scattered 16-byte read and one or two memory-side atomics (10M keys:
// clear 0.05 + insert 0.48 + lookup 0.22 ms).  A shard that is resolved in ONE call needs no table that outlives it: the keys
// are partitioned by their top bits into buckets of ~1300..2600 keys (a counting pass, two small scans, a scatter of
// {hash, local index} rows -- no global atomic anywhere: every workgroup owns a contiguous range of the keys and its own column
// of the count matrix), and each bucket is resolved by one workgroup in a 4096-slot table in LDS (64-bit LDS compare-and-swap
// on the key, LDS atomic min on the index).  What stays scattered is the answer: first_seen[i] and keep[i] go to the record's
// own place.  A bucket with more than BKT_MAX keys (all records equal, say) raises a flag and the whole shard takes the HBM
// table behind it -- same answers; its three kernels return at once otherwise.  Scratch: the table's own memory (rows, count
// matrix, bucket totals and bases all fit the slots circkit_uniq_reset sizes for n keys).
#ifndef CK_BKT_NW
#define CK_BKT_NW 512
#endif
constexpr uint32_t BKT_NW = CK_BKT_NW, BKT_SLOTS = 4096, BKT_MAX = 3072, BKT_KEYS = 2600, BKT_MAX_LOG2 = 13;      // (mean bucket 1300..2600 keys; 2200 -- twice the buckets at 10M keys -- measured 1 % slower)
__device__ __forceinline__ uint32_t bkt_of(uint64_t h, uint32_t log2b) { return (uint32_t)(h >> (64 - log2b)); }
// counts[w][b] = keys of workgroup w's range in bucket b; also zeroes the fallback flag and the table-overflow counter
__global__ __launch_bounds__(1024) void uniq_bkt_count_kernel(const uint64_t* __restrict__ hash, uint64_t n, uint64_t per, uint32_t log2b, uint32_t* counts,
                                                              uint32_t* flag, uint32_t* overflow)
{
    extern __shared__ uint32_t bkt_lds[];
    const uint32_t B = 1u << log2b;
    for (uint32_t b = threadIdx.x; b < B; b += 1024) bkt_lds[b] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { *flag = 0; *overflow = 0; }
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    for (uint64_t i = i0 + threadIdx.x; i < i1; i += 1024) atomicAdd(&bkt_lds[bkt_of(hash[i], log2b)], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += 1024) counts[(uint64_t)blockIdx.x * B + b] = bkt_lds[b];
}
// per bucket: exclusive prefix of the counts over the workgroups (in place), the bucket's total.  A workgroup takes 64 buckets,
// sixteen threads per bucket with a 32nd of the column each (one thread per bucket walking all 512 counts: 65 us of latency)
__global__ __launch_bounds__(1024) void uniq_bkt_colscan_kernel(uint32_t* counts, uint32_t B, uint32_t* tot)
{
    __shared__ uint32_t seg[16][64];
    constexpr uint32_t PER = BKT_NW / 16;
    const uint32_t bl = threadIdx.x & 63, sg = threadIdx.x >> 6, b = blockIdx.x * 64 + bl;      // (B is a multiple of 64)
    uint32_t sum = 0;
    for (uint32_t w = sg * PER; w < (sg + 1) * PER; ++w) sum += counts[(uint64_t)w * B + b];
    seg[sg][bl] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (uint32_t k = 0; k < sg; ++k) run += seg[k][bl];
    for (uint32_t w = sg * PER; w < (sg + 1) * PER; ++w) {
        const uint64_t k = (uint64_t)w * B + b;
        const uint32_t cnt = counts[k];
        counts[k] = run;
        run += cnt;
    }
    if (sg == 15) tot[b] = run;
}
// base[b] = first row of bucket b, base[B] = n (one workgroup; B <= 8192)
__global__ __launch_bounds__(1024) void uniq_bkt_basescan_kernel(const uint32_t* __restrict__ tot, uint32_t B, uint32_t* base)
{
    __shared__ uint32_t part[1024];
    const uint32_t items = (B + 1023) / 1024, b0 = threadIdx.x * items;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < items; ++k) if (b0 + k < B) sum += tot[b0 + k];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {                   // inclusive scan of the partial sums
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t k = 0; k < items; ++k) if (b0 + k < B) { base[b0 + k] = run; run += tot[b0 + k]; }
    if (threadIdx.x == 1023) base[B] = part[1023];
}
// rows[base[b] + ...] = {hash, local index}; a workgroup's keys of bucket b start at base[b] + its prefix in the count matrix
__global__ __launch_bounds__(1024) void uniq_bkt_scatter_kernel(const uint64_t* __restrict__ hash, uint64_t n, uint64_t per, uint32_t log2b,
                                                                const uint32_t* __restrict__ counts, const uint32_t* __restrict__ base, uint64_t* rows,
                                                                uint64_t* first_seen, uint64_t base_index)
{
    // (every record starts out as its own first occurrence, written here in record order -- coalesced; the resolve kernel then
    // only has scattered answers for the records that are NOT: half the batch in BASELINE config 3, next to none in real inputs)
    extern __shared__ uint32_t bkt_lds[];
    const uint32_t B = 1u << log2b;
    for (uint32_t b = threadIdx.x; b < B; b += 1024) bkt_lds[b] = base[b] + counts[(uint64_t)blockIdx.x * B + b];
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    typedef unsigned long long v2 __attribute__((ext_vector_type(2)));
    constexpr int U = 4;                                    // keys in flight per thread (one at a time the loop is a chain of load latencies)
    uint64_t i = i0 + threadIdx.x;
    for (; i + (U - 1) * 1024 < i1; i += U * 1024) {
        uint64_t h[U];
#pragma unroll
        for (int u = 0; u < U; ++u) h[u] = hash[i + u * 1024];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t pos = atomicAdd(&bkt_lds[bkt_of(h[u], log2b)], 1u);
            *reinterpret_cast<v2*>(rows + 2 * (uint64_t)pos) = v2{ h[u], i + u * 1024 };
            first_seen[i + u * 1024] = base_index + i + u * 1024;
        }
    }
    for (; i < i1; i += 1024) {
        const uint64_t h = hash[i];
        const uint32_t pos = atomicAdd(&bkt_lds[bkt_of(h, log2b)], 1u);
        *reinterpret_cast<v2*>(rows + 2 * (uint64_t)pos) = v2{ h, i };
        first_seen[i] = base_index + i;
    }
}
// one workgroup per bucket: smallest local index per key in an LDS table, then every row's answer to its record's place
#ifndef BKT_RESOLVE_T
#define BKT_RESOLVE_T 1024     // threads per bucket: 48 KB of LDS allow two workgroups per CU -- at 1024 threads that is every wave slot (256: 12 of 32)
#endif
__global__ __launch_bounds__(BKT_RESOLVE_T) void uniq_bkt_resolve_kernel(const uint64_t* __restrict__ rows, const uint32_t* __restrict__ base, uint64_t base_index,
                                                               uint64_t* first_seen, uint8_t* keep, uint32_t* flag)
{
    __shared__ unsigned long long keys[BKT_SLOTS];
    __shared__ uint32_t idx[BKT_SLOTS], special;
    const uint32_t r0 = base[blockIdx.x], cnt = base[blockIdx.x + 1] - r0;
    if (cnt == 0) return;
    if (cnt > BKT_MAX) { if (threadIdx.x == 0) atomicExch(flag, 1u); return; }      // the HBM table has this shard (every record of it)
    for (uint32_t s = threadIdx.x; s < BKT_SLOTS; s += BKT_RESOLVE_T) { keys[s] = UNIQ_EMPTY; idx[s] = 0xFFFFFFFFu; }
    if (threadIdx.x == 0) special = 0xFFFFFFFFu;
    __syncthreads();
    // a thread's rows stay in registers between the two passes (cnt <= BKT_MAX = 3 x 1024)
    constexpr uint32_t RPT = (BKT_MAX + BKT_RESOLVE_T - 1) / BKT_RESOLVE_T;
    uint64_t hk[RPT];
    uint32_t ik[RPT];
#pragma unroll
    for (uint32_t k = 0; k < RPT; ++k) {
        const uint32_t r = threadIdx.x + k * BKT_RESOLVE_T;
        hk[k] = 0; ik[k] = 0xFFFFFFFFu;
        if (r < cnt) { const UniqSlot row = uniq_peek(reinterpret_cast<const UniqSlot*>(rows) + r0 + r); hk[k] = row.key; ik[k] = (uint32_t)row.val; }
    }
#pragma unroll
    for (uint32_t k = 0; k < RPT; ++k) {
        if (ik[k] == 0xFFFFFFFFu) continue;
        const uint64_t h = hk[k];
        if (h == UNIQ_EMPTY) { atomicMin(&special, ik[k]); continue; }
        uint32_t s = (uint32_t)(h ^ (h >> 29)) & (BKT_SLOTS - 1);
        for (;;) {
            const unsigned long long old = atomicCAS(&keys[s], (unsigned long long)UNIQ_EMPTY, (unsigned long long)h);
            if (old == UNIQ_EMPTY || old == h) { atomicMin(&idx[s], ik[k]); break; }
            s = (s + 1) & (BKT_SLOTS - 1);
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < RPT; ++k) {
        if (ik[k] == 0xFFFFFFFFu) continue;
        const uint64_t h = hk[k];
        uint32_t m;
        if (h == UNIQ_EMPTY) m = special;
        else {
            uint32_t s = (uint32_t)(h ^ (h >> 29)) & (BKT_SLOTS - 1);
            while (keys[s] != h) s = (s + 1) & (BKT_SLOTS - 1);
            m = idx[s];
        }
        if (m != ik[k]) first_seen[ik[k]] = base_index + m;     // (the one scattered store, for records that repeat an earlier one; keep[] follows, coalesced)
    }
}
// keep[i] = 1 iff record i is the first with its hash
__global__ __launch_bounds__(256) void uniq_keep_kernel(const uint64_t* __restrict__ first_seen, uint64_t n, uint64_t base, uint8_t* keep, const uint32_t* skip_if)
{
    if (skip_if && *skip_if != 0) return;                   // (the fallback writes its own)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) keep[i] = first_seen[i] == base + i;
}

// ------------------------------------------------------------------------------------------------
// LDS stages of a batch behind the rescue pass (+ 276 dwords per workgroup: deferral counter, decode table, N patch table).
// A 2-bit record needs its ONE stored strand, n / 16 + 2 dwords (+ n / 32 for the candidate bitmask only if the minimal key
// ties); with a few N the strand + n / 32 of N bitmask; 4-bit and byte mode two strands + the bitmask.  In every stage a
// record goes to ONE wave if it fits that wave's slice, to the workgroup's waves as a TEAM if it is 2-bit material and fits
// their slices together, else to wave 0 ALONE with the workgroup's whole LDS (canon_core.h team_pass, canon_team_kernel).
//   A: 4 waves x 5 KiB (7 workgroups per CU)      one wave: 2-bit up to ~20.4 kb (all of BASELINE config 4), few N ~13.6 kb,
//                                                 4-bit ~8.2 kb (one stored strand since round 3); team: 2-bit ~81 kb, few N ~54 kb, 4-bit ~41 kb; alone: 4-bit ~33 kb
//   C: 4 waves x 9.7 KiB (4 workgroups per CU)    team: 2-bit up to ~160 kb, few N ~106 kb, 4-bit ~80 kb; alone: 4-bit ~64 kb, bytes ~17 kb
//   team stage: 16 waves x 157 KiB (the CU)       team: 2-bit up to ~640 kb, few N ~420 kb, 4-bit ~320 kb; alone: 4-bit ~250 kb, bytes ~70 kb
//   beyond: canon_global_kernel, the same over slices of a global-memory scratch (one more launch of every batch)
// (One-wave slices of 7.4 / 13 / 39 / 157 KiB -- stages of their own in earlier versions -- remain for single-record calls.)
#ifndef CK_RESCUE_BPC
#define CK_RESCUE_BPC 8      // workgroups per CU of the rescue pass's persistent grid
#endif
#ifndef CK_TIER_BPC
#define CK_TIER_BPC 128    // workgroups launched per CU by the LDS tiers at most (they walk the list segments); 128 = one per segment
#endif
#ifndef CK_TIER_BPC_IDLE
#define CK_TIER_BPC_IDLE 8 // ...when the previous batch left them (next to) nothing to do
#endif
// A 2-bit record is admitted by its one stored strand alone, (n + 15) / 16 + 2 dwords (the candidate bitmask is only needed
// on a tie of the minimal key; a record that ties without room for it moves on).  Tier A: 5 KiB per wave = records up to
// 20.4 kb -- all of BASELINE config 4 in the four-wave tier, 7 workgroups = 28 waves per CU by LDS and by VGPRs alike (6 by SGPRs for the mixed-length kernels, see there)
// (measured on config 4, one box: A = 4 KiB + B1 = 7.4 KiB with the bitmask counted in 2.41 ms, without it 2.30,
// A = 2.5 KiB + B1 = 5 KiB 2.41, A = 5 KiB 2.20, A = 5.5 KiB 2.47).
#ifndef CK_TIER_A
#define CK_TIER_A 1280
#endif
#ifndef CK_TIER_B1
#define CK_TIER_B1 1900     // (one-wave slice sizes of single-record calls: launch_single)
#endif
#ifndef CK_TIER_B2
#define CK_TIER_B2 3324
#endif
constexpr int N_TIERS = 5;
constexpr uint32_t TIER_DW[N_TIERS] = { CK_TIER_A, CK_TIER_B1, CK_TIER_B2, 9980, CK_LUT_STRIDE == 1 ? 40188u : 31996u };     // + TIER_EXTRA_DW per workgroup
constexpr uint32_t TIER_EXTRA_DW = 4 + ck::FAST_LUT_DW + ck::FAST_LUTN_DW;        // counter, decode table, N patch table
constexpr uint32_t TIER_D_DW = TIER_DW[N_TIERS - 1];
// A batch's LDS stages: A (four waves x 5 KiB), C (four waves x 9.7 KiB) and the team stage (sixteen waves, the whole LDS).
// In each of them a record goes to one wave if it fits that wave's slice, to all waves of the workgroup as a team if it is
// 2-bit material and fits their slices together, else to wave 0 alone with the workgroup's whole LDS (canon_core.h
// team_pass, canon_team_kernel) -- which is what the one-wave tiers B1 / B2 / C / D of 7.4 / 13 / 39 / 157 KiB did with a
// launch each; those sizes remain for single-record calls (launch_single).
constexpr int BATCH_TIERS = 3;
constexpr uint32_t BATCH_SLICE_DW[BATCH_TIERS - 1] = { CK_TIER_A, 9980 / 4 };
constexpr uint32_t TEAM_SLICE_DW = TIER_D_DW / 16;
constexpr int N_CU = 256;

}  // namespace

// ------------------------------------------------------------------------------------------------
// ctx
// ------------------------------------------------------------------------------------------------
struct circkit_ctx {
    int device = -1;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_order = nullptr;       // circkit_ctx_set_stream: the new stream waits for what the ctx queued on the old one
    // host-buffer batches in parts (host_batch): copy-in and copy-out streams of the ctx's own and one event pair per part, so
    // that part k + 1 arrives and part k - 1 leaves while part k computes (PCIe is full duplex)
    static constexpr int MAX_PARTS = 32;
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[MAX_PARTS] = {}, ev_done[MAX_PARTS] = {}, ev_head = nullptr;
    bool timed = false;
    std::string err;
    uint8_t* d_comp = nullptr;
    uint32_t* d_counters = nullptr;      // [3] unprocessed records; [4] uniq table overflow
    // segmented deferral lists: streaming kernel -> rescue pass / mixed kernel -> stage A -> stage C -> team stage -> global
    // stage (one segment per producing workgroup)
    uint32_t* d_lists[N_TIERS + 1] = {}; // [0] streaming kernel -> rescue pass, [i + 1] input list of tier i (output of the stage before)
    uint32_t* d_seg_counts = nullptr;    // [(N_TIERS + 1) * seg_alloc]
    uint64_t list_cap = 0, seg_alloc = 0;
    // host-batch staging (grow only)
    uint8_t *d_in = nullptr, *d_out = nullptr, *d_strand = nullptr;
    uint64_t* d_off = nullptr; uint32_t* d_idx = nullptr; uint64_t* d_hash = nullptr;
    uint64_t cap_bytes = 0, cap_rec = 0;
    uint32_t* d_view = nullptr; uint64_t cap_view = 0;        // hash-only batches: strand + rotation of the records whose hash is not fused
    uint8_t* d_hashed = nullptr; uint64_t cap_hashed = 0;     // per record: hash already written by the streaming kernel
    // records beyond the last LDS tier run the same per-record code over slices of this global-memory scratch
    // (grow-only; the device API allocates the default on first use, the host API sizes it for the batch's longest record)
    uint32_t* d_gscratch = nullptr; uint64_t cap_gscratch = 0;    // bytes
    uint64_t gscratch_default = 256ull << 20;
    // the previous batch's mode (1 / 2 / 3, see stream_mode): only picks which build gets the full-size grid
    uint64_t* h_off = nullptr;           // page-locked staging of a host batch's offsets and per-record outputs (host_batch): h_off_cap x (8 + 8 + 4 + 1) bytes
    uint64_t h_off_cap = 0;
    volatile uint32_t* h_mode = nullptr; // pinned host word the rescue kernel writes the batch's mode to (d_mode = its device address)
    uint32_t* d_mode = nullptr;
    // uniq table
    UniqSlot* d_table = nullptr;         // [uniq_mask + 2]
    uint64_t uniq_mask = 0, uniq_count = 0;   // slots - 1; upper bound of the keys folded in so far
    uint32_t mode_seen = 0;                   // launch_canon: the mode word as the previous device batch's launch read it (MODE_GUESS)
    bool uniq_local = false;                  // the table holds circkit_uniq_resolve_device's split local values
    bool uniq_lost = false;                   // a rehash failed half-way: the stream's earlier batches are gone -- every uniq call fails until circkit_uniq_reset
};

namespace {

int fail(circkit_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define CK_HIP(c, call)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(c, e_ == hipErrorOutOfMemory ? CIRCKIT_ERR_OOM : CIRCKIT_ERR_HIP,           \
                        "%s failed: %s", #call, hipGetErrorString(e_));                             \
    } while (0)

int ensure_lists(circkit_ctx* c, uint64_t entries, uint64_t segs)
{
    if (entries > c->list_cap) {
        c->list_cap = 0;
        for (int i = 0; i < N_TIERS + 1; ++i) {
            if (c->d_lists[i]) { (void)hipFree(c->d_lists[i]); c->d_lists[i] = nullptr; }
            CK_HIP(c, hipMalloc(&c->d_lists[i], entries * sizeof(uint32_t)));
        }
        c->list_cap = entries;
    }
    if (segs > c->seg_alloc) {
        if (c->d_seg_counts) { (void)hipFree(c->d_seg_counts); c->d_seg_counts = nullptr; }
        c->seg_alloc = 0;
        CK_HIP(c, hipMalloc(&c->d_seg_counts, (N_TIERS + 1) * segs * sizeof(uint32_t)));
        c->seg_alloc = segs;
    }
    return CIRCKIT_OK;
}

int ensure_gscratch(circkit_ctx* c, uint64_t bytes)
{
    if (bytes <= c->cap_gscratch) return CIRCKIT_OK;
    if (c->d_gscratch) { (void)hipFree(c->d_gscratch); c->d_gscratch = nullptr; c->cap_gscratch = 0; }
    CK_HIP(c, hipMalloc(&c->d_gscratch, bytes));
    c->cap_gscratch = bytes;
    return CIRCKIT_OK;
}

// LDS dwords (or scratch dwords) the byte-wide mode needs for a record of n symbols: the largest of the three modes
// (ck::need_dw<8>), with 64 dwords of slack
inline uint64_t worst_case_dw(uint64_t n) { return 2 * ((n + 3) / 4 + 2) + (n + 31) / 32 + 1 + 64; }
constexpr uint64_t GSLICE1_DW = 1ull << 20;     // stage 1 of the global-scratch pass: up to 64 waves x 4 MiB

// Enqueues one batch: every kernel, nothing else -- no host-side follow-up is needed once the stream has run them
// (records beyond the on-chip tiers included: the last two stages take them in a global-memory scratch).
// host_mode: 1 / 2 / 3 when the host has seen the offsets (stream_mode), 0 = the device decides.
int launch_canon(circkit_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n, uint8_t* d_out,
                 uint32_t* d_idx, uint8_t* d_strand, uint64_t* d_hash, uint32_t flags, uint32_t host_mode = 0, bool keep_status = false)
{
#ifdef CK_FORCE_HOST_MODE
    if (!host_mode) host_mode = CK_FORCE_HOST_MODE;      // experiment: no count kernel, no second build
#endif
    if (n >= (1ull << 30)) return fail(c, CIRCKIT_ERR_INVALID_ARG, "n_records must be < 2^30");      // (list entries: 30 bits of record index)
    CK_HIP(c, hipSetDevice(c->device));
    if (n == 0) { c->timed = false; return CIRCKIT_OK; }
    // MODE_GUESS.  Every mode computes the same results -- the mode only says which builds are the fast ones for the batch's
    // lengths and alphabet.  A device batch whose mode the host does not know used to launch every build (the ones the
    // device-side decision does not name return at once): eleven idle kernels per batch, ~5 us each, 3-4 % of a 1.8 ms
    // batch.  Now the mode the last batches REPORTED (the pinned word the device writes, read here without waiting) is taken as
    // this batch's mode, exactly as if the host had seen the offsets: only that mode's kernels are launched.  The count
    // kernel still runs and the kernels report ITS answer (bit 31 of the mode argument), so the guess follows the data: a
    // batch of another kind runs correctly on the wrong builds (as slow as round 2's general path at worst), the launch after
    // the one that sees the new report decides on the device again, the one after that guesses the new mode.
    // CIRCKIT_NO_MODE_GUESS=1: every batch decides on the device, as before.
    const bool device_decides = host_mode == 0;
    uint32_t guess_flag = 0;
    if (device_decides) {
        static const bool no_guess = getenv("CIRCKIT_NO_MODE_GUESS") != nullptr;
        const uint32_t seen = *c->h_mode & 15u;
        if (!no_guess && (seen & 3u) && seen == c->mode_seen) { host_mode = seen; guess_flag = 0x80000000u; }
        c->mode_seen = seen;
        static const bool dbg = getenv("CIRCKIT_DEBUG_MODE_GUESS") != nullptr;
        if (dbg) fprintf(stderr, "[mode guess] n=%llu report=%u -> %s\n", (unsigned long long)n, seen, guess_flag ? "guessed" : "decided on the device");
    }
    const uint32_t kmode = host_mode | guess_flag;          // what the kernels get
    // launch geometry: G virtual workgroups for the streaming kernel, the rescue pass and tier A (segment b of a list
    // belongs to workgroup b); stage C and the team stage take several segments per workgroup each
    const bool aux = d_idx || d_strand || (flags & ck::CK_FLAG_FWD_ONLY);
    const uint64_t per_step = aux ? StreamCAux::GROUP : StreamC::GROUP;    // records a workgroup takes per iteration
    const uint64_t blocks = (n + per_step - 1) / per_step;
    unsigned G = (unsigned)(blocks < (uint64_t)N_CU * CK_FAST_BPC ? blocks : (uint64_t)N_CU * CK_FAST_BPC);
    // A small batch gets more segments than the streaming kernel has workgroups with work (the others publish an empty
    // segment and leave): the stages that walk all records of a mode-3 batch deal them out all_cap at a time, and a list
    // segment is one workgroup of every stage behind -- 2000 records of 250 kb in 125 segments of 16 left half the chip idle.
    const unsigned G_small = (unsigned)(n < (uint64_t)N_CU * 16 ? n : (uint64_t)N_CU * 16);
    if (G < G_small) G = G_small;
    uint32_t cap = (uint32_t)(per_step * ((blocks + G - 1) / G));     // records one workgroup can see = entries of a list segment
    uint32_t all_cap = (uint32_t)((n + G - 1) / G);
    // the walking stages' segments taper off at the end (canon_core.h seg_records): three generations of G/16 segments with
    // a half, a quarter, an eighth of the records of the ones before
    uint32_t taper_seg0 = 0, taper_log2 = 0;
#ifndef CK_NO_TAPER
#ifndef CK_TAPER_DIV
#define CK_TAPER_DIV 16
#endif
    if (G >= 4096 && all_cap >= (2u << CK_TAPER_GENS)) {
        while ((2u << taper_log2) <= G / CK_TAPER_DIV) ++taper_log2;
        const uint64_t gen = 1ull << taper_log2;
        taper_seg0 = G - (uint32_t)(CK_TAPER_GENS * gen);
        auto tapered = [&](uint32_t A) { uint64_t t = 0; for (int j = 1; j <= CK_TAPER_GENS; ++j) t += A >> j; return t; };
        while ((uint64_t)taper_seg0 * all_cap + gen * tapered(all_cap) < n) ++all_cap;
        if (cap < all_cap) cap = all_cap;
    }
#endif
    // + 1024 segments of slack: a stage that merges k segments per workgroup addresses up to k - 1 segments past the end
    int rc = ensure_lists(c, (uint64_t)G * cap + 1024ull * cap, G < 256u ? 256u : G);
    if (rc) return rc;
    if (!c->d_gscratch && (rc = ensure_gscratch(c, c->gscratch_default))) return rc;
    uint32_t* view = nullptr;
    if (d_hash && !d_out) {
        // hash-only (uniq without --canonicalize, src/uniq.rs:55-60): no canonical bytes are written anywhere.  A record
        // whose hash is not fused leaves its canonical form as a view of the input -- strand + rotation, one u32 in a ctx
        // array sized by n (known here: no look at the device, no synchronisation) -- and the xxh3 pass hashes that view.
        if (n > c->cap_view) {
            if (c->d_view) { (void)hipFree(c->d_view); c->d_view = nullptr; c->cap_view = 0; }
            CK_HIP(c, hipMalloc(&c->d_view, (n + n / 8 + 64) * sizeof(uint32_t)));
            c->cap_view = n + n / 8 + 64;
        }
        view = c->d_view;
    }
    if (d_hash) {
        if (n > c->cap_hashed) {
            if (c->d_hashed) { (void)hipFree(c->d_hashed); c->d_hashed = nullptr; c->cap_hashed = 0; }
            CK_HIP(c, hipMalloc(&c->d_hashed, n + n / 8 + 64));
            c->cap_hashed = n + n / 8 + 64;
        }
        CK_HIP(c, hipMemsetAsync(c->d_hashed, 0, n, c->stream));
    }
    // d_counters: [0] records the last LDS tier passed on, [2] arrival ticket of the global-scratch kernel (leaves it zero),
    // [3] records nothing could take, [4] uniq table overflow,
    // [5] the batch's mode (device-side decision), [8..10] the count kernel's counters and ticket (it leaves them zero).
    // A device-side decision zeroes [0] and [3] in its count kernel; the host-side one with a memset.
    // (keep_status: a later part of a host batch -- counters[3], the records nothing could take, adds up over the parts)
    if (!device_decides) CK_HIP(c, hipMemsetAsync(c->d_counters, 0, (keep_status ? 3 : 4) * sizeof(uint32_t), c->stream));
    CK_HIP(c, hipEventRecord(c->ev0, c->stream));
    ck::CanonArgs a{};
    a.bytes = d_bytes; a.offsets = d_offsets; a.n_records = n;
    a.out_bytes = d_out; a.out_index = d_idx; a.out_strand = d_strand; a.out_hash = d_hash; a.hashed = c->d_hashed; a.out_view = view;
    a.comp_lut = c->d_comp; a.status = c->d_counters + 3; a.flags = flags;
    // streaming kernel over every record; what it cannot take goes down the LDS tiers
    a.list = nullptr; a.list_count = nullptr; a.all_seg_cap = all_cap; a.taper_seg0 = taper_seg0; a.taper_log2 = taper_log2;
    a.defer_list = c->d_lists[0]; a.defer_count = c->d_seg_counts; a.out_seg_cap = cap;
    a.slice_dw = 0;
    const uint32_t* counts = c->d_counters + 5;     // the mode word
    {
        // three builds of the streaming kernel: canonical bytes only (the headline), + fused XXH3 (uniq), and the
        // general one for callers that also want the rotation index / strand or the forward-only variant (lmsr);
        // each for ROWS = 1 and ROWS = 2.  The host's answer launches exactly one of the two; a device-side decision
        // launches both, full-size where the previous batch's mode says it will run.
        if (device_decides) {
            const uint64_t ns = count_samples(n);
            const unsigned cgrid = (unsigned)((ns + 1023) / 1024 < 128 ? (ns + 1023) / 1024 : 128);
            hipLaunchKernelGGL(stream_count_kernel, dim3(cgrid), dim3(1024), 0, c->stream, d_bytes, d_offsets, n, c->d_counters + 8, c->d_counters + 5, c->d_counters, (d_hash && !aux) ? 1u : 0u);
        }
        // the mode of whichever earlier batch last reported: a hint for the grid sizes, nothing else
        const uint32_t seen = *c->h_mode;
        const uint32_t expect = host_mode ? host_mode & 3 : ((seen & 3) ? seen & 3 : 1u);
        const unsigned small = G < 2u * N_CU ? G : 2u * N_CU;
        const dim3 block(StreamC::WPB * 64), block_aux(StreamCAux::WPB * 64), block_h(StreamCH::WPB * 64), block_hp(StreamCHP::WPB * 64), block_a(StreamCA::WPB * 64), block_2(StreamC2::WPB * 64);
        // (rows, alphabet) builds: ROWS = 1 lean, ROWS = 1 with the 4-bit routine, ROWS = 2.  The host's answer launches
        // exactly one; a device-side decision launches all three, full-size where the previous batch's mode says it will run
        const uint32_t expect_alpha = host_mode ? host_mode & MODE_ALPHA : seen & MODE_ALPHA;
        for (uint32_t rows = 1; rows <= 2; ++rows) {
            if (host_mode && (host_mode & 3) != rows) continue;
            for (uint32_t alpha = 0; alpha <= (rows == 1 && !aux ? 1u : 0u); ++alpha) {
                if (host_mode && rows == 1 && !aux && ((host_mode & MODE_ALPHA) != 0) != (alpha != 0)) continue;
                const bool full = expect == rows && (rows != 1 || aux || (expect_alpha != 0) == (alpha != 0));
                const dim3 grid(full ? G : small);
#define CK_LAUNCH_STREAM(CFG, H, A, BLK, AL)                                                                                         \
                do {                                                                                                                \
                    if (full) hipLaunchKernelGGL((canon_stream_kernel<CFG, H, A, false, AL>), grid, BLK, 0, c->stream, a, counts, kmode, G); \
                    else hipLaunchKernelGGL((canon_stream_kernel<CFG, H, A, true, AL>), grid, BLK, 0, c->stream, a, counts, kmode, G);       \
                } while (0)
                if (rows == 1) {
                    if (aux) CK_LAUNCH_STREAM(StreamCAux, true, true, block_aux, false);
                    else if (d_hash) { if (alpha) CK_LAUNCH_STREAM(StreamCH, true, false, block_h, true); else CK_LAUNCH_STREAM(StreamCHP, true, false, block_hp, false); }
                    else if (alpha) CK_LAUNCH_STREAM(StreamCA, false, false, block_a, true);
                    else {
                        // bytes only, pure ACGT: one record per wave, or -- MODE_SHORT -- two; the host's answer launches one of the builds,
                        // a device-side decision both (the one the previous batch did not use with the small grid)
                        const bool expect_short = ((host_mode ? host_mode : seen) & MODE_SHORT) != 0;
                        for (int sh = 0; sh < 2; ++sh) {
                            if (host_mode && (sh != 0) != expect_short) continue;
                            const bool full_sh = full && (sh != 0) == expect_short;
                            const dim3 grid_sh(full_sh ? G : small);
                            if (sh) { if (full_sh) hipLaunchKernelGGL((canon_stream_kernel<StreamCHP, false, false, false, false>), grid_sh, block_hp, 0, c->stream, a, counts, kmode, G);
                                      else hipLaunchKernelGGL((canon_stream_kernel<StreamCHP, false, false, true, false>), grid_sh, block_hp, 0, c->stream, a, counts, kmode, G); }
                            else { if (full_sh) hipLaunchKernelGGL((canon_stream_kernel<StreamC, false, false, false, false>), grid_sh, block, 0, c->stream, a, counts, kmode, G);
                                   else hipLaunchKernelGGL((canon_stream_kernel<StreamC, false, false, true, false>), grid_sh, block, 0, c->stream, a, counts, kmode, G); }
                        }
                    }
                } else {
                    if (aux) CK_LAUNCH_STREAM(StreamCAux2, true, true, block_aux, false);
                    else if (d_hash) CK_LAUNCH_STREAM(StreamC2, true, false, block_2, false);
                    else CK_LAUNCH_STREAM(StreamC2, false, false, block_2, false);
                }
#undef CK_LAUNCH_STREAM
            }
        }
    }
    unsigned nseg = G;              // segments / capacity of the list the next stage consumes
    uint32_t seg_cap = cap;
    {
        // rescue pass over the streaming kernel's leftovers: segment for segment into the next list
        const unsigned grid = nseg < (unsigned)N_CU * CK_RESCUE_BPC ? nseg : (unsigned)N_CU * CK_RESCUE_BPC;
        a.list = c->d_lists[0]; a.list_count = c->d_seg_counts;
        a.in_nseg = nseg; a.in_seg_cap = seg_cap; a.segs_per_block = 1;
        a.defer_list = c->d_lists[1]; a.defer_count = c->d_seg_counts + c->seg_alloc; a.out_seg_cap = seg_cap;
#ifdef CK_NO_PINNED_MODE
        uint32_t* mode_out = c->d_counters + 6;
#else
        uint32_t* mode_out = c->d_mode;               // straight into pinned host memory: no copy-back, no event
#endif
        // both alphabets' builds unless the host has decided; the one the mode does not name returns at once
        const bool mixed_has_it = host_mode && !aux && (host_mode & 3) == 3;
        const bool lean = (!host_mode || !(host_mode & MODE_ALPHA)) && !mixed_has_it, alpha = (!host_mode || (host_mode & MODE_ALPHA)) && !mixed_has_it;
        if (aux) hipLaunchKernelGGL((canon_rescue_kernel<true, true, false>), dim3(grid), dim3(256), 0, c->stream, a, counts, kmode, mode_out, c->d_counters + 1);
        else if (d_hash) {
            if (lean) hipLaunchKernelGGL((canon_rescue_kernel<true, false, false>), dim3(grid), dim3(256), 0, c->stream, a, counts, kmode, mode_out, c->d_counters + 1);
            if (alpha) hipLaunchKernelGGL((canon_rescue_kernel<true, false, true>), dim3(grid), dim3(256), 0, c->stream, a, counts, kmode, mode_out, c->d_counters + 1);
        } else {
            if (lean) hipLaunchKernelGGL((canon_rescue_kernel<false, false, false>), dim3(grid), dim3(256), 0, c->stream, a, counts, kmode, mode_out, c->d_counters + 1);
            if (alpha) hipLaunchKernelGGL((canon_rescue_kernel<false, false, true>), dim3(grid), dim3(256), 0, c->stream, a, counts, kmode, mode_out, c->d_counters + 1);
        }
    }
    if (!aux && (!host_mode || (host_mode & 3) == 3)) {
        // mode 3: canon_mixed_kernel takes every record (the rescue pass above stood out); same lists, same segments.  Both
        // alphabets' builds unless the host has decided; one workgroup per segment for the build the previous batch used,
        // a small walking grid for the other (it returns at once unless the expectation was wrong).
        const uint32_t seen = *c->h_mode;
        const unsigned walking = nseg < (unsigned)N_CU * CK_RESCUE_BPC ? nseg : (unsigned)N_CU * CK_RESCUE_BPC;
#ifdef CK_NO_PINNED_MODE
        uint32_t* mode_out = c->d_counters + 6;
#else
        uint32_t* mode_out = c->d_mode;
#endif
        for (uint32_t nm = 0; nm < 2; ++nm) {
            if (host_mode && ((host_mode & MODE_ALPHA) != 0) != (nm != 0)) continue;
            const bool expected = host_mode ? true : ((seen & 3) == 3 && ((seen & MODE_ALPHA) != 0) == (nm != 0));
            // N builds: a resident grid (five workgroups per CU) fed by ticket; pure builds: one workgroup per segment
            const unsigned resident = (unsigned)N_CU * (20u / (unsigned)mixed_wpb(true, false));      // (20 waves per CU by LDS)
#ifdef CK_MIXED_N_STATIC
            const unsigned full = nseg;                    // EXPERIMENT: the N builds on the static mapping too
#else
            const unsigned full = nm ? (nseg < resident ? nseg : resident) : nseg;
#endif
            const unsigned grid = expected ? full : walking;
#ifdef CK_MIXED_N_STATIC
            uint32_t* ticket = nullptr;
#else
            uint32_t* ticket = nm ? c->d_counters + 16 : nullptr;   // [16], [17]: self-zeroing
#endif
            // the N build keeps one N bit per symbol (and lean_resolve_n's candidates) behind the strand: half as much again, so
            // that a 20 kb record of config 4 fits a slice (n / 16 + n / 32 + 24 dwords) -- five workgroups = 20 waves per CU
            a.slice_dw = nm ? CK_MIXED_N_SLICE : TIER_DW[0];
            const int mwpb = nm ? mixed_wpb(true, d_hash != nullptr) : (d_hash ? mixed_wpb(false, true) : mixed_wpb(false, false));
            const size_t shmem = ((size_t)mwpb * a.slice_dw + 4 + ck::FAST_LUT_DW + ck::LEAN_LUTN_DW + (d_hash ? ck::LEAN_HASH_TABLE_DW : 0)) * 4;
            if (d_hash) {
                if (nm) hipLaunchKernelGGL(canon_mixed_nh_kernel, dim3(grid), dim3(64 * mwpb), shmem, c->stream, a, counts, kmode, mode_out, c->d_counters + 1, ticket);
                else hipLaunchKernelGGL(canon_mixed_h_kernel, dim3(grid), dim3(64 * mwpb), shmem, c->stream, a, counts, kmode, mode_out, c->d_counters + 1, ticket);
            } else {
                if (nm) hipLaunchKernelGGL(canon_mixed_n_kernel, dim3(grid), dim3(64 * mwpb), shmem, c->stream, a, counts, kmode, mode_out, c->d_counters + 1, ticket);
                else hipLaunchKernelGGL(canon_mixed_kernel, dim3(grid), dim3(64 * mwpb), shmem, c->stream, a, counts, kmode, mode_out, c->d_counters + 1, ticket);
            }
        }
    }
#ifdef CK_DEBUG_DUMP
    {   // experiment builds only (tools/build_variant.sh -DCK_DEBUG_DUMP): what the front kernels handed to stage A
        (void)hipStreamSynchronize(c->stream);
        std::vector<uint32_t> cnt(nseg);
        (void)hipMemcpy(cnt.data(), c->d_seg_counts + c->seg_alloc, nseg * 4, hipMemcpyDeviceToHost);
        uint64_t total = 0; unsigned shown = 0;
        if (getenv("CK_DUMP_HIST")) {       // length histogram of the deferred records (powers of two), flagged / not
            std::vector<uint32_t> ents((size_t)nseg * seg_cap);
            (void)hipMemcpy(ents.data(), c->d_lists[1], ents.size() * 4, hipMemcpyDeviceToHost);
            std::vector<uint64_t> offs(n + 1);
            (void)hipMemcpy(offs.data(), d_offsets, (n + 1) * 8, hipMemcpyDeviceToHost);
            uint64_t hist[2][32] = {}, bytes[2] = {};
            for (unsigned sg = 0; sg < nseg; ++sg)
                for (uint32_t k = 0; k < cnt[sg]; ++k) {
                    const uint32_t e = ents[(size_t)sg * seg_cap + k]; const uint64_t len = offs[(e & ck::ENTRY_REC) + 1] - offs[e & ck::ENTRY_REC];
                    hist[e >> 31][63 - __builtin_clzll(len | 1)]++; bytes[e >> 31] += len;
                }
            for (int f = 0; f < 2; ++f) { fprintf(stderr, "[dump] flag %d (%llu bytes):", f, (unsigned long long)bytes[f]); for (int b = 0; b < 32; ++b) if (hist[f][b]) fprintf(stderr, " 2^%d:%llu", b, (unsigned long long)hist[f][b]); fprintf(stderr, "\n"); }
        }
        for (unsigned sg = 0; sg < nseg; ++sg) {
            total += cnt[sg];
            for (uint32_t k = 0; k < cnt[sg] && shown < 12; ++k, ++shown) {
                uint32_t e = 0; uint64_t o[2];
                (void)hipMemcpy(&e, c->d_lists[1] + (uint64_t)sg * seg_cap + k, 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(o, d_offsets + (e & ck::ENTRY_REC), 16, hipMemcpyDeviceToHost);
                fprintf(stderr, "[dump] seg %u entry %u: rec %u flag %u len %llu\n", sg, k, e & ck::ENTRY_REC, e >> 31, (unsigned long long)(o[1] - o[0]));
            }
        }
        fprintf(stderr, "[dump] stage A input: %llu records in %u segments\n", (unsigned long long)total, nseg);
    }
#endif
    const bool tiers_idle = c->h_mode[1] == 1;       // the previous batch's tiers found (next to) nothing: small grids for this one
    // (What the idle tiers cost live, measured by not launching them: 82 us of the 3.70 ms headline step, 116 us of uniq's
    // 5.26; with the parallel look at the counts in canon_kernel ~40 us remain -- five dependent launches.)
    for (int t = 0; t < BATCH_TIERS; ++t) {
        const bool last = t == BATCH_TIERS - 1;
        // stage C merges four segments per workgroup when there are plenty (its workgroups are four to a CU); a small batch
        // keeps one segment per workgroup -- 4000 records of 120 kb are 250 segments: 63 workgroups would leave the chip idle
        const unsigned spb = t == 0 ? 1 : (last ? (nseg + N_CU - 1) / N_CU : (nseg >= 16u * N_CU ? 4 : 1));
        const unsigned grid = (nseg + spb - 1) / spb;
        a.list = c->d_lists[t + 1]; a.list_count = c->d_seg_counts + (uint64_t)(t + 1) * c->seg_alloc;
        a.in_nseg = nseg; a.in_seg_cap = seg_cap; a.segs_per_block = spb;
        // the last tier's leftovers go into the first list (long consumed by now): the global-scratch stages pick them up
        a.defer_list = last ? c->d_lists[0] : c->d_lists[t + 2];
        a.defer_count = last ? c->d_seg_counts : c->d_seg_counts + (uint64_t)(t + 2) * c->seg_alloc;
        a.out_seg_cap = spb * seg_cap;
        a.slice_dw = last ? TEAM_SLICE_DW : BATCH_SLICE_DW[t];
        // `grid` virtual workgroups; launched: a few times what is resident at once (dispatch order balances the rest)
        const unsigned bpc = tiers_idle ? CK_TIER_BPC_IDLE : CK_TIER_BPC;
        const unsigned launched = grid < (unsigned)N_CU * bpc ? grid : (unsigned)N_CU * bpc;
        if (last) hipLaunchKernelGGL(canon_team_kernel, dim3(launched), dim3(TEAM_WAVES * 64), (TEAM_WAVES * TEAM_SLICE_DW + TIER_EXTRA_DW) * 4, c->stream, a, grid, c->d_counters);
        else if (tiers_idle) hipLaunchKernelGGL((canon_kernel<4, false>), dim3(launched), dim3(256), (4 * a.slice_dw + TIER_EXTRA_DW) * 4, c->stream, a, grid, (uint32_t*)nullptr);
        else hipLaunchKernelGGL((canon_kernel<4, true>), dim3(launched), dim3(256), (4 * a.slice_dw + TIER_EXTRA_DW) * 4, c->stream, a, grid, (uint32_t*)nullptr);
        nseg = grid;
        seg_cap = spb * seg_cap;
    }
    {
        // Records no LDS tier can hold (2-bit beyond ~640 kb, byte-mode beyond ~70 kb): the same code, one wave per
        // record, with a slice of the global scratch in place of the LDS slice -- one more launch.  Phase 1: up to 64
        // waves x 4 MiB (2-bit records up to ~16.7 Mb); phase 2 (the last workgroup to finish): one wave with the whole
        // scratch (256 MiB by default: 2-bit up to ~1 Gb -- ~700 Mb if the minimal key ties --, bytes up to ~126 MB;
        // circkit_ctx_set_long_record_scratch).
        // Beyond: counted in status[0].
        const uint64_t cap_dw = c->cap_gscratch / 4;
        const uint64_t slice1 = cap_dw < GSLICE1_DW ? cap_dw : GSLICE1_DW;
        const unsigned max_w1 = (unsigned)(cap_dw / slice1 < 64 ? cap_dw / slice1 : 64);
        const unsigned spb1 = (nseg + max_w1 - 1) / max_w1, w1 = (nseg + spb1 - 1) / spb1;
        a.list = c->d_lists[0]; a.list_count = c->d_seg_counts;
        a.in_nseg = nseg; a.in_seg_cap = seg_cap; a.segs_per_block = spb1;
        a.defer_list = c->d_lists[1]; a.defer_count = c->d_seg_counts + c->seg_alloc; a.out_seg_cap = spb1 * seg_cap;
        a.slice_dw = (uint32_t)slice1;
        ck::CanonArgs a2 = a;
        a2.list = c->d_lists[1]; a2.list_count = c->d_seg_counts + c->seg_alloc;
        a2.in_nseg = w1; a2.in_seg_cap = spb1 * seg_cap; a2.segs_per_block = w1;
        a2.defer_list = nullptr; a2.defer_count = nullptr; a2.out_seg_cap = 0;
        a2.slice_dw = (uint32_t)(cap_dw < 0xFFFFFFFFull ? cap_dw : 0xFFFFFFFFull);
        hipLaunchKernelGGL(canon_global_kernel, dim3(w1), dim3(TEAM_WAVES * 64), 0, c->stream, a, a2, c->d_gscratch, c->d_counters + 2, (const uint32_t*)c->d_counters,
                           (const uint32_t*)(c->d_counters + 1), c->d_mode + 1);
    }
    if (d_hash) {
        // the records whose hash was not fused: from their canonical bytes, or (hash-only batch) from the input through their views
        if (view) hipLaunchKernelGGL(xxh3_kernel, dim3(G < 2048u ? G : 2048u), dim3(256), 0, c->stream, d_bytes, d_offsets, n, d_hash, (const uint8_t*)c->d_hashed,
                                     (const uint32_t*)view, (const uint8_t*)c->d_comp);
        else hipLaunchKernelGGL(xxh3_kernel, dim3(G < 2048u ? G : 2048u), dim3(256), 0, c->stream, d_out, d_offsets, n, d_hash, (const uint8_t*)c->d_hashed,
                                (const uint32_t*)nullptr, (const uint8_t*)nullptr);
    }
    CK_HIP(c, hipEventRecord(c->ev1, c->stream));
    CK_HIP(c, hipGetLastError());
    c->timed = true;
    return CIRCKIT_OK;
}

// One record through the host API (the lib-crate mirrors circkit_lmsr_index / _lmsr / _canonicalize): a single launch
// of the general per-record kernel with the smallest slice that holds the record in any mode, instead of the batch
// pipeline's dozen launches.
int launch_single(circkit_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t len, uint8_t* d_out, uint32_t* d_idx,
                  uint8_t* d_strand, uint64_t* d_hash, uint32_t flags)
{
    ck::CanonArgs a{};
    a.bytes = d_bytes; a.offsets = d_offsets; a.n_records = 1;
    a.out_bytes = d_out; a.out_index = d_idx; a.out_strand = d_strand; a.out_hash = nullptr; a.hashed = nullptr;
    a.comp_lut = c->d_comp; a.status = c->d_counters + 3; a.flags = flags;
    CK_HIP(c, hipMemsetAsync(c->d_counters + 3, 0, sizeof(uint32_t), c->stream));
    const uint64_t need = worst_case_dw(len);
    int t = 0;
    while (t < N_TIERS && TIER_DW[t] < need) ++t;
    if (t < N_TIERS) {
        a.slice_dw = TIER_DW[t];
        hipLaunchKernelGGL(canon_kernel<1>, dim3(1), dim3(64), (TIER_DW[t] + TIER_EXTRA_DW) * 4, c->stream, a, 1u, (uint32_t*)nullptr);
    } else {
        int rc = ensure_gscratch(c, need * 4);
        if (rc) return rc;
        a.slice_dw = (uint32_t)(need < 0xFFFFFFFFull ? need : 0xFFFFFFFFull);
        hipLaunchKernelGGL(canon_global_one_kernel, dim3(1), dim3(64), 0, c->stream, a, c->d_gscratch);
    }
    if (d_hash) hipLaunchKernelGGL(xxh3_kernel, dim3(1), dim3(256), 0, c->stream, d_out, d_offsets, (uint64_t)1, d_hash, (const uint8_t*)nullptr, (const uint32_t*)nullptr, (const uint8_t*)nullptr);
    CK_HIP(c, hipGetLastError());
    c->timed = false;
    return CIRCKIT_OK;
}

template <class T>
int grow(circkit_ctx* c, T*& p, uint64_t want_elems)
{
    if (p) { (void)hipFree(p); p = nullptr; }
    CK_HIP(c, hipMalloc(&p, want_elems * sizeof(T)));
    return CIRCKIT_OK;
}

int ensure_staging(circkit_ctx* c, uint64_t bytes, uint64_t recs)
{
    if (bytes + 64 > c->cap_bytes) {
        const uint64_t nb = bytes + bytes / 8 + 4096;
        c->cap_bytes = 0;
        int rc;
        if ((rc = grow(c, c->d_in, nb))) return rc;
        if ((rc = grow(c, c->d_out, nb))) return rc;
        c->cap_bytes = nb;
    }
    if (recs + 1 > c->cap_rec) {
        const uint64_t nr = recs + recs / 8 + 1024;
        c->cap_rec = 0;
        int rc;
        if ((rc = grow(c, c->d_off, nr))) return rc;
        if ((rc = grow(c, c->d_idx, nr))) return rc;
        if ((rc = grow(c, c->d_strand, nr))) return rc;
        if ((rc = grow(c, c->d_hash, nr))) return rc;
        c->cap_rec = nr;
    }
    return CIRCKIT_OK;
}

int host_batch_enqueue(circkit_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t n, uint8_t* out,
                       uint32_t* idx, uint8_t* strand, uint64_t* hash, uint32_t flags);
// Every error exit of the host-buffer path drains the ctx's streams before it returns: copies of earlier parts into the caller's
// `out` and out of the ctx's page-locked staging may still be in flight when a later part fails, and the caller is free to
// release its buffers the moment the call is back.
int host_batch(circkit_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t n, uint8_t* out,
               uint32_t* idx, uint8_t* strand, uint64_t* hash, uint32_t flags)
{
    if (!c) return CIRCKIT_ERR_INVALID_ARG;
    const int rc = host_batch_enqueue(c, bytes, offsets, n, out, idx, strand, hash, flags);
    if (rc != CIRCKIT_OK) {
        if (c->s_in) (void)hipStreamSynchronize(c->s_in);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->s_out) (void)hipStreamSynchronize(c->s_out);
    }
    return rc;
}
int host_batch_enqueue(circkit_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t n, uint8_t* out,
                       uint32_t* idx, uint8_t* strand, uint64_t* hash, uint32_t flags)
{
    if (n && (!offsets || offsets[0] != 0)) return fail(c, CIRCKIT_ERR_INVALID_ARG, "offsets[0] must be 0");
    if (n == 0) return CIRCKIT_OK;
    const uint64_t total = offsets[n];
    if (total && !bytes) return fail(c, CIRCKIT_ERR_INVALID_ARG, "bytes is NULL");
    CK_HIP(c, hipSetDevice(c->device));
    int rc = ensure_staging(c, total, n);
    if (rc) return rc;
    const bool need_bytes = out || hash;
    static const bool dbg_t = getenv("CIRCKIT_DEBUG_TIMING") != nullptr;
    auto dbg_t0 = std::chrono::steady_clock::now();
    auto dbg_lap = [&](const char* what) {
        if (!dbg_t) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  host_batch %-18s %.3f ms\n", what, std::chrono::duration<double>(now - dbg_t0).count() * 1e3);
        dbg_t0 = now;
    };
    uint64_t two_word = 0, longer = 0, shorter = 0, max_len = 0;   // the host has the offsets: it picks the streaming kernel's build
    for (uint64_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return fail(c, CIRCKIT_ERR_INVALID_ARG, "offsets decrease at record %llu", (unsigned long long)i);
        const uint64_t len = offsets[i + 1] - offsets[i];
        two_word += len > ck::FAST_MAX_N && len <= ck::FAST2_MAX_N;
        longer += len > ck::FAST2_MAX_N;
        shorter += len <= SHORT_MAX_N;
        max_len = len > max_len ? len : max_len;
    }
    if (max_len >> 31) return fail(c, CIRCKIT_ERR_TOO_LONG, "a record of 2^31 symbols or more (cyclic positions are 32-bit)");
    if (n == 1 && (max_len + 15) / 16 + 2 <= TIER_DW[0]) {     // (a longer record is worth the batch pipeline: its team stages)
        if (total) CK_HIP(c, hipMemcpyAsync(c->d_in, bytes, total, hipMemcpyHostToDevice, c->stream));
        CK_HIP(c, hipMemcpyAsync(c->d_off, offsets, (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
        rc = launch_single(c, c->d_in, c->d_off, max_len, need_bytes ? c->d_out : nullptr, idx ? c->d_idx : nullptr,
                           strand ? c->d_strand : nullptr, hash ? c->d_hash : nullptr, flags);
        if (rc) return rc;
        if (out && total) CK_HIP(c, hipMemcpyAsync(out, c->d_out, total, hipMemcpyDeviceToHost, c->stream));
        if (idx) CK_HIP(c, hipMemcpyAsync(idx, c->d_idx, n * 4, hipMemcpyDeviceToHost, c->stream));
        if (strand) CK_HIP(c, hipMemcpyAsync(strand, c->d_strand, n, hipMemcpyDeviceToHost, c->stream));
        if (hash) CK_HIP(c, hipMemcpyAsync(hash, c->d_hash, n * 8, hipMemcpyDeviceToHost, c->stream));
        uint32_t unprocessed = 0;
        CK_HIP(c, hipMemcpyAsync(&unprocessed, c->d_counters + 3, 4, hipMemcpyDeviceToHost, c->stream));
        CK_HIP(c, hipStreamSynchronize(c->stream));
        if (unprocessed) return fail(c, CIRCKIT_ERR_TOO_LONG, "%u record(s) could not be processed", unprocessed);
        return CIRCKIT_OK;
    }
    // ...sizes the global scratch for the longest record, whatever mode it turns out to need
    if (worst_case_dw(max_len) > TIER_D_DW && (rc = ensure_gscratch(c, worst_case_dw(max_len) * 4))) return rc;
    // The batch goes through the device in PARTS of >= 16 MB (up to sixteen): part k + 1 is copied in while part k - 1 is copied
    // out on a stream of the ctx's own; with page-locked buffers (circkit_host_alloc) the two directions of the link work at
    // the same time.  One part = the round-2 behaviour: in, compute, out, one after the other
    // (1 GB of 1 kb records: 37.9 ms per call, 53 GB/s for both directions together).
    dbg_lap("offset scan");
    // (measured, tools/probe_host_batch.py: 64 MB in four parts 1.93 ms; 256 MB in eight 6.48 ms, in sixteen 6.84; 1 GB in eight
    // 25.2 ms, in sixteen 24.3, in thirty-two 24.4-24.8: up to eight parts of 16 MB and more, beyond 512 MB sixteen of 64 MB)
    int parts = (int)(total / (16ull << 20));
    parts = parts < 1 ? 1 : (parts > 8 ? 8 : parts);
    if (total > (512ull << 20)) { parts = (int)(total / (64ull << 20)); parts = parts > 16 ? 16 : parts; }
    parts = parts > circkit_ctx::MAX_PARTS ? circkit_ctx::MAX_PARTS : parts;
    if ((uint64_t)parts > n) parts = (int)n;
    if (getenv("CIRCKIT_HOST_BATCH_PARTS")) { const int p = atoi(getenv("CIRCKIT_HOST_BATCH_PARTS")); if (p >= 1 && p <= circkit_ctx::MAX_PARTS && (uint64_t)p <= n) parts = p; }
    // The copy-in rides the ctx stream itself, part k + 1 queued behind part k's kernels (a few hundred microseconds next to the
    // milliseconds of a part's copy); the copy-out has a stream of the ctx's own, at a priority of its own (the runtime keeps
    // a separate set of hardware queues per priority: no queue shared with the ctx stream or any default-priority stream).
    // With a plain copy-in and a plain copy-out stream -- round 3's first arrangement -- the two directions of a call
    // overlapped or took turns (1 GB: 24.8 or 38-40 ms) depending on how many streams the process had made before: one extra
    // stream ahead of the ctx was slow in 2 runs of 2 (tools/probe_stream_luck.py), and so were bench.py's `uniq` processes.
    // In this arrangement 130 fresh processes in the same probes were fast but one -- which had run straight behind a process
    // that had just released tens of GB of device memory: for some seconds after such an exit the copy engines are busy with
    // the driver's handling of that memory and the directions of every process's copies take turns
    // (tools/probe_after_big_process.sh); nothing a library can do about, bench.py's end-to-end leg waits it out.  Moving the bytes out by a copy kernel that stores across
    // the link instead of the DMA engine: 28 ms, and it took turns as well when it did.
    // CIRCKIT_HOST_BATCH_PLAIN_STREAMS=1: the first arrangement, for comparison.
    static const bool plain_streams = getenv("CIRCKIT_HOST_BATCH_PLAIN_STREAMS") != nullptr;
    const bool inline_h2d = !plain_streams;
    if (!c->s_out) {
        if (plain_streams) {
            CK_HIP(c, hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
            CK_HIP(c, hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
        } else {
            int prio_least = 0, prio_greatest = 0;
            CK_HIP(c, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
            CK_HIP(c, hipStreamCreateWithPriority(&c->s_out, hipStreamNonBlocking, prio_greatest));
        }
        CK_HIP(c, hipEventCreateWithFlags(&c->ev_head, hipEventDisableTiming));
        for (int k = 0; k < circkit_ctx::MAX_PARTS; ++k) {
            CK_HIP(c, hipEventCreateWithFlags(&c->ev_in[k], hipEventDisableTiming));
            CK_HIP(c, hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming));
        }
    }
    // part boundaries: record indices, cut where the payload passes k / parts of its bytes
    uint64_t cut[circkit_ctx::MAX_PARTS + 1];
    cut[0] = 0; cut[parts] = n;
    for (int k = 1; k < parts; ++k) {
        const uint64_t want = total / parts * k;
        uint64_t lo = cut[k - 1] + 1, hi = n - (parts - k);               // keep every part non-empty
        if (lo > hi) lo = hi;
        while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (offsets[mid] < want) lo = mid + 1; else hi = mid; }
        cut[k] = lo;
    }
    // whatever the caller has queued on the ctx stream comes first; the offsets go with the first part
    CK_HIP(c, hipEventRecord(c->ev_head, c->stream));
    if (!inline_h2d) CK_HIP(c, hipStreamWaitEvent(c->s_in, c->ev_head, 0));
    // (through a page-locked copy of them: from the caller's pageable array the runtime stages the copy and the call blocks
    // until it is through -- 0.9 ms of a 64 MB batch's 4.2)
    // The per-record outputs (hash, index, strand: 13 bytes a record) come back the same way: into page-locked staging part by
    // part, to the caller's arrays once at the end -- a copy-out to pageable memory would hold up every part behind it
    // (uniq on 1M x 1 kb: 38.6 ms per call with the hashes going straight to a pageable array, 23.7 without hashes).
    if (n + 1 > c->h_off_cap) {
        if (c->h_off) { (void)hipHostFree((void*)c->h_off); c->h_off = nullptr; c->h_off_cap = 0; }
        const uint64_t cap = n + 1 + (n + 1) / 8 + 1024;
        CK_HIP(c, hipHostMalloc((void**)&c->h_off, cap * (8 + 8 + 4 + 1), hipHostMallocDefault));
        c->h_off_cap = cap;
    }
    uint64_t* const st_hash = c->h_off + c->h_off_cap;
    uint32_t* const st_idx = (uint32_t*)(st_hash + c->h_off_cap);
    uint8_t* const st_strand = (uint8_t*)(st_idx + c->h_off_cap);
    memcpy(c->h_off, offsets, (n + 1) * 8);
    hipStream_t sin = inline_h2d ? c->stream : c->s_in;
    CK_HIP(c, hipMemcpyAsync(c->d_off, c->h_off, (n + 1) * 8, hipMemcpyHostToDevice, sin));
    auto copy_in = [&](int k) -> int {
        const uint64_t b0 = offsets[cut[k]], b1 = offsets[cut[k + 1]];
        if (b1 > b0) CK_HIP(c, hipMemcpyAsync(c->d_in + b0, bytes + b0, b1 - b0, hipMemcpyHostToDevice, sin));
        if (!inline_h2d) CK_HIP(c, hipEventRecord(c->ev_in[k], sin));
        return CIRCKIT_OK;
    };
    for (int k = 0; k < (inline_h2d ? 1 : parts); ++k) if ((rc = copy_in(k))) return rc;
    dbg_lap("enqueue copy-in");
    // ...and, while the first part is on its way, samples the content like stream_count_kernel does (a byte outside ACGT in
    // the first 1008 of a sampled record) -- on an eighth of that kernel's sample: the rule asks whether a sixteenth of the
    // records hold such a byte, and either answer gives the same output (it picks the faster build).  Up front and over all
    // 4096 samples this loop was 1.5-2.4 ms of EVERY call, more than the 16 MB batch it was deciding about took to cross the link.
    const uint64_t nc = n < CONTENT_SAMPLES / 8 ? n : CONTENT_SAMPLES / 8, cstep = n / nc;
    uint64_t bad = 0;
    {
        static const struct Acgt { uint8_t no[256]; Acgt() { memset(no, 1, sizeof no); no['A'] = no['C'] = no['G'] = no['T'] = 0; } } acgt;
        for (uint64_t k = 0; k < nc; ++k) {
            const uint64_t o = offsets[k * cstep], len = offsets[k * cstep + 1] - o, m = len < ck::FAST_MAX_N ? len : ck::FAST_MAX_N;
            uint32_t b = 0;
            for (uint64_t i = 0; i < m; ++i) b |= acgt.no[bytes[o + i]];
            bad += b;
        }
    }
    const uint32_t host_mode = batch_mode_of(two_word, longer, n, bad, nc, shorter, hash != nullptr && !idx && !strand && !(flags & ck::CK_FLAG_FWD_ONLY));
    dbg_lap("content sample");
    volatile uint32_t* unprocessed = c->h_mode + 4;                       // (pinned)
    for (int k = 0; k < parts; ++k) {
        const uint64_t r0 = cut[k], nk = cut[k + 1] - r0, b0 = offsets[r0], b1 = offsets[cut[k + 1]];
        if (!inline_h2d) CK_HIP(c, hipStreamWaitEvent(c->stream, c->ev_in[k], 0));
        rc = launch_canon(c, c->d_in, c->d_off + r0, nk, need_bytes ? c->d_out : nullptr, idx ? c->d_idx + r0 : nullptr,
                          strand ? c->d_strand + r0 : nullptr, hash ? c->d_hash + r0 : nullptr, flags, host_mode, k > 0);
        if (rc) return rc;                                  // (host_batch drains the streams)
        CK_HIP(c, hipEventRecord(c->ev_done[k], c->stream));
        CK_HIP(c, hipStreamWaitEvent(c->s_out, c->ev_done[k], 0));
        if (inline_h2d && k + 1 < parts && (rc = copy_in(k + 1))) return rc;
        if (out && b1 > b0) CK_HIP(c, hipMemcpyAsync(out + b0, c->d_out + b0, b1 - b0, hipMemcpyDeviceToHost, c->s_out));
        if (idx) CK_HIP(c, hipMemcpyAsync(st_idx + r0, c->d_idx + r0, nk * 4, hipMemcpyDeviceToHost, c->s_out));
        if (strand) CK_HIP(c, hipMemcpyAsync(st_strand + r0, c->d_strand + r0, nk, hipMemcpyDeviceToHost, c->s_out));
        if (hash) CK_HIP(c, hipMemcpyAsync(st_hash + r0, c->d_hash + r0, nk * 8, hipMemcpyDeviceToHost, c->s_out));
    }
    // the count of records nothing could take, added up over the parts (only part 0's launch zeroes it): read once, behind the
    // last part -- a copy of it per part made every part's kernels wait for the copy-out two parts back
    unprocessed[0] = 0;
    CK_HIP(c, hipMemcpyAsync((void*)unprocessed, c->d_counters + 3, 4, hipMemcpyDeviceToHost, c->s_out));
    dbg_lap("enqueue parts");
    CK_HIP(c, hipStreamSynchronize(c->s_out));
    dbg_lap("sync copy-out");
    CK_HIP(c, hipStreamSynchronize(c->stream));
    dbg_lap("sync stream");
    if (idx) memcpy(idx, st_idx, n * 4);
    if (strand) memcpy(strand, st_strand, n);
    if (hash) memcpy(hash, st_hash, n * 8);
    const uint32_t lost = unprocessed[0];
    if (lost) return fail(c, CIRCKIT_ERR_TOO_LONG, "%u record(s) could not be processed", lost);
    return CIRCKIT_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* circkit_version(void) { return "circkit-mi355x 0.1.0 (gfx950)"; }

int circkit_ctx_create(int device, circkit_ctx** out)
{
    if (!out) return CIRCKIT_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
        return CIRCKIT_ERR_NO_DEVICE;
    circkit_ctx* c = new circkit_ctx();
    c->device = device;
    *out = c;   // handed out even on failure below so the caller can read last_error, then destroy
    CK_HIP(c, hipSetDevice(device));
    CK_HIP(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    CK_HIP(c, hipEventCreate(&c->ev0));
    CK_HIP(c, hipEventCreate(&c->ev1));
    CK_HIP(c, hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
    CK_HIP(c, hipHostMalloc((void**)&c->h_mode, 256, hipHostMallocMapped));      // [0] mode, [1] tiers-busy flag, [4 .. 4 + MAX_PARTS) host_batch's per-part status
    c->h_mode[0] = 0; c->h_mode[1] = 0;
    CK_HIP(c, hipHostGetDevicePointer((void**)&c->d_mode, (void*)c->h_mode, 0));
    CK_HIP(c, hipMalloc(&c->d_comp, 256));
    CK_HIP(c, hipMalloc(&c->d_counters, 32 * sizeof(uint32_t)));
    CK_HIP(c, hipMemset(c->d_counters, 0, 32 * sizeof(uint32_t)));
    uint8_t comp[256];
    for (int v = 0; v < 256; ++v) comp[v] = (uint8_t)v;
    // bio 1.3.1 alphabets::dna complement table (call site lib/src/canonicalize.rs:56)
    const char *x = "AGCTYRWSKMDVHBN", *y = "TCGARYWSMKHBDVN";
    for (int i = 0; x[i]; ++i) { comp[(uint8_t)x[i]] = (uint8_t)y[i]; comp[(uint8_t)x[i] + 32] = (uint8_t)(y[i] + 32); }
    CK_HIP(c, hipMemcpy(c->d_comp, comp, 256, hipMemcpyHostToDevice));
    CK_HIP(c, hipFuncSetAttribute((const void*)canon_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)((TIER_D_DW + TIER_EXTRA_DW) * 4)));
    CK_HIP(c, hipFuncSetAttribute((const void*)canon_team_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)((TEAM_WAVES * TEAM_SLICE_DW + TIER_EXTRA_DW) * 4)));
    return CIRCKIT_OK;
}

int circkit_ctx_destroy(circkit_ctx* c)
{
    if (!c) return CIRCKIT_OK;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    void* ptrs[] = { c->d_comp, c->d_counters, c->d_seg_counts, c->d_in, c->d_out, c->d_strand, c->d_off, c->d_idx, c->d_hash, c->d_table,
                     c->d_view, c->d_hashed, c->d_gscratch };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (uint32_t* p : c->d_lists) if (p) (void)hipFree(p);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    if (c->ev_head) (void)hipEventDestroy(c->ev_head);
    for (hipEvent_t e : c->ev_in) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_done) if (e) (void)hipEventDestroy(e);
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    if (c->h_mode) (void)hipHostFree((void*)c->h_mode);
    if (c->h_off) (void)hipHostFree((void*)c->h_off);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return CIRCKIT_OK;
}

const char* circkit_last_error(const circkit_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

// The ctx's lists, counters, hash flags and table are shared by everything it enqueues, and a caller may hand a later call
// tensors an earlier one has just written: work on the new stream must not overtake what is still queued on the old one.
// An event on the old stream, waited for by the new one -- no host-side wait; switching to the stream already bound costs
// nothing (circkit_amd/uniq.py rebinds before every table call).
static int switch_stream(circkit_ctx* c, hipStream_t s)
{
    if (s == c->stream) return CIRCKIT_OK;
    CK_HIP(c, hipSetDevice(c->device));
    // the stream being left may be gone already (an external stream its owner destroyed): the ordering then falls back to a
    // host-side wait for the device, and the ctx still moves -- it must not stay bound to a dead stream for good
    if (hipEventRecord(c->ev_order, c->stream) != hipSuccess || hipStreamWaitEvent(s, c->ev_order, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        (void)hipGetLastError();
    }
    c->stream = s;
    return CIRCKIT_OK;
}

int circkit_ctx_set_stream(circkit_ctx* c, void* s)
{
    if (!c) return CIRCKIT_ERR_INVALID_ARG;
    return switch_stream(c, (hipStream_t)s);      // NULL is a real stream: HIP's default (null) stream
}

int circkit_ctx_use_own_stream(circkit_ctx* c)
{
    if (!c) return CIRCKIT_ERR_INVALID_ARG;
    return switch_stream(c, c->own_stream);
}

int circkit_ctx_synchronize(circkit_ctx* c)
{
    if (!c) return CIRCKIT_ERR_INVALID_ARG;
    CK_HIP(c, hipSetDevice(c->device));
    CK_HIP(c, hipStreamSynchronize(c->stream));
    return CIRCKIT_OK;
}

int circkit_ctx_last_kernel_ms(circkit_ctx* c, float* ms)
{
    if (!c || !ms) return CIRCKIT_ERR_INVALID_ARG;
    if (!c->timed) return fail(c, CIRCKIT_ERR_INVALID_ARG, "no timed batch yet");
    CK_HIP(c, hipEventSynchronize(c->ev1));
    CK_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return CIRCKIT_OK;
}

int circkit_ctx_batch_status(circkit_ctx* c, uint32_t* n_unprocessed)
{
    if (!c || !n_unprocessed) return CIRCKIT_ERR_INVALID_ARG;
    CK_HIP(c, hipSetDevice(c->device));
    *n_unprocessed = 0;
    CK_HIP(c, hipMemcpyAsync(n_unprocessed, c->d_counters + 3, 4, hipMemcpyDeviceToHost, c->stream));
    CK_HIP(c, hipStreamSynchronize(c->stream));
    if (*n_unprocessed)
        return fail(c, CIRCKIT_ERR_TOO_LONG, "%u record(s) were not processed: longer than the long-record scratch (%llu MiB, "
                    "circkit_ctx_set_long_record_scratch) or than 2^31 symbols", *n_unprocessed, (unsigned long long)(c->cap_gscratch >> 20));
    return CIRCKIT_OK;
}

int circkit_ctx_set_long_record_scratch(circkit_ctx* c, uint64_t bytes)
{
    if (!c) return CIRCKIT_ERR_INVALID_ARG;
    CK_HIP(c, hipSetDevice(c->device));
    if (bytes < (1ull << 20)) bytes = 1ull << 20;
    c->gscratch_default = bytes;
    CK_HIP(c, hipStreamSynchronize(c->stream));        // a batch in flight may be using the old one
    return ensure_gscratch(c, bytes);
}

int circkit_ctx_last_batch_mode(circkit_ctx* c, uint32_t* mode)
{
    if (!c || !mode) return CIRCKIT_ERR_INVALID_ARG;
    CK_HIP(c, hipSetDevice(c->device));
    CK_HIP(c, hipStreamSynchronize(c->stream));
    *mode = *c->h_mode & 3;
    return CIRCKIT_OK;
}

int circkit_canonicalize_batch_device(circkit_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets,
                                      uint64_t n, uint8_t* d_out, uint32_t* d_idx, uint8_t* d_strand,
                                      uint64_t* d_hash)
{
    if (!c) return CIRCKIT_ERR_INVALID_ARG;
    if (n && !d_offsets) return fail(c, CIRCKIT_ERR_INVALID_ARG, "d_offsets is NULL");
    return launch_canon(c, d_bytes, d_offsets, n, d_out, d_idx, d_strand, d_hash, 0);
}

int circkit_lmsr_batch_device(circkit_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n,
                              uint8_t* d_out, uint32_t* d_idx)
{
    if (!c) return CIRCKIT_ERR_INVALID_ARG;
    if (n && !d_offsets) return fail(c, CIRCKIT_ERR_INVALID_ARG, "d_offsets is NULL");
    return launch_canon(c, d_bytes, d_offsets, n, d_out, d_idx, nullptr, nullptr, ck::CK_FLAG_FWD_ONLY);
}

int circkit_canonicalize_batch(circkit_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t n,
                               uint8_t* out, uint32_t* idx, uint8_t* strand, uint64_t* hash)
{
    return host_batch(c, bytes, offsets, n, out, idx, strand, hash, 0);
}

void* circkit_host_alloc(size_t bytes)
{
    void* p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void circkit_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

static int check_ascii(circkit_ctx* c, const uint8_t* s, size_t n)
{
    if (n && !s) return fail(c, CIRCKIT_ERR_INVALID_ARG, "s is NULL");
    if (n >= (1ull << 31)) return fail(c, CIRCKIT_ERR_TOO_LONG, "record too long");
    for (size_t i = 0; i < n; ++i)
        if (s[i] & 0x80) return fail(c, CIRCKIT_ERR_NOT_ASCII, "non-ASCII byte at %zu (the reference panics on non-UTF-8 / indexes by char)", i);
    return CIRCKIT_OK;
}

int circkit_lmsr_index(circkit_ctx* c, const uint8_t* s, size_t n, size_t* out_index)
{
    if (!c || !out_index) return CIRCKIT_ERR_INVALID_ARG;
    int rc = check_ascii(c, s, n);
    if (rc) return rc;
    *out_index = 0;
    if (n == 0) return CIRCKIT_OK;                      // lib/src/canonicalize.rs:8-11: res stays 0
    uint64_t off[2] = { 0, (uint64_t)n };
    uint32_t idx = 0;
    rc = host_batch(c, s, off, 1, nullptr, &idx, nullptr, nullptr, ck::CK_FLAG_FWD_ONLY);
    *out_index = idx;
    return rc;
}

int circkit_lmsr(circkit_ctx* c, const uint8_t* s, size_t n, uint8_t* out)
{
    if (!c || (n && !out)) return CIRCKIT_ERR_INVALID_ARG;
    int rc = check_ascii(c, s, n);
    if (rc || n == 0) return rc;
    uint64_t off[2] = { 0, (uint64_t)n };
    return host_batch(c, s, off, 1, out, nullptr, nullptr, nullptr, ck::CK_FLAG_FWD_ONLY);
}

int circkit_canonicalize(circkit_ctx* c, const uint8_t* s, size_t n, uint8_t* out)
{
    if (!c || (n && !out)) return CIRCKIT_ERR_INVALID_ARG;
    int rc = check_ascii(c, s, n);
    if (rc || n == 0) return rc;
    uint64_t off[2] = { 0, (uint64_t)n };
    return host_batch(c, s, off, 1, out, nullptr, nullptr, nullptr, 0);
}

int circkit_xxh3_64(circkit_ctx* c, const uint8_t* s, size_t n, uint64_t* out_hash)
{
    if (!c || !out_hash || (n && !s)) return CIRCKIT_ERR_INVALID_ARG;
    CK_HIP(c, hipSetDevice(c->device));
    int rc = ensure_staging(c, n, 1);
    if (rc) return rc;
    uint64_t off[2] = { 0, (uint64_t)n };
    if (n) CK_HIP(c, hipMemcpyAsync(c->d_in, s, n, hipMemcpyHostToDevice, c->stream));
    CK_HIP(c, hipMemcpyAsync(c->d_off, off, 16, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(xxh3_kernel, dim3(1), dim3(256), 0, c->stream, c->d_in, c->d_off, (uint64_t)1, c->d_hash, (const uint8_t*)nullptr, (const uint32_t*)nullptr, (const uint8_t*)nullptr);
    CK_HIP(c, hipMemcpyAsync(out_hash, c->d_hash, 8, hipMemcpyDeviceToHost, c->stream));
    CK_HIP(c, hipStreamSynchronize(c->stream));
    return CIRCKIT_OK;
}

int circkit_xxh3_batch_device(circkit_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t n,
                              uint64_t* d_hash)
{
    if (!c || (n && (!d_offsets || !d_hash))) return CIRCKIT_ERR_INVALID_ARG;
    if (n == 0) return CIRCKIT_OK;
    CK_HIP(c, hipSetDevice(c->device));
    const uint64_t blocks = (n + 3) / 4;
    const unsigned grid = (unsigned)(blocks < (uint64_t)N_CU * 8 ? blocks : (uint64_t)N_CU * 8);
    hipLaunchKernelGGL(xxh3_kernel, dim3(grid), dim3(256), 0, c->stream, d_bytes, d_offsets, n, d_hash, (const uint8_t*)nullptr, (const uint32_t*)nullptr, (const uint8_t*)nullptr);
    CK_HIP(c, hipGetLastError());
    return CIRCKIT_OK;
}

#ifndef CK_UNIQ_BPC
#define CK_UNIQ_BPC 8        // workgroups of 256 threads per CU for the table kernels (one key per thread and trip)
#endif
#ifndef CK_UNIQ_LOAD_PCT
#define CK_UNIQ_LOAD_PCT 70       // circkit_uniq_reset sizes the table for at most this load with `expected_keys` distinct keys
#endif
// the table sized (allocated) for `expected_keys` distinct keys at CK_UNIQ_LOAD_PCT; contents untouched
static int uniq_size(circkit_ctx* c, uint64_t expected_keys)
{
    CK_HIP(c, hipSetDevice(c->device));
    uint64_t cap = 1024;
    while (cap * CK_UNIQ_LOAD_PCT < expected_keys * 100) cap <<= 1;
    if (cap - 1 != c->uniq_mask || !c->d_table) {
        c->uniq_mask = 0;
        int rc;
        if ((rc = grow(c, c->d_table, cap + 1))) return rc;
        c->uniq_mask = cap - 1;
    }
    return CIRCKIT_OK;
}
int circkit_uniq_reset(circkit_ctx* c, uint64_t expected_keys)
{
    if (!c) return CIRCKIT_ERR_INVALID_ARG;
    const int rcs = uniq_size(c, expected_keys);
    if (rcs) return rcs;
    const uint64_t cap = c->uniq_mask + 1;
    hipLaunchKernelGGL(uniq_clear_kernel, dim3(N_CU * 8), dim3(256), 0, c->stream, c->d_table, cap + 1, c->d_counters + 4);
    CK_HIP(c, hipGetLastError());
    c->uniq_count = 0;
    c->uniq_local = false;
    c->uniq_lost = false;
    return CIRCKIT_OK;
}

// Host-buffer form for streaming hosts (the CLI): folds this batch's hashes into the table, growing (and
// rehashing) it as the stream gets longer, and returns the winners.  Synchronizes.
int circkit_uniq_first_seen(circkit_ctx* c, const uint64_t* hash, uint64_t n, uint64_t base_index, uint64_t* first_seen)
{
    if (!c || (n && (!hash || !first_seen))) return CIRCKIT_ERR_INVALID_ARG;
    if (n == 0) return CIRCKIT_OK;
    CK_HIP(c, hipSetDevice(c->device));
    if (c->uniq_lost)
        return fail(c, CIRCKIT_ERR_HIP, "the uniq table was lost in a failed rehash: the earlier batches of this stream are gone "
                    "(circkit_uniq_reset starts a new one)");
    if (c->uniq_local) { const int rc0 = circkit_uniq_reset(c, n); if (rc0) return rc0; }      // a resolve result is not a stream's table
    if (!c->d_table || (c->uniq_count + n) * 2 > c->uniq_mask + 1) {
        uint64_t cap = 1 << 16;
        while (cap < 4 * (c->uniq_count + n)) cap <<= 1;
        UniqSlot* nt = nullptr;
        CK_HIP(c, hipMalloc(&nt, (cap + 1) * sizeof(UniqSlot)));
        hipLaunchKernelGGL(uniq_clear_kernel, dim3(N_CU * 8), dim3(256), 0, c->stream, nt, cap + 1, (uint32_t*)nullptr);
        if (c->d_table) {
            hipLaunchKernelGGL(uniq_rehash_kernel, dim3(N_CU * 8), dim3(256), 0, c->stream, (const UniqSlot*)c->d_table, c->uniq_mask + 2, nt, cap - 1);
            hipError_t e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess && getenv("CIRCKIT_TEST_FAIL_REHASH")) e = hipErrorUnknown;      // fault injection (tests/test_gpu_parity.py)
            (void)hipFree(c->d_table); c->d_table = nullptr;
            if (e != hipSuccess) {
                // the old table is gone and the new one holds who knows what: not a state to go on from silently -- every
                // later call fails until the caller starts over with circkit_uniq_reset
                (void)hipFree(nt);
                c->uniq_mask = 0; c->uniq_count = 0; c->uniq_lost = true;
                return fail(c, CIRCKIT_ERR_HIP, "uniq table rehash failed: %s", hipGetErrorString(e));
            }
        } else {
            CK_HIP(c, hipMemsetAsync(c->d_counters + 4, 0, 4, c->stream));
        }
        c->d_table = nt; c->uniq_mask = cap - 1;
    }
    int rc = ensure_staging(c, 0, n);
    if (rc) return rc;
    CK_HIP(c, hipMemcpyAsync(c->d_hash, hash, n * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(uniq_insert_kernel, dim3(N_CU * CK_UNIQ_BPC), dim3(256), 0, c->stream, (const uint64_t*)c->d_hash, (const uint64_t*)nullptr, n, base_index,
                       c->d_table, c->uniq_mask, c->d_counters + 4);
    uint64_t* d_fs = (uint64_t*)c->d_off;          // staging reuse: offsets buffer holds >= n + 1 u64
    hipLaunchKernelGGL(uniq_lookup_kernel, dim3(N_CU * CK_UNIQ_BPC), dim3(256), 0, c->stream, (const uint64_t*)c->d_hash, n, (const UniqSlot*)c->d_table,
                       c->uniq_mask, d_fs, (uint8_t*)nullptr, (uint64_t)0);
    CK_HIP(c, hipGetLastError());
    CK_HIP(c, hipMemcpyAsync(first_seen, d_fs, n * 8, hipMemcpyDeviceToHost, c->stream));
    CK_HIP(c, hipStreamSynchronize(c->stream));
    c->uniq_count += n;
    return CIRCKIT_OK;
}

int circkit_uniq_insert_device(circkit_ctx* c, const uint64_t* d_hash, uint64_t n, uint64_t base_index)
{
    if (!c || (n && !d_hash)) return CIRCKIT_ERR_INVALID_ARG;
    if (!c->d_table || c->uniq_local) return fail(c, CIRCKIT_ERR_INVALID_ARG, "circkit_uniq_reset has not been called (since the last circkit_uniq_resolve_device)");
    if (n == 0) return CIRCKIT_OK;
    CK_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(uniq_insert_kernel, dim3(N_CU * CK_UNIQ_BPC), dim3(256), 0, c->stream, d_hash, (const uint64_t*)nullptr, n, base_index,
                       c->d_table, c->uniq_mask, c->d_counters + 4);
    CK_HIP(c, hipGetLastError());
    c->uniq_count += n;
    return CIRCKIT_OK;
}

int circkit_uniq_insert_pairs_device(circkit_ctx* c, const uint64_t* d_hash, const uint64_t* d_index, uint64_t n)
{
    if (!c || (n && (!d_hash || !d_index))) return CIRCKIT_ERR_INVALID_ARG;
    if (!c->d_table || c->uniq_local) return fail(c, CIRCKIT_ERR_INVALID_ARG, "circkit_uniq_reset has not been called (since the last circkit_uniq_resolve_device)");
    if (n == 0) return CIRCKIT_OK;
    CK_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(uniq_insert_kernel, dim3(N_CU * CK_UNIQ_BPC), dim3(256), 0, c->stream, d_hash, d_index, n, (uint64_t)0, c->d_table,
                       c->uniq_mask, c->d_counters + 4);
    CK_HIP(c, hipGetLastError());
    c->uniq_count += n;
    return CIRCKIT_OK;
}

// Enqueues the lookups; a table overflow of the inserts before it is reported by circkit_uniq_status (or by the next
// synchronising uniq call) -- this call itself does not wait for the GPU.
int circkit_uniq_lookup_device(circkit_ctx* c, const uint64_t* d_hash, uint64_t n, uint64_t* d_first_seen)
{
    if (!c || (n && (!d_hash || !d_first_seen))) return CIRCKIT_ERR_INVALID_ARG;
    if (!c->d_table || c->uniq_local) return fail(c, CIRCKIT_ERR_INVALID_ARG, "circkit_uniq_reset has not been called (since the last circkit_uniq_resolve_device)");
    if (n == 0) return CIRCKIT_OK;
    CK_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(uniq_lookup_kernel, dim3(N_CU * CK_UNIQ_BPC), dim3(256), 0, c->stream, d_hash, n, (const UniqSlot*)c->d_table, c->uniq_mask, d_first_seen,
                       (uint8_t*)nullptr, (uint64_t)0);
    CK_HIP(c, hipGetLastError());
    return CIRCKIT_OK;
}

// ---- the multi-GPU exchange's device steps (circkit_amd/uniq.py; include/circkit.h)
int circkit_uniq_partition_device(circkit_ctx* c, const uint64_t* d_hash, uint64_t n, uint64_t base_index, uint32_t world, uint64_t* d_rows,
                                  uint64_t* d_counts, uint32_t* d_slot)
{
    if (!c || !d_counts || world == 0 || world > UNIQ_MAX_WORLD || (n && (!d_hash || !d_rows || !d_slot))) return CIRCKIT_ERR_INVALID_ARG;
    if (n >= 0xFFFFFFFFull) return fail(c, CIRCKIT_ERR_INVALID_ARG, "circkit_uniq_partition_device: n must be < 2^32 - 1");
    if ((uintptr_t)d_rows & 15) return fail(c, CIRCKIT_ERR_INVALID_ARG, "d_rows must be 16-byte aligned");
    CK_HIP(c, hipSetDevice(c->device));
    int rc = ensure_staging(c, 0, UNIQ_MAX_WORLD);             // d_hash: the scatter's cursors (64 x u64)
    if (rc) return rc;
    CK_HIP(c, hipMemsetAsync(d_counts, 0, world * sizeof(uint64_t), c->stream));
    CK_HIP(c, hipMemsetAsync(c->d_hash, 0, UNIQ_MAX_WORLD * sizeof(uint64_t), c->stream));
    if (n == 0) return CIRCKIT_OK;
    hipLaunchKernelGGL(uniq_partition_count_kernel, dim3(N_CU * 8), dim3(256), 0, c->stream, d_hash, n, world, (unsigned long long*)d_counts);
    hipLaunchKernelGGL(uniq_partition_scatter_kernel, dim3(N_CU * 8), dim3(256), 0, c->stream, d_hash, n, base_index, world,
                       (const unsigned long long*)d_counts, (unsigned long long*)c->d_hash, d_rows, d_slot);
    CK_HIP(c, hipGetLastError());
    return CIRCKIT_OK;
}

int circkit_uniq_insert_rows_device(circkit_ctx* c, const uint64_t* d_rows, uint64_t n)
{
    if (!c || (n && !d_rows)) return CIRCKIT_ERR_INVALID_ARG;
    if ((uintptr_t)d_rows & 15) return fail(c, CIRCKIT_ERR_INVALID_ARG, "d_rows must be 16-byte aligned");
    if (!c->d_table || c->uniq_local) return fail(c, CIRCKIT_ERR_INVALID_ARG, "circkit_uniq_reset has not been called (since the last circkit_uniq_resolve_device)");
    if (n == 0) return CIRCKIT_OK;
    CK_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(uniq_insert_rows_kernel, dim3(N_CU * CK_UNIQ_BPC), dim3(256), 0, c->stream, d_rows, n, c->d_table, c->uniq_mask, c->d_counters + 4);
    CK_HIP(c, hipGetLastError());
    c->uniq_count += n;
    return CIRCKIT_OK;
}

int circkit_uniq_lookup_rows_device(circkit_ctx* c, const uint64_t* d_rows, uint64_t n, uint64_t* d_answers)
{
    if (!c || (n && (!d_rows || !d_answers))) return CIRCKIT_ERR_INVALID_ARG;
    if ((uintptr_t)d_rows & 15) return fail(c, CIRCKIT_ERR_INVALID_ARG, "d_rows must be 16-byte aligned");
    if (!c->d_table || c->uniq_local) return fail(c, CIRCKIT_ERR_INVALID_ARG, "circkit_uniq_reset has not been called (since the last circkit_uniq_resolve_device)");
    if (n == 0) return CIRCKIT_OK;
    CK_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(uniq_lookup_rows_kernel, dim3(N_CU * CK_UNIQ_BPC), dim3(256), 0, c->stream, d_rows, n, (const UniqSlot*)c->d_table, c->uniq_mask, d_answers);
    CK_HIP(c, hipGetLastError());
    return CIRCKIT_OK;
}

int circkit_uniq_gather_device(circkit_ctx* c, const uint64_t* d_answers, const uint32_t* d_slot, uint64_t n, uint64_t base_index, uint64_t* d_first_seen,
                               uint8_t* d_keep)
{
    if (!c || (n && (!d_answers || !d_slot || !d_first_seen))) return CIRCKIT_ERR_INVALID_ARG;
    if (n == 0) return CIRCKIT_OK;
    CK_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(uniq_gather_kernel, dim3(N_CU * CK_UNIQ_BPC), dim3(256), 0, c->stream, d_answers, d_slot, n, base_index, d_first_seen, d_keep);
    CK_HIP(c, hipGetLastError());
    return CIRCKIT_OK;
}

// One shard, one call: table reset, insert, lookup and the keep flags -- the whole first-seen resolution of a batch whose
// record i has global index base_index + i (what `circkit uniq` decides per record on one GPU).  Only enqueues work.
int circkit_uniq_resolve_device(circkit_ctx* c, const uint64_t* d_hash, uint64_t n, uint64_t base_index, uint64_t* d_first_seen, uint8_t* d_keep)
{
    if (!c || (n && (!d_hash || !d_first_seen))) return CIRCKIT_ERR_INVALID_ARG;
    if (n >= 0xFFFFFFFFull) return fail(c, CIRCKIT_ERR_INVALID_ARG, "circkit_uniq_resolve_device: n must be < 2^32 - 1");     // before the table is touched
#ifndef CK_BKT_KEYS
#define CK_BKT_KEYS BKT_KEYS
#endif
#ifndef CK_UNIQ_BUCKET_MIN
#define CK_UNIQ_BUCKET_MIN (1u << 19)      // shards below this take the HBM table directly (the bucketed path is five launches)
#endif
    static const bool no_buckets = getenv("CIRCKIT_UNIQ_NO_BUCKETS") != nullptr;
    // (shards beyond 2600 x 8192 = 21M keys would overfill their buckets -- the count / scatter histograms hold 8192 bins in LDS --
    // and take the HBM table directly, like the small ones)
    if (n >= CK_UNIQ_BUCKET_MIN && n <= ((uint64_t)CK_BKT_KEYS << BKT_MAX_LOG2) && !no_buckets) {
        // LDS-sized buckets (see uniq_bkt_*_kernel); the table's memory is the scratch, the table itself only the fallback
        int rc = uniq_size(c, n);
        if (rc) return rc;
        uint32_t log2b = 6;
        while (log2b < BKT_MAX_LOG2 && ((uint64_t)CK_BKT_KEYS << log2b) < n) ++log2b;
        const uint32_t B = 1u << log2b;
        const uint64_t per = (n + BKT_NW - 1) / BKT_NW;
        uint8_t* scratch = reinterpret_cast<uint8_t*>(c->d_table);
        uint64_t* rows = reinterpret_cast<uint64_t*>(scratch);
        uint32_t* counts = reinterpret_cast<uint32_t*>(scratch + ((n * 16 + 255) & ~255ull));
        uint32_t* tot = counts + (uint64_t)BKT_NW * B;
        uint32_t* base = tot + B;
        if ((uint64_t)(reinterpret_cast<uint8_t*>(base + B + 1) - scratch) > (c->uniq_mask + 2) * sizeof(UniqSlot))
            return fail(c, CIRCKIT_ERR_INVALID_ARG, "internal: bucket scratch does not fit the table");
        uint32_t* flag = c->d_counters + 20;
        hipLaunchKernelGGL(uniq_bkt_count_kernel, dim3(BKT_NW), dim3(1024), B * 4, c->stream, d_hash, n, per, log2b, counts, flag, c->d_counters + 4);
        hipLaunchKernelGGL(uniq_bkt_colscan_kernel, dim3(B / 64), dim3(1024), 0, c->stream, counts, B, tot);
        hipLaunchKernelGGL(uniq_bkt_basescan_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)tot, B, base);
        hipLaunchKernelGGL(uniq_bkt_scatter_kernel, dim3(BKT_NW), dim3(1024), B * 4, c->stream, d_hash, n, per, log2b, (const uint32_t*)counts, (const uint32_t*)base, rows, d_first_seen, base_index);
        hipLaunchKernelGGL(uniq_bkt_resolve_kernel, dim3(B), dim3(BKT_RESOLVE_T), 0, c->stream, (const uint64_t*)rows, (const uint32_t*)base, base_index, d_first_seen, d_keep, flag);
        if (d_keep) hipLaunchKernelGGL(uniq_keep_kernel, dim3(N_CU * CK_UNIQ_BPC), dim3(256), 0, c->stream, (const uint64_t*)d_first_seen, n, base_index, d_keep, (const uint32_t*)flag);
        // the fallback (a bucket beyond BKT_MAX keys): the HBM table over the whole shard; three launches that return at once otherwise
        hipLaunchKernelGGL(uniq_clear_kernel, dim3(N_CU * 8), dim3(256), 0, c->stream, c->d_table, c->uniq_mask + 2, (uint32_t*)nullptr, (const uint32_t*)flag);
        hipLaunchKernelGGL(uniq_resolve_insert_kernel, dim3(N_CU * CK_UNIQ_BPC), dim3(256), 0, c->stream, d_hash, n, c->d_table, c->uniq_mask, c->d_counters + 4, (const uint32_t*)flag);
        hipLaunchKernelGGL(uniq_resolve_lookup_kernel, dim3(N_CU * CK_UNIQ_BPC), dim3(256), 0, c->stream, d_hash, n, (const UniqSlot*)c->d_table, c->uniq_mask,
                           d_first_seen, d_keep, base_index, (const uint32_t*)flag);
        CK_HIP(c, hipGetLastError());
        c->uniq_local = true;
        c->uniq_count = n;
        c->uniq_lost = false;
        return CIRCKIT_OK;
    }
    int rc = circkit_uniq_reset(c, n);
    if (rc || n == 0) return rc;
    hipLaunchKernelGGL(uniq_resolve_insert_kernel, dim3(N_CU * CK_UNIQ_BPC), dim3(256), 0, c->stream, d_hash, n, c->d_table, c->uniq_mask, c->d_counters + 4);
    hipLaunchKernelGGL(uniq_resolve_lookup_kernel, dim3(N_CU * CK_UNIQ_BPC), dim3(256), 0, c->stream, d_hash, n, (const UniqSlot*)c->d_table, c->uniq_mask,
                       d_first_seen, d_keep, base_index);
    CK_HIP(c, hipGetLastError());
    c->uniq_local = true;       // the table now holds split local values: circkit_uniq_reset before any other use
    c->uniq_count = n;
    // (Clearing the table for the NEXT call here, on a side stream behind the lookup, so that the 268 MB of stores run under
    // the next batch's hash kernel, was tried: 5.21 -> 5.18 ms, inside the noise -- that kernel has no bandwidth to spare.)
    return CIRCKIT_OK;
}

int circkit_uniq_status(circkit_ctx* c, uint32_t* n_overflowed)
{
    if (!c) return CIRCKIT_ERR_INVALID_ARG;
    CK_HIP(c, hipSetDevice(c->device));
    uint32_t overflow = 0;
    CK_HIP(c, hipMemcpyAsync(&overflow, c->d_counters + 4, 4, hipMemcpyDeviceToHost, c->stream));
    CK_HIP(c, hipStreamSynchronize(c->stream));
    if (n_overflowed) *n_overflowed = overflow;
    if (overflow) return fail(c, CIRCKIT_ERR_OOM, "uniq table overflow: %u key(s) found no slot (more distinct keys than circkit_uniq_reset sized it for)", overflow);
    return CIRCKIT_OK;
}

int circkit_synth_fill_device(circkit_ctx* c, uint64_t seed, uint64_t first_base, uint64_t n_bases, uint8_t* d_bytes)
{
    if (!c || (n_bases && !d_bytes)) return CIRCKIT_ERR_INVALID_ARG;
    if (n_bases == 0) return CIRCKIT_OK;
    CK_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(synth_fill_kernel, dim3(N_CU * 8), dim3(256), 0, c->stream, seed, first_base, n_bases, d_bytes);
    CK_HIP(c, hipGetLastError());
    return CIRCKIT_OK;
}

int circkit_fixed_offsets_device(circkit_ctx* c, uint64_t base, uint64_t len, uint64_t n, uint64_t* d_offsets)
{
    if (!c || !d_offsets) return CIRCKIT_ERR_INVALID_ARG;
    CK_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(fixed_offsets_kernel, dim3(N_CU * 4), dim3(256), 0, c->stream, base, len, n, d_offsets);
    CK_HIP(c, hipGetLastError());
    return CIRCKIT_OK;
}

// Measurement helper: enqueues one plain copy of `bytes` bytes (rounded down to 16) on the ctx stream.
int circkit_bench_copy_device(circkit_ctx* c, const void* d_src, void* d_dst, uint64_t bytes, uint32_t variant)
{
    if (!c || !d_src || !d_dst || variant >= 5) return CIRCKIT_ERR_INVALID_ARG;
    if (((uintptr_t)d_src | (uintptr_t)d_dst) & 15) return fail(c, CIRCKIT_ERR_INVALID_ARG, "circkit_bench_copy_device: 16-byte aligned buffers");
    CK_HIP(c, hipSetDevice(c->device));
    const uint64_t n16 = bytes / 16;
    const copy_v4* in = (const copy_v4*)d_src; copy_v4* out = (copy_v4*)d_dst;
    switch (variant) {
    case 0: hipLaunchKernelGGL(bench_copy_kernel<4>, dim3(2048), dim3(256), 0, c->stream, in, out, n16); break;
    case 1: hipLaunchKernelGGL(bench_copy_kernel<8>, dim3(2048), dim3(256), 0, c->stream, in, out, n16); break;
    case 2: hipLaunchKernelGGL(bench_copy_kernel<4>, dim3(8192), dim3(256), 0, c->stream, in, out, n16); break;
    case 3: hipLaunchKernelGGL(bench_copy_kernel<2>, dim3(65536), dim3(256), 0, c->stream, in, out, n16); break;
    default: CK_HIP(c, hipMemcpyAsync(d_dst, d_src, n16 * 16, hipMemcpyDeviceToDevice, c->stream)); break;      // the runtime's own copy
    }
    CK_HIP(c, hipGetLastError());
    return CIRCKIT_OK;
}

#ifdef CK_MIXED_TIMELINE
int circkit_debug_timeline(circkit_ctx* c, uint64_t* host, uint32_t n_wg)
{
    if (!c || !host || n_wg > 65536) return CIRCKIT_ERR_INVALID_ARG;
    CK_HIP(c, hipSetDevice(c->device));
    CK_HIP(c, hipStreamSynchronize(c->stream));
    CK_HIP(c, hipMemcpyFromSymbol(host, HIP_SYMBOL(g_timeline), (size_t)n_wg * 24));
    return CIRCKIT_OK;
}
#endif

#ifdef CK_DEBUG_POISON
// Test build only (tests/poison.py; not in include/circkit.h): records of all batches since ctx creation whose staged chunk was
// read before its DMA had landed (canon_stream.h stream_poison).  Synchronizes.
int circkit_debug_poison_count(circkit_ctx* c, uint32_t* n)
{
    if (!c || !n) return CIRCKIT_ERR_INVALID_ARG;
    CK_HIP(c, hipSetDevice(c->device));
    CK_HIP(c, hipStreamSynchronize(c->stream));
    CK_HIP(c, hipMemcpy(n, c->d_counters + 12, 4, hipMemcpyDeviceToHost));
    return CIRCKIT_OK;
}
#endif

// needletail 0.5.1 sequence::normalize(seq, false) -- host logic of the CSR packer.
size_t circkit_normalize(const uint8_t* s, size_t n, uint8_t* out, int* changed)
{
    const uint8_t* lut = ckhost::normalize_lut();     // fasta_host.cpp: one table, built once (thread-safe)
    size_t m = 0; int ch = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t c = s[i], o = lut[c];
        ch |= (o != c);
        out[m] = o;
        m += (o != 0);
    }
    if (changed) *changed = ch;
    return m;
}

}  // extern "C"
