// canon_stream.h -- the streaming kernel with workgroup-staged input.
//
// Same per-record routine as canon_fast.h (registers, DPP, fused XXH3); this file is how record bytes reach the
// lanes.  Measured on MI355X (tools/microbench/*copy_bench.hip, 10M x 1000 B pure copies): one 16 B/lane request per
// record per wave at the record's own alignment tops out at 4.4 ms, while flat 16-byte-aligned loads of 8 KiB+
// contiguous spans by the whole workgroup reach 3.7-3.9 ms -- the record-per-wave LOADS are what costs bandwidth,
// the stores are not.  So: a workgroup takes GROUP consecutive records, its waves DMA the group's byte span into an
// LDS image as aligned 16-byte chunks (global_load_lds_dwordx4; a ring of NBUF images, NBUF-1 groups in flight), and
// after one barrier every wave pulls its record(s) out of the image: one aligned ds_read_b128 per lane, pack to
// 2 bits, and the record's offset mod 16 is removed on the PACKED words (one DPP shift + one 64-bit shift).  Output
// goes straight from registers to global memory.
// Default geometry (circkit_hip.hip): 16 waves, one record each, two 16 KiB images -- what matters most is running
// 32 waves per CU; deeper rings with fewer waves measured slower (Measured).
#pragma once
#include "canon_fast.h"
#include "canon_pair.h"

namespace ck {

// Geometry of one workgroup: WPB waves, RPW (1 or 2) records per wave per group, NBUF images in the LDS ring (NBUF-1
// groups in flight).  A group is RPW*WPB consecutive records; its image is ROWS KiB per record (1008 / 2032 B at most
// + the 16 B alignment slack of the span start), so every wave issues ROWS*RPW DMA instructions per group.  The
// ROWS == 2 build also takes records of 1009..2032 bases (fast_canon2, two packed words per lane); it is a separate
// build because carrying that path costs the one-word path ~12 % more instructions (measured), so launch_canon picks
// it only for batches whose mean record length lies in that range.
// The iteration's scalar loads of the offsets are cold misses all the way to HBM (80 MB per 10M-record batch, every line used
// by one iteration of one workgroup) and every wave of the workgroup waits for them right behind the barrier, before anything
// else can be issued.  With CK_STREAM_TOUCH the line(s) the NEXT iteration's loads will ask for are fetched one iteration
// ahead by an LDS-DMA into a dump area (wave_prims.h glds4_touch): the scalar loads then find them in the L2.
#ifndef CK_STREAM_TOUCH
#define CK_STREAM_TOUCH 1
#endif
template <int WPB_, int NBUF_, int RPW_ = 2, int ROWS_ = 1>
struct StreamCfg {
    static constexpr int WPB = WPB_, RPW = RPW_, NBUF = NBUF_;
    static constexpr int ROWS = ROWS_;                          // packed words per lane a record may take: 1 (<= 1008 b) or 2 (<= 2032 b)
    static constexpr uint32_t GROUP = WPB * RPW;                // records per group
    static constexpr int DPW = ROWS * RPW;                      // DMA instructions (1 KiB each) per wave per group
    static constexpr uint32_t SPAN = GROUP * 1024 * ROWS;       // bytes of one image: ROWS KiB per record (incl. alignment slack)
    static constexpr uint32_t BUF_DW = (SPAN + 64) / 4;
    // ring, then the decode table, the deferral counter and (ROWS == 2) 1 KiB of padding: a record's lanes past its
    // end read up to 63 / 127 chunks beyond the image (never used), which for the last buffer lands in the table and
    // the padding
    static constexpr uint32_t PF = CK_STREAM_TOUCH;             // 1: every iteration touches the offsets it will load two iterations on
    static constexpr uint32_t LDS_DW = NBUF * BUF_DW + FAST_LUT_DW + 4 + (ROWS == 2 ? 256 : 0) + 64 * PF;     // (+ the touches' dump area, last)
};

// 1 KiB slot of the image that DMA instruction i of wave w fills: the first RPW instructions of every wave tile the
// first half of the image, the rest the second half
template <class C>
CK_DEV uint32_t stream_slot(uint32_t w, uint32_t i)
{
    return i < (uint32_t)C::RPW ? w * C::RPW + i : C::GROUP + w * C::RPW + (i - C::RPW);     // (second half: ROWS == 2 only)
}

struct StreamGroup {
    uint32_t base_lo;   // low 32 bits of the byte offset (into a.bytes) of the image's first byte
    bool ok;            // staged (else: every record of the group goes to the general kernel)
};

// DMA of the group whose first / one-past-last offsets are s, e.  Every wave issues exactly RPW instructions (chunks
// (wave*RPW + i)*64 + lane of the image; c16[i] = 16 * that chunk index, precomputed), whatever the group looks like
// -- the vmcnt bookkeeping of the loop depends on it.  Lanes past the span re-fetch its last chunk; unstaged groups
// fetch the offsets array (always readable).  All but the per-lane clamp is scalar work.
// Per-kernel constants of the DMA addressing: the payload pointer pulled back to a 16-byte boundary, and by how much.
struct StreamBase {
    const uint8_t* abase;     // a.bytes - mis
    uint32_t mis;             // (uintptr_t)a.bytes & 15
    uint32_t misaligned;      // mis != 0 ? 1 : 0
};
CK_DEV StreamBase stream_base(const CanonArgs& a)
{
    StreamBase b;
    b.mis = (uint32_t)(uintptr_t)a.bytes & 15u;
    b.abase = a.bytes - b.mis;
    b.misaligned = b.mis != 0 ? 1u : 0u;
    return b;
}
// The scalar side is written for instruction count (the hash builds of the streaming kernel are bound by instruction issue,
// scalar and vector alike; tools/probe_variants.py): everything is relative to the ALIGNED pointer, the three conditions are
// folded into one word that must be zero, the 64-bit compares are done on the halves (there is no s_cmp_lt_u64, the compiler
// moves them to the vector unit), and the lane offset is clamped against a scalar that is already zero for an unstaged group.
template <class C>
CK_DEV StreamGroup stream_issue(const CanonArgs& a, const StreamBase& sb, uint32_t out_of_range, uint64_t s, uint64_t e, uint32_t* buf, const uint32_t (&c16)[C::DPW])
{
    static_assert((C::SPAN & (C::SPAN - 1)) == 0, "image size must be a power of two");
    const uint64_t s1 = s + sb.mis;                   // the group's first byte, counted from abase
    const uint32_t base_lo = (uint32_t)s1 & ~15u, s1_hi = (uint32_t)(s1 >> 32);
    const uint64_t base1 = ((uint64_t)s1_hi << 32) | base_lo;
    const uint64_t nb1 = e + sb.mis - base1 - 1;      // image bytes - 1; wraps to huge for an empty span
    // not staged: a group whose first chunk would start before the payload (unaligned d_bytes: base1 == 0 reads the `mis` bytes
    // in front of it), empty or oversized spans (and, by the caller, the batch's last group: its final chunk would read past
    // the payload)
    // (flags as 0 / 1 words, not bools: a bool with two uses is materialised as a lane mask and its 0 / 1 value comes back
    // through the vector unit, taking everything behind it along)
    uint32_t bad = (uint32_t)(nb1 >> 32) | ((uint32_t)nb1 / C::SPAN) | out_of_range;
    bad |= (base_lo | s1_hi) == 0 ? sb.misaligned : 0u;
    StreamGroup grp;
    grp.ok = bad == 0;
    grp.base_lo = base_lo - sb.mis;                   // low 32 bits of the image's first byte as an offset into a.bytes
    const uint32_t last16 = grp.ok ? (uint32_t)nb1 & ~15u : 0u;
    const uint8_t* src = grp.ok ? sb.abase + base1 : (const uint8_t*)a.offsets;
    // DMA i of wave w covers chunks (i*WPB*RPW + w*RPW + i%RPW...) -- laid out so that the first RPW instructions of all
    // waves together cover the first half of the image: groups of records up to 1 KiB (the common case) need only
    // those, and the second half is skipped.  Allowed because with one group in flight (NBUF == 2) no vmcnt wait
    // counts DMA instructions; deeper rings always issue all of them.
    const uint32_t w = wave_in_block();
    const bool second_half = C::ROWS == 2 && (C::NBUF > 2 || last16 >= C::SPAN / 2);
#pragma unroll
    for (uint32_t i = 0; i < (uint32_t)C::DPW; ++i)
        if (i < (uint32_t)C::RPW || second_half)
            glds16_async_s(buf + stream_slot<C>(w, i) * 256, src, c16[i] < last16 ? c16[i] : last16);
    return grp;
}

#ifdef CK_DEBUG_POISON
// Test build (tests/poison.py, never the product library): a guard for the hand-counted vmcnt protocol below.  Right before
// a ring image's DMA is re-issued every wave overwrites its slots of that image with a pattern no payload byte can form
// (the writes are waited for, so the DMA data lands on top of them); a record whose aligned chunk still reads the pattern
// when it is packed was consumed before its DMA had landed -- counted in status[9] (d_counters[12]), read back by
// circkit_debug_poison_count.  On the CPU emulator the DMA is a synchronous memcpy, so only the GPU can tell.
constexpr uint32_t CK_POISON = 0xFEFEFEFEu;
template <class C>
CK_DEV void stream_poison(uint32_t* buf)
{
    const uint32_t w = wave_in_block(), t = lane_id();
#pragma unroll
    for (uint32_t i = 0; i < (uint32_t)C::DPW; ++i) lds_store16(buf + stream_slot<C>(w, i) * 256 + 4 * t, u32x4{ CK_POISON, CK_POISON, CK_POISON, CK_POISON });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
CK_DEV void stream_poison_check(const CanonArgs& a, u32x4 v, uint64_t lanes)
{
    const uint64_t hit = ballot(v.x == CK_POISON || v.y == CK_POISON || v.z == CK_POISON || v.w == CK_POISON) & lanes;
    if (hit != 0 && lane_id() == 0) atomic_add_u32(a.status + 9, 1u);
}
#endif

// loop of one wave of a workgroup; every wave of the workgroup runs the same number of iterations (barriers inside).
// GH: XXH3 is finished per record group by one wave (canon_fast.h, group_hash_*); gh = its LDS area (gh_lds_dw<GROUP>()
// dwords, constants initialised by group_hash_init).
// ALPHA: records with a byte outside ACGT take the 4-bit register routine here (fast_canonw<4>) instead of being
// deferred -- the build for batches whose mode carries MODE_ALPHA.
template <class C, bool HASH, bool AUX, bool GH = false, bool ALPHA = false>
CK_DEV void canon_stream_wave_loop(const CanonArgs& a, const uint32_t* lut, uint32_t* ring, uint32_t* blk_count, uint32_t block,
                                   uint32_t nblocks, uint32_t* gh = nullptr)
{
    static_assert(!ALPHA || !AUX, "index / strand outputs: the 4-bit records take the LDS tiers");
    static_assert(C::RPW == 1 || C::RPW == 2, "one or two records per wave per group");
    static_assert(!GH || (HASH && !AUX && C::ROWS == 1 && C::GROUP <= 16), "the group merger takes up to 16 records of one packed word per 16 symbols");
    // GH with two records per wave: canon_pair.h -- one record per half-wave, both at once
    // ... and the bytes-only build of such a geometry as well (gh = PAIR_SCRATCH_DW dwords per record then): the build for batches of
    // SHORT records (circkit_hip.hip MODE_SHORT -- a record of 300 symbols keeps 19 of a wave's 64 lanes busy, two of them 20 of 32
    // each: 20M x 200 b 6.14 -> 4.84 ms, 10M x 400 b 3.34 -> 2.68; from ~850 symbols on one record per wave is faster)
    constexpr bool PAIR_B = !HASH && !AUX && !ALPHA && C::ROWS == 1 && C::RPW == 2;
    constexpr bool PAIR = (GH && C::RPW == 2) || PAIR_B;
    static_assert(!PAIR || !ALPHA, "the pair routine takes pure ACGT only (the ALPHA build keeps one record per wave)");
    constexpr int SPW = PAIR ? 1 : C::RPW;            // vector-memory stores every wave is SURE to issue per iteration (pair: cell 0 of a finished record, or a deferral)
    constexpr int D = C::NBUF - 1;                    // groups in flight
    // the touches pay in the builds with the fused XXH3, which are bound by instruction issue (same box, 10M x 1 kb: bytes + hash
    // 4.16 -> 4.05 ms); the bytes-only build sits on the memory system and LOSES when its loads go out earlier (3.60 -> 4.00 ms
    // with the touches, 3.60 -> 3.79 with offsets that need no load at all: tools/probe_variants.py)
    constexpr int PF = HASH ? (int)C::PF : 0;
    const uint32_t N = (uint32_t)a.n_records, n_groups = (N + C::GROUP - 1) / C::GROUP;
    if (N == 0) return;
    const uint32_t n_staged = n_groups - 1;           // the batch's last group is never staged
    const uint32_t w = wave_in_block(), t = lane_id();
    if (block == n_staged % nblocks) {                // ...its records go to the general kernel
        for (uint32_t rec = n_staged * C::GROUP + C::RPW * w; rec < n_staged * C::GROUP + C::RPW * (w + 1) && rec < N; ++rec)
            defer_record(a, blk_count, block, rec);
    }
    if (block >= n_staged) return;
    FastHashConst hc{};
    if (HASH) hc = fast_hash_const();
    FastShape shape;
    PairShape pshape;
    // does every record issue at least one store (bytes, hash, index or its deferral)?  The vmcnt arithmetic below counts
    // on it.  Not with the group merger on a hash-only batch: a record then leaves nothing but LDS writes.
    const bool stores = GH ? a.out_bytes != nullptr : (a.out_bytes || a.out_hash || a.out_index || a.out_strand);
    const uint32_t* gh_const = GH ? gh + 2 * C::GROUP * GH_STRIDE_DW : nullptr;
    uint32_t c16[C::DPW];
#pragma unroll
    for (int i = 0; i < C::DPW; ++i) c16[i] = (stream_slot<C>(w, (uint32_t)i) * 64 + t) * 16;
    // ring state in scalars: q[0] = the group being processed, q[1..D-1] = the ones in flight behind it
    const StreamBase sb = stream_base(a);
    StreamGroup q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const uint32_t g = block + d * nblocks, gg = g < n_staged ? g : 0;
        uint64_t s, e;
        sload_2u64(a.offsets + (uint64_t)gg * C::GROUP, a.offsets + (uint64_t)gg * C::GROUP + C::GROUP, s, e);
#ifdef CK_DEBUG_POISON
        stream_poison<C>(ring + d * C::BUF_DW);
#endif
        q[d] = stream_issue<C>(a, sb, g < n_staged ? 0u : 1u, s, e, ring + d * C::BUF_DW, c16);
    }
    vmem_wait<(D - 1) * C::DPW>();                    // the first group's DMAs; the later ones may still fly
    block_barrier();
    // loop-carried scalars, advanced by addition (no index -> address arithmetic per iteration): the offsets of the group to
    // prefetch and of this wave's record(s), the ring position of q[0] and of the image freed by the previous iteration
    const uint64_t stride = (uint64_t)nblocks * C::GROUP;
    const uint64_t* p_grp = a.offsets + ((uint64_t)block + (uint64_t)D * nblocks) * C::GROUP;
    const uint64_t* p_rec = a.offsets + ((uint64_t)block * C::GROUP + C::RPW * w);
    uint32_t gf = block + D * nblocks, ra = block * C::GROUP + C::RPW * w;
    uint32_t img_dw = 0, free_dw = (C::NBUF - 1) * C::BUF_DW, it = 0;
    const uint32_t pf_off = (t >> 5) * (C::GROUP * 8);
    uint32_t oor = (n_staged - 1u - gf) >> 31;                  // 1: the group to prefetch does not exist (both < 2^28)
#ifdef CK_EXP_UNIFORM
    uint64_t exp_a, exp_b;
    sload_2u64(a.offsets, a.offsets + 1, exp_a, exp_b);
    const uint32_t exp_len = (uint32_t)(exp_b - exp_a);
#endif
    for (uint32_t g = block; g < n_staged; g += nblocks, ++it) {
        // one scalar round trip per iteration: the span of the group to prefetch and this wave's record offsets
        uint64_t s, e, o0, o1, o2;
#ifdef CK_EXP_UNIFORM
        // EXPERIMENT build only: offsets of a batch of equal-length records computed from the (runtime) length
        { const uint64_t L = exp_len; s = (uint64_t)(oor ? 0 : gf) * C::GROUP * L; e = s + C::GROUP * L; o0 = (uint64_t)ra * L; o1 = o0 + L; o2 = o1 + L; }
#elif defined(CK_EXP_FIXED1000)
        // EXPERIMENT build only (tools/probe_variants.py): every record 1000 bytes, offsets computed instead of loaded -- an upper
        // bound of what leaner scalar bookkeeping at the head of the loop could buy
        s = (uint64_t)(oor ? 0 : gf) * C::GROUP * 1000; e = s + C::GROUP * 1000; o0 = (uint64_t)ra * 1000; o1 = o0 + 1000; o2 = o1 + 1000;
#else
#ifdef CK_STREAM_INDEXED
        sload_group<(int)C::GROUP>(a.offsets + (uint64_t)(oor ? 0u : gf) * C::GROUP, a.offsets + ra, s, e, o0, o1, o2);
#else
        sload_group<(int)C::GROUP>(oor ? a.offsets : p_grp, p_rec, s, e, o0, o1, o2);
#endif
#endif
#ifdef CK_DEBUG_POISON
        stream_poison<C>(ring + free_dw);
#endif
#ifdef CK_EXP_DMA_SLEEP
        if (!HASH) __builtin_amdgcn_s_sleep(CK_EXP_DMA_SLEEP);      // EXPERIMENT: the bytes-only build loses when its loads go out earlier -- does it gain when they go out later?
#endif
        const StreamGroup fut = stream_issue<C>(a, sb, oor, s, e, ring + free_dw, c16);
        // the next iteration's group to prefetch
        gf += nblocks;
#ifndef CK_STREAM_INDEXED
        p_grp += stride;
#endif
        oor = (n_staged - 1u - gf) >> 31;
        if constexpr (PF) {
            // the offsets the next iteration loads: the span of that group -- its first offset (lanes 0..31) and the one behind
            // its last (lanes 32..63: the next 128 bytes) -- which are also the lines of this wave's record two iterations on
#ifdef CK_STREAM_INDEXED
            glds4_touch(ring + (C::LDS_DW - 64), a.offsets + (uint64_t)(oor ? 0u : gf) * C::GROUP, pf_off);
#else
            glds4_touch(ring + (C::LDS_DW - 64), oor ? a.offsets : p_grp, pf_off);
#endif
        }
        const uint32_t* img = ring + img_dw;
        uint32_t* slot = GH ? gh + ((it & 1) * C::GROUP + C::RPW * w) * GH_STRIDE_DW : PAIR_B ? gh + C::RPW * w * PAIR_SCRATCH_DW : nullptr;
        if constexpr (PAIR) {
            // records ra, ra + 1 in the two halves of the wave
            const uint32_t nA = (uint32_t)o1 - (uint32_t)o0, nB = (uint32_t)o2 - (uint32_t)o1;
            const uint32_t elig = q[0].ok ? (fast_eligible(nA) ? 1u : 0u) | (fast_eligible(nB) ? 2u : 0u) : 0u;
            uint32_t done = 0;
#ifdef CK_DEBUG_POISON
            if (elig) {     // the chunks pair_canon is about to pack (same addressing)
                const uint32_t hbp = t >> 5, up = t & 31, relp = (uint32_t)o0 - q[0].base_lo + (hbp ? (uint32_t)(o1 - o0) : 0u);
                const uint32_t nchp = ((relp & 15) + (hbp ? nB : nA) + 15) >> 4;
                const bool el = ((elig >> hbp) & 1) != 0;
                const uint32_t* cpp = img + 4 * ((relp >> 4) + 2 * up);
                stream_poison_check(a, lds_load16(cpp), ballot(el && 2 * up < nchp));
                stream_poison_check(a, lds_load16(cpp + 4), ballot(el && 2 * up + 1 < nchp));
            }
#endif
            if (elig) done = pair_canon<!PAIR_B>(a, lut, PAIR_B ? nullptr : gh_const + GH_CONST_DW, pshape, img, q[0].base_lo, ra, o0, o1, (elig & 1) ? nA : 64u, (elig & 2) ? nB : 64u, elig, slot);
            else if (!PAIR_B) { group_hash_invalidate(slot); group_hash_invalidate(slot + GH_STRIDE_DW); }
            if (!(done & 1)) defer_record(a, blk_count, block, ra);
            if (!(done & 2)) defer_record(a, blk_count, block, ra + 1);
        }
        if (GH && !PAIR) group_hash_invalidate(slot);
#pragma unroll
        for (int k = 0; k < (PAIR ? 0 : C::RPW); ++k) {
            const uint64_t off = k ? o1 : o0;                           // (o2 is loaded but unused when RPW == 1)
            const uint32_t n = (uint32_t)(k ? o2 : o1) - (uint32_t)off;
            const uint32_t rec = ra + k;
            bool done = false, tried4 = false;
            if (C::ROWS == 2 && q[0].ok && fast2_eligible(n)) {
                // 1009..2032 bases: two chunks per lane (c0 + 2t, c0 + 2t + 1); the word behind them is lane t+1's first
                const uint32_t rel = (uint32_t)off - q[0].base_lo, a16 = rel & 15, nch = (a16 + n + 15) >> 4;     // <= 128
                const uint32_t* cp = img + 4 * ((rel >> 4) + 2 * t);
                uint32_t m0, m1;
                const uint32_t P0 = fast_pack(lds_load16(cp), m0), P1 = fast_pack(lds_load16(cp + 4), m1);
                const uint64_t bad0 = ballot(m0 != 0) & (~0ull >> (64 - ((nch + 1) >> 1)));       // chunk 2t   < nch
                const uint64_t bad1 = ballot(m1 != 0) & ((nch >> 1) >= 64 ? ~0ull : (1ull << (nch >> 1)) - 1);   // chunk 2t+1 < nch
#ifdef CK_DEBUG_POISON
                stream_poison_check(a, lds_load16(cp), ~0ull >> (64 - ((nch + 1) >> 1)));
                stream_poison_check(a, lds_load16(cp + 4), (nch >> 1) >= 64 ? ~0ull : (1ull << (nch >> 1)) - 1);
#endif
                const uint32_t sh2 = 32 - 2 * a16;
                done = fast_canon2<HASH, AUX>(a, lut, hc, rec, off, n, lshr64(P0, P1, sh2), lshr64(P1, wave_shl1(P0), sh2), (bad0 | bad1) != 0);
            } else if (q[0].ok && fast_eligible(n)) {
                // lane t packs the aligned chunk c0 + t of the image; the record starts a16 bytes into chunk c0, so
                // the byte funnel is done on the packed words: 2*a16 bits, with the next lane's word behind
                const uint32_t rel = (uint32_t)off - q[0].base_lo, a16 = rel & 15, nch = (a16 + n + 15) >> 4;
                const u32x4 v = lds_load16(img + 4 * ((rel >> 4) + t));
                const uint64_t chunks = ~0ull >> (64 - nch);                            // 3 <= nch <= 64
#ifdef CK_DEBUG_POISON
                stream_poison_check(a, v, chunks);
#endif
                // chunks 0 and nch-1 also hold bytes of the neighbouring records: an invalid byte there sends this
                // record to the general kernel for nothing, which is harmless
                uint64_t bad;                       // lanes with a byte the 2-bit routines cannot take
                if constexpr (ALPHA && !AUX && (!HASH || GH)) {
                    // N is the usual stranger: ONE packing pass gives the 2-bit words with N as G plus the mask, and the
                    // register routine's N-mask variant takes the record (220 VALU per record against 255 with the 4-bit
                    // routine below, and 16-symbol keys instead of 8: hardly a tie).  A batch of this build holds an N in
                    // most records; the few without take the plain routine.  What the variant refuses -- an N among the
                    // deciding symbols -- and records with a gap go to the 4-bit routine.  (With the XXH3, end of round 4: the
                    // hash is fused for records with N too -- group_hash_put_bytes; before, every such record took the 4-bit
                    // routine and a pass of the xxh3 kernel over its bytes: `uniq` on 10M x 1 kb with 1 % N 10.8 ms.)
                    uint32_t nm, miss2;
                    const uint32_t Pn = fast_pack_n2(v, nm, miss2);
                    bad = ballot(miss2 != 0) & chunks;
                    const uint64_t with_n = ballot(nm != 0) & chunks;
                    if (bad == 0) {
                        const uint32_t Fw = lshr64(Pn, wave_shl1(Pn), 32 - 2 * a16);
                        // (with the XXH3 every record of the build leaves its last stripe as bytes -- the merger is built for that --, so
                        // the few without an N take the N-mask variant with an empty mask)
                        if (!HASH && with_n == 0) done = fast_canon<HASH, false, GH, true, false, true>(a, lut, hc, shape, rec, off, n, Fw, 0, slot);
                        else done = fast_canon<HASH, false, GH, true, true, true>(a, lut, hc, shape, rec, off, n, Fw, 0, slot, lshr64(nm, wave_shl1(nm), 32 - 2 * a16));
                    }
                    bad |= with_n;                  // (what the 4-bit routine is for)
                } else {
                    uint32_t miss;
                    const uint32_t P = fast_pack(v, miss);
                    bad = ballot(miss != 0) & chunks;
                    if (!ALPHA || bad == 0) done = fast_canon<HASH, AUX, GH, true, false, ALPHA || C::ROWS == 2>(a, lut, hc, shape, rec, off, n, lshr64(P, wave_shl1(P), 32 - 2 * a16), bad, slot);
                }
                if (ALPHA && bad != 0 && !done) {
                    // {-,A,C,G,N,T} at 4 bits per symbol: 64 bits per lane, the record's offset in its first chunk
                    // removed by a 128-bit funnel with the next lane's pair (4 * a16 bits)
                    uint32_t H, L, bad4;
                    fast_pack4(v, H, L, bad4);
                    const uint64_t b4 = ballot(bad4 != 0) & chunks;
                    const uint64_t X = ((uint64_t)H << 32) | L, Xn = ((uint64_t)wave_shl1(H) << 32) | wave_shl1(L);
                    const uint64_t W = a16 ? (X << (4 * a16)) | (Xn >> (64 - 4 * a16)) : X;
                    done = fast_canonw<4, HASH, false>(a, lut, hc, rec, off, n, (uint32_t)(W >> 32), (uint32_t)W, b4 != 0);
                    tried4 = true;
                }
            }
            // The entry's flag: the 4-bit register routine has had this record (a tied 8-symbol key, mostly) -- the rescue
            // pass, which would run the same routine again, hands it on.  (Otherwise no alphabet flag: the edge chunks hold
            // neighbours' bytes too.  A flagged record that is pure ACGT itself takes the N-mask mode of the tiers: correct.)
            // (Not the entries' second flag: the register routine's N variant refuses every N among the deciding symbols, the
            // tiers' N-mask mode has the prefix rule for them and may still succeed.  Setting it here as well measured the same
            // on 1 % and 10 % N, fixed and mixed lengths: tools/ruled_ab.sh.)
            if (!done) defer_record(a, blk_count, block, rec, ALPHA && tried4);
        }
        // the previous group's hashes: its slots were complete at the barrier that ended the previous iteration
        if constexpr (GH) {
            if (it > 0 && w == ((it - 1) & (C::WPB - 1)))
                group_hash_merge<(int)C::GROUP, PAIR ? 2 : 4, GH && ALPHA>(a, lut, gh_const, gh + (((it - 1) & 1) * C::GROUP) * GH_STRIDE_DW, (g - nblocks) * C::GROUP);
        }
        // the next group's DMAs (issued D-1 iterations ago) must have landed.  Younger vector-memory instructions:
        // the DMAs of the D-1 groups issued since (DPW each), and the stores of this and the D-1 previous iterations
        // (>= RPW each: every record stores its bytes, hash or index, or its deferral).  While the next group is still
        // one of the prologue's (it + 1 < D) there are fewer stores: only this iteration's are counted on.
#ifdef CK_DEBUG_POISON_BREAK
        vmem_wait<63>();                                  // negative control of the poison guard: no wait at all -- the count must NOT stay zero
#else
        // (touches: one per iteration, issued right behind the iteration's DMAs -- D of them are younger than the DMAs waited
        // for, one at least while the next group is still one of the prologue's)
        if (!stores) { if (it + 1 < (uint32_t)D) vmem_wait<(D - 1) * C::DPW + PF>(); else vmem_wait<(D - 1) * C::DPW + D * PF>(); }
        else if (it + 1 < (uint32_t)D) vmem_wait<(D - 1) * C::DPW + SPW + PF>();
        else vmem_wait<(D - 1) * C::DPW + D * SPW + D * PF>();
#endif
        block_barrier();
#pragma unroll
        for (int d = 0; d + 1 < D; ++d) q[d] = q[d + 1];
        q[D - 1] = fut;
        free_dw = img_dw;
        img_dw = img_dw + C::BUF_DW == C::NBUF * C::BUF_DW ? 0 : img_dw + C::BUF_DW;
        ra += nblocks * C::GROUP;
#ifndef CK_STREAM_INDEXED
        p_rec += stride;
#endif
    }
    if constexpr (GH) {
        if (it > 0 && w == ((it - 1) & (C::WPB - 1))) {             // the last group's hashes (behind the loop's final barrier)
            const uint32_t g_last = block + (it - 1) * nblocks;
            group_hash_merge<(int)C::GROUP, PAIR ? 2 : 4, GH && ALPHA>(a, lut, gh_const, gh + (((it - 1) & 1) * C::GROUP) * GH_STRIDE_DW, g_last * C::GROUP);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Rescue pass: eligible records that the staged loop could not take because their GROUP was not staged -- one long
// record is enough to push a group's span past its image, so in batches of mixed lengths most short records end up
// here -- are run through the same register routine, fetched straight from memory: 16 bytes per lane at the record's
// own alignment (lanes past the last full word re-read the record's LAST 16 bytes, in bounds, and shift their tail
// symbols up).  One wave per record of the streaming kernel's deferral list; what is still not eligible (other
// alphabets, ties, longer records) goes on to the LDS tiers.  A separate kernel on purpose: the same fallback inside
// the staged loop cost the staged path 5 %.
template <bool HASH, bool AUX>
struct RescueState {        // per-wave constants of the rescue pass (kept across list segments)
    FastHashConst hc{};
    FastShape shape;
};
// one wave's share of list segment `seg_index`; failures are appended to the SAME segment index of the output list.
// all_records: the streaming kernel did not run (a batch with so many long records that hardly any group could be
// staged): segment s then stands for records [s * all_seg_cap, (s + 1) * all_seg_cap).
// The wave takes the segment's entries in chunks of RESCUE_CHUNK; list entries and offsets of a chunk arrive in one
// round trip (lane L: entry L, handed out by v_readlane), and each record's bytes are requested while the record
// before it is being processed -- taken one at a time, a record is three dependent round trips (list entry -> offsets ->
// bytes) with nothing to hide them behind (measured: 5.4 us per record and wave, 6.6 ms for 10M records).
constexpr uint32_t RESCUE_CHUNK = 8;
struct RescueMeta { uint32_t rec, len; uint64_t off; };      // lane L < count: entry L of the chunk (rec: the raw entry, flag included)
CK_DEV RescueMeta rescue_meta(const CanonArgs& a, const uint32_t* seg, uint64_t first, uint32_t c0, uint32_t count, bool all_records)
{
    const uint32_t t = lane_id();
    RescueMeta m{ 0, 0, 0 };
    if (t < RESCUE_CHUNK && c0 + t < count) {
        m.rec = all_records ? (uint32_t)first + c0 + t : seg[c0 + t];
        const uint32_t r = m.rec & ENTRY_REC;
        m.off = a.offsets[r];
        const uint64_t len = a.offsets[r + 1] - m.off;
        m.len = len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len;
    }
    return m;
}
// the 16 bytes lane t packs of an eligible record (lanes past the last full word re-read the record's LAST 16 bytes)
CK_DEV u32x4 rescue_fetch(const CanonArgs& a, const RescueMeta& m, uint32_t l)
{
    const uint32_t n = readlane(m.len, l), t = lane_id();
    const uint64_t off = ((uint64_t)readlane((uint32_t)(m.off >> 32), l) << 32) | readlane((uint32_t)m.off, l);
    if (!fast_eligible(n) || (readlane(m.rec, l) & ENTRY_NOT_ACGT)) return u32x4{ 0, 0, 0, 0 };
    return load16(a.bytes + off + (t >= (n >> 4) ? n - 16 : 16 * t));
}
// entry l of a chunk, its 16 bytes per lane in v
template <bool HASH, bool AUX>
CK_DEV void rescue_one(const CanonArgs& a, const uint32_t* lut, RescueState<HASH, AUX>& st, uint32_t* seg_count, uint32_t seg_index,
               const RescueMeta& cur, uint32_t l, u32x4 v)
{
    const uint32_t t = lane_id();
    const uint32_t entry = readlane(cur.rec, l), rec = entry & ENTRY_REC, n = readlane(cur.len, l);
    const uint64_t off = ((uint64_t)readlane((uint32_t)(cur.off >> 32), l) << 32) | readlane((uint32_t)cur.off, l);
    bool done = false, not_acgt = (entry & ENTRY_NOT_ACGT) != 0;       // flagged: the streaming kernel's 4-bit routine has had it
    if (!not_acgt && fast_eligible(n)) {
        const uint32_t nwf = n >> 4;
        const uint32_t tail_syms = t >= nwf ? (16 - (n & 15)) & 15 : 0;       // the tail lane's symbols move up by this much
        uint32_t miss;
        const uint32_t F = fast_pack(v, miss);
        const uint64_t bad = ballot(miss != 0);
        not_acgt = bad != 0;                              // (exact here: the lanes hold this record's bytes only)
        if (AUX || bad == 0) {
            done = fast_canon<HASH, AUX>(a, lut, st.hc, st.shape, rec, off, n, F << (2 * tail_syms), bad);
        } else {
            // a byte outside ACGT: the CLI alphabet {-,A,C,G,N,T} at 4 bits per symbol, two words per lane, still in
            // registers (builds that report index / strand leave these records to the LDS tiers)
            uint32_t H, L, bad4;
            fast_pack4(v, H, L, bad4);
            const uint64_t x = ((((uint64_t)H) << 32) | L) << (4 * tail_syms);
            done = fast_canonw<4, HASH, false>(a, lut, st.hc, rec, off, n, (uint32_t)(x >> 32), (uint32_t)x, ballot(bad4 != 0) != 0);
        }
    }
    if (!done) defer_record(a, seg_count, seg_index, rec, not_acgt, (entry & ENTRY_NO_2N) != 0);
}
// one pure-ACGT record of 48..1008 bases through the register routine, 16 bytes per lane straight from memory
template <bool HASH, bool AUX>
CK_DEV bool rescue_direct(const CanonArgs& a, const uint32_t* lut, RescueState<HASH, AUX>& st, uint32_t rec, uint64_t off, uint64_t len, bool& not_acgt)
{
    if (len > FAST_MAX_N || !fast_eligible((uint32_t)len)) return false;
    const uint32_t n = (uint32_t)len, nwf = n >> 4, t = lane_id();
    uint32_t miss;
    uint32_t F = fast_pack(load16(a.bytes + off + (t >= nwf ? n - 16 : 16 * t)), miss);
    F <<= t >= nwf ? ((16 - (n & 15)) & 15) * 2 : 0;
    const uint64_t bad = ballot(miss != 0);
    not_acgt = bad != 0;
    return fast_canon<HASH, AUX>(a, lut, st.hc, st.shape, rec, off, n, F, bad);
}
template <bool HASH, bool AUX, bool ALPHA>
CK_DEV void canon_rescue_segment(const CanonArgs& a, const uint32_t* lut, RescueState<HASH, AUX>& st, uint32_t* seg_count, uint32_t seg_index,
                                 uint32_t wib, uint32_t wpb, bool all_records)
{
    const uint32_t t = lane_id();
    uint64_t first = 0;                                                   // (all_records only)
    uint32_t count;
    if (all_records) seg_records(a, seg_index, first, count);
    else count = a.list_count[seg_index];
    const uint32_t* seg = a.list + (uint64_t)seg_index * a.in_seg_cap;
    if constexpr (!ALPHA) {
        // the lean build: one record at a time, pure-ACGT records only (what is left after the streaming kernel of an
        // ordinary batch is a handful of records per segment; prefetching bought nothing there, measured)
        for (uint32_t i = wib; i < count; i += wpb) {
            const uint32_t rec = all_records ? (uint32_t)first + i : seg[i] & ENTRY_REC;
            const uint64_t off = a.offsets[rec], len = a.offsets[rec + 1] - off;
            bool not_acgt = false;
            if (!rescue_direct<HASH, AUX>(a, lut, st, rec, off, len, not_acgt)) defer_record(a, seg_count, seg_index, rec, not_acgt);
        }
        return;
    }
    uint32_t c0 = wib * RESCUE_CHUNK;
    if (c0 >= count) return;
    const uint32_t step = wpb * RESCUE_CHUNK;
    RescueMeta cur = rescue_meta(a, seg, first, c0, count, all_records);
    RescueMeta nxt = rescue_meta(a, seg, first, c0 + step, count, all_records);          // (all lanes idle past the end)
    u32x4 v_next = rescue_fetch(a, cur, 0);
    for (; c0 < count; c0 += step) {
        const uint32_t m = count - c0 < RESCUE_CHUNK ? count - c0 : RESCUE_CHUNK;
        for (uint32_t l = 0; l < m; ++l) {
            const u32x4 v = v_next;
            if (l + 1 < m) v_next = rescue_fetch(a, cur, l + 1);
            else if (c0 + step < count) v_next = rescue_fetch(a, nxt, 0);
            rescue_one<HASH, AUX>(a, lut, st, seg_count, seg_index, cur, l, v);
        }
        cur = nxt;
        nxt = rescue_meta(a, seg, first, c0 + 2 * step, count, all_records);
    }
}

}  // namespace ck
