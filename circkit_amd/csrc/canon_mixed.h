// canon_mixed.h -- the per-record routines of canon_mixed_kernel: batches of mixed lengths (BASELINE config 4), ONE kernel
// over all records, one wave per record.
//
// Same reference functions as canon_core.h (lib/src/canonicalize.rs:5-63) on their common path only: a record whose
// minimal 16-symbol key is owned by one position and differs between the strands.  Everything else -- ties, periods,
// reverse-complement palindromes, gaps, bytes outside ACGTN, records that do not fit the wave's LDS slice -- is appended to
// the list of LDS stage A, whose kernel carries the general routine.  Keeping the general routine OUT of this kernel is the
// point: with it inlined next to the hot loops (round 2) the kernel was 14k instructions, spilled 55 SGPRs and a VGPR, and
// issued 1091 vector instructions per record of config 4.
//
// What is different from canon_record_mode<2> (canon_core.h):
//  * ALIGNED loads.  The strand is built from 16-byte-ALIGNED chunks of the payload (a wave's row = 1 KiB of whole cache
//    lines; the record-shaped 16-byte loads at the record's own alignment are what held the round-2 kernel at the
//    4.4 TB/s of a copy of that shape) and stored as it comes: LDS symbol index s = record position + a16, a16 = the
//    record's offset in its first chunk.  The symbols in front of the record in word 0 and behind it in the last word are
//    then replaced by the record's own tail / head (periodic extension on BOTH sides, one word more on either side), so
//    every window that starts anywhere in the stored words is the key of a real rotation.
//  * ONE scan loop for both strands.  The reverse-complement key that ends where forward word w begins is
//    funnel(rc(E[w]), rc(E[w-1]), 2b): with R = rc_word(E[w]) both strands' sixteen keys per word come from three aligned
//    LDS words -- no second pass, no mirrored address arithmetic, no cross-lane traffic.
//  * Full 16-byte stores only (the last window is pulled back to end at n, as in the register routine).
//  * NM (batches with N, MODE_ALPHA): N packed as G, canon_record_mode2n's rules (a key with an N is below its true value;
//    the winner stands if its window is N-free, or by the prefix rule).  The N are kept as ONE bit per symbol next to the
//    strand, in the strand's own coordinates (16 bits per strand word, straight from the packing step, extended periodically
//    like the strand): the scans never look at them; the winning window, the exact comparison of sharers and the output read
//    16-symbol windows of it.  The output is patched IN REGISTERS before its 16-byte stores (round 3 kept a list of N
//    positions instead -- a 50th of the LDS -- and patched single bytes behind the stores: +15 % of HBM traffic from the
//    partial writes, a vmcnt(0) per record, a list walk per winner, and no fused XXH3 for records with N).
#pragma once
#include "canon_core.h"
#include "canon_fast.h"
#include "canon_stream.h"

namespace ck {

// LDS dwords of the lean routine's strand for a record of n symbols (any alignment): pre-extension word, ceil((15 + n) / 16)
// strand words, post-extension word, one word more (NM: the N list's counter).  NM: what is left of the slice behind it is
// the N list, one 16-bit record position per N.
CK_DEV uint32_t lean_strand_dw(uint32_t n) { return 1 + (n + 30) / 16 + 2; }

struct LeanGeom { uint32_t a16, n, T, nW; };      // T = a16 + n: LDS symbol index of the record's end; nW = strand words
// NM: dwords of the N bits -- one 16-bit entry per strand word E[-1] .. E[nW] (bit 15 = the word's first symbol)
CK_DEV uint32_t lean_mask_dw(uint32_t n) { return ((n + 30) / 16 + 3 + 1) / 2; }
// the N bits of the 16 symbols at LDS symbol index s (bit 15 = the first); Mk[-1] .. Mk[nW] are valid
CK_DEV uint32_t lean_mask_window(const uint16_t* Mk, int32_t s)
{
    const uint32_t two = ((uint32_t)Mk[s >> 4] << 16) | Mk[(s >> 4) + 1];
    return (two >> (16 - ((uint32_t)s & 15))) & 0xFFFFu;
}

// the 16 symbols at LDS symbol index s (first in the top bits); E[-1] .. E[nW] are valid
// (a 64-bit shift, not funnel(): with a lane-varying shift the compiler turns funnel()'s "sh ? alignbit : hi" into an exec-mask
// region around the second LDS read and a v_mul_lo_u32 by 30 for the shift amount -- a quarter-rate instruction per window)
// (end of round 4: not the 64-bit shift either -- v_lshrrev_b64 issues at half rate and wants a register pair, tools/microbench/
// valu_rate.hip.  With q = s + 15 the window is the words E[(q >> 4) - 1], E[q >> 4] shifted right by 30 - 2 (q & 15) = 0..30 bits:
// ONE v_alignbit_b32, whose five shift bits are those of (2q) ^ 30 (canon_pair.h pair_window).  s >= 0; E[-1] is valid.)
CK_DEV uint32_t lean_window(const uint32_t* E, int32_t s)
{
    const uint32_t q = (uint32_t)s + 15;
    const uint32_t* w = E - 1 + (q >> 4);
    return alignbit(w[0], w[1], (q << 1) ^ 30u);
}

// Strand from aligned chunks; returns false when a chunk holds a byte outside the alphabet (the first and last chunk also
// hold the neighbours' bytes: a stranger there sends the record to stage A for nothing, which is harmless).
// NM: N is packed as G and the chunk's 16 N bits go to Mk[w], as the packing step yields them.
template <bool NM>
CK_DEV bool lean_build(const uint8_t* base, const LeanGeom& g, uint32_t* E, uint16_t* Mk)
{
    const uint32_t lane = lane_id();
    uint32_t bad = 0;
    constexpr int U = CK_BUILD_ROWS;
    for (uint32_t w0 = lane; w0 < g.nW; w0 += 64 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t w = w0 + 64 * u;
            if (w < g.nW) v[u] = load16(base + 16 * (uint64_t)w);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t w = w0 + 64 * u;
            if (w < g.nW) {
                uint32_t nm = 0, miss = 0;
                if constexpr (NM) { E[w] = fast_pack_n(v[u], nm, miss); Mk[w] = (uint16_t)nm; }
                else E[w] = fast_pack(v[u], miss);
                bad |= miss;
            }
        }
    }
    return ballot(bad != 0) == 0;
}

// Periodic extension on both sides (see the file comment); a handful of lanes, every value computed from the words as
// built before any is stored.
template <bool NM = false>
CK_DEV void lean_extend(const LeanGeom& g, uint32_t* E, uint16_t* Mk = nullptr)
{
    const uint32_t lane = lane_id();
    const uint32_t jl = g.nW - 1, rl = g.T - 16 * jl;                    // last word, its symbols that belong to the record (1..16)
    uint32_t val = 0, dst = 0, mval = 0;
    bool on = false;
    if (lane == 0) {                // word 0: its first a16 symbols become the record's last a16 (LDS indices n .. T)
        const uint32_t w0 = E[0];
        val = g.a16 ? bfi(~(0xFFFFFFFFu >> (2 * g.a16)), lean_window(E, (int32_t)g.n), w0) : w0;
        if constexpr (NM) mval = g.a16 ? bfi(0xFFFFu >> g.a16, Mk[0], lean_mask_window(Mk, (int32_t)g.n)) : Mk[0];
        dst = 0; on = true;
    } else if (lane == 1) {         // E[-1]: the 16 symbols in front of LDS index 0 = record positions n - 16 - a16 .., at LDS n - 16
        val = lean_window(E, (int32_t)g.n - 16);
        if constexpr (NM) mval = lean_mask_window(Mk, (int32_t)g.n - 16);
        dst = ~0u; on = true;
    } else if (lane == 2) {         // last word: behind its rl record symbols the record's head (LDS index a16 ..)
        const uint32_t wl = E[jl];
        val = rl < 16 ? bfi(~(0xFFFFFFFFu >> (2 * rl)), wl, lean_window(E, (int32_t)g.a16) >> (2 * rl)) : wl;
        if constexpr (NM) mval = rl < 16 ? bfi(0xFFFFu >> rl, lean_mask_window(Mk, (int32_t)g.a16) >> rl, Mk[jl]) : Mk[jl];
        dst = jl; on = true;
    } else if (lane == 3) {         // E[nW]: LDS indices 16 nW .. = the same symbols one period earlier
        val = lean_window(E, (int32_t)(16 * g.nW - g.n));
        if constexpr (NM) mval = lean_mask_window(Mk, (int32_t)(16 * g.nW - g.n));
        dst = g.nW; on = true;
    }
    wave_sync();
    if (on) {
        if (dst == ~0u) { E[-1] = val; if constexpr (NM) Mk[-1] = (uint16_t)mval; }
        else { E[dst] = val; if constexpr (NM) Mk[dst] = (uint16_t)mval; }
    }
    wave_sync();
}

// per-lane running minimum over the lane's words (lane l holds words l, l + 64, ...): key, the word that holds it first, and
// `second`, the smallest minimum among the lane's OTHER words -- second == key says that more than one of the lane's words
// holds the minimum (mostly ONE position seen twice: in word 0 or 1 and, as its periodic twin, in the last words, which share
// a lane when the record is a little longer than a multiple of 1024 symbols; lean_locate then walks the lane's words again),
// and the N build's prefix rule needs it anyway.  Five instructions per word and strand (v_max, v_min, v_cmp, v_cndmask,
// v_min; round 3 counted the ties: six, nine in the N build).
struct LeanBest { uint32_t key = ~0u, word, second = ~0u; };
CK_DEV void lean_update(LeanBest& b, uint32_t m, uint32_t w)
{
    const uint32_t hi = m > b.key ? m : b.key;
    b.second = hi < b.second ? hi : b.second;
    // (key starts at ~0 and word at the lane's first word: a first minimum of ~0 leaves the initial state, which names the
    // right word already)
    b.word = m < b.key ? w : b.word;
    b.key = m < b.key ? m : b.key;
}

// the keys that start (forward) / end (reverse) in word j, one per lane & 15: forward key at LDS index 16 j + b, reverse key
// of the forward window at LDS index 16 j - b
CK_DEV uint32_t lean_key_at(const uint32_t* E, uint32_t j, uint32_t b, bool fwd)
{
    const uint32_t hi = fwd ? E[j] : rc_word<2>(E[j]), lo = fwd ? E[j + 1] : rc_word<2>(E[(int32_t)j - 1]);
    return lshr64(hi, lo, 32 - 2 * b);
}

// Two rotations whose 16-symbol keys tie -- of one strand, or one of either strand (f1, f2: forward strand?) --, given by the
// record positions q1, q2 of the forward windows behind their keys: -1 / +1 as the first / second is the smaller rotation,
// 0 if they are equal (a period; across the strands: a reverse-complement palindrome).  The rotation at forward window q
// reads on as the windows at q + 16 t (forward strand) / q - 16 t, reverse-complemented (reverse strand).
// A handful of records per million-record batch of config 4 get here (a minimal 16-mer owned twice: ~1; a minimal 16-mer
// inside a reverse-complement palindrome of 18+, which both strands then own: ~6) -- and used to cost the batch a one-wave
// pass of stage A's general routine behind everything else (~40-70 us of a 1.9 ms step).
CK_DEV int lean_cmp_rot(const uint32_t* E, const LeanGeom& g, bool f1, uint32_t q1, bool f2, uint32_t q2)
{
    const uint32_t lane = lane_id();
    for (uint32_t base = 0; base < g.n; base += 64 * 16) {
        const uint32_t t = base + 16 * lane;
        uint32_t x = 0, y = 0;
        if (t < g.n) {
            const uint32_t t1 = f1 ? (q1 + t >= g.n ? q1 + t - g.n : q1 + t) : (q1 >= t ? q1 - t : q1 + g.n - t);
            const uint32_t t2 = f2 ? (q2 + t >= g.n ? q2 + t - g.n : q2 + t) : (q2 >= t ? q2 - t : q2 + g.n - t);
            x = lean_window(E, (int32_t)(t1 + g.a16));
            y = lean_window(E, (int32_t)(t2 + g.a16));
            x = f1 ? x : rc_word<2>(x);
            y = f2 ? y : rc_word<2>(y);
        }
        const uint64_t bal = ballot(x != y);
        if (bal) {
            const uint32_t l = (uint32_t)ffs64(bal);
            return readlane(x, l) < readlane(y, l) ? -1 : 1;
        }
    }
    return 0;
}

// The record position of the forward window behind one strand's minimal rotation, from the scan's per-lane results (b) and
// the wave minimum M: mostly one owner, possibly seen again in the extension at either end (same record position).  Several
// owners (pure build): the smallest of their rotations, as long as they are few and no two of them equal.  -1: stage A's.
// K16 (pure builds, round 4): the scan compared 8-symbol prefixes (word_min_key16), M is one of them -- the positions that share
// it are the rivals, settled by full comparison below like any other tie.
template <bool NM, bool K16 = false>
CK_DEV int32_t lean_locate(const uint32_t* E, const LeanGeom& g, bool fwd, uint32_t M, const LeanBest& b)
{
    const uint32_t lane = lane_id();
    uint64_t hm = ballot(lane < g.nW && b.key == M);              // (lanes without a word keep the initial key)
    if (popc64(hm) > 8) return -1;
    int32_t Q = -1;
    uint32_t rivals = 0;
    while (hm) {
        const uint32_t l = (uint32_t)ffs64(hm);
        hm &= hm - 1;
        // the owner lane's word with the minimum -- or, when two of its words hold it, every word of the lane from that one on
        const uint32_t j0 = readlane(b.word, l), j1 = readlane(b.second, l) != M ? j0 + 1 : g.nW;
        for (uint32_t j = j0; j < j1; j += 64) {
            const uint32_t key = lean_key_at(E, j, lane & 15, fwd);
            uint64_t pm = ballot(lane < 16 && (K16 ? key >> 16 : key) == M);
            if (NM && j == j0 && popc64(pm) != 1) return -1;
            while (pm) {
                const int32_t bit = ffs64(pm);
                pm &= pm - 1;
                int32_t q = (fwd ? (int32_t)(16 * j) + bit : (int32_t)(16 * j) - bit) - (int32_t)g.a16;     // record position of the forward window
                q = q < 0 ? q + (int32_t)g.n : (q >= (int32_t)g.n ? q - (int32_t)g.n : q);
                if (Q < 0) { Q = q; continue; }
                if (q == Q) continue;
                if (NM || ++rivals > 8) return -1;
                const int c = lean_cmp_rot(E, g, fwd, (uint32_t)q, fwd, (uint32_t)Q);
                if (c == 0) return -1;
                if (c < 0) Q = q;
            }
        }
    }
    return Q;
}

// N build, the rare case (about one record in fifty at 1 % N): the packed minimal key's window holds an N and other
// rotations share the symbols in front of it, so the packed order no longer decides.  Every rotation that does NOT share
// them is above the winner in the true order too (canon_record_mode2n's argument), so the true minimum is among the sharers:
// the rotations of either strand whose packed key is <= thr.  They are collected (one more pass over the words, a candidate
// list of LEAN_CAND_MAX in LDS) and compared EXACTLY on their first 32 symbols, ranked A 0, C 2, G 4, N 5, T 6 -- packed
// symbol x 2, or 5 where the N list marks the position.  Returns rotation | strand << 31 (record position of the forward
// window behind the key), or ~0: too many sharers, or two of them equal on 32 symbols -- stage A's.
// Before this, such a record cost stage A a 2-bit attempt, then the 4-bit mode on one wave behind everything else: 0.38 ms
// of config 4's 2.4 ms step with 1 % N, for 1 % of its bytes.
constexpr uint32_t LEAN_CAND_MAX = 16, LEAN_CAND_DW = LEAN_CAND_MAX + 2;        // + counter, + the candidate's N mask
// (g BY VALUE: a reference parameter of a non-inlined function makes the caller keep the struct in scratch memory -- one 1 KiB
// scratch store per record of the N build, +15 % of its write traffic by PMC, for a call one record in fifty makes)
CK_DEV_NOINLINE uint32_t lean_resolve_n(const uint32_t* E, const LeanGeom g, const uint16_t* Mk, uint32_t thr, uint32_t* cand)
{
    const uint32_t lane = lane_id(), n = g.n;
    uint32_t* count = cand + LEAN_CAND_MAX;
    if (lane == 0) *count = 0;
    wave_sync();
    // forward windows: LDS indices a16 .. T - 1.  Reverse keys: the loop yields the forward windows at 16 w - b, i.e. indices
    // -15 .. 16 (nW - 1); n consecutive ones of them, (16 (nW - 1) - n, 16 (nW - 1)], are every rotation once
    const int32_t r_hi = (int32_t)(16 * (g.nW - 1)), r_lo = r_hi - (int32_t)n;
    for (uint32_t w = lane; w < g.nW; w += 64) {
        const uint32_t cur = E[w], nxt = E[w + 1], R = rc_word<2>(cur), Rp = rc_word<2>(E[(int32_t)w - 1]);
        const bool hit_f = word_min_key<2>(cur, nxt) <= thr, hit_r = word_min_key<2>(R, Rp) <= thr;
        if (hit_f || hit_r) {
#pragma nounroll
            for (uint32_t b = 0; b < 16; ++b) {
                const uint32_t sf = 16 * w + b;
                const int32_t sr = (int32_t)(16 * w) - (int32_t)b;
                if (hit_f && funnel(cur, nxt, 2 * b) <= thr && sf >= g.a16 && sf < g.T) {
                    const uint32_t k = lds_atomic_inc(count);
                    if (k < LEAN_CAND_MAX) cand[k] = sf - g.a16;
                }
                if (hit_r && funnel(R, Rp, 2 * b) <= thr && sr > r_lo && sr <= r_hi) {
                    const int32_t q0 = sr - (int32_t)g.a16, q = q0 < 0 ? q0 + (int32_t)n : q0;
                    const uint32_t k = lds_atomic_inc(count);
                    if (k < LEAN_CAND_MAX) cand[k] = (uint32_t)q | 0x80000000u;
                }
            }
        }
    }
    wave_sync();
    const uint32_t c = *count;
    if (c == 0 || c > LEAN_CAND_MAX) return ~0u;
    uint32_t best = ~0u, best_rank = 0;
    for (uint32_t k = 0; k < c; ++k) {
        const uint32_t v = cand[k], q = v & 0x7FFFFFFFu;
        const bool f = (v >> 31) == 0;
        // which of the rotation's first 32 symbols are N: symbol i is forward position q + i, or (reverse strand) q + 15 - i --
        // the N bits of the same two windows the symbols come from (bit 15 = the window's first forward position)
        // the symbol itself: two 16-symbol windows (the second one 16 further on the strand that is being read)
        const uint32_t q2 = f ? (q + 16 >= n ? q + 16 - n : q + 16) : (q >= 16 ? q - 16 : q + n - 16);
        uint32_t x0 = lean_window(E, (int32_t)(q + g.a16)), x1 = lean_window(E, (int32_t)(q2 + g.a16));
        uint32_t m0 = lean_mask_window(Mk, (int32_t)(q + g.a16)), m1 = lean_mask_window(Mk, (int32_t)(q2 + g.a16));
        if (!f) { x0 = rc_word<2>(x0); x1 = rc_word<2>(x1); }
        else { m0 = bitrev(m0) >> 16; m1 = bitrev(m1) >> 16; }                // bit i = symbol i of the rotation, either way
        const uint32_t nm = m0 | (m1 << 16);
        const uint32_t sym = ((lane & 16) ? x1 : x0) >> (30 - 2 * (lane & 15)) & 3u;
        const uint32_t rank = lane < 32 ? (((nm >> lane) & 1u) ? 5u : 2u * sym) : 0u;
        if (best == ~0u) { best = v; best_rank = rank; continue; }
        const uint64_t diff = ballot(rank != best_rank);
        if (diff == 0) return ~0u;
        const uint32_t l = (uint32_t)ffs64(diff);
        if (readlane(rank, l) < readlane(best_rank, l)) { best = v; best_rank = rank; }
    }
    return best;
}

// ---- XXH3-64 fused into the lean routine's output loop (HASH builds, pure records of more than 240 symbols) ----------------
// Replaces `xxh3_64(canonicalized)` (src/uniq.rs:45) for the records of a mixed-length batch: `circkit uniq` on contigs of
// mixed lengths went rescue pass + general LDS routine + a pass of the xxh3 kernel over the output (round 2 / 3 first half:
// 5.0 ms for config 4's batch against 1.8 ms for the bytes alone).  The output loop's row of 64 chunks IS one 1024-byte XXH3
// block -- lane = (stripe lane >> 2, accumulator pair lane & 3), the layout of canon_fast.h fast_hash -- so every row adds
// its cells' products to the accumulators, full blocks are followed by XXH3's scramble, and the last stripe, the merge and
// the avalanche follow behind the loop.  Per-pair constants live in a 64-dword LDS table (lean_hash_table_init: last-stripe
// and merge secrets, initial accumulators, scramble secrets); the per-lane stripe secrets stay in registers.
// N builds: nibble of N bits (bit 3 = the first of four output bytes) -> 'G' ^ 'N' = 0x09 (forward strand, entries 0..15) or
// 'C' ^ 'N' = 0x0D (reverse strand, entries 16..31) in the bytes the bits name
constexpr uint32_t LEAN_LUTN_DW = 32;
CK_DEV void lean_lutn_init(uint32_t* tab, uint32_t tid, uint32_t nthreads)
{
    for (uint32_t x = tid; x < 32; x += nthreads) {
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) o |= ((x >> (3 - k)) & 1u) << (8 * k);
        tab[x] = o * (x < 16 ? 0x09u : 0x0Du);
    }
}
constexpr uint32_t LEAN_HASH_TABLE_DW = 64;          // per pair j (16 dwords): l0 l1 m0 m1 i0 i1 sc0 sc1
CK_DEV void lean_hash_table_init(uint32_t* tab, uint32_t tid)
{
    if (tid < 4) {
        const uint32_t j = tid;
        const uint64_t v[8] = { xsec64(121 + 16 * j), xsec64(129 + 16 * j), xsec64(11 + 16 * j), xsec64(19 + 16 * j),
                                j == 0 ? XP32_3 : j == 1 ? XP64_2 : j == 2 ? XP64_4 : XP64_5,
                                j == 0 ? XP64_1 : j == 1 ? XP64_3 : j == 2 ? XP32_2 : XP32_1,
                                xsec64(128 + 16 * j), xsec64(136 + 16 * j) };
#pragma unroll
        for (int k = 0; k < 8; ++k) { tab[16 * j + 2 * k] = (uint32_t)v[k]; tab[16 * j + 2 * k + 1] = (uint32_t)(v[k] >> 32); }
    }
}
CK_DEV uint64_t lean_hash_const(const uint32_t* tab, uint32_t k) { const uint32_t j = lane_id() & 3; return ((uint64_t)tab[16 * j + 2 * k + 1] << 32) | tab[16 * j + 2 * k]; }
// the register routine's per-lane constants other than the stripe secrets, fetched from the table when a short record is at
// hand instead of living in sixteen registers across the kernel's loop
CK_DEV void lean_hash_refill(FastHashConst& hc, const uint32_t* tab)
{
    hc.l0 = lean_hash_const(tab, 0); hc.l1 = lean_hash_const(tab, 1);
    hc.m0 = lean_hash_const(tab, 2); hc.m1 = lean_hash_const(tab, 3);
    hc.i0 = lean_hash_const(tab, 4); hc.i1 = lean_hash_const(tab, 5);
}
struct LeanHash { uint64_t a0, a1, k0, k1; };         // accumulators of pair lane & 3 (meaningful where (lane & 15) >= 12), the lane's stripe secrets
// one row (= block `b`) of cells: `cell` = this lane's 16 output bytes at chunk 64 b + lane, counted iff its stripe is one of
// the `total_stripes` full stripes in front of the last one
CK_DEV void lean_hash_row(LeanHash& h, const uint32_t* tab, u32x4 cell, uint32_t w, uint32_t total_stripes, bool scramble)
{
    const uint64_t d0 = ((uint64_t)cell.y << 32) | cell.x, d1 = ((uint64_t)cell.w << 32) | cell.z;
    const uint64_t x0 = d0 ^ h.k0, x1 = d1 ^ h.k1;
    const bool on = (w >> 2) < total_stripes;
    uint64_t c0 = on ? d1 + (uint64_t)(uint32_t)x0 * (x0 >> 32) : 0;
    uint64_t c1 = on ? d0 + (uint64_t)(uint32_t)x1 * (x1 >> 32) : 0;
    dpp_rowsum4_u64x2(c0, c1);
    c0 = shfl_xor_add64(c0, 16); c1 = shfl_xor_add64(c1, 16);
    c0 = shfl_xor_add64(c0, 32); c1 = shfl_xor_add64(c1, 32);
    h.a0 += c0; h.a1 += c1;
    if (scramble) {                                   // (uniform) a full block: scramble with the last 64 secret bytes
        h.a0 = (h.a0 ^ (h.a0 >> 47) ^ lean_hash_const(tab, 6)) * XP32_1;
        h.a1 = (h.a1 ^ (h.a1 >> 47) ^ lean_hash_const(tab, 7)) * XP32_1;
    }
}
// last stripe (the final 64 bytes: pair j = bytes [n - 64 + 16 j, + 16), given as `cell`), merge, avalanche; same in every lane
CK_DEV uint64_t lean_hash_finish(LeanHash& h, const uint32_t* tab, u32x4 cell, uint32_t n)
{
    const uint64_t d0 = ((uint64_t)cell.y << 32) | cell.x, d1 = ((uint64_t)cell.w << 32) | cell.z;
    const uint64_t x0 = d0 ^ lean_hash_const(tab, 0), x1 = d1 ^ lean_hash_const(tab, 1);
    h.a0 += d1 + (uint64_t)(uint32_t)x0 * (x0 >> 32);
    h.a1 += d0 + (uint64_t)(uint32_t)x1 * (x1 >> 32);
    uint64_t r = xfold(h.a0 ^ lean_hash_const(tab, 2), h.a1 ^ lean_hash_const(tab, 3));
    r = dpp_quadsum_u64(r);
    const uint64_t v = xaval3((uint64_t)n * XP64_1 + r);
    return ((uint64_t)readlane((uint32_t)(v >> 32), 15) << 32) | readlane((uint32_t)v, 15);
}

// The scan of both strands in one pass over the words + the decision: strand (fwd) and the record position Q of the forward
// window behind the winning key; 0, or 2 = stage A's (a tie the routine does not settle).  K16: 8-symbol prefixes, two per
// v_pk_min_u16 (15 instructions per word and strand instead of 23); positions that share the minimal prefix -- one record in
// eight at 4 kb, one in two at 20 kb of random sequence -- are lean_locate's rivals, compared in full there.  The N build's
// prefix rule argues with 16-symbol keys and keeps them.
template <bool NM, bool K16>
CK_DEV int lean_scan_locate(const uint32_t* E, const LeanGeom& g, LeanBest& bF, LeanBest& bC, uint32_t& MF, uint32_t& MC, bool& fwd, int32_t& Q)
{
    const uint32_t lane = lane_id();
    bF = LeanBest{}; bC = LeanBest{};
    bF.word = bC.word = lane;
    for (uint32_t w = lane; w < g.nW; w += 64) {
        const uint32_t cur = E[w], nxt = E[w + 1], prv = E[(int32_t)w - 1];
        if constexpr (K16) {
            lean_update(bF, word_min_key16(cur, nxt), w);
            lean_update(bC, word_min_key16(rc_word<2>(cur), rc_word<2>(prv)), w);
        } else {
            lean_update(bF, word_min_key<2>(cur, nxt), w);
            lean_update(bC, word_min_key<2>(rc_word<2>(cur), rc_word<2>(prv)), w);
        }
    }
    wave_min2_u32(bF.key, bC.key, MF, MC);
    fwd = MF < MC;
    if (MF == MC) {
        // equal minimal keys: the two minimal rotations are compared in full (lib/src/canonicalize.rs:58-62: forward only if
        // strictly smaller; equal = a reverse-complement palindrome, either strand's bytes are the same) -- pure build only
        if (NM) return 2;
        const int32_t QF = lean_locate<NM, K16>(E, g, true, MF, bF), QC = lean_locate<NM, K16>(E, g, false, MC, bC);
        if (QF < 0 || QC < 0) return 2;
        fwd = lean_cmp_rot(E, g, true, (uint32_t)QF, false, (uint32_t)QC) < 0;
        Q = fwd ? QF : QC;
    } else {
        Q = fwd ? lean_locate<NM, K16>(E, g, true, MF, bF) : lean_locate<NM, K16>(E, g, false, MC, bC);
        if (Q < 0) return 2;
    }
    return 0;
}

// One record of more than FAST_MAX_N symbols in the wave's slice.  0: done; 1: not this routine's alphabet (stage A is told);
// 2: pure as far as seen, but a tie / equal strands (stage A's general routine); 3: not tried -- no room in the slice, or (N build)
// chunks outside the payload.
#ifndef CK_LEAN_MIN_PREFIX
#define CK_LEAN_MIN_PREFIX 6      // prefix rule: with fewer deciding symbols than this some other rotation shares them anyway (4^6 against ~10^4 rotations)
#endif
template <bool NM, bool HASH = false>
CK_DEV int canon_lean_record(const CanonArgs& a, uint32_t rec, uint64_t off, uint32_t n, uint64_t payload_end, uint32_t* slice,
                             const uint32_t* lut, const uint32_t* htab = nullptr, uint64_t hk0 = 0, uint64_t hk1 = 0, const uint32_t* lutn = nullptr)
{
    const uint32_t lane = lane_id();
    LeanGeom g;
    g.a16 = ((uint32_t)(uintptr_t)a.bytes + (uint32_t)off) & 15;
    g.n = n; g.T = g.a16 + n; g.nW = (g.T + 15) >> 4;
    // the aligned chunks must lie inside the payload: not in front of it (a misaligned payload's first record), not
    // behind it (the batch's last record(s)).  Such a record is built from 16-byte loads at its own alignment instead
    // (build_packed<2>: the last word re-reads the record's last 16 bytes), i.e. as a record with a16 = 0 -- the pure build
    // only; the N build leaves it to stage A.
    const bool inside = off >= g.a16 && off - g.a16 + 16ull * g.nW <= payload_end;
    if (!inside) {
        if (NM) return 3;
        g.a16 = 0; g.T = n; g.nW = (n + 15) >> 4;
    }
    const uint32_t strand_dw = lean_strand_dw(n);
    if (strand_dw + (NM ? lean_mask_dw(n) + LEAN_CAND_DW : 0) > a.slice_dw) return 3;
    uint32_t* E = slice + 1;
    // (NM: the N bits behind the strand, lean_resolve_n's candidate list in the slice's last LEAN_CAND_DW dwords)
    uint16_t* Mk = reinterpret_cast<uint16_t*>(slice + strand_dw) + 1;
    if (inside) {
        if (!lean_build<NM>(a.bytes + (off - g.a16), g, E, Mk)) return 1;
    } else {
        if (!build_packed<2>(a.bytes + off, n, E, E)) return 1;         // (its own extension behind the end; lean_extend repeats it)
    }
    lean_extend<NM>(g, E, Mk);
    // both strands' minimal keys in one pass over the words, the winner located (lean_scan_locate).  Pure builds: first on
    // 8-symbol prefixes; a record whose minimal prefix has too many owners for lean_locate (a run of 16 A, say: nine positions
    // share AAAAAAAA) is scanned again with 16-symbol keys -- low complexity costs that record a second pass, not a trip to stage A.
    LeanBest bF, bC;
    uint32_t MF = 0, MC = 0;
    bool fwd = false;
    int32_t Q = -1;
    int rs = lean_scan_locate<NM, !NM>(E, g, bF, bC, MF, MC, fwd, Q);
    if constexpr (!NM) { if (rs != 0) rs = lean_scan_locate<NM, false>(E, g, bF, bC, MF, MC, fwd, Q); }
    if (rs != 0) return rs;
    const uint32_t M = fwd ? MF : MC;
    const uint32_t bkey = fwd ? bF.key : bC.key, bword = fwd ? bF.word : bC.word, bsecond = fwd ? bF.second : bC.second;
    const uint64_t owners = ballot(lane < g.nW && bkey == M);
    // rotation index on the winning strand; the forward window behind reverse position p starts at n - 16 - p
    uint32_t idx = fwd ? (uint32_t)Q : (uint32_t)((int32_t)n - 16 - Q < 0 ? 2 * (int32_t)n - 16 - Q : (int32_t)n - 16 - Q);
    if constexpr (NM) {
        // An N inside the winning window -- the forward positions Q .. Q + 15 on either strand -- is where the packed key
        // differs from the true one: canon_record_mode2n's prefix rule (canon_core.h).  Offset of the first N in the
        // winner's own reading direction: d on the forward strand, 15 - d on the reverse one.
        const uint32_t mw = lean_mask_window(Mk, Q + (int32_t)g.a16);
        const uint32_t first = mw == 0 ? 16u : (fwd ? (uint32_t)clz32(mw) - 16u : (uint32_t)ffs32(mw));
        if (first < 16) {
            const uint32_t plen = fwd ? first + 1 : first;
            if (plen == 0) return 1;
            // "No other rotation of either strand shares the winner's first plen packed symbols" = no other key is <= thr (M
            // is the smallest of all).  The scan has kept every lane's smallest word minimum on both strands and, on top, the
            // smallest among its OTHER words: nothing but the owner words may reach down to thr, and inside them only the
            // winner itself -- decided without another pass over the strand.
            const uint32_t sh = 32 - 2 * plen, thr = M | (sh ? 0xFFFFFFFFu >> (32 - sh) : 0u);
            const bool mine = ((owners >> lane) & 1) != 0;
            const uint32_t lowest_other = fwd ? bC.key : bF.key, lowest_w = mine ? bsecond : bkey;
            bool alone = plen >= CK_LEAN_MIN_PREFIX && ballot(lowest_other <= thr || lowest_w <= thr) == 0;
            for (uint64_t h2 = owners; alone && h2; h2 &= h2 - 1) {
                const uint32_t l = (uint32_t)ffs64(h2), j = readlane(bword, l);
                if (readlane(bsecond, l) == M || popc64(ballot(lane < 16 && lean_key_at(E, j, lane & 15, fwd) <= thr)) != 1) alone = false;
            }
            if (!alone) {
                // other rotations share those symbols: the true minimum is one of the sharers (lean_resolve_n)
                const uint32_t v = lean_resolve_n(E, g, Mk, thr, slice + a.slice_dw - LEAN_CAND_DW);
                if (v == ~0u) return 1;
                fwd = (v >> 31) == 0;
                Q = (int32_t)(v & 0x7FFFFFFFu);
                idx = fwd ? (uint32_t)Q : (uint32_t)((int32_t)n - 16 - Q < 0 ? 2 * (int32_t)n - 16 - Q : (int32_t)n - 16 - Q);
            }
        }
    }
    // XXH3 fused (HASH builds): records of more than 240 symbols, with or without N (round 4: the N are patched into the cells
    // in registers, so the hash sees the bytes that are stored); the short-input classes are the xxh3 pass's, from the bytes
    // or, when no bytes were asked for, from the view
    const bool fused = HASH && a.out_hash != nullptr && n > 240;
    if (HASH && !fused && a.out_view && lane == 0) a.out_view[rec] = fwd ? idx : (idx | 0x80000000u);
    if (a.out_bytes || fused) {
        uint8_t* out = a.out_bytes ? a.out_bytes + off : nullptr;
        // the 16 output bytes at output offset o (the strand is the same for the whole record: two loops, a select inside one
        // loop computed both strands' words)
        const auto cell_at = [&](uint32_t o, bool f) {
            uint32_t p = idx + o;
            p = p >= n ? p - n : p;
            // LDS index of the forward window behind the 16 bytes (reverse strand: read backwards)
            const uint32_t s = f ? p + g.a16 : (p + 16 <= n ? n - 16 - p : 2 * n - 16 - p) + g.a16;
            u32x4 c = fast_decode(lut, f ? lean_window(E, (int32_t)s) : rc_word<2>(lean_window(E, (int32_t)s)));
            if constexpr (NM) {
                // the decoded G (forward) / C (reverse) of every N becomes N: four bits of the window's mask -> the XOR constant
                // in the bytes they name, by a 16-entry table per strand (lean_lutn_init); output byte j is the window's symbol
                // j (forward) or 15 - j (reverse).  No branch around it: at 1 % N some lane of a row always has one.
                uint32_t m = lean_mask_window(Mk, (int32_t)s);
                if (!f) m = bitrev(m) >> 16;
                const uint32_t* tab = lutn + (f ? 0 : 16);
                c.x ^= tab[m >> 12]; c.y ^= tab[(m >> 8) & 15]; c.z ^= tab[(m >> 4) & 15]; c.w ^= tab[m & 15];
            }
            return c;
        };
        if (!fused) {
            if (fwd) { for (uint32_t w = lane; 16 * w < n; w += 64) { const uint32_t o = 16 * w + 16 <= n ? 16 * w : n - 16; store16(out + o, cell_at(o, true)); } }
            else { for (uint32_t w = lane; 16 * w < n; w += 64) { const uint32_t o = 16 * w + 16 <= n ? 16 * w : n - 16; store16(out + o, cell_at(o, false)); } }
        } else if constexpr (HASH) {
            LeanHash h;
            h.a0 = lean_hash_const(htab, 4); h.a1 = lean_hash_const(htab, 5);
            h.k0 = hk0; h.k1 = hk1;                                                // (the lane's stripe secrets: FastHashConst::k0 / k1)
            const uint32_t total_stripes = (n - 1) >> 6, nb = (n - 1) >> 10;
            for (uint32_t b = 0; b <= nb; ++b) {                                   // (uniform: a row of 64 chunks = one XXH3 block)
                const uint32_t w = 64 * b + lane;
                u32x4 cell{ 0, 0, 0, 0 };
                if (16 * w < n) {
                    const uint32_t o = 16 * w + 16 <= n ? 16 * w : n - 16;         // the last window is pulled back to end at n
                    cell = fwd ? cell_at(o, true) : cell_at(o, false);
                    if (out) store16(out + o, cell);
                }
                lean_hash_row(h, htab, cell, w, total_stripes, b < nb);           // (a counted cell is a full chunk: o = 16 w)
            }
            const u32x4 last = fwd ? cell_at(n - 64 + 16 * (lane & 3), true) : cell_at(n - 64 + 16 * (lane & 3), false);
            const uint64_t hv = lean_hash_finish(h, htab, last, n);
            if (lane == 0) { a.out_hash[rec] = hv; a.hashed[rec] = 1; }
        }
    }
    return 0;
}

// A record's offsets and -- if it is one of 48..1008 symbols, the register routine's -- the 16 bytes this lane packs.
// (Fetching them while the record BEFORE is being processed -- a short record is two dependent round trips, offsets -> bytes,
// in front of ~500 cycles of work -- was tried: nothing gained on config 4, 1.87 -> 1.89 ms; the CU's 24 waves hide it, and
// the N build lost 9 % to the registers.)
struct MixedNext { uint64_t off, len; u32x4 v; };
CK_DEV MixedNext mixed_fetch(const CanonArgs& a, uint32_t rec)
{
    MixedNext m;
    m.off = a.offsets[rec];
    m.len = a.offsets[rec + 1] - m.off;
    m.v = u32x4{ 0, 0, 0, 0 };
    if (m.len >= FAST_MIN_N && m.len <= FAST_MAX_N) {
        const uint32_t n = (uint32_t)m.len, t = lane_id();
        m.v = load16(a.bytes + m.off + (t >= (n >> 4) ? n - 16 : 16 * t));      // lanes past the last full word re-read the record's last 16 bytes
    }
    return m;
}
// XXH3-64 of a record of <= 240 symbols as a view of the input; a call of its own (inlined, the short-input recipe cost the
// kernel around it 35 spilled registers)
CK_DEV_NOINLINE uint64_t mixed_short_hash(const uint8_t* s, uint32_t n, uint32_t view, const uint8_t* comp)
{
    return xxh3_short(XView{ s, n, view & 0x7FFFFFFFu, (view >> 31) != 0, comp }, n);
}
// a pure-ACGT record of 48..1008 symbols through the register routine (canon_stream.h rescue_direct with the bytes at hand)
template <bool HASH>
CK_DEV bool mixed_short(const CanonArgs& a, const uint32_t* lut, RescueState<HASH, false>& st, uint32_t rec, const MixedNext& m, bool& not_acgt, const uint32_t* htab)
{
    if constexpr (HASH) lean_hash_refill(st.hc, htab);
    const uint32_t n = (uint32_t)m.len, nwf = n >> 4, t = lane_id();
    uint32_t miss;
    uint32_t F = fast_pack(m.v, miss);
    F <<= t >= nwf ? ((16 - (n & 15)) & 15) * 2 : 0;
    const uint64_t bad = ballot(miss != 0);
    not_acgt = bad != 0;
    uint32_t view = 0;
    const bool done = fast_canon<HASH, false>(a, lut, st.hc, st.shape, rec, m.off, n, F, bad, nullptr, 0, &view);
    if constexpr (HASH) {
        // XXH3's short-input classes (<= 240 bytes) are not fused into the register routine: hashed right here from the view of
        // the input (the bytes are in the cache) instead of by the xxh3 pass behind everything -- config 4 has 4 % such records,
        // and the pass took 0.29 ms of a 2.7 ms step for them
        if (done && n <= 240 && a.out_hash) {
            const uint64_t hv = mixed_short_hash(a.bytes + m.off, n, view, a.comp_lut);
            if (t == 0) { a.out_hash[rec] = hv; a.hashed[rec] = 1; }
        }
    }
    return done;
}
// the same over ACGTN (the rescue pass's rescue_one without the list): the N-mask variant of fast_canon, then the 4-bit
// routine for what it refuses
template <bool HASH>
CK_DEV bool mixed_short_n(const CanonArgs& a, const uint32_t* lut, RescueState<HASH, false>& st, uint32_t rec, const MixedNext& m, const uint32_t* htab)
{
    if constexpr (HASH) lean_hash_refill(st.hc, htab);
    const uint32_t n = (uint32_t)m.len, nwf = n >> 4, t = lane_id();
    const uint32_t tail_syms = t >= nwf ? (16 - (n & 15)) & 15 : 0;       // the tail lane's symbols move up by this much
    uint32_t nm, miss;
    const uint32_t F = fast_pack_n2(m.v, nm, miss) << (2 * tail_syms);
    nm <<= 2 * tail_syms;
    const uint64_t bad = ballot(miss != 0), with_n = ballot(nm != 0);
    bool done = false;
    if (bad == 0) {
        // (HASH builds: the N-mask variant of the register routine writes bytes only -- records with an N take the 4-bit
        // routine, whose bytes the xxh3 pass hashes)
        if (with_n == 0) done = fast_canon<HASH, false>(a, lut, st.hc, st.shape, rec, m.off, n, F, 0);
        else if constexpr (!HASH) done = fast_canon<false, false, false, false, true>(a, lut, st.hc, st.shape, rec, m.off, n, F, 0, nullptr, nm);
    }
    if (!done && (bad | with_n) != 0) {
        uint32_t H, L, bad4;
        fast_pack4(m.v, H, L, bad4);
        const uint64_t x = ((((uint64_t)H) << 32) | L) << (4 * tail_syms);
        done = fast_canonw<4, HASH, false>(a, lut, st.hc, rec, m.off, n, (uint32_t)(x >> 32), (uint32_t)x, ballot(bad4 != 0) != 0);
    }
    return done;
}

// One wave's share of segment `sgm` of a mode-3 batch (seg_records), every wpb-th record from wib on.
template <bool NM, bool HASH = false>
CK_DEV void canon_mixed_segment(const CanonArgs& a, uint32_t* slice, const uint32_t* lut, RescueState<HASH, false>& st,
                                uint32_t* blk_count, uint32_t sgm, uint32_t wib, uint32_t wpb, uint64_t payload_end, const uint32_t* htab = nullptr,
                                const uint32_t* lutn = nullptr)
{
    uint64_t first;
    uint32_t count;
    seg_records(a, sgm, first, count);
    // (handing the records out one at a time through an LDS counter, so that no wave waits for the one that drew the long
    // records, measured slower: 1.86 -> 1.91 ms)
    for (uint32_t i = wib; i < count; i += wpb) {
        const uint32_t rec = (uint32_t)first + i;
        const MixedNext cur = mixed_fetch(a, rec);
        const uint64_t off = cur.off, len = cur.len;
        // ruled: the N-mask rule with its prefix rule has had the record (entry bit 30) -- the lean routine's; NOT the register
        // routine's N variant, which refuses every N among the deciding symbols where the tiers' N-mask mode may still succeed
        bool not_acgt = false, tried = false, ruled = false;
        if (len >= FAST_MIN_N && len <= FAST_MAX_N) {
            if constexpr (NM) { if (mixed_short_n<HASH>(a, lut, st, rec, cur, htab)) continue; tried = true; }
            else { if (mixed_short<HASH>(a, lut, st, rec, cur, not_acgt, htab)) continue; }
        }
        // longer records -- and (pure build) the few short ones the register routine leaves (a tied minimal key, a minimal key
        // both strands own): the lean LDS routine
        if (len >= FAST_MIN_N && len < (1ull << 31) && !not_acgt && !tried) {
            const int r = canon_lean_record<NM, HASH>(a, rec, off, (uint32_t)len, payload_end, slice, lut, htab, st.hc.k0, st.hc.k1, lutn);
            wave_sync();                                    // every lane is done with the slice before the next record's build
            if (r == 0) continue;
            not_acgt = NM || r == 1;
            ruled = NM && r != 3;
        }
        not_acgt = not_acgt || tried;
        defer_record(a, blk_count, sgm, rec, not_acgt, ruled);
    }
}

}  // namespace ck
