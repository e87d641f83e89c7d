// canon_fast.h -- the per-record routine of the streaming kernel of the canonicalize path.
//
// Handles the records that make up BASELINE's headline workload -- pure ACGT, 48..1008 bases -- with one
// packed word per lane held in REGISTERS (no LDS memory; cross-lane access by ds_bpermute / v_readlane,
// reductions by DPP) and straight-line code.  How the bytes reach the lanes (workgroup-staged LDS images, the
// software pipeline) is canon_stream.h.  Anything else (other alphabets, longer or shorter records, a repeated
// minimal key, equal minimal keys on both strands) is appended to a list for the general LDS kernel of
// canon_core.h.  Same reference functions as there (lib/src/canonicalize.rs:5-63).
#pragma once
#include "canon_core.h"
#include "xxh3_core.h"

namespace ck {

constexpr uint32_t FAST_MIN_N = 48, FAST_MAX_N = 1008;

// the 16 symbols at cyclic symbol position p (lane-varying, p < 2n) of a one-word-per-lane strand
CK_DEV uint32_t reg_sym_word(uint32_t E, uint32_t p, uint32_t n)
{
    p = p >= n ? p - n : p;
    const uint32_t wi = p >> 4, sh = (p & 15) * 2;
    return lshr64(shfl(E, wi), shfl(E, wi + 1), 32 - sh);
}

// Position of the minimal key M, and whether exactly one valid position owns it.  hm = lanes whose word
// minimum equals M.  Straight-line for the common single-hit case; a second hit (the duplicate of word 0's
// first positions behind the record end, or a real tie) takes one extra round; more hits = not unique.
// shv = 32 - 2*(lane & 15).
CK_DEV uint32_t fast2_locate(uint32_t E, uint32_t En, uint64_t hm, uint32_t M, uint32_t n, uint32_t shv, bool& unique)
{
    const uint32_t l = (uint32_t)ffs64(hm);                                 // hm != 0: some valid word owns the minimum
    const uint32_t k = lshr64(readlane(E, l), readlane(En, l), shv);
    const uint32_t left = n - 16 * l;                                       // valid positions in word l (>= 1)
    uint32_t pm = (uint32_t)ballot(k == M) & low_mask16(left);
    uint32_t cnt = (uint32_t)popc32(pm);
    uint32_t pos = 16 * l + (uint32_t)ffs32_or_neg(pm);                     // pm == 0: garbage, and cnt says so
    if ((uint32_t)popc64(hm) != 1u) {                                       // rare
        hm &= hm - 1;
        const uint32_t l2 = (uint32_t)ffs64(hm);
        const uint32_t k2 = lshr64(readlane(E, l2), readlane(En, l2), shv);
        const uint32_t left2 = n - 16 * l2;
        const uint32_t pm2 = (uint32_t)ballot(k2 == M) & low_mask16(left2);
        cnt += (uint32_t)popc32(pm2) + ((hm & (hm - 1)) ? 2u : 0u);         // a third hit lane: give up
        if (pm == 0) pos = 16 * l2 + (uint32_t)ffs32_or_neg(pm2);
    }
    unique = cnt == 1;
    return pos;
}

// records the streaming kernel takes: pure ACGT (checked while packing), one 16-symbol word per lane
CK_DEV bool fast_eligible(uint32_t n) { return n - FAST_MIN_N <= FAST_MAX_N - FAST_MIN_N; }

// ---- XXH3-64 fused into the streaming kernel (records of 241..1008 bytes: one 1024-byte block, no scramble) ----
// Replaces `xxh3_64(canonicalized)` (src/uniq.rs:45) without re-reading the canonical bytes: lane t already holds
// bytes [16t, 16t+16) of the output = cell (stripe t>>2, accumulator pair t&3) of XXH3's long-input loop.
// Everything that depends only on the lane is computed once per wave.
struct FastHashConst {
    uint64_t k0, k1;    // secret words of this lane's cell:            offset 8*(t>>2) + 16*(t&3), +8
    uint64_t l0, l1;    // secret words of the last stripe, pair t&3:   offset 121 + 16*(t&3), +8
    uint64_t m0, m1;    // secret words of the final merge, pair t&3:   offset 11 + 16*(t&3), +8
    uint64_t i0, i1;    // accumulator initial values of pair t&3
};
CK_DEV FastHashConst fast_hash_const()
{
    const uint32_t t = lane_id(), j = t & 3;
    FastHashConst h;
    h.k0 = xsec64(8 * (t >> 2) + 16 * j); h.k1 = xsec64(8 * (t >> 2) + 16 * j + 8);
    h.l0 = xsec64(121 + 16 * j); h.l1 = xsec64(129 + 16 * j);
    h.m0 = xsec64(11 + 16 * j); h.m1 = xsec64(19 + 16 * j);
    h.i0 = j == 0 ? XP32_3 : j == 1 ? XP64_2 : j == 2 ? XP64_4 : XP64_5;
    h.i1 = j == 0 ? XP64_1 : j == 1 ? XP64_3 : j == 2 ? XP32_2 : XP32_1;
    return h;
}
CK_DEV uint64_t shfl_xor_add64(uint64_t v, uint32_t m)
{
    const uint32_t src = lane_id() ^ m;
    return add64_parts(v, shfl((uint32_t)v, src), shfl((uint32_t)(v >> 32), src));
}
// `cell` = this lane's 16 canonical bytes [16t, 16t+16) (valid for t < 4*stripes); E / idx = winning strand and
// rotation; returns the hash (same value in every lane).
CK_DEV uint64_t fast_hash(const FastHashConst& hc, const uint32_t* lut, u32x4 cell, uint32_t E, uint32_t idx, uint32_t n)
{
    const uint32_t t = lane_id();
    const uint32_t stripes = (n - 1) >> 6;                       // full 64-byte stripes before the last one
    uint64_t c0 = 0, c1 = 0;
    {
        const uint64_t d0 = ((uint64_t)cell.y << 32) | cell.x, d1 = ((uint64_t)cell.w << 32) | cell.z;
        const uint64_t x0 = d0 ^ hc.k0, x1 = d1 ^ hc.k1;
        const bool on = t < 4 * stripes;
        c0 = on ? d1 + (uint64_t)(uint32_t)x0 * (x0 >> 32) : 0;
        c1 = on ? d0 + (uint64_t)(uint32_t)x1 * (x1 >> 32) : 0;
    }
    // sum over the 16 stripes: lanes t, t+4, t+8, t+12 inside a row (DPP), then the four rows (LDS crossbar)
    dpp_rowsum4_u64x2(c0, c1);                                   // lanes 12..15 of each row: row sums of pair t&3
    c0 = shfl_xor_add64(c0, 16); c1 = shfl_xor_add64(c1, 16);
    c0 = shfl_xor_add64(c0, 32); c1 = shfl_xor_add64(c1, 32);
    uint64_t a0 = hc.i0 + c0, a1 = hc.i1 + c1;                   // meaningful in lanes with (t & 15) >= 12
    {   // last stripe: the final 64 bytes, pair t&3 = bytes [n-64+16j, n-48+16j)
        const uint32_t w = reg_sym_word(E, idx + (n - 64) + 16 * (t & 3), n);
        const u32x4 b = fast_decode(lut, w);
        const uint64_t d0 = ((uint64_t)b.y << 32) | b.x, d1 = ((uint64_t)b.w << 32) | b.z;
        const uint64_t x0 = d0 ^ hc.l0, x1 = d1 ^ hc.l1;
        a0 += d1 + (uint64_t)(uint32_t)x0 * (x0 >> 32);
        a1 += d0 + (uint64_t)(uint32_t)x1 * (x1 >> 32);
    }
    uint64_t r = xfold(a0 ^ hc.m0, a1 ^ hc.m1);                  // merge: sum over the four pairs (one quad)
    r = dpp_quadsum_u64(r);
    const uint64_t h = xaval3((uint64_t)n * XP64_1 + r);
    return ((uint64_t)readlane((uint32_t)(h >> 32), 15) << 32) | readlane((uint32_t)h, 15);
}

// ---- XXH3-64 merged per record GROUP (the 16-wave streaming build) -----------------------------------------------
// fast_hash() above spends most of its instructions on work that is the same for every record and uses 4 lanes of
// 64: the cross-row butterfly (8 ds_bpermute), the last stripe (2 ds_bpermute + a 16-byte decode), the 64x64->128
// merge fold and the avalanche (~18 quarter-rate multiplies).  In the 16-wave build every wave therefore only
// leaves its per-row partial sums (lanes 12..15 of each 16-lane row), its packed winning strand and (n, idx) in an
// LDS slot; after the iteration's barrier ONE wave (a different one each iteration) finishes all 16 records at once,
// lane t = (record t>>2, accumulator pair t&3), and stores 16 hashes with one instruction.  Slots are double-buffered
// by iteration parity: the merger reads group g while the others fill the slots of group g+1.
//   slot (GH_STRIDE_DW dwords): [0,64) partial sums: entry (row, pair) = {c0, c1}; [64,128) packed words of the
//   winning strand (extension included); [128..130] n, idx, valid.  576 B apart: the merger's 16-byte reads of one
//   entry from 16 slots are bank-conflict-free.
constexpr uint32_t GH_STRIDE_DW = 144;
constexpr uint32_t GH_CONST_DW = 48;            // per accumulator pair j: l0 l1 m0 m1 i0 i1 (6 x u64)
constexpr uint32_t GH_SECRET_DW = 48;           // canon_pair.h only: XXH3's 192-byte secret behind the constants (the lanes' cell words are read from it)
template <int GROUP, bool PAIR = false>
constexpr uint32_t gh_lds_dw() { return 2 * GROUP * GH_STRIDE_DW + GH_CONST_DW + (PAIR ? GH_SECRET_DW : 0); }
CK_DEV void group_hash_secret_init(uint32_t* sec, uint32_t tid)
{
    if (tid < GH_SECRET_DW) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 3; i >= 0; --i) v = (v << 8) | XXH3_SECRET[4 * tid + i];
        sec[tid] = v;
    }
}
CK_DEV void group_hash_init(uint32_t* ghc, uint32_t tid)
{
    if (tid < 4) {
        const uint32_t j = tid;
        const uint64_t v[6] = { xsec64(121 + 16 * j), xsec64(129 + 16 * j), xsec64(11 + 16 * j), xsec64(19 + 16 * j),
                                j == 0 ? XP32_3 : j == 1 ? XP64_2 : j == 2 ? XP64_4 : XP64_5,
                                j == 0 ? XP64_1 : j == 1 ? XP64_3 : j == 2 ? XP32_2 : XP32_1 };
#pragma unroll
        for (int k = 0; k < 6; ++k) { ghc[12 * j + 2 * k] = (uint32_t)v[k]; ghc[12 * j + 2 * k + 1] = (uint32_t)(v[k] >> 32); }
    }
}
CK_DEV void group_hash_invalidate(uint32_t* slot) { if (lane_id() == 0) slot[130] = 0; }
// producer: `cell` = this lane's 16 canonical bytes [16t, 16t+16); k0 / k1 = secret words of the lane's cell
CK_DEV void group_hash_put(uint32_t* slot, uint64_t k0, uint64_t k1, u32x4 cell, uint32_t E, uint32_t idx, uint32_t n)
{
    const uint32_t t = lane_id();
    const uint32_t stripes = (n - 1) >> 6;                       // full 64-byte stripes before the last one
    const uint64_t d0 = ((uint64_t)cell.y << 32) | cell.x, d1 = ((uint64_t)cell.w << 32) | cell.z;
    const uint64_t x0 = d0 ^ k0, x1 = d1 ^ k1;
    const bool on = t < 4 * stripes;
    uint64_t c0 = on ? d1 + (uint64_t)(uint32_t)x0 * (x0 >> 32) : 0;
    uint64_t c1 = on ? d0 + (uint64_t)(uint32_t)x1 * (x1 >> 32) : 0;
    dpp_rowsum4_u64x2(c0, c1);                                   // lanes 12..15 of each row: row sums of pair t&3
    if ((t & 15) >= 12) lds_store16(slot + (t >> 4) * 16 + (t & 3) * 4, u32x4{ (uint32_t)c0, (uint32_t)(c0 >> 32), (uint32_t)c1, (uint32_t)(c1 >> 32) });
    slot[64 + t] = E;
    if (t == 0) { slot[128] = n; slot[129] = idx; slot[130] = 1; }
}
// ... for a record with N (the streaming build for MODE_ALPHA batches): the strand in the slot would decode to G / C where the
// record holds N, so the producer leaves the last stripe's 64 BYTES (patched like the cells it stores) in [64,80) instead of the
// strand: `last` = the 16 bytes at output offset n - 64 + 16 * (lane & 3), the same in every lane of a quad
CK_DEV void group_hash_put_bytes(uint32_t* slot, uint64_t k0, uint64_t k1, u32x4 cell, u32x4 last, uint32_t n)
{
    const uint32_t t = lane_id();
    const uint32_t stripes = (n - 1) >> 6;
    const uint64_t d0 = ((uint64_t)cell.y << 32) | cell.x, d1 = ((uint64_t)cell.w << 32) | cell.z;
    const uint64_t x0 = d0 ^ k0, x1 = d1 ^ k1;
    const bool on = t < 4 * stripes;
    uint64_t c0 = on ? d1 + (uint64_t)(uint32_t)x0 * (x0 >> 32) : 0;
    uint64_t c1 = on ? d0 + (uint64_t)(uint32_t)x1 * (x1 >> 32) : 0;
    dpp_rowsum4_u64x2(c0, c1);
    if ((t & 15) >= 12) lds_store16(slot + (t >> 4) * 16 + (t & 3) * 4, u32x4{ (uint32_t)c0, (uint32_t)(c0 >> 32), (uint32_t)c1, (uint32_t)(c1 >> 32) });
    if (t < 4) lds_store16(slot + 64 + 4 * t, last);
    if (t == 0) { slot[128] = n; slot[129] = 0; slot[130] = 1; }
}
// merger: finishes the GROUP records of `slots` (records rec0 .. rec0 + GROUP - 1 of the batch)
// LASTB: the producers are group_hash_put_bytes
// NROWS: 16-lane rows of partial sums a producer leaves per record (4: one record per wave; 2: canon_pair.h, one per half-wave)
template <int GROUP, int NROWS = 4, bool LASTB = false>
CK_DEV void group_hash_merge(const CanonArgs& a, const uint32_t* lut, const uint32_t* ghc, const uint32_t* slots, uint32_t rec0)
{
    static_assert(GROUP <= 16, "one lane per (record, accumulator pair)");
    const uint32_t t = lane_id(), r = t >> 2, j = t & 3;
    const uint32_t* s = slots + (r < (uint32_t)GROUP ? r : 0u) * GH_STRIDE_DW;      // (smaller groups: the upper lanes idle)
    const bool valid = r < (uint32_t)GROUP && s[130] != 0;
    const uint32_t n = valid ? s[128] : 64u, idx = valid ? s[129] : 0u;       // (a dummy that keeps the reads inside the slot)
    const u32x4 cl = lds_load16(ghc + 12 * j), cm = lds_load16(ghc + 12 * j + 4), ci = lds_load16(ghc + 12 * j + 8);
    uint64_t a0 = ((uint64_t)ci.y << 32) | ci.x, a1 = ((uint64_t)ci.w << 32) | ci.z;
#pragma unroll
    for (uint32_t row = 0; row < (uint32_t)NROWS; ++row) {
        const u32x4 v = lds_load16(s + row * 16 + j * 4);
        a0 += ((uint64_t)v.y << 32) | v.x;
        a1 += ((uint64_t)v.w << 32) | v.z;
    }
    {   // last stripe: the final 64 bytes, pair j = bytes [n-64+16j, n-48+16j)
        u32x4 b;
        if constexpr (LASTB) b = lds_load16(s + 64 + 4 * j);
        else {
            uint32_t p = idx + (n - 64) + 16 * j;
            p = p >= n ? p - n : p;
            const uint32_t wi = p >> 4;
            b = fast_decode(lut, lshr64(s[64 + wi], s[64 + wi + 1], 32 - (p & 15) * 2));
        }
        const uint64_t d0 = ((uint64_t)b.y << 32) | b.x, d1 = ((uint64_t)b.w << 32) | b.z;
        const uint64_t x0 = d0 ^ (((uint64_t)cl.y << 32) | cl.x), x1 = d1 ^ (((uint64_t)cl.w << 32) | cl.z);
        a0 += d1 + (uint64_t)(uint32_t)x0 * (x0 >> 32);
        a1 += d0 + (uint64_t)(uint32_t)x1 * (x1 >> 32);
    }
    uint64_t q = xfold(a0 ^ (((uint64_t)cm.y << 32) | cm.x), a1 ^ (((uint64_t)cm.w << 32) | cm.z));
    q = dpp_quadsum_u64(q);                                                   // sum over the four pairs of the record
    const uint64_t h = xaval3((uint64_t)n * XP64_1 + q);
    if (valid && j == 0) { a.out_hash[rec0 + r] = h; a.hashed[rec0 + r] = 1; }
}

// First level of the scan: the minimum over a word's 16 positions of the FIRST 8 symbols of their keys.  The 32-bit
// window that starts at position b holds that 16-bit prefix of position b in its upper half and the one of position
// b + 8 in its lower half, so eight windows carry all sixteen prefixes and v_pk_min_u16 takes two minima at a time:
// 7 + 7 + 3 instructions instead of the 15 + 8 of the full-width scan.  Random DNA: the minimal 8-mer of ~1000
// positions is owned by one position in ~98.5 % of the records; the others repeat the scan at full width.
CK_DEV uint32_t word_min_key16(uint32_t cur, uint32_t nxt)
{
    uint32_t m = cur;
#pragma unroll
    for (int b = 1; b < 8; ++b) m = pk_min_u16(m, funnel(cur, nxt, 2 * b));
    const uint32_t hi = m >> 16, lo = m & 0xFFFFu;
    return hi < lo ? hi : lo;
}
// fast2_locate() on the 16-bit prefixes
CK_DEV uint32_t fast2_locate16(uint32_t E, uint32_t En, uint64_t hm, uint32_t M16, uint32_t n, uint32_t shv, bool& unique)
{
    // (scalar instruction count matters here -- the streaming kernel issues about as many scalar as vector instructions per
    // record: the position mask is (1 << min(left, 16)) - 1 = s_min + s_bfm, the first set bit one s_ff1 that yields -1 for an
    // empty mask, the lane count a 32-bit compare)
    const uint32_t l = (uint32_t)ffs64(hm);
    const uint32_t k = lshr64(readlane(E, l), readlane(En, l), shv) >> 16;
    const uint32_t left = n - 16 * l;
    uint32_t pm = (uint32_t)ballot(k == M16) & low_mask16(left);
    uint32_t cnt = (uint32_t)popc32(pm);
    uint32_t pos = 16 * l + (uint32_t)ffs32_or_neg(pm);
    if ((uint32_t)popc64(hm) != 1u) {                                       // rare
        hm &= hm - 1;
        const uint32_t l2 = (uint32_t)ffs64(hm);
        const uint32_t k2 = lshr64(readlane(E, l2), readlane(En, l2), shv) >> 16;
        const uint32_t left2 = n - 16 * l2;
        const uint32_t pm2 = (uint32_t)ballot(k2 == M16) & low_mask16(left2);
        cnt += (uint32_t)popc32(pm2) + ((hm & (hm - 1)) ? 2u : 0u);
        if (pm == 0) pos = 16 * l2 + (uint32_t)ffs32_or_neg(pm2);
    }
    unique = cnt == 1;
    return pos;
}

// Canonicalizes one eligible record held as packed words: lane t = symbols [16t, 16t+16) (whatever follows the
// record in the tail word is replaced by the periodic extension), bad = wave mask of lanes holding a byte outside
// ACGT.  Returns false (nothing written) when the record must go to the general kernel: an invalid byte, a minimal
// key that is not unique, or equal minimal keys on the two strands.  Single exit: the rare failures are folded into
// one flag instead of early returns, which keeps the scalar unit's branch / mask bookkeeping off the hot path.
// Lane constants that depend on the record length only (source words and funnel shifts of the periodic extension
// and of the reverse strand, the byte window each lane stores): recomputed only when the length changes, which on a
// batch of equal-length records is once per wave.
struct FastShape {
    uint32_t n = 0;                 // length these constants are for (0 = none)
    uint32_t ext_lane;              // t - ceil(n/16): source word of the extension
    uint32_t rc_lane, rc_sh;        // reverse strand: source word and funnel shift
    uint32_t out_o;                 // byte offset of this lane's 16 output bytes
    uint32_t inv;                   // ~0 in the lanes past the record's last word (their scan results must lose every minimum), else 0
};
CK_DEV void fast_shape(FastShape& sh, uint32_t n)
{
    const uint32_t t = lane_id(), r = n & 15, nwv = (n >> 4) + (r ? 1u : 0u);
    sh.n = n;
    sh.ext_lane = t - nwv;
    // rc word t = comp(reverse(forward symbols [n - 16(t+1), n - 16t) mod n))
    const int32_t p0 = (int32_t)n - 16 * (int32_t)(t + 1);
    const uint32_t p = (uint32_t)(p0 + ((p0 >> 31) & (int32_t)n));
    sh.rc_lane = p >> 4;
    sh.rc_sh = 32 - (p & 15) * 2;
    // every lane stores a full 16 bytes: the last lane's window is pulled back to end exactly at n, so it overlaps
    // its neighbour's with identical bytes -- one store instruction, no partial-store branches
    sh.out_o = 16 * t + 16 <= n ? 16 * t : n - 16;
    sh.inv = t < nwv ? 0u : ~0u;
}

// HASH = false compiles the fused XXH3 out; AUX = false compiles out what only some callers ask for (rotation index
// and strand outputs, forward-only mode).
// GH: the hash is finished by the workgroup's merger (group_hash_put into gh_slot), else here (fast_hash).
// K16: two-level scan (8-symbol prefixes first).  The staged streaming kernel is bound by vector-instruction issue and
// gains from it; the latency-bound rescue pass lost a third of its speed to it (mixed lengths: 225 -> 330 us) and keeps
// the one-level scan.
// NM (bytes-only builds): the record may hold N.  F has them packed as G -- the reverse strand then shows C --, Nm marks
// them (0b11 per N, the strand's layout).  Either way a key with an N is BELOW its true value and every other key exact
// (C < G < N), so the winner stands if the symbols that decided -- 8 at level 1, 16 at level 2 -- hold no N; else false,
// and the caller's 4-bit routine has the record.  `bad` = lanes with a byte outside ACGTN.  (canon_core.h
// canon_record_mode2n is the same idea in the LDS tiers.)
// TIGHT: the build is at its register limit (the ALPHA and ROWS = 2 streaming builds: 64 registers at 16 waves per workgroup):
// the scans of the lanes past the record stay behind exec masks -- straight-line, the scheduler interleaves the two strands'
// scans and keeps ~15 more registers live, which those builds pay for in scratch spills (round 4, measured by PMC: +2 vector
// stores and +1 load per record in the ALPHA build).
template <bool HASH, bool AUX, bool GH = false, bool K16 = false, bool NM = false, bool TIGHT = false>
CK_DEV bool fast_canon(const CanonArgs& a, const uint32_t* lut, const FastHashConst& hc, FastShape& sh, uint32_t rec, uint64_t off,
                       uint32_t n, uint32_t F, uint64_t bad, uint32_t* gh_slot = nullptr, uint32_t Nm = 0, uint32_t* view_out = nullptr)
{
    static_assert(!NM || (!AUX && (!HASH || GH)), "the N-mask variant: bytes, and the XXH3 only through the group merger (group_hash_put_bytes)");
    const uint32_t t = lane_id();
    const uint32_t nwf = n >> 4, r = n & 15, nwv = nwf + (r ? 1u : 0u);
    if (sh.n != n) fast_shape(sh, n);
    // periodic extension (lanes >= nwf): E[nwf] = r tail symbols ++ head, E[nwv + e] = head shifted by r
    {
        const uint32_t A = shfl(F, sh.ext_lane), B = shfl(F, sh.ext_lane + 1);
        const uint32_t ext = lshr64(A, B, 32 - ((16 - r) & 15) * 2);
        const uint32_t fix = bfi(~(0xFFFFFFFFu >> (2 * r)), F, B >> (2 * r));
        F = t >= nwv ? ext : (t == nwf ? fix : F);        // r == 0: nwf == nwv, `fix` is never selected
        if constexpr (NM) {
            const uint32_t An = shfl(Nm, sh.ext_lane), Bn = shfl(Nm, sh.ext_lane + 1);
            const uint32_t extn = lshr64(An, Bn, 32 - ((16 - r) & 15) * 2);
            const uint32_t fixn = bfi(~(0xFFFFFFFFu >> (2 * r)), Nm, Bn >> (2 * r));
            Nm = t >= nwv ? extn : (t == nwf ? fixn : Nm);
        }
    }
    const bool fwd_only = AUX && (a.flags & CK_FLAG_FWD_ONLY) != 0;
    // reverse-complement strand from the extended forward words:
    // rc word t = comp(reverse(forward symbols [n - 16(t+1), n - 16t) mod n))
    uint32_t C;
    {
        const uint32_t g = ~lshr64(shfl(F, sh.rc_lane), shfl(F, sh.rc_lane + 1), sh.rc_sh);
        const uint32_t v = bitrev(g);                       // reverses bits; swap the two bits of every symbol back
        C = bfi(0x55555555u, v >> 1, v << 1);
    }
    const uint32_t Fn = wave_shl1(F), Cn = wave_shl1(C);
    const bool valid = t < nwv;
    const uint32_t shv = 32 - 2 * (t & 15);
    if (bad != 0) return false;
    // Level 1: 8-symbol prefixes (word_min_key16).  They decide the strand and the rotation whenever the two strands'
    // minimal prefixes differ and the winner's is owned by one position -- the minimal keys ARE the first symbols of the
    // two minimal rotations (lib/src/canonicalize.rs:58-62: forward only if strictly smaller).
    bool fwd = true, tie = true, uE = false, uF = true;
    uint32_t idx = 0, iF = 0, decided = 16;                // symbols of the winner's key that decided (NM)
    if constexpr (K16) {
        decided = 8;
        // every lane scans (a VALU instruction costs the same with any exec mask); the lanes past the record are ORed out of
        // the minimum instead of being branched around: `valid ? scan : 0xFFFF` compiled to two exec-mask regions
        // (s_and_saveexec / s_cbranch_execz / s_or: 6 scalar instructions + 2 branches per record)
        uint32_t mF, mC;
        if constexpr (TIGHT) { mF = valid ? word_min_key16(F, Fn) : 0xFFFFu; mC = valid ? word_min_key16(C, Cn) : 0xFFFFu; }
        else { mF = word_min_key16(F, Fn) | sh.inv; mC = word_min_key16(C, Cn) | sh.inv; }
        uint32_t MF, MC;
        wave_min2_u32(mF, mC, MF, MC);
        fwd = fwd_only || MF < MC;
        tie = !fwd_only && MF == MC;
        idx = fast2_locate16(fwd ? F : C, fwd ? Fn : Cn, fwd ? ballot(mF == MF) : ballot(mC == MC), fwd ? MF : MC, n, shv, uE);
        iF = idx;
        if (AUX && a.out_index && !fwd) iF = fast2_locate16(F, Fn, ballot(mF == MF), MF, n, shv, uF);
    }
    if (tie || !uE || !uF) {
        // Level 2 (rare): full 16-symbol keys.  Only the winning strand's rotation has to be located -- unless the
        // rotation index is asked for, which for the reverse strand is counted from the forward strand's minimal
        // rotation.  Equal keys on both strands (reverse-complement palindromes), or a minimal key that is still not
        // unique: left to the general kernel.
        decided = 16;
        uint32_t mF = valid ? word_min_key<2>(F, Fn) : ~0u, mC = valid ? word_min_key<2>(C, Cn) : ~0u;
        uint32_t MF, MC;
        wave_min2_u32(mF, mC, MF, MC);
        fwd = fwd_only || MF <= MC;
        tie = !fwd_only && MF == MC;
        uF = true;
        idx = fast2_locate(fwd ? F : C, fwd ? Fn : Cn, fwd ? ballot(mF == MF) : ballot(mC == MC), fwd ? MF : MC, n, shv, uE);
        iF = idx;
        if (AUX && a.out_index && !fwd) iF = fast2_locate(F, Fn, ballot(mF == MF), MF, n, shv, uF);
        if constexpr (!AUX && !NM) {
            if (tie) {
                // Equal minimal keys on the two strands -- a minimal 16-mer inside a reverse-complement palindrome of 18+, which
                // both strands then own: ~10 records per million of random 1 kb DNA.  The two minimal rotations are compared in
                // full right here, one 16-symbol word per lane (lib/src/canonicalize.rs:58-62: forward only if strictly smaller;
                // equal = the record is its own reverse complement, either strand's bytes are the same).  Left to the general
                // kernel these few records were a ~30 us pass of LDS stage A behind every batch of the headline workload.
                bool uC;
                const uint32_t iC = fast2_locate(C, Cn, ballot(mC == MC), MC, n, shv, uC);     // (idx = the forward strand's: fwd was MF <= MC)
                if (!uE || !uC) return false;
                const uint32_t x = reg_sym_word(F, idx + (valid ? 16 * t : 0), n), y = reg_sym_word(C, iC + (valid ? 16 * t : 0), n);
                const uint64_t d = ballot(valid && x != y);
                fwd = false;
                if (d != 0) { const uint32_t l = (uint32_t)ffs64(d); fwd = readlane(x, l) < readlane(y, l); }
                idx = fwd ? idx : iC;
                tie = false;
            }
        }
        if (tie || !uE || !uF) return false;
    }
    const uint32_t E = fwd ? F : C;
    uint32_t Em = Nm;
    if constexpr (NM) {
        if (!fwd) Em = bitrev(lshr64(shfl(Nm, sh.rc_lane), shfl(Nm, sh.rc_lane + 1), sh.rc_sh));      // the mask in the reverse strand's order
        if ((reg_sym_word(Em, idx, n) >> (32 - 2 * decided)) != 0) return false;                       // an N among the symbols that decided
    }
    {
        const uint32_t o = sh.out_o;
        const bool hash = HASH && a.out_hash != nullptr && n > 240;     // XXH3's long-input path; shorter: xxh3 pass
        const bool store = a.out_bytes != nullptr;
        if (store || hash) {
            // the 16 output bytes at output offset q (NM: the decoded 'G' (forward) or 'C' (reverse) of every marked symbol becomes
            // 'N': bit 2j of a reversed mask byte -> 0x01 in byte j by one multiplication (1 + 2^6 + 2^12 + 2^18), times the XOR constant)
            const auto cell_at = [&](uint32_t q) {
                u32x4 c = fast_decode(lut, reg_sym_word(E, idx + q, n));
                if constexpr (NM) {
                    const uint32_t m = reg_sym_word(Em, idx + q, n);
                    if (m) {
                        const uint32_t rm = bitrev(m), fix = fwd ? 0x09u : 0x0Du;
                        c.x ^= (((rm & 0x55u) * 0x41041u) & 0x01010101u) * fix;
                        c.y ^= ((((rm >> 8) & 0x55u) * 0x41041u) & 0x01010101u) * fix;
                        c.z ^= ((((rm >> 16) & 0x55u) * 0x41041u) & 0x01010101u) * fix;
                        c.w ^= ((((rm >> 24) & 0x55u) * 0x41041u) & 0x01010101u) * fix;
                    }
                }
                return c;
            };
            const u32x4 cell = cell_at(o);
            if (store && valid) store16(a.out_bytes + off + o, cell);
            if (hash) {
                if constexpr (GH && NM) {
                    group_hash_put_bytes(gh_slot, hc.k0, hc.k1, cell, cell_at(n - 64 + 16 * (t & 3)), n);      // (n > 240: the offset is in range)
                } else if constexpr (GH) {
                    group_hash_put(gh_slot, hc.k0, hc.k1, cell, E, idx, n);
                } else {
                    const uint64_t h = fast_hash(hc, lut, cell, E, idx, n);
                    if (t == 0) { a.out_hash[rec] = h; a.hashed[rec] = 1; }
                }
            }
        }
    }
    if (AUX && t == 0) {
        // unique minimum => period n; idx + iF < 2n
        if (a.out_index) a.out_index[rec] = fwd ? idx : (idx + iF >= n ? idx + iF - n : idx + iF);
        if (a.out_strand) a.out_strand[rec] = fwd ? 0 : 1;
    }
    if (HASH && a.out_view && n <= 240 && t == 0) a.out_view[rec] = fwd ? idx : (idx | 0x80000000u);      // hash not fused: the xxh3 pass reads the view
    if (view_out) *view_out = fwd ? idx : (idx | 0x80000000u);     // (callers that hash the short-input classes themselves: canon_mixed.h)
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// Two words per lane.  BITS = 2: records of 1009..2032 bases (the 1.0-1.7 kb circular RNAs -- obelisks, deltaviruses
// -- sit here), lane t = symbols [32t, 32t+32).  BITS = 4: records of 48..1008 symbols over the CLI alphabet
// {-,A,C,G,N,T} -- the reference treats N and '-' like any other byte (lib/src/canonicalize.rs:50-53), and one N no
// longer sends a 1 kb record to the LDS tiers -- lane t = symbols [16t, 16t+16).  Word q of a strand lives in lane
// q>>1, register q&1, and every cross-lane access goes through word2().  Same algorithm and exits as fast_canon; keys
// are one word = S = 32/BITS symbols (8 for BITS = 4: two positions share the minimal 8-mer in ~1 % of random 1 kb
// records, which then take the LDS tier).  XXH3: fast_hash2 for BITS = 2; the 4-bit records leave theirs to the xxh3 pass.
constexpr uint32_t FAST2_MAX_N = 2032;
CK_DEV bool fast2_eligible(uint32_t n) { return n - (FAST_MAX_N + 1) <= FAST2_MAX_N - (FAST_MAX_N + 1); }

CK_DEV uint32_t word2(uint32_t W0, uint32_t W1, uint32_t q)
{
    const uint32_t a = shfl(W0, q >> 1), b = shfl(W1, q >> 1);
    return (q & 1) ? b : a;
}
template <int BITS>
CK_DEV uint32_t reg_sym_wordw(uint32_t W0, uint32_t W1, uint32_t p, uint32_t n)    // p < 2n, lane-varying
{
    constexpr uint32_t S = 32 / BITS;
    p = p >= n ? p - n : p;
    const uint32_t wi = p / S;
    return lshr64(word2(W0, W1, wi), word2(W0, W1, wi + 1), 32 - (p % S) * BITS);
}
CK_DEV uint32_t reg_sym_word2(uint32_t W0, uint32_t W1, uint32_t p, uint32_t n) { return reg_sym_wordw<2>(W0, W1, p, n); }
// adds the valid positions of word 2l+k (l = first lane of hm) whose key equals M; cur / nxt = that register and the
// word behind it, as lane vectors; shv = 32 - BITS * (lane & (S - 1))
template <int BITS>
CK_DEV void locate2_word(uint32_t cur, uint32_t nxt, uint32_t l, uint32_t k, uint32_t M, uint32_t n, uint32_t shv, uint32_t& cnt, uint32_t& pos)
{
    constexpr uint32_t S = 32 / BITS;
    const uint32_t w = 2 * l + k;
    const uint32_t key = lshr64(readlane(cur, l), readlane(nxt, l), shv);
    const uint32_t left = n - S * w;
    const uint32_t pm = (uint32_t)ballot(key == M) & ((1u << (left < S ? left : S)) - 1u);     // lanes 0..S-1: the word's positions
    cnt += (uint32_t)popc64(pm);
    if (pm) pos = S * w + (uint32_t)ffs64(pm);
}
// position of the minimal key M in a two-words-per-lane strand and whether exactly one valid position owns it; at
// most two hit words are examined (the second is normally the wrapped duplicate behind the record end)
template <int BITS>
CK_DEV uint32_t fast2x_locate(uint32_t E0, uint32_t E1, uint32_t E0n, uint64_t hm0, uint64_t hm1, uint32_t M, uint32_t n, uint32_t shv,
                              bool& unique)
{
    uint32_t cnt = 0, pos = 0;
    const uint32_t hits = (uint32_t)(popc64(hm0) + popc64(hm1));
    if (hm0) locate2_word<BITS>(E0, E1, (uint32_t)ffs64(hm0), 0, M, n, shv, cnt, pos);
    if (hm1) locate2_word<BITS>(E1, E0n, (uint32_t)ffs64(hm1), 1, M, n, shv, cnt, pos);
    if (hits == 2 && (hm0 == 0 || hm1 == 0)) {            // both hits in the same register: its second lane
        if (hm0) { hm0 &= hm0 - 1; locate2_word<BITS>(E0, E1, (uint32_t)ffs64(hm0), 0, M, n, shv, cnt, pos); }
        else { hm1 &= hm1 - 1; locate2_word<BITS>(E1, E0n, (uint32_t)ffs64(hm1), 1, M, n, shv, cnt, pos); }
    }
    unique = cnt == 1 && hits <= 2;
    return pos;
}

// XXH3-64 of a two-words-per-lane record (1009..2032 bytes: one or two 1024-byte blocks), fused like fast_hash: per
// block the lanes re-fetch the canonical bytes in the hash's own layout (lane t = cell (stripe t>>2, pair t&3) of the
// block), accumulate, sum over the stripes, and the first block is followed by XXH3's scramble; last stripe, merge and
// avalanche as in fast_hash.
CK_DEV uint64_t fast_hash2(const FastHashConst& hc, const uint32_t* lut, uint32_t E0, uint32_t E1, uint32_t idx, uint32_t n)
{
    const uint32_t t = lane_id(), j = t & 3;
    const uint32_t nb = (n - 1) >> 10;                                   // 0 or 1 full blocks before the last one
    uint64_t a0 = hc.i0, a1 = hc.i1;                                     // meaningful in lanes with (t & 15) >= 12
    for (uint32_t b = 0; b <= nb; ++b) {
        const uint32_t stripes = b < nb ? 16u : ((n - 1) - 1024 * nb) / 64;
        const u32x4 cell = fast_decode(lut, reg_sym_word2(E0, E1, idx + 1024 * b + 16 * t, n));
        const uint64_t d0 = ((uint64_t)cell.y << 32) | cell.x, d1 = ((uint64_t)cell.w << 32) | cell.z;
        const uint64_t x0 = d0 ^ hc.k0, x1 = d1 ^ hc.k1;
        const bool on = t < 4 * stripes;
        uint64_t c0 = on ? d1 + (uint64_t)(uint32_t)x0 * (x0 >> 32) : 0;
        uint64_t c1 = on ? d0 + (uint64_t)(uint32_t)x1 * (x1 >> 32) : 0;
        dpp_rowsum4_u64x2(c0, c1);
        c0 = shfl_xor_add64(c0, 16); c1 = shfl_xor_add64(c1, 16);
        c0 = shfl_xor_add64(c0, 32); c1 = shfl_xor_add64(c1, 32);
        a0 += c0; a1 += c1;
        if (b < nb) {                                                    // scramble with the last 64 secret bytes
            a0 = (a0 ^ (a0 >> 47) ^ xsec64(128 + 16 * j)) * XP32_1;
            a1 = (a1 ^ (a1 >> 47) ^ xsec64(128 + 16 * j + 8)) * XP32_1;
        }
    }
    {   // last stripe: the final 64 bytes, pair j = bytes [n-64+16j, n-48+16j)
        const u32x4 bb = fast_decode(lut, reg_sym_word2(E0, E1, idx + (n - 64) + 16 * j, n));
        const uint64_t d0 = ((uint64_t)bb.y << 32) | bb.x, d1 = ((uint64_t)bb.w << 32) | bb.z;
        const uint64_t x0 = d0 ^ hc.l0, x1 = d1 ^ hc.l1;
        a0 += d1 + (uint64_t)(uint32_t)x0 * (x0 >> 32);
        a1 += d0 + (uint64_t)(uint32_t)x1 * (x1 >> 32);
    }
    uint64_t r = xfold(a0 ^ hc.m0, a1 ^ hc.m1);
    r = dpp_quadsum_u64(r);
    const uint64_t h = xaval3((uint64_t)n * XP64_1 + r);
    return ((uint64_t)readlane((uint32_t)(h >> 32), 15) << 32) | readlane((uint32_t)h, 15);
}

// W0 / W1: lane t = words 2t / 2t+1 of the record (garbage past n); bad = any lane of the record's chunks holds a
// byte outside the mode's alphabet.
template <int BITS, bool HASH, bool AUX>
CK_DEV bool fast_canonw(const CanonArgs& a, const uint32_t* lut, const FastHashConst& hc, uint32_t rec, uint64_t off, uint32_t n,
                        uint32_t W0, uint32_t W1, bool bad)
{
    static_assert(BITS == 2 || (BITS == 4 && !AUX), "4-bit records: canonical bytes only (index / strand requests take the LDS tier)");
    constexpr uint32_t S = 32 / BITS;
    const uint32_t t = lane_id();
    const uint32_t nwf = n / S, r = n % S, nwv = nwf + (r ? 1u : 0u);
    const uint32_t w0 = 2 * t, w1 = 2 * t + 1;
    // periodic extension (words >= nwf), from the original words: E[nwf] = r tail symbols ++ head, E[nwv + e] = head
    // shifted by r.  Words up to index nwv are needed (scan of the last word, reverse strand); nwv <= 127.
    {
        const uint32_t esh = 32 - ((S - r) % S) * BITS;
        const uint32_t A0 = word2(W0, W1, w0 - nwv), B0 = word2(W0, W1, w0 - nwv + 1), B1 = word2(W0, W1, w1 - nwv + 1);
        const uint32_t keep = ~(0xFFFFFFFFu >> (BITS * r));
        const uint32_t n0 = w0 >= nwv ? lshr64(A0, B0, esh) : (w0 == nwf ? bfi(keep, W0, B0 >> (BITS * r)) : W0);
        const uint32_t n1 = w1 >= nwv ? lshr64(B0, B1, esh) : (w1 == nwf ? bfi(keep, W1, B1 >> (BITS * r)) : W1);   // word(w1-nwv) = B0
        W0 = n0; W1 = n1;
    }
    const bool fwd_only = AUX && (a.flags & CK_FLAG_FWD_ONLY) != 0;
    // reverse-complement strand: rc word w = comp(reverse(forward symbols [n - S(w+1), n - S w) mod n))
    uint32_t C0, C1;
    {
        uint32_t c[2];
#pragma unroll
        for (uint32_t k = 0; k < 2; ++k) {
            const int32_t p0 = (int32_t)n - (int32_t)S * (int32_t)(2 * t + k + 1);
            const uint32_t p = (uint32_t)(p0 + ((p0 >> 31) & (int32_t)n));      // (garbage for words past nwv + 1: unused)
            c[k] = rc_word<BITS>(lshr64(word2(W0, W1, p / S), word2(W0, W1, p / S + 1), 32 - (p % S) * BITS));
        }
        C0 = c[0]; C1 = c[1];
    }
    const uint32_t W0n = wave_shl1(W0), C0n = wave_shl1(C0);          // word 2t+2 = the word behind register 1
    const bool v0 = w0 < nwv, v1 = w1 < nwv;
    const uint32_t mF0 = v0 ? word_min_key<BITS>(W0, W1) : ~0u, mF1 = v1 ? word_min_key<BITS>(W1, W0n) : ~0u;
    const uint32_t mC0 = v0 ? word_min_key<BITS>(C0, C1) : ~0u, mC1 = v1 ? word_min_key<BITS>(C1, C0n) : ~0u;
    uint32_t MF, MC;
    wave_min2_u32(mF0 < mF1 ? mF0 : mF1, mC0 < mC1 ? mC0 : mC1, MF, MC);
    const bool fwd = fwd_only || MF <= MC;
    const bool tie = !fwd_only && MF == MC;
    const uint32_t shv = 32 - BITS * (t & (S - 1));
    const uint32_t E0 = fwd ? W0 : C0, E1 = fwd ? W1 : C1;
    bool uE, uF = true;
    const uint32_t idx = fwd ? fast2x_locate<BITS>(W0, W1, W0n, ballot(mF0 == MF), ballot(mF1 == MF), MF, n, shv, uE)
                             : fast2x_locate<BITS>(C0, C1, C0n, ballot(mC0 == MC), ballot(mC1 == MC), MC, n, shv, uE);
    uint32_t iF = idx;
    if (AUX && a.out_index && !fwd) iF = fast2x_locate<BITS>(W0, W1, W0n, ballot(mF0 == MF), ballot(mF1 == MF), MF, n, shv, uF);
    if (bad || tie || !uE || !uF) return false;
    const bool hash = BITS == 2 && HASH && a.out_hash != nullptr;
    if (a.out_bytes != nullptr) {
        if (BITS == 2) {
#pragma unroll
            for (uint32_t k = 0; k < 2; ++k) {
                // every stored window is a full 16 bytes: the record's last window is pulled back to end exactly at n
                const uint32_t ob = 32 * t + 16 * k, o = ob + 16 <= n ? ob : n - 16;
                const u32x4 cell = fast_decode(lut, reg_sym_wordw<2>(E0, E1, idx + o, n));
                if (ob < n) store16(a.out_bytes + off + o, cell);
            }
        } else {
            const uint32_t ob = 16 * t, o = ob + 16 <= n ? ob : n - 16;
            u32x4 cell;
            decode4(reg_sym_wordw<4>(E0, E1, idx + o, n), cell.x, cell.y);
            decode4(reg_sym_wordw<4>(E0, E1, idx + o + 8, n), cell.z, cell.w);
            if (ob < n) store16(a.out_bytes + off + o, cell);
        }
    }
    if (hash) {
        const uint64_t h = fast_hash2(hc, lut, E0, E1, idx, n);
        if (t == 0) { a.out_hash[rec] = h; a.hashed[rec] = 1; }
    }
    if (AUX && t == 0) {
        if (a.out_index) a.out_index[rec] = fwd ? idx : (idx + iF >= n ? idx + iF - n : idx + iF);
        if (a.out_strand) a.out_strand[rec] = fwd ? 0 : 1;
    }
    if (HASH && !hash && a.out_view && t == 0) a.out_view[rec] = fwd ? idx : (idx | 0x80000000u);         // (the 4-bit records: hash not fused)
    return true;
}
template <bool HASH, bool AUX>
CK_DEV bool fast_canon2(const CanonArgs& a, const uint32_t* lut, const FastHashConst& hc, uint32_t rec, uint64_t off, uint32_t n,
                        uint32_t W0, uint32_t W1, bool bad)
{
    return fast_canonw<2, HASH, AUX>(a, lut, hc, rec, off, n, W0, W1, bad);
}

// 16 ASCII bytes -> two 4-bit words (symbols [0,8) and [8,16)); bad != 0 iff a byte is outside {-,A,C,G,N,T}
CK_DEV void fast_pack4(u32x4 v, uint32_t& H, uint32_t& L, uint32_t& bad)
{
    bad = 0;
    H = pack4_fwd(v.x, v.y, bad);
    L = pack4_fwd(v.z, v.w, bad);
}

}  // namespace ck
