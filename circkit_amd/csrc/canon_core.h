// canon_core.h -- one wavefront canonicalizes one circular record.
//
// Replaces, per record (reference = Benjamin-Lee/circkit):
//   lib/src/canonicalize.rs:5-36   lmsr_index  -> find_min_rot()  (smallest minimal-rotation index)
//   lib/src/canonicalize.rs:41-47  lmsr        -> emit()          (rotation is materialised only once)
//   lib/src/canonicalize.rs:54-63  canonicalize -> canon_record() (both strands, lexicographic select)
//   bio 1.3.1 alphabets::dna::revcomp (call site lib/src/canonicalize.rs:56) -> build_*() reverse strand
//
// Method (not the reference's sequential Duval loop, which is a ~2n-long dependent chain):
//   1. Each strand is re-coded order-preservingly and packed big-endian into 32-bit words in LDS
//      (2 bits/symbol for pure ACGT, 4 bits for the CLI alphabet {-,A,C,G,N,T}, 8 bits for anything
//      else), followed by two words of periodic extension so no window ever has to wrap.
//   2. Candidate elimination: every start position's first 32 bits form a key; each lane scans the
//      32/BITS positions of its word with v_alignbit + v_min, a wave-wide min gives the smallest key M.
//      For non-repetitive DNA exactly one position owns M -> that is the answer.
//   3. Otherwise (repeats, low complexity, tiny alphabets): positions whose key == M form a candidate
//      bitmask; a two-pointer duel over the candidates, each duel a wave-parallel longest-common-prefix
//      (64 words per step), eliminates whole runs at once and also yields the period, so the SMALLEST
//      minimal index is returned (the contract pinned by lib/src/canonicalize.rs:154-164,218-221).
//   4. The two strands' minimal rotations are compared with the same wave-parallel LCP; the winner is
//      decoded back to bytes straight from the packed words and stored with 16-byte lanes.
#pragma once
#include "wave_prims.h"

#ifndef CK_BUILD_ROWS
#define CK_BUILD_ROWS 6     // rows of 64 packed words whose loads are in flight together in build_packed
#endif

namespace ck {

struct CanonArgs {
    const uint8_t* bytes;        // CSR payload (already normalized by the host packer)
    const uint64_t* offsets;     // [n_records + 1]
    uint64_t n_records;
    uint8_t* out_bytes;          // nullable; same offsets as the input
    uint32_t* out_index;         // nullable
    uint8_t* out_strand;         // nullable
    uint64_t* out_hash;          // nullable; XXH3-64 of the canonical bytes (written by the streaming kernel where it can)
    uint8_t* hashed;             // [n_records] set to 1 where out_hash was written; the xxh3 pass does the rest
    // hash-only batches (out_hash without out_bytes: `uniq` without --canonicalize, src/uniq.rs:55-60): a record whose hash is
    // not fused leaves its canonical form as a VIEW of the input for the xxh3 pass -- bit 31 = reverse strand, bits 0..30 =
    // rotation index on that strand OF THE RECORD ITSELF (not the reference-visible out_index, which for the reverse strand
    // counts from the forward strand's minimal rotation) -- instead of writing bytes nobody has asked for
    uint32_t* out_view;          // nullable; [n_records]
    // Work lists are SEGMENTED per producing workgroup: a workgroup appends the records it cannot take to its own
    // segment with an LDS counter and publishes the count when it ends -- no global atomics on the data path
    // (one shared counter serialises at ~10 ns per append: 7 ms for the 760k deferrals of BASELINE config 4).
    const uint32_t* list;        // nullable: input list (nullptr = all records 0..n_records-1, grid-stride)
    const uint32_t* list_count;  // [in_nseg] entries per input segment
    uint32_t in_nseg, in_seg_cap, segs_per_block;   // workgroup b consumes input segments [b*k, b*k + k), k = segs_per_block
    uint32_t all_seg_cap;        // stages that walk ALL records of a batch (mode 3): segment s = records [s * all_seg_cap, (s + 1) * all_seg_cap)
    // ...tapered at the end (seg_records): from segment taper_seg0 on three generations of 2^taper_log2 segments hold a half,
    // a quarter and an eighth of all_seg_cap records each (the third takes every segment behind it).  taper_log2 == 0: none.
    uint32_t taper_seg0, taper_log2;
    uint32_t* defer_list;        // nullable: output list, one segment per workgroup of THIS launch
    uint32_t* defer_count;       // [gridDim.x]
    uint32_t out_seg_cap;
    uint32_t* status;            // [0] = number of records no tier could take
    const uint8_t* comp_lut;     // 256-entry complement table (bio 1.3.1 semantics)
    uint32_t slice_dw;           // LDS dwords available to one wave
    uint32_t flags;              // CK_FLAG_*
};
// Records of segment s for the stages that walk all records of a batch.  Workgroups are dispatched in segment order and a
// kernel ends when its LAST workgroup does: with equal segments the chip drains for one segment's duration (config 4: 32 768
// segments of ~80 us, 65 us = 3.6 % of the kernel with fewer and fewer CUs at work).  The last segments are therefore smaller,
// generation by generation, so that whatever is still running when the list runs out is short.
#ifndef CK_TAPER_GENS
#define CK_TAPER_GENS 3
#endif
CK_DEV void seg_records(const CanonArgs& a, uint32_t s, uint64_t& first, uint32_t& count)
{
    uint64_t f = (uint64_t)s * a.all_seg_cap;
    uint32_t cap = a.all_seg_cap;
    if (a.taper_log2 != 0 && s >= a.taper_seg0) {
        const uint32_t k = (s - a.taper_seg0) >> a.taper_log2, g = k < CK_TAPER_GENS - 1 ? k : CK_TAPER_GENS - 1;
        f = (uint64_t)a.taper_seg0 * a.all_seg_cap;
        for (uint32_t j = 0; j < g; ++j) f += (uint64_t)(a.all_seg_cap >> (j + 1)) << a.taper_log2;
        cap = a.all_seg_cap >> (g + 1);
        f += (uint64_t)(s - a.taper_seg0 - (g << a.taper_log2)) * cap;
    }
    first = f;
    count = (uint32_t)(f >= a.n_records ? 0 : (a.n_records - f < cap ? a.n_records - f : cap));
}
constexpr uint32_t CK_FLAG_FWD_ONLY = 1u;   // lmsr(): forward strand only (lib/src/canonicalize.rs:41-47)


struct RotResult { uint32_t idx; uint32_t period; };
struct Lcp { uint32_t k; int cmp; };

// ------------------------------------------------------------------------------------------------
// packed-word helpers
// ------------------------------------------------------------------------------------------------
// The 32/BITS symbols that start at cyclic symbol position p (0 <= p < 2n), first symbol in the top bits.
template <int BITS>
CK_DEV uint32_t sym_word(const uint32_t* E, uint32_t p, uint32_t n)
{
    constexpr uint32_t S = 32 / BITS;
    p = p >= n ? p - n : p;
    uint32_t w = p / S, sh = (p % S) * BITS;
    return funnel(E[w], E[w + 1], sh);
}

// reverse complement of one packed word: symbol order reversed, every symbol complemented
template <int BITS>
CK_DEV uint32_t rc_word(uint32_t g)
{
    if (BITS == 2) {
        const uint32_t v = bitrev(~g);                      // reverses bits; swap the two bits of every symbol back
        return bfi(0x55555555u, v >> 1, v << 1);
    }
    // 4-bit codes '-'0 A1 C2 G3 N4 T5: complement by v_perm as an 8-entry table (0 5 3 2 4 1), the nibbles of a byte
    // swapped while they are apart, then the bytes reversed
    const uint32_t lo = perm(0x00000104u, 0x02030500u, g & 0x0F0F0F0Fu), hi = perm(0x00000104u, 0x02030500u, (g >> 4) & 0x0F0F0F0Fu);
    return perm(0u, (lo << 4) | hi, 0x00010203u);
}

// A strand as the LDS tiers see it: packed words in E, or -- RCV, 2-bit records only -- the reverse complement of the
// strand in E without a copy of its own: the S symbols at position p of rc(s) are the complement, in reverse order, of
// the forward symbols [n - S - p, n - p) (mod n), so a view costs one forward window + rc_word().  Not keeping the
// second strand halves the LDS a 2-bit record needs, i.e. doubles the records (waves) a CU holds in the one-wave tiers.
template <int BITS, bool RCV>
CK_DEV uint32_t view_word(const uint32_t* E, uint32_t p, uint32_t n)     // p < 2n
{
    if (!RCV) return sym_word<BITS>(E, p, n);
    constexpr uint32_t S = 32 / BITS;
    p = p >= n ? p - n : p;
    const int32_t s0 = (int32_t)n - (int32_t)S - (int32_t)p;
    return rc_word<BITS>(sym_word<BITS>(E, (uint32_t)(s0 + ((s0 >> 31) & (int32_t)n)), n));
}

// min over the S keys that start inside word `cur` (next word `nxt`)
template <int BITS>
CK_DEV uint32_t word_min_key(uint32_t cur, uint32_t nxt)
{
    constexpr int S = 32 / BITS;
    uint32_t m = cur;
#pragma unroll
    for (int b = 1; b < S; ++b) {
        uint32_t k = funnel(cur, nxt, b * BITS);
        m = k < m ? k : m;
    }
    return m;
}

// bit b set iff the key starting at symbol b of word `cur` equals M
template <int BITS>
CK_DEV uint32_t word_eq_mask(uint32_t cur, uint32_t nxt, uint32_t M)
{
    constexpr int S = 32 / BITS;
    uint32_t m = (cur == M) ? 1u : 0u;
#pragma unroll
    for (int b = 1; b < S; ++b) m |= (funnel(cur, nxt, b * BITS) == M ? 1u : 0u) << b;
    return m;
}

// ------------------------------------------------------------------------------------------------
// byte -> code conversion with v_perm_b32 as an 8-entry LUT.
// Index h = (c >> 1) & 7 separates the CLI alphabet: A->0 C->1 T->2 G->3 '-'->6 N->7.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t HASH_MASK = 0x07070707u;
constexpr uint32_t CHK2_LO = 0x47544341u;  // h0..3 -> 'A','C','T','G'
constexpr uint32_t CHK4_HI = 0x4E2D0000u;  // h6 -> '-', h7 -> 'N'

// 2-bit codes A0 C1 G2 T3, first byte -> most significant.
// 16 ASCII bytes -> 16 two-bit codes (first byte in the top bits); miss != 0 iff this lane holds a byte that is not
// A/C/G/T.  Per dword: shift+mask to a 3-bit selector, v_perm for the check byte and for the code, v_sad_u8 to
// accumulate the mismatch (keeps the ORs off the scalar unit), one v_dot4_u32_u8 (weights 64,16,4,1) to gather the
// four codes into a byte.
CK_DEV uint32_t fast_pack(u32x4 v, uint32_t& miss)
{
    const uint32_t d[4] = { v.x, v.y, v.z, v.w };
    uint32_t u[4];
    miss = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t sel = (d[k] >> 1) & HASH_MASK;
        miss = sad_u8(perm(0u, CHK2_LO, sel), d[k], miss);
        u[k] = udot4(perm(0u, 0x02030100u, sel), 0x01041040u, 0u);
    }
    // the four code bytes into one word by v_perm (three instructions; as shifts and ORs the compiler emits six).  Through the
    // builtin: a dot4 result read by inline assembly misses the wait states only the compiler's hazard recognizer inserts
    // (wave_prims.h udot4; seen again in round 4 with v_lshl_or_b32 as the reader: stale registers)
    return perm(perm(u[0], u[1], 0x0c0c0400u), perm(u[2], u[3], 0x0c0c0400u), 0x05040100u);
}

// 256-entry LDS table: packed byte (4 symbols, first in the top bits) -> its 4 ASCII bytes.  Replaces ~6 VALU
// per output dword (spread the 2-bit fields into bytes, v_perm) by shift + mask + one ds_read_b32.
// CK_LUT_STRIDE = 32 (experiment): 32 copies of the table, entry x of copy c at [32 x + c], each lane reads copy
// lane & 31 -- a wave's data-dependent ds_read_b32 then hits 32 different banks per half by construction (one copy: the
// 64 random entries collide, SQ_LDS_BANK_CONFLICT ~20 cycles per record in the headline kernel).
#ifndef CK_LUT_STRIDE
#define CK_LUT_STRIDE 1
#endif
constexpr uint32_t FAST_LUT_DW = 256 * CK_LUT_STRIDE;
CK_DEV void fast_lut_init(uint32_t* lut, uint32_t tid, uint32_t nthreads)
{
    for (uint32_t i = tid; i < FAST_LUT_DW; i += nthreads) {
        const uint32_t x = i / CK_LUT_STRIDE;
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) o |= ((0x54474341u >> (8 * ((x >> (6 - 2 * k)) & 3))) & 0xFFu) << (8 * k);
        lut[i] = o;
    }
}
CK_DEV u32x4 fast_decode(const uint32_t* lut, uint32_t w)
{
    if (CK_LUT_STRIDE == 1) return u32x4{ lut[w >> 24], lut[(w >> 16) & 0xFF], lut[(w >> 8) & 0xFF], lut[w & 0xFF] };
    const uint32_t* l = lut + (lane_id() & (CK_LUT_STRIDE - 1));
    return u32x4{ l[(w >> 24) * CK_LUT_STRIDE], l[((w >> 16) & 0xFF) * CK_LUT_STRIDE], l[((w >> 8) & 0xFF) * CK_LUT_STRIDE], l[(w & 0xFF) * CK_LUT_STRIDE] };
}

// 4-bit codes: '-'0 A1 C2 G3 N4 T5.  8 bytes (two dwords) per packed word.
CK_DEV uint32_t pack4_fwd(uint32_t d0, uint32_t d1, uint32_t& bad)
{
    uint32_t s0 = (d0 >> 1) & HASH_MASK, s1 = (d1 >> 1) & HASH_MASK;
    uint32_t c0 = perm(0x04000000u, 0x03050201u, s0), c1 = perm(0x04000000u, 0x03050201u, s1);
    bad |= (perm(CHK4_HI, CHK2_LO, s0) ^ d0) | (perm(CHK4_HI, CHK2_LO, s1) ^ d1);
    uint32_t t0 = c0 | (c0 << 12), t1 = c1 | (c1 << 12);  // byte1 = [b0 b1], byte3 = [b2 b3]
    return perm(t0, t1, 0x05070103u);
}
CK_DEV uint32_t pack4_rc(uint32_t d0, uint32_t d1)
{
    uint32_t s0 = (d0 >> 1) & HASH_MASK, s1 = (d1 >> 1) & HASH_MASK;
    // comp rank << 4: A->T(5) C->G(3) T->A(1) G->C(2) '-'->0 N->4
    uint32_t c0 = perm(0x40000000u, 0x20103050u, s0), c1 = perm(0x40000000u, 0x20103050u, s1);
    uint32_t t0 = c0 | (c0 << 4), t1 = c1 | (c1 << 4);    // byte3 = [b3 b2], byte1 = [b1 b0]
    return perm(t0, t1, 0x03010705u);
}

// ------------------------------------------------------------------------------------------------
// strand builders.  Ef / Er receive nwv + 2 words each.  Return true when every byte was in the
// alphabet of the mode.
// ------------------------------------------------------------------------------------------------
// team > 1 (2-bit only): wave `member` of `team` builds every team-th trip of rows; the caller joins the waves'
// verdicts and adds the periodic extension (build_extension2) behind a workgroup barrier.
template <int BITS>
CK_DEV bool build_packed(const uint8_t* src, uint32_t n, uint32_t* Ef, uint32_t* Er, uint32_t member = 0, uint32_t team = 1)
{
    constexpr uint32_t S = 32 / BITS;           // bytes consumed per packed word
    constexpr bool RC_FROM_FWD = true;          // the reverse strand is a VIEW of Ef (view_word), never stored (round 3: the 4-bit mode too)
    const uint32_t lane = lane_id();
    const uint32_t nwf = n / S, r = n % S, nwv = nwf + (r ? 1u : 0u);
    uint32_t bad = 0;
    // Rows of 64 words, eight rows (8 KiB of a 2-bit record) per trip with all the 16-byte loads issued before the
    // first is consumed: a long record is otherwise one exposed HBM round trip per row (measured: tier B of
    // BASELINE config 4 spent 61 % of its wave cycles in s_waitcnt with the VALU 20 % busy).
    constexpr int U = CK_BUILD_ROWS;
    for (uint32_t w0 = member * 64 * U + lane; w0 < nwv; w0 += team * 64 * U) {
        u32x4 vf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t w = w0 + 64 * u;
            if (w < nwv) {
                const bool tail = w >= nwf;
                // tail words read the last S bytes of the record (in bounds: n >= 3S) and are shifted up
                const uint32_t fa = tail ? n - S : w * S;
                if (BITS == 2) vf[u] = load16(src + fa);
                else vf[u] = u32x4{ load4(src + fa), load4(src + fa + 4), 0, 0 };
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t w = w0 + 64 * u;
            if (w < nwv) {
                const uint32_t sh = w >= nwf ? (S - r) * BITS : 0u;
                if (BITS == 2) {
                    uint32_t miss;
                    Ef[w] = fast_pack(vf[u], miss) << sh;
                    bad |= miss;
                } else {
                    Ef[w] = pack4_fwd(vf[u].x, vf[u].y, bad) << sh;
                }
            }
        }
    }
    const bool ok = ballot(bad != 0) == 0;
    if (team > 1) return ok;
    wave_sync();
    // periodic extension: E[nwf] gets the head symbols behind the r tail symbols; E[nwv], E[nwv+1] follow.
    if (lane < 3) {
#pragma unroll
        for (int s = 0; s < (RC_FROM_FWD ? 1 : 2); ++s) {
            uint32_t* E = s ? Er : Ef;
            if (lane == 0) {
                if (r) E[nwf] = E[nwf] | (E[0] >> (r * BITS));
            } else {
                const uint32_t e = lane - 1;
                E[nwv + e] = r ? funnel(E[e], E[e + 1], (S - r) * BITS) : E[e];
            }
        }
    }
    wave_sync();
    return ok;
}

// 8-bit mode: any bytes, any n >= 1 (also the path for records shorter than 3 packed words).
CK_DEV void build_bytes(const uint8_t* src, uint32_t n, uint32_t* Ef, uint32_t* Er, const uint8_t* comp)
{
    const uint32_t nwv = (n + 3) / 4;
    for (uint32_t w = lane_id(); w < nwv + 2; w += 64) {
        uint32_t f = 0, c = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4; ++b) {
            const uint32_t x = (4 * w + b) % n;
            f = (f << 8) | src[x];
            c = (c << 8) | comp[src[n - 1 - x]];
        }
        Ef[w] = f;
        Er[w] = c;
    }
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// wave-parallel longest common prefix of rotation i of A and rotation j of B (both length n)
// ------------------------------------------------------------------------------------------------
template <int BITS, bool RA, bool RB>
CK_DEV Lcp lcp_rot(const uint32_t* A, const uint32_t* B, uint32_t i, uint32_t j, uint32_t n)
{
    constexpr uint32_t S = 32 / BITS;
    const uint32_t lane = lane_id();
    for (uint32_t base = 0; base < n; base += 64 * S) {
        const uint32_t off = base + lane * S;
        uint32_t a = 0, b = 0;
        if (off < n) {
            a = view_word<BITS, RA>(A, i + off, n);
            b = view_word<BITS, RB>(B, j + off, n);
        }
        const uint32_t d = a ^ b;
        const uint64_t bal = ballot(d != 0);
        if (bal) {
            const uint32_t l = (uint32_t)ffs64(bal);
            const uint32_t dd = readlane(d, l), aa = readlane(a, l), bb = readlane(b, l);
            const uint32_t k = base + l * S + (uint32_t)clz32(dd) / BITS;
            if (k >= n) return Lcp{ n, 0 };
            return Lcp{ k, aa > bb ? 1 : -1 };
        }
    }
    return Lcp{ n, 0 };
}

// lowest set bit >= x in the candidate bitmask (n bits, no bits >= n set); n if none
CK_DEV uint32_t next_cand(const uint32_t* bm, uint32_t x, uint32_t n)
{
    if (x >= n) return n;
    const uint32_t nbm = (n + 31) / 32, first = x / 32, lane = lane_id();
    for (uint32_t wb = first; wb < nbm; wb += 64) {
        const uint32_t idx = wb + lane;
        uint32_t v = idx < nbm ? bm[idx] : 0u;
        if (idx == first) v &= ~0u << (x % 32);
        const uint64_t bal = ballot(v != 0);
        if (bal) {
            const uint32_t l = (uint32_t)ffs64(bal);
            return (wb + l) * 32 + (uint32_t)ffs32(readlane(v, l));
        }
    }
    return n;
}

// ------------------------------------------------------------------------------------------------
// dense scan of one strand and the unique-minimum shortcut; `word_at(p)` = the S symbols at cyclic position p < 2n
// ------------------------------------------------------------------------------------------------
struct ScanMin { uint32_t M; uint64_t hm; uint32_t bestw, ties; };     // wave minimum key, lanes holding it; per lane: its word, ties
// per-lane minimum key, which word owns it, and how many of the lane's words tie.  Every word is fetched once: the
// word behind it is the next lane's (DPP), lane 63's the first word of the next row.
template <int BITS, class WordAt>
CK_DEV ScanMin dense_scan(WordAt word_at, uint32_t n)
{
    constexpr uint32_t S = 32 / BITS;
    const uint32_t lane = lane_id();
    const uint32_t nwv = (n + S - 1) / S;
    uint32_t best = ~0u, bestw = 0, ties = 0;
    uint32_t cur = lane < nwv + 1 ? word_at(lane * S) : 0u;
    for (uint32_t w0 = 0; w0 < nwv; w0 += 64) {
        const uint32_t w = w0 + lane;
        const uint32_t ahead = w + 64 < nwv + 1 ? word_at((w + 64) * S) : 0u;                     // this lane's word of the next row
        const uint32_t first_of_next_row = readlane(ahead, 0), neighbour = wave_shl1(cur);      // (collectives: every lane, unconditionally)
        const uint32_t nxt = lane == 63 ? first_of_next_row : neighbour;
        if (w < nwv) {
            const uint32_t m = word_min_key<BITS>(cur, nxt);
            if (m < best || ties == 0) { best = m; bestw = w; ties = 1; }
            else if (m == best) ++ties;
        }
        cur = ahead;
    }
    ScanMin r;
    r.M = wave_min_u32(best);
    r.hm = ballot(ties != 0 && best == r.M);
    r.bestw = bestw; r.ties = ties;
    return r;
}
// at most two words hold M (the second is normally the duplicate of word 0's positions that sits behind the record
// end in the last word) and exactly one valid position owns it: that position
template <int BITS, class WordAt>
CK_DEV bool locate_unique(WordAt word_at, uint32_t n, const ScanMin& sm, uint32_t& pos)
{
    constexpr uint32_t S = 32 / BITS;
    const uint32_t lane = lane_id();
    if (popc64(sm.hm) > 2) return false;
    uint32_t cnt = 0;
    uint64_t mm = sm.hm;
    while (mm) {
        const uint32_t l = (uint32_t)ffs64(mm);
        mm &= mm - 1;
        const uint32_t tl = readlane(sm.ties, l), wl = readlane(sm.bestw, l);
        if (tl != 1) return false;
        const uint32_t b = lane % S;
        const uint32_t k = word_at(wl * S + b);
        const uint64_t pm = ballot(lane < S && k == sm.M && wl * S + b < n);
        cnt += (uint32_t)popc64(pm);
        if (pm) pos = wl * S + (uint32_t)ffs64(pm);
    }
    return cnt == 1;
}

// ------------------------------------------------------------------------------------------------
// smallest index of the lexicographically minimal rotation + the rotation period
// ------------------------------------------------------------------------------------------------
// bm_ok: the LDS slice has room for the candidate bitmask.  It is only needed on a tie of the minimal key, which ordinary
// DNA of these lengths practically never has -- so the 2-bit tiers admit records by the strand alone (a 20 kb record:
// 4.9 KiB instead of 7.4) and a record that does tie without room answers idx = NO_ROOM and moves on to the next tier.
constexpr uint32_t NO_ROOM = 0xFFFFFFFFu;
template <int BITS, bool RCV>
CK_DEV RotResult find_min_rot(const uint32_t* E, uint32_t n, uint32_t* bm, bool bm_ok = true)
{
    constexpr uint32_t S = 32 / BITS;
    const uint32_t lane = lane_id();
    const uint32_t nwv = (n + S - 1) / S;
    const auto word_at = [&](uint32_t p) { return view_word<BITS, RCV>(E, p, n); };
    const ScanMin sm = dense_scan<BITS>(word_at, n);
    const uint32_t M = sm.M;
    {
        uint32_t pos = 0;
        if (locate_unique<BITS>(word_at, n, sm, pos)) return RotResult{ pos, n };
    }
    if (!bm_ok) return RotResult{ NO_ROOM, n };

    // general path.  1: candidate bitmask = positions whose key equals M
    const uint32_t nbm = (n + 31) / 32;
    for (uint32_t t = lane; t < nbm; t += 64) bm[t] = 0;
    wave_sync();
    for (uint32_t w = lane; w < nwv; w += 64) {
        uint32_t m = word_eq_mask<BITS>(view_word<BITS, RCV>(E, w * S, n), view_word<BITS, RCV>(E, (w + 1) * S, n), M);
        const uint32_t valid = n - w * S;
        if (valid < S) m &= (1u << valid) - 1u;
        const uint32_t p = w * S;
        if (m) lds_atomic_or(&bm[p / 32], m << (p % 32));
    }
    wave_sync();
    // 2: duel.  Invariant: alive candidates = {i} U {candidates >= j}; a minimal start is never killed.
    uint32_t i = next_cand(bm, 0, n);
    uint32_t j = next_cand(bm, i + 1, n);
    uint32_t period = n;
    while (j < n) {
        const Lcp c = lcp_rot<BITS, RCV, RCV>(E, E, i, j, n);
        if (c.k >= n) { period = j - i; break; }       // equal rotations: i is the smallest minimal start
        if (c.cmp > 0) {                                // rotation j is smaller: i .. i+k are dead
            const uint32_t ni = (i + c.k + 1 <= j) ? j : next_cand(bm, i + c.k + 1, n);
            if (ni >= n) break;                         // unreachable: a minimal start always survives
            i = ni;
            j = next_cand(bm, i + 1, n);
        } else {                                        // rotation i is smaller: j .. j+k are dead
            j = next_cand(bm, j + c.k + 1, n);
        }
    }
    return RotResult{ i, period };
}

// ------------------------------------------------------------------------------------------------
// decode the winner back to bytes and store it
// ------------------------------------------------------------------------------------------------
CK_DEV void store_bytes(uint8_t* p, u32x4 v, uint32_t nb)   // nb = 1..16 bytes of v
{
    if (nb >= 16) { store16(p, v); return; }
    if (nb & 8) { store8(p, v.x, v.y); p += 8; v.x = v.z; v.y = v.w; }
    if (nb & 4) { store4(p, v.x); p += 4; v.x = v.y; }
    if (nb & 2) { p[0] = (uint8_t)v.x; p[1] = (uint8_t)(v.x >> 8); p += 2; v.x >>= 16; }
    if (nb & 1) { p[0] = (uint8_t)v.x; }
}

// 16 symbols (2 bits each, first in the top bits) -> 16 ASCII bytes
CK_DEV u32x4 decode2(uint32_t v)
{
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t x = (v >> (24 - 8 * k)) & 0xFFu;
        const uint32_t t = x | (x << 10);
        const uint32_t z = t | (t << 20);
        o[k] = perm(0u, 0x54474341u, (z >> 6) & 0x03030303u);   // 0..3 -> 'A','C','G','T'
    }
    return u32x4{ o[0], o[1], o[2], o[3] };
}
// 8 symbols (4 bits each) -> 8 ASCII bytes in (lo, hi)
CK_DEV void decode4(uint32_t v, uint32_t& lo, uint32_t& hi)
{
    const uint32_t a = perm(0u, v, 0x02020303u), b = perm(0u, v, 0x00000101u);
    const uint32_t sa = ((a >> 4) & 0x000F000Fu) | (a & 0x0F000F00u);
    const uint32_t sb = ((b >> 4) & 0x000F000Fu) | (b & 0x0F000F00u);
    lo = perm(0x0000544Eu, 0x4743412Du, sa);                       // 0..5 -> '-','A','C','G','N','T'
    hi = perm(0x0000544Eu, 0x4743412Du, sb);
}

// lut: the 256-entry packed-byte -> 4 ASCII bytes table of fast_lut_init() in LDS (2-bit mode), or nullptr
template <int BITS, bool RCV>
CK_DEV u32x4 decode_word(const uint32_t* E, uint32_t p, uint32_t n, const uint32_t* lut)      // the S symbols at position p as ASCII
{
    const uint32_t v = view_word<BITS, RCV>(E, p, n);
    u32x4 o{ 0, 0, 0, 0 };
    if (BITS == 2) {
        o = lut ? fast_decode(lut, v) : decode2(v);
    } else if (BITS == 4) {
        decode4(v, o.x, o.y);
    } else {
        o.x = perm(0u, v, 0x00010203u);
    }
    return o;
}
// (Cutting the output at the 16-byte boundaries of the destination -- aligned dwordx4 stores -- and building the 2-bit
// strand from aligned loads + a funnel on the packed words were both tried on BASELINE config 4: 2.39 -> 2.8 ms.)
template <int BITS, bool RCV>
CK_DEV void emit(const uint32_t* E, uint32_t idx, uint32_t n, uint8_t* out, const uint32_t* lut)
{
    constexpr uint32_t S = 32 / BITS;
    const uint32_t nwv = (n + S - 1) / S;
    for (uint32_t w = lane_id(); w < nwv; w += 64) {
        const u32x4 o = decode_word<BITS, RCV>(E, idx + w * S, n, lut);
        const uint32_t left = n - w * S;
        store_bytes(out + w * S, o, left < S ? left : S);
    }
}

// ------------------------------------------------------------------------------------------------
// one record, one mode.  Returns false if the record's bytes are outside the mode's alphabet.
// ------------------------------------------------------------------------------------------------
template <int BITS>
CK_DEV uint32_t need_dw(uint32_t n)
{
    constexpr uint32_t S = 32 / BITS;
    return (BITS == 8 ? 2 : 1) * ((n + S - 1) / S + 2) + (n + 31) / 32 + 1;      // one stored strand in the 2- and 4-bit modes, two in byte mode
}
CK_DEV uint32_t need_dw_strand2(uint32_t n) { return (n + 15) / 16 + 2; }         // 2-bit mode without the candidate bitmask

// 0: done; 1: a byte outside the mode's alphabet; 2 (2-bit mode only): the minimal key ties and the slice has no room for
// the candidate bitmask
template <int BITS>
CK_DEV int canon_record_mode(const CanonArgs& a, uint64_t rec, const uint8_t* src, uint64_t off, uint32_t n,
                              uint32_t* lds, const uint32_t* lut)
{
    constexpr uint32_t S = 32 / BITS;
    constexpr bool RCV = BITS != 8;              // reverse strand = a view of the forward words (2- and 4-bit modes)
    const uint32_t nwv = (n + S - 1) / S;
    uint32_t* Ef = lds;
    uint32_t* Er = RCV ? lds : lds + (nwv + 2);
    uint32_t* bm = lds + (RCV ? 1 : 2) * (nwv + 2);
    if (BITS == 8) {
        build_bytes(src, n, Ef, Er, a.comp_lut);
    } else {
        if (!build_packed<BITS>(src, n, Ef, Er)) return 1;
    }
    const bool bm_ok = BITS != 2 || need_dw<2>(n) <= a.slice_dw;
    const RotResult f = find_min_rot<BITS, false>(Ef, n, bm, bm_ok);
    if (f.idx == NO_ROOM) return 2;
    RotResult r{ 0, n };
    bool fwd = true;
    if (!(a.flags & CK_FLAG_FWD_ONLY)) {
        r = find_min_rot<BITS, RCV>(Er, n, bm, bm_ok);
        if (r.idx == NO_ROOM) return 2;
        // lib/src/canonicalize.rs:58-62: forward only if strictly smaller
        fwd = lcp_rot<BITS, false, RCV>(Ef, Er, f.idx, r.idx, n).cmp < 0;
    }
    if (a.out_bytes) {
        if (fwd) emit<BITS, false>(Ef, f.idx, n, a.out_bytes + off, lut);
        else emit<BITS, RCV>(Er, r.idx, n, a.out_bytes + off, lut);
    }
    if (lane_id() == 0) {
        // index as the reference would see it: lmsr_index(s) for the forward strand,
        // lmsr_index(revcomp(lmsr(s))) for the reverse strand (rotation by f.idx, modulo the period)
        if (a.out_index) a.out_index[rec] = fwd ? f.idx : (r.idx + f.idx) % f.period;
        if (a.out_strand) a.out_strand[rec] = fwd ? 0 : 1;
        if (a.out_view) a.out_view[rec] = fwd ? f.idx : (r.idx | 0x80000000u);
    }
    return 0;
}


// ------------------------------------------------------------------------------------------------
// 2-bit mode for records with a FEW N ("2N"): the reference sorts N like any other byte (between G and T,
// lib/src/canonicalize.rs:50-53), but one N in 20 kb should not push the record into the 4-bit mode at 2-3x the cost.
// N is packed as G -- so the reverse-strand view shows a C there -- and remembered in a bitmask, one bit per symbol.
// On either strand every key that holds an N is then SMALLER than its true value (C < G < N) and every other key is
// exact, so:
//   * if the smaller of the two strands' minimal keys is owned by exactly one position whose 16-symbol window is
//     N-free, that position is the true minimal rotation of the true winning strand (any window with an N has a true
//     key above its packed key >= that minimum; the other strand's true minimum is above its packed minimum);
//   * anything else -- equal minimal keys, a tie, an N inside the winning window -- is left to the 4-bit mode.
// The scans never look at the mask (the 2-bit scan as it is); it is read once at the winning position and along the
// output, where the decoded 'G' (forward) or 'C' (reverse) is patched to 'N' through a 16-entry table.
// LDS: the strand + n / 32 + 2 dwords (the first version kept a second STRAND of masks and corrected the reverse view
// with it in every scan step: twice the LDS, so that a 20 kb record with one N fell to the 13 KiB tier).
// ------------------------------------------------------------------------------------------------
// 16 ASCII bytes -> 16 two-bit codes with N -> G, the N mask (bit 15 = first symbol), miss != 0 iff a byte is outside ACGTN
CK_DEV uint32_t fast_pack_n(u32x4 v, uint32_t& nmask16, uint32_t& miss)
{
    const uint32_t d[4] = { v.x, v.y, v.z, v.w };
    uint32_t u[4], m[4];
    miss = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t sel = (d[k] >> 1) & HASH_MASK;                       // A0 C1 T2 G3 ... N7
        miss = sad_u8(perm(0x4E000000u, CHK2_LO, sel), d[k], miss);
        u[k] = udot4(perm(0x02000000u, 0x02030100u, sel), 0x01041040u, 0u);
        m[k] = udot4((sel >> 2) & 0x01010101u, 0x01020408u, 0u);            // selector bit 2: N (everything else up there is refused)
    }
    nmask16 = (((m[0] << 4 | m[1]) << 4 | m[2]) << 4) | m[3];
    return (((u[0] << 8 | u[1]) << 8 | u[2]) << 8) | u[3];
}
// the same with the mask in the strand's own layout (0b11 at every N, first symbol in the top bits): the register routine
// moves it through the shuffles and funnels its strand words take
CK_DEV uint32_t fast_pack_n2(u32x4 v, uint32_t& nmask, uint32_t& miss)
{
    const uint32_t d[4] = { v.x, v.y, v.z, v.w };
    uint32_t u[4], m[4];
    miss = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t sel = (d[k] >> 1) & HASH_MASK;                       // A0 C1 T2 G3 ... N7
        miss = sad_u8(perm(0x4E000000u, CHK2_LO, sel), d[k], miss);
        u[k] = udot4(perm(0x02000000u, 0x02030100u, sel), 0x01041040u, 0u);
        m[k] = udot4((sel >> 2) & 0x01010101u, 0x030C30C0u, 0u);
    }
    nmask = (((m[0] << 8 | m[1]) << 8 | m[2]) << 8) | m[3];
    return (((u[0] << 8 | u[1]) << 8 | u[2]) << 8) | u[3];
}
// 16-entry LDS table: 4 mask bits (bit 3 = first symbol) -> 0x01 in the byte of every N (times 'G' ^ 'N' or 'C' ^ 'N')
CK_DEV void fast_lutn_init(uint32_t* lutn, uint32_t tid, uint32_t nthreads)
{
    for (uint32_t x = tid; x < 16; x += nthreads) {
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) o |= ((x >> (3 - k)) & 1u) << (8 * k);
        lutn[x] = o;
    }
}
constexpr uint32_t FAST_LUTN_DW = 16;
// the mask bits of the 16 symbols at cyclic position p (0 <= p < 2n), first symbol in bit 15
CK_DEV uint32_t nmask16_at(const uint32_t* Nb, uint32_t p, uint32_t n)
{
    p = p >= n ? p - n : p;
    return funnel(Nb[p >> 5], Nb[(p >> 5) + 1], p & 31) >> 16;
}
// forward words (+ periodic extension) and the N bitmask (32 symbols per dword, first symbol in the top bit, extended
// periodically by 32 symbols or more); returns false when a byte is outside ACGTN
// the N bitmask's periodic extension (one lane; strand-sized masks complete and visible): symbols n .. n + 32 (or more)
// repeat symbols 0 ..; the bits past n in dword n / 32 are zero so far
CK_DEV void build_extension_nb(uint32_t* Nb, uint32_t n)
{
    const uint32_t q = n & 31, nd = n >> 5, a0 = Nb[0];
    const uint32_t x = q ? (Nb[nd] | (a0 >> q)) : a0;
    const uint32_t a1 = nd == 1 ? x : Nb[1];
    Nb[nd] = x;
    Nb[nd + 1] = q ? ((a0 << (32 - q)) | (a1 >> q)) : a1;
}
// team > 1: wave `member` builds every team-th trip of rows and leaves the extensions to the caller (team mode)
CK_DEV bool build_packed2n(const uint8_t* src, uint32_t n, uint32_t* Ef, uint32_t* Nb, uint32_t member = 0, uint32_t team = 1)
{
    const uint32_t lane = lane_id();
    const uint32_t nwf = n >> 4, r = n & 15, nwv = nwf + (r ? 1u : 0u);
    uint32_t bad = 0;
    constexpr int U = CK_BUILD_ROWS;
    for (uint32_t w0 = member * 64 * U; w0 < nwv; w0 += team * 64 * U) {             // (wave-uniform trip count: the mask halves meet by DPP)
        u32x4 vf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t w = w0 + lane + 64 * u;
            if (w < nwv) vf[u] = load16(src + (w >= nwf ? n - 16 : w * 16));        // the tail word reads the record's last 16 bytes
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t w = w0 + lane + 64 * u;
            uint32_t nm = 0;
            if (w < nwv) {
                const uint32_t sh = w >= nwf ? 16 - r : 0u;                         // tail: the last r symbols move to the top
                uint32_t miss;
                Ef[w] = fast_pack_n(vf[u], nm, miss) << (2 * sh);
                nm = (nm << sh) & 0xFFFFu;
                bad |= miss;
            }
            const uint32_t nx = wave_shl1(nm);                                      // the odd neighbour's 16 bits (0 past the record)
            if (!(lane & 1) && w < nwv) Nb[w >> 1] = (nm << 16) | nx;
        }
    }
    const bool ok = ballot(bad != 0) == 0;
    if (team > 1) return ok;
    wave_sync();
    if (lane < 3) {
        if (lane == 0) {
            if (r) Ef[nwf] = Ef[nwf] | (Ef[0] >> (r * 2));
        } else {
            const uint32_t e = lane - 1;
            Ef[nwv + e] = r ? funnel(Ef[e], Ef[e + 1], (16 - r) * 2) : Ef[e];
        }
    } else if (lane == 3) {
        build_extension_nb(Nb, n);
    }
    wave_sync();
    return ok;
}
CK_DEV uint32_t need_dw_2n(uint32_t n) { return (n + 15) / 16 + 2 + (n >> 5) + 2; }

// how many of the strand's n positions start a key that agrees with K in its first 32 - sh bits (simple loop: rare path)
// (member / team: this wave's share of the rows in team mode)
template <class WordAt>
CK_DEV uint32_t count_prefix2(WordAt word_at, uint32_t n, uint32_t K, uint32_t sh, uint32_t member = 0, uint32_t team = 1)
{
    const uint32_t nwv = (n + 15) / 16;
    uint32_t cnt = 0;
    for (uint32_t w = member * 64 + lane_id(); w < nwv; w += team * 64) {
        const uint32_t cur = word_at(w * 16), nxt = word_at((w + 1) * 16);
        const uint32_t valid = n - w * 16 < 16 ? n - w * 16 : 16;
        for (uint32_t b = 0; b < valid; ++b) cnt += ((funnel(cur, nxt, 2 * b) ^ K) >> sh) == 0 ? 1u : 0u;
    }
    return (uint32_t)wave_sum_u64(cnt);
}

// 0: done; 1: leave it to the 4-bit mode (a byte outside ACGTN, or one of the cases listed above)
CK_DEV int canon_record_mode2n(const CanonArgs& a, uint64_t rec, const uint8_t* src, uint64_t off, uint32_t n, uint32_t* lds,
                               const uint32_t* lut, const uint32_t* lutn)
{
    const uint32_t nwv = (n + 15) / 16;
    uint32_t* Ef = lds;
    uint32_t* Nb = lds + (nwv + 2);
    if (!build_packed2n(src, n, Ef, Nb)) return 1;
    const auto mirror = [&](uint32_t p) {                   // start of the forward window behind reverse-strand position p
        p = p >= n ? p - n : p;
        const int32_t s0 = (int32_t)n - 16 - (int32_t)p;
        return (uint32_t)(s0 + ((s0 >> 31) & (int32_t)n));
    };
    const auto fwd_at = [&](uint32_t p) { return sym_word<2>(Ef, p, n); };
    const auto rc_at = [&](uint32_t p) { return rc_word<2>(sym_word<2>(Ef, mirror(p), n)); };
    const bool fwd_only = (a.flags & CK_FLAG_FWD_ONLY) != 0;
    const ScanMin sf = dense_scan<2>(fwd_at, n);
    ScanMin sc = sf;
    if (!fwd_only) {
        sc = dense_scan<2>(rc_at, n);
        if (sf.M == sc.M) return 1;
    }
    const bool fwd = fwd_only || sf.M < sc.M;
    uint32_t pos = 0, fpos = 0;
    const bool uq = fwd ? locate_unique<2>(fwd_at, n, sf, pos) : locate_unique<2>(rc_at, n, sc, pos);
    if (!uq) return 1;
    {
        // An N inside the winning window, first at offset j.  Every other key is above K in the packed order; one that
        // differs from K before offset j stays above it in the true order (K's symbols there are exact, the other key's can
        // only grow).  Forward winner (N packed as G): a key that shares the first j symbols and differs at offset j holds
        // a T there, and T > N -- so the position still wins if nothing else, on either strand, shares its first j + 1
        // packed symbols.  Reverse winner (its N shows as C): a sharer of the first j symbols may hold a true G at offset
        // j, below N -- so nothing else may share the first j symbols.  With 1 % N one window in ten holds an N, mostly
        // far enough back for the prefix to be unique: without this a tenth of the long records went on to the 4-bit
        // mode, which for 20 kb means the 39 KiB tier at four waves per CU.
        const uint32_t mw = fwd ? nmask16_at(Nb, pos, n) : bitrev(nmask16_at(Nb, mirror(pos), n)) >> 16;
        if (mw != 0) {
            const uint32_t j = (uint32_t)clz32(mw) - 16, plen = fwd ? j + 1 : j;
            if (plen == 0) return 1;
            const uint32_t sh = 32 - 2 * plen, K = fwd ? sf.M : sc.M;
            uint32_t c = count_prefix2(fwd_at, n, K, sh);
            if (!fwd_only) c += count_prefix2(rc_at, n, K, sh);
            if (c != 1) return 1;
        }
    }
    if (!fwd && a.out_index) {                                                  // the reference-visible index counts from the forward minimum
        if (!locate_unique<2>(fwd_at, n, sf, fpos) || nmask16_at(Nb, fpos, n) != 0) return 1;
    }
    if (a.out_bytes) {
        uint8_t* out = a.out_bytes + off;
        const uint32_t fix = fwd ? 0x09u : 0x0Du;                               // 'G' ^ 'N', 'C' ^ 'N'
        for (uint32_t w = lane_id(); w < nwv; w += 64) {
            const uint32_t p = pos + w * 16;
            const uint32_t v = fwd ? fwd_at(p) : rc_at(p);
            const uint32_t m = fwd ? nmask16_at(Nb, p, n) : bitrev(nmask16_at(Nb, mirror(p), n)) >> 16;
            u32x4 o = fast_decode(lut, v);
            if (m) {
                o.x ^= lutn[m >> 12] * fix; o.y ^= lutn[(m >> 8) & 15] * fix; o.z ^= lutn[(m >> 4) & 15] * fix; o.w ^= lutn[m & 15] * fix;
            }
            const uint32_t left = n - w * 16;
            store_bytes(out + w * 16, o, left < 16 ? left : 16);
        }
    }
    if (lane_id() == 0) {
        if (a.out_index) a.out_index[rec] = fwd ? pos : (pos + fpos) % n;        // unique minimum: period n
        if (a.out_strand) a.out_strand[rec] = fwd ? 0 : 1;
        if (a.out_view) a.out_view[rec] = fwd ? pos : (pos | 0x80000000u);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Team mode: the waves of a workgroup canonicalize ONE pure-ACGT record together, its strand in the workgroup's whole LDS
// (all the per-wave slices side by side).  For records too long for one wave's slice that would otherwise go down to the
// one-wave tiers, where 7 to 157 KiB of LDS per wave leave 18 to 1 waves per CU: here the CU keeps its 28 waves, four
// to a record.  Rows of 64 words are dealt to the waves in turn (build, scans, output); the waves meet in three LDS words
// (minimal key by atomic min, number of positions that own it, the smallest of them) between workgroup barriers.  Same
// answers as canon_record_mode<2> on its unique-minimum path; a tie, equal strands or a byte outside ACGT leave the
// record untouched for the tiers behind.  Every wave of the workgroup runs through the same barriers: all decisions are
// taken on values read back from LDS.
// ------------------------------------------------------------------------------------------------
template <int BITS>
CK_DEV void build_extension(uint32_t* Ef, uint32_t n)      // lanes 0..2 of one wave, the strand complete and visible
{
    constexpr uint32_t S = 32 / BITS;
    const uint32_t lane = lane_id(), nwf = n / S, r = n % S, nwv = nwf + (r ? 1u : 0u);
    if (lane == 0) {
        if (r) Ef[nwf] = Ef[nwf] | (Ef[0] >> (r * BITS));
    } else if (lane < 3) {
        const uint32_t e = lane - 1;
        Ef[nwv + e] = r ? funnel(Ef[e], Ef[e + 1], (S - r) * BITS) : Ef[e];
    }
}
CK_DEV void build_extension2(uint32_t* Ef, uint32_t n) { build_extension<2>(Ef, n); }
// one strand: minimal key M, how many valid positions own it, the smallest of them -- and what this lane knows about the
// owners (team_settle)
struct TeamOwn { uint32_t M, owners, pos, lane_word, lane_mask; };
template <int BITS, bool RCV>
CK_DEV TeamOwn team_scan_locate(const uint32_t* E, uint32_t n, uint32_t member, uint32_t team, uint32_t* comm)
{
    constexpr uint32_t S = 32 / BITS;
    const uint32_t lane = lane_id(), nwv = (n + S - 1) / S;
    if (member == 0 && lane == 0) { comm[0] = ~0u; comm[1] = 0; comm[2] = ~0u; }
    block_barrier();
    uint32_t best = ~0u, bestw = 0, ties = 0;
    for (uint32_t w = member * 64 + lane; w < nwv; w += team * 64) {
        const uint32_t m = word_min_key<BITS>(view_word<BITS, RCV>(E, w * S, n), view_word<BITS, RCV>(E, (w + 1) * S, n));
        if (m < best || ties == 0) { best = m; bestw = w; ties = 1; }
        else if (m == best) ++ties;
    }
    const uint32_t Mw = wave_min_u32(best);
    if (lane == 0) lds_atomic_min(comm, Mw);
    block_barrier();
    TeamOwn r;
    r.M = comm[0]; r.lane_word = bestw; r.lane_mask = 0;
    uint32_t cnt = 0, p = ~0u;
    if (ties != 0 && best == r.M) {
        if (ties > 1) {
            cnt = 64;                                           // several of this lane's words hold M: more than the team settles
        } else {
            uint32_t mask = word_eq_mask<BITS>(view_word<BITS, RCV>(E, bestw * S, n), view_word<BITS, RCV>(E, (bestw + 1) * S, n), r.M);
            const uint32_t valid = n - bestw * S;
            if (valid < S) mask &= (1u << valid) - 1u;          // the last word's positions behind the record end repeat word 0's
            r.lane_mask = mask;
            cnt = (uint32_t)popc32(mask);
            if (mask) p = bestw * S + (uint32_t)ffs32(mask);
        }
    }
    const uint32_t cw = (uint32_t)wave_sum_u64(cnt), pw = wave_min_u32(p);
    if (lane == 0 && cw) { lds_atomic_add(comm + 1, cw); lds_atomic_min(comm + 2, pw); }
    block_barrier();
    r.owners = comm[1]; r.pos = comm[2];
    block_barrier();                                            // everybody has the answers before the words are reset
    return r;
}
// A minimal key with a few owners -- a homopolymer run of 17 or more is a row of them, and a 600 kb record has one as its
// minimal key once in a few thousand; the general routine then is one wave with a candidate bitmask that no longer fits
// the LDS, a millisecond in the global-memory stage.  The owners are taken in ascending order (an LDS atomic min per round)
// and the smallest rotation kept (lcp_rot: every wave computes the same).  false: too many owners, or two equal rotations
// (a period): the general routine's.
constexpr uint32_t TEAM_MAX_OWNERS = 8;
template <int BITS, bool RCV>
CK_DEV bool team_settle(const uint32_t* E, uint32_t n, uint32_t member, uint32_t* comm, TeamOwn& o)
{
    constexpr uint32_t S = 32 / BITS;
    if (o.owners == 1) return true;
    if (o.owners > TEAM_MAX_OWNERS) return false;
    const uint32_t lane = lane_id(), base = o.lane_word * S;
    uint32_t best = o.pos, last = o.pos;
    for (uint32_t k = 1; k < o.owners; ++k) {
        if (member == 0 && lane == 0) comm[2] = ~0u;
        block_barrier();
        uint32_t m = o.lane_mask;                               // this lane's owners behind `last`
        if (m && last >= base) m = last - base >= S - 1 ? 0u : m & ~((2u << (last - base)) - 1u);
        const uint32_t pw = wave_min_u32(m ? base + (uint32_t)ffs32(m) : ~0u);
        if (lane == 0 && pw != ~0u) lds_atomic_min(comm + 2, pw);
        block_barrier();
        const uint32_t nxt = comm[2];
        block_barrier();
        if (nxt == ~0u) return false;
        const Lcp c = lcp_rot<BITS, RCV, RCV>(E, E, best, nxt, n);
        if (c.k >= n) return false;
        if (c.cmp > 0) best = nxt;
        last = nxt;
    }
    o.pos = best;
    return true;
}
// lds: the workgroup's slices (team * slice_dw dwords); comm: three LDS words
// returns 0: done; 1: a byte outside the mode's alphabet (ACGT; BITS = 4: -ACGNT); 2: a tie / equal strands (the general routine's business)
// BITS = 4 (round 4): the same team for what the 2-bit modes refuse -- an N inside the winning window behind a prefix other
// windows share, gaps -- with 8-symbol keys (so: more often a few owners of the minimal key, team_settle's); before it such a
// record beyond a slice was wave 0's alone (a batch of 40 000 records of 20..80 kb with 1 % N: 3.28 -> 2.13 ms)
template <int BITS>
CK_DEV int canon_record_team(const CanonArgs& a, uint64_t rec, uint32_t* lds, const uint32_t* lut, uint32_t* comm, uint32_t member, uint32_t team)
{
    constexpr uint32_t S = 32 / BITS;
    const uint64_t off = a.offsets[rec];
    const uint32_t n = (uint32_t)(a.offsets[rec + 1] - off), nwv = (n + S - 1) / S, lane = lane_id();
    const uint8_t* src = a.bytes + off;
    uint32_t* E = lds;
    if (member == 0 && lane == 0) comm[0] = 0;
    block_barrier();
    if (!build_packed<BITS>(src, n, E, E, member, team) && lane == 0) lds_atomic_or(comm, 1u);
    block_barrier();
    const bool not_acgt = comm[0] != 0;
    if (member == 0) build_extension<BITS>(E, n);
    block_barrier();                                            // (also: comm[0] has been read by everybody)
    if (not_acgt) return 1;
    const bool fwd_only = (a.flags & CK_FLAG_FWD_ONLY) != 0;
    TeamOwn F = team_scan_locate<BITS, false>(E, n, member, team, comm), C = F;
    if (!fwd_only) C = team_scan_locate<BITS, true>(E, n, member, team, comm);
    // lib/src/canonicalize.rs:58-62: forward only if strictly smaller.  Equal minimal keys: the two rotations are compared
    // in full; a minimal key with a few owners: the smallest of their rotations.  More owners, periods: the general routine.
    bool fwd = fwd_only || F.M < C.M;
    if (!fwd_only && F.M == C.M) {
        if (!team_settle<BITS, false>(E, n, member, comm, F) || !team_settle<BITS, true>(E, n, member, comm, C)) return 2;
        fwd = lcp_rot<BITS, false, true>(E, E, F.pos, C.pos, n).cmp < 0;
    } else {
        if (!(fwd ? team_settle<BITS, false>(E, n, member, comm, F) : team_settle<BITS, true>(E, n, member, comm, C))) return 2;
        if (!fwd && a.out_index && !team_settle<BITS, false>(E, n, member, comm, F)) return 2;     // the reference-visible index counts from the forward minimum
    }
    const uint32_t pF = F.pos, pC = C.pos;
    const uint32_t idx = fwd ? pF : pC;
    if (a.out_bytes) {
        uint8_t* out = a.out_bytes + off;
        for (uint32_t w = member * 64 + lane; w < nwv; w += team * 64) {
            const u32x4 o = fwd ? decode_word<BITS, false>(E, idx + w * S, n, lut) : decode_word<BITS, true>(E, idx + w * S, n, lut);
            const uint32_t left = n - w * S;
            store_bytes(out + w * S, o, left < S ? left : S);
        }
    }
    if (member == 0 && lane == 0) {
        if (a.out_index) a.out_index[rec] = fwd ? pF : (pC + pF) % n;               // unique minima: period n
        if (a.out_strand) a.out_strand[rec] = fwd ? 0 : 1;
        if (a.out_view) a.out_view[rec] = fwd ? pF : (pC | 0x80000000u);
    }
    block_barrier();                                            // the strand is read to the end before the next record's build
    return 0;
}
CK_DEV int canon_record_team2(const CanonArgs& a, uint64_t rec, uint32_t* lds, const uint32_t* lut, uint32_t* comm, uint32_t member, uint32_t team)
{
    return canon_record_team<2>(a, rec, lds, lut, comm, member, team);
}
// (As a call of its own -- __noinline__, arguments by value -- the 4-bit team took the tier kernel around it from 72 to 87
// vector registers; inlined it costs that kernel five spilled registers on its own path.)
CK_DEV int canon_record_team4(const CanonArgs& a, uint64_t rec, uint32_t* lds, uint32_t* comm, uint32_t member, uint32_t team)
{
    return canon_record_team<4>(a, rec, lds, nullptr, comm, member, team);
}
// a record the team takes: too long for one wave's slice, short enough for all of them together
CK_DEV bool team_takes(uint32_t n, uint32_t slice_dw, uint32_t team) { return need_dw_strand2(n) > slice_dw && need_dw_strand2(n) <= team * slice_dw; }

// The same for a record with a few N (canon_record_mode2n's rules: N packed as G, one mask bit per symbol, scans that never
// read the mask, the prefix rule for an N inside the winning window).  false: left for the 4-bit mode of the tiers behind.
CK_DEV bool canon_record_team2n(const CanonArgs& a, uint64_t rec, uint32_t* lds, const uint32_t* lut, const uint32_t* lutn, uint32_t* comm,
                                uint32_t member, uint32_t team)
{
    const uint64_t off = a.offsets[rec];
    const uint32_t n = (uint32_t)(a.offsets[rec + 1] - off), nwv = (n + 15) / 16, lane = lane_id();
    const uint8_t* src = a.bytes + off;
    uint32_t* E = lds;
    uint32_t* Nb = lds + (nwv + 2);
    if (member == 0 && lane == 0) comm[0] = 0;
    block_barrier();
    if (!build_packed2n(src, n, E, Nb, member, team) && lane == 0) lds_atomic_or(comm, 1u);
    block_barrier();
    const bool refused = comm[0] != 0;
    if (member == 0) { build_extension2(E, n); if (lane == 3) build_extension_nb(Nb, n); }
    block_barrier();
    if (refused) return false;
    const auto mirror = [&](uint32_t p) {
        p = p >= n ? p - n : p;
        const int32_t s0 = (int32_t)n - 16 - (int32_t)p;
        return (uint32_t)(s0 + ((s0 >> 31) & (int32_t)n));
    };
    const bool fwd_only = (a.flags & CK_FLAG_FWD_ONLY) != 0;
    const TeamOwn F = team_scan_locate<2, false>(E, n, member, team, comm);
    TeamOwn C = F;
    if (!fwd_only) C = team_scan_locate<2, true>(E, n, member, team, comm);
    const uint32_t MF = F.M, oF = F.owners, pF = F.pos, MC = fwd_only ? ~0u : C.M, oC = fwd_only ? 1u : C.owners, pC = C.pos;
    if (!fwd_only && MF == MC) return false;
    const bool fwd = fwd_only || MF < MC;
    if ((fwd ? oF : oC) != 1) return false;
    const uint32_t pos = fwd ? pF : pC;
    const uint32_t mw = fwd ? nmask16_at(Nb, pos, n) : bitrev(nmask16_at(Nb, mirror(pos), n)) >> 16;
    if (mw != 0) {                                              // the prefix rule (see canon_record_mode2n)
        const uint32_t j = (uint32_t)clz32(mw) - 16, plen = fwd ? j + 1 : j;
        if (plen == 0) return false;
        const uint32_t sh = 32 - 2 * plen, K = fwd ? MF : MC;
        if (member == 0 && lane == 0) comm[1] = 0;
        block_barrier();
        uint32_t c = count_prefix2([&](uint32_t p) { return view_word<2, false>(E, p, n); }, n, K, sh, member, team);
        if (!fwd_only) c += count_prefix2([&](uint32_t p) { return view_word<2, true>(E, p, n); }, n, K, sh, member, team);
        if (lane == 0 && c) lds_atomic_add(comm + 1, c);
        block_barrier();
        const uint32_t sharers = comm[1];
        block_barrier();
        if (sharers != 1) return false;
    }
    if (!fwd && a.out_index && (oF != 1 || nmask16_at(Nb, pF, n) != 0)) return false;
    if (a.out_bytes) {
        uint8_t* out = a.out_bytes + off;
        const uint32_t fix = fwd ? 0x09u : 0x0Du;                               // 'G' ^ 'N', 'C' ^ 'N'
        for (uint32_t w = member * 64 + lane; w < nwv; w += team * 64) {
            const uint32_t p = pos + w * 16;
            const uint32_t m = fwd ? nmask16_at(Nb, p, n) : bitrev(nmask16_at(Nb, mirror(p), n)) >> 16;
            u32x4 o = fwd ? decode_word<2, false>(E, p, n, lut) : decode_word<2, true>(E, p, n, lut);
            if (m) {
                o.x ^= lutn[m >> 12] * fix; o.y ^= lutn[(m >> 8) & 15] * fix; o.z ^= lutn[(m >> 4) & 15] * fix; o.w ^= lutn[m & 15] * fix;
            }
            const uint32_t left = n - w * 16;
            store_bytes(out + w * 16, o, left < 16 ? left : 16);
        }
    }
    if (member == 0 && lane == 0) {
        if (a.out_index) a.out_index[rec] = fwd ? pF : (pC + pF) % n;
        if (a.out_strand) a.out_strand[rec] = fwd ? 0 : 1;
        if (a.out_view) a.out_view[rec] = fwd ? pF : (pC | 0x80000000u);
    }
    block_barrier();
    return true;
}
CK_DEV bool team_takes_2n(uint32_t n, uint32_t slice_dw, uint32_t team) { return n >= 48 && need_dw_2n(n) > slice_dw && need_dw_2n(n) <= team * slice_dw; }
// ...and what is left for the 4-bit team (the one-wave 4-bit mode did not fit its slice: strand + candidate bitmask): the strand alone
CK_DEV uint32_t need_dw_strand4(uint32_t n) { return (n + 7) / 8 + 2; }
CK_DEV bool team_takes_4(uint32_t n, uint32_t slice_dw, uint32_t team) { return n >= 48 && need_dw_strand4(n) <= team * slice_dw; }

// List entries: bits 0..29 = record index, bit 31 = "holds a byte outside ACGT" (set by whichever stage found out, so
// that the stages behind do not build the 2-bit strand of that record again just to stumble over the same byte), bit 30 (with
// bit 31) = "the N-mask rule, prefix rule included, has had this record and refused it" -- the lean routine's N variant in the
// kernel in front, canon_record_mode2n / the N-mask team in an earlier stage: the stages behind go straight to the 4-bit mode
// (late round 4; before, stage A ran the N-mask mode again on every leftover of the N builds: build, two scans, the same refusal).
constexpr uint32_t ENTRY_NOT_ACGT = 0x80000000u, ENTRY_NO_2N = 0x40000000u, ENTRY_REC = 0x3FFFFFFFu;

// Processes one record; returns false if it does not fit this tier's LDS slice (not_acgt then says what was learnt).
// no_2n: in -- skip the N-mask mode (it has refused this record before); out -- it has now
CK_DEV bool canon_record(const CanonArgs& a, uint64_t rec, uint32_t* lds, const uint32_t* lut, const uint32_t* lutn, bool& not_acgt, bool& no_2n)
{
    const uint64_t off = a.offsets[rec], len = a.offsets[rec + 1] - off;
    if (len >> 31) return false;                 // 32-bit cyclic positions (p < 2n): a record of 2 Gi symbols or more fits nowhere
    const uint32_t n = (uint32_t)len;
    const uint8_t* src = a.bytes + off;
    if (n == 0) {
        if (lane_id() == 0) {
            if (a.out_index) a.out_index[rec] = 0;
            if (a.out_strand) a.out_strand[rec] = 1;
        }
        return true;
    }
    if (n >= 48) {
        if (!not_acgt) {
            if (need_dw_strand2(n) > a.slice_dw) return false;
            // a look at the first KiB before the strand is built: with N sprinkled in at 1 %, that is where a long
            // record shows it (its bytes are in the cache for the builder afterwards)
            uint32_t miss = 0;
            if (16 * lane_id() + 16 <= n) (void)fast_pack(load16(src + 16 * lane_id()), miss);
            not_acgt = ballot(miss != 0) != 0;
            if (!not_acgt) {
                const int r2 = canon_record_mode<2>(a, rec, src, off, n, lds, lut);
                if (r2 == 0) return true;
                if (r2 == 2) return false;          // pure ACGT, tied minimal key, no room for the bitmask: next tier
                not_acgt = true;
            }
        }
        // a few N: 2-bit words + an N bitmask (0 = done; else the 4-bit mode decides, here or in a bigger stage)
        if (lutn && !no_2n && need_dw_2n(n) <= a.slice_dw) {
            if (canon_record_mode2n(a, rec, src, off, n, lds, lut, lutn) == 0) return true;
            no_2n = true;
            wave_sync();            // every lane has read the strands before the next mode overwrites them
        }
        if (need_dw<4>(n) > a.slice_dw) return false;
        if (canon_record_mode<4>(a, rec, src, off, n, lds, lut) == 0) return true;
    }
    if (need_dw<8>(n) > a.slice_dw) return false;
    canon_record_mode<8>(a, rec, src, off, n, lds, lut);
    return true;
}

CK_DEV bool canon_record(const CanonArgs& a, uint64_t rec, uint32_t* lds, const uint32_t* lut, const uint32_t* lutn, bool& not_acgt)
{
    bool no_2n = false;
    return canon_record(a, rec, lds, lut, lutn, not_acgt, no_2n);
}

// append a record this launch cannot take to the workgroup's output segment (blk_count lives in LDS)
CK_DEV void defer_record(const CanonArgs& a, uint32_t* blk_count, uint32_t block, uint32_t rec, bool not_acgt = false, bool no_2n = false)
{
    if (lane_id() == 0) {
        if (a.defer_list) a.defer_list[(uint64_t)block * a.out_seg_cap + lds_atomic_inc(blk_count)] = rec | (not_acgt ? ENTRY_NOT_ACGT | (no_2n ? ENTRY_NO_2N : 0u) : 0u);
        else {
            // nothing can take this record (CIRCKIT_ERR_TOO_LONG through circkit_ctx_batch_status): a hash-only batch must not
            // find a stale view of an earlier batch in its place (out-of-bounds reads in the xxh3 pass)
            atomic_add_u32(a.status, 1u);
            if (a.out_view) a.out_view[rec] = 0;
        }
    }
}

// loop of one wave (wave `wib` of `wpb` in workgroup `block` of `nblocks`) over its share of the work.
// (Loading the next record's list entry and offsets one record ahead was tried and measured: no gain.)
// (Handing a segment's entries out dynamically -- an LDS counter per workgroup instead of the fixed stride, so that no wave
// waits for the one that drew the long records -- was tried on BASELINE config 4: 2.29 -> 2.41 ms.)
CK_DEV void canon_wave_loop(const CanonArgs& a, uint32_t* lds, const uint32_t* lut, uint32_t* blk_count, uint32_t block,
                            uint32_t nblocks, uint32_t wib, uint32_t wpb, const uint32_t* lutn = nullptr)
{
    if (!a.list) {
        for (uint64_t rec = (uint64_t)block * wpb + wib; rec < a.n_records; rec += (uint64_t)nblocks * wpb) {
            bool not_acgt = false, no_2n = false;
            if (!canon_record(a, rec, lds, lut, lutn, not_acgt, no_2n)) defer_record(a, blk_count, block, (uint32_t)rec, not_acgt, no_2n);
            wave_sync();
        }
        return;
    }
    for (uint32_t s = block * a.segs_per_block; s < (block + 1) * a.segs_per_block && s < a.in_nseg; ++s) {
        const uint32_t count = a.list_count[s];
        const uint32_t* seg = a.list + (uint64_t)s * a.in_seg_cap;
        for (uint32_t i = wib; i < count; i += wpb) {
            const uint32_t rec = seg[i] & ENTRY_REC;
            bool not_acgt = (seg[i] & ENTRY_NOT_ACGT) != 0, no_2n = (seg[i] & ENTRY_NO_2N) != 0;
            if (!canon_record(a, rec, lds, lut, lutn, not_acgt, no_2n)) defer_record(a, blk_count, block, rec, not_acgt, no_2n);
            wave_sync();
        }
    }
}

// Team pass of a workgroup over the deferral segment it has just written (every wave, behind a workgroup barrier):
// entries the team can take (team_takes, pure ACGT as far as anybody knows) are canonicalized by all waves together and
// leave the segment, the others move up.  blk_count[0] = entries, blk_count[1..3] = the team's three words.
CK_DEV void team_pass(const CanonArgs& a, uint32_t* lds, const uint32_t* lut, const uint32_t* lutn, uint32_t* blk_count, uint32_t block, uint32_t wib, uint32_t wpb,
                      bool allow_solo = true, bool team4 = true)
{
    const uint32_t cnt = *blk_count;
    if (cnt == 0 || !a.defer_list) return;
    uint32_t* seg = a.defer_list + (uint64_t)block * a.out_seg_cap;
    uint32_t kept = 0;
    for (uint32_t k = 0; k < cnt; ++k) {
        uint32_t entry = seg[k];                                     // written by this workgroup before the barrier
        const uint32_t entry0 = entry, rec = entry & ENTRY_REC;
        bool done = false;
        const uint64_t len = a.offsets[rec + 1] - a.offsets[rec];
        if (len < (1ull << 31)) {
            // an entry without the alphabet flag is pure ACGT as far as anybody knows (if not, the 2-bit team finds out and
            // the N-mask team has the next look); a flagged one holds an N, a gap or worse
            int why = (entry & ENTRY_NOT_ACGT) ? 1 : 3;            // 3: not tried
            if (why == 3 && team_takes((uint32_t)len, a.slice_dw, wpb)) { why = canon_record_team2(a, rec, lds, lut, blk_count + 1, wib, wpb); done = why == 0; }
            if (why == 1 && lutn && !(entry & ENTRY_NO_2N) && team_takes_2n((uint32_t)len, a.slice_dw, wpb)) {
                done = canon_record_team2n(a, rec, lds, lut, lutn, blk_count + 1, wib, wpb);
                entry |= ENTRY_NOT_ACGT | ENTRY_NO_2N;             // (if it stays on the list: the N-mask rule has had it)
            }
            // what the N-mask team refuses (an N in the winning window behind a shared prefix), gaps: the 4-bit team
            if (team4 && !done && why == 1 && lutn && team_takes_4((uint32_t)len, a.slice_dw, wpb)) done = canon_record_team4(a, rec, lds, blk_count + 1, wib, wpb) == 0;
        }
        if (!done && lutn && allow_solo) {
            // not the team's (4-bit and byte modes, periods): wave 0 alone with the workgroup's whole LDS, the general
            // routine -- what a one-wave tier with a slice of that size would do, without a launch of its own
            CanonArgs solo = a;
            solo.slice_dw = wpb * a.slice_dw;
            if (wib == 0) {
                bool na = (entry & ENTRY_NOT_ACGT) != 0, n2 = (entry & ENTRY_NO_2N) != 0;
                const bool ok = canon_record(solo, rec, lds, lut, lutn, na, n2);
                if (n2) entry |= ENTRY_NOT_ACGT | ENTRY_NO_2N;     // (wave 0 writes the list)
                if (lane_id() == 0) blk_count[1] = ok ? 1u : 0u;
            }
            block_barrier();
            done = blk_count[1] != 0;
            block_barrier();
        }
        if (!done) {
            if (wib == 0 && lane_id() == 0 && (kept != k || entry != entry0)) seg[kept] = entry;
            ++kept;
        }
    }
    block_barrier();
    if (wib == 0 && lane_id() == 0) *blk_count = kept;
    block_barrier();
}

}  // namespace ck
