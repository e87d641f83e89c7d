// canon_pair.h -- the per-record routine of the streaming builds with the fused XXH3 (`uniq`), two records per wave.
//
// Why: those builds are bound by instruction issue, scalar and vector alike (Measured: ~161 vector + ~173 scalar
// instructions per record with one record per wave, the CU's one scalar unit ~70 % busy).  Nearly all of the scalar side is per
// WAVE and per iteration -- ring and pointer bookkeeping, offset loads, DMA issue, waits, barrier, dispatch -- so a wave that
// takes two records per iteration halves it, and the vector work that does not depend on the record's bytes (reductions, the
// XXH3 row sums, address arithmetic) is shared as well.  Layout: record A in lanes 0..31, record B in lanes 32..63; lane u of a
// half packs the aligned chunks 2u and 2u+1 of its record (32 bytes), i.e. holds words 2u and 2u+1 of the strand -- the word
// behind a lane's first word is its own second word, the one behind that its neighbour's first (one DPP move).  Everything that
// canon_fast.h does with ds_bpermute / v_readlane on one word per lane (periodic extension, reverse strand, rotation fetch) goes
// through a copy of the strand in LDS here -- the record's XXH3 slot, which the workgroup's merger needs filled with the winning
// strand anyway (canon_fast.h, group_hash_*): a window at any symbol position is two consecutive dwords and one 64-bit shift.
// Same reference functions (lib/src/canonicalize.rs:5-63, src/uniq.rs:45), same exits: a record whose minimal key is not owned
// by exactly one position, or whose strands tie, goes to the deferral list.
#pragma once
#include "canon_fast.h"
#ifndef CK_PAIR_SCALAR_VALID
#define CK_PAIR_SCALAR_VALID 0
#endif

namespace ck {

// The 16 symbols at symbol position p (0 <= p, not wrapped) of a strand copy S: with q = p + 15 the words S[(q >> 4) - 1] and
// S[q >> 4] shifted right by 30 - 2 * (q & 15) bits -- a shift of 0..30, so ONE v_alignbit_b32 (p on a word boundary: shift 0 of
// (S[w - 1], S[w]) = S[w]; the natural form, a shift of 32 - 2 * (p & 15) = 2..32 of (S[w], S[w + 1]), needs the half-rate 64-bit
// shift).  30 - x == 30 ^ x for the even x <= 30, and the instruction reads five bits of the shift: (2q) ^ 30 is enough.
// S must be readable one word in front.  pair_window_code(p): word index + 1 and shift of a position known in advance.
CK_DEV uint32_t pair_window(const uint32_t* S, uint32_t p)
{
    const uint32_t q = p + 15;
    const uint32_t* w = S - 1 + (q >> 4);
    return alignbit(w[0], w[1], (q << 1) ^ 30u);
}
CK_DEV uint32_t pair_window_code(uint32_t p) { const uint32_t q = p + 15; return ((q >> 4) << 6) | (((q << 1) ^ 30u) & 31u); }
CK_DEV uint32_t pair_window_at(const uint32_t* S, uint32_t code) { const uint32_t* w = S - 1 + (code >> 6); return alignbit(w[0], w[1], code); }

// lane constants that depend on the two record lengths only (recomputed when a length changes: once per wave on a batch of
// equal-length records)
struct PairShape {
    uint32_t nA = 0, nB = 0;
    uint32_t n;             // this lane's record's length
    int32_t e0;             // periodic extension: first of the three words of the strand copy this lane reads (>= -3)
    uint32_t sft;           // ... and the shift of the 64-bit window (2r)
    uint32_t m0, m1;        // bits of words 2u / 2u+1 that stay: all below the record's last word, its r tail symbols in it, none behind
    uint32_t rc0, rc1;      // reverse strand, words 2u / 2u+1: ((source word + 1) << 6) | window shift (pair_window)
    // wave masks (scalars): lanes whose output cell k lies inside the record; lanes whose cell k belongs to XXH3's full stripes
    uint64_t in_rec[2], on[2];
};
CK_DEV void pair_shape(PairShape& sh, uint32_t nA, uint32_t nB)
{
    const uint32_t t = lane_id(), u = t & 31;
    const uint32_t n = t < 32 ? nA : nB;
    const uint32_t r = n & 15, nwf = n >> 4, nwv = nwf + (r ? 1u : 0u);
    sh.nA = nA; sh.nB = nB; sh.n = n;
    const uint32_t q0 = 2 * u, q1 = q0 + 1;
    // word q >= nwf of the extended strand = bfi(m, F[q], window(S[q - nwv], S[q - nwv + 1]) >> 2r): the r tail symbols of word
    // nwf stay and the record's head follows (q - nwv == -1: only the low word of the window counts), the words behind are the
    // head shifted by r symbols -- ONE formula, one shift for both (canon_fast.h computes `fix` and `ext` apart)
    // (windows are v_alignbit_b32: shifts of 0..31 -- r == 0, where the window IS its first word, reads one word earlier and
    // shifts by 0, which yields the second)
    const int32_t e = (int32_t)q0 - (int32_t)nwv - (r ? 0 : 1);
    sh.e0 = e < -3 ? -3 : e;
    sh.sft = 2 * r;
    const uint32_t keep = ~(0xFFFFFFFFu >> (2 * r));                  // (r == 0: nwf == nwv, never selected)
    sh.m0 = q0 < nwf ? ~0u : (q0 < nwv ? keep : 0u);
    sh.m1 = q1 < nwf ? ~0u : (q1 < nwv ? keep : 0u);
    // rc word q = comp(reverse(forward symbols [n - 16(q+1), n - 16q) mod n)); lanes far behind the record: any word in range
    uint32_t rc[2];
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {
        const int32_t p0 = (int32_t)n - 16 * (int32_t)(q0 + k + 1);
        int32_t p = p0 + ((p0 >> 31) & (int32_t)n);
        p = p < 0 ? 0 : p;
        rc[k] = pair_window_code((uint32_t)p);
    }
    sh.rc0 = rc[0]; sh.rc1 = rc[1];
    // output: cell k of lane u = bytes [16u + 512k, + 16) of the canonical record -- one store instruction covers 512 contiguous
    // bytes per record -- = XXH3 cell (stripe (u >> 2) + 8k, accumulator pair u & 3)
    const uint32_t stripes = (n - 1) >> 6;
    sh.in_rec[0] = ballot(16 * u < n); sh.in_rec[1] = ballot(16 * u + 512 < n);
    sh.on[0] = ballot(u < 4 * stripes); sh.on[1] = ballot(u + 32 < 4 * stripes);
}

CK_DEV uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// which of the 32 positions of lane lv's two words (lv: a lane of the reader's own half) own the key Mv: bit t of the result =
// position 32 * (lv & 31) + (t & 31) of the reader's record does
template <bool K16>
CK_DEV uint64_t pair_positions(uint32_t E0, uint32_t E1, uint32_t En, uint32_t lv, uint32_t Mv, uint32_t n)
{
    const uint32_t u = lane_id() & 31;
    const uint32_t c0 = shfl(E0, lv), c1 = shfl(E1, lv), c2 = shfl(En, lv);
    const uint32_t x = u < 16 ? c0 : c1, y = u < 16 ? c1 : c2;
    uint32_t key = funnel(x, y, 2 * (u & 15));
    if (K16) key >>= 16;
    const uint32_t pos = 32 * (lv & 31) + u;
    return ballot(key == Mv && pos < n);
}
// Position of the minimal key of each half's strand (E0 / E1 = the lane's words, En = the word behind them, mE = the lane's
// minimum, Mv = its half's minimum) and whether exactly one position of the record owns it.  The first owner lane of each
// half is examined straight-line; a second one (the duplicate of the record's first positions behind its end, or a real tie)
// in a rare extra round for both halves; a third: not unique.
struct PairPos { uint32_t posA, posB; bool uA, uB; };
template <bool K16>
CK_DEV PairPos pair_locate(uint32_t E0, uint32_t E1, uint32_t En, uint32_t mE, uint32_t Mv, uint32_t n, uint32_t hb)
{
    const uint64_t own = ballot(mE == Mv);
    uint32_t ownA = (uint32_t)own, ownB = (uint32_t)(own >> 32);              // both != 0: some lane holds its half's minimum
    const uint32_t lA = (uint32_t)ffs32(ownA), lB = (uint32_t)ffs32(ownB);
    const uint64_t pm = pair_positions<K16>(E0, E1, En, lA + (hb ? 32 + lB - lA : 0u), Mv, n);
    const uint32_t pmA = (uint32_t)pm, pmB = (uint32_t)(pm >> 32);
    uint32_t cntA = (uint32_t)popc32(pmA), cntB = (uint32_t)popc32(pmB);
    PairPos r;
    r.posA = 32 * lA + (uint32_t)ffs32_or_neg(pmA);                            // (empty mask: garbage, and the count says so)
    r.posB = 32 * lB + (uint32_t)ffs32_or_neg(pmB);
    ownA &= ownA - 1; ownB &= ownB - 1;
    if ((ownA | ownB) != 0) {                                                  // rare
        const uint32_t l2A = ownA ? (uint32_t)ffs32(ownA) : lA, l2B = ownB ? (uint32_t)ffs32(ownB) : lB;
        const uint64_t pm2 = pair_positions<K16>(E0, E1, En, l2A + (hb ? 32 + l2B - l2A : 0u), Mv, n);
        const uint32_t pm2A = (uint32_t)pm2, pm2B = (uint32_t)(pm2 >> 32);
        if (ownA) {
            cntA += (uint32_t)popc32(pm2A) + ((ownA & (ownA - 1)) ? 2u : 0u);  // a third owner lane: give up
            if (pmA == 0) r.posA = 32 * l2A + (uint32_t)ffs32_or_neg(pm2A);
        }
        if (ownB) {
            cntB += (uint32_t)popc32(pm2B) + ((ownB & (ownB - 1)) ? 2u : 0u);
            if (pmB == 0) r.posB = 32 * l2B + (uint32_t)ffs32_or_neg(pm2B);
        }
    }
    r.uA = cntA == 1; r.uB = cntB == 1;
    return r;
}
// the two 16-bit halves of a scan's running minimum (word_min_key16) for words 2u and 2u+1, merged before the last step
CK_DEV uint32_t word_pkmin16(uint32_t cur, uint32_t nxt)
{
    uint32_t m = cur;
#pragma unroll
    for (int b = 1; b < 8; ++b) m = pk_min_u16(m, funnel(cur, nxt, 2 * b));
    return m;
}
CK_DEV uint32_t pair_min16(uint32_t a, uint32_t b, uint32_t c, const PairShape& sh)     // words (a, b), (b, c) -> the lane's minimal 8-symbol prefix
{
    // (a word behind the record's last one -- exactly those of which nothing stays, m == 0 -- must lose every minimum)
    const uint32_t m = pk_min_u16(sh.m0 != 0 ? word_pkmin16(a, b) : ~0u, sh.m1 != 0 ? word_pkmin16(b, c) : ~0u);
    return umin32(m >> 16, m & 0xFFFFu);
}
CK_DEV uint32_t pair_min32(uint32_t a, uint32_t b, uint32_t c, const PairShape& sh)
{
    return umin32(sh.m0 != 0 ? word_min_key<2>(a, b) : ~0u, sh.m1 != 0 ? word_min_key<2>(b, c) : ~0u);
}

// Canonicalizes records rec0 (lanes 0..31) and rec0 + 1 (lanes 32..63) of a staged group: img = the group's LDS image, base_lo =
// the low 32 bits of its first byte's offset, offA / offB = the records' byte offsets, nA / nB their lengths -- for a half whose
// bit in `elig` is clear (a length outside 48..1008) the caller passes a harmless dummy length, and the half only goes through
// the motions.  slot0 = record A's XXH3 slot, record B's follows (canon_fast.h: [0,64) row sums, [64,128) the winning strand,
// [128..130] n, rotation, valid); every call leaves both slots' valid words set or cleared.  Returns bit h = record h is done;
// the others are the caller's to defer.
// HASHING = false (the bytes-only pair build): no XXH3, and `slot0` is the wave's scratch for the two strand copies, PAIR_SCRATCH_DW
// dwords per record (four in front of the copy: its words -3 .. -1 are read, never used).
constexpr uint32_t PAIR_SCRATCH_DW = 72;
template <bool HASHING = true>
CK_DEV uint32_t pair_canon(const CanonArgs& a, const uint32_t* lut, const uint32_t* sec, PairShape& sh, const uint32_t* img, uint32_t base_lo,
                           uint32_t rec0, uint64_t offA, uint64_t offB, uint32_t nA, uint32_t nB, uint32_t elig, uint32_t* slot0)
{
    const uint32_t t = lane_id(), u = t & 31;
    const uint32_t hb = t >> 5;                                       // 0 / 1: the lane's half; `hb ? x : 0` with x uniform is one instruction
    if (sh.nA != nA || sh.nB != nB) pair_shape(sh, nA, nB);
    const uint32_t n = sh.n;
    const uint32_t dAB = (uint32_t)(offB - offA);                     // record B starts this far behind record A (the true length of A)
    const uint32_t voff = hb ? dAB : 0u;
    const uint32_t relA = (uint32_t)offA - base_lo, rel = relA + voff, a16 = rel & 15;
    const uint32_t* cp = img + 4 * ((rel >> 4) + 2 * u);
    uint32_t ms0, ms1;
    const uint32_t P0 = fast_pack(lds_load16(cp), ms0), P1 = fast_pack(lds_load16(cp + 4), ms1);
#if CK_PAIR_SCALAR_VALID
    // Which lanes' chunks belong to their record -- chunk 2u / 2u+1 of nch = (a16 + n + 15) / 16 <= 64 --: wave masks from the
    // two records' scalars.  (EXPERIMENT, off: eight more scalar registers live across the loop, which the compiler pays for in
    // v_writelane / v_readlane spill traffic -- vector instructions: hash only 3.39 -> 3.53 ms.)
    const uint32_t nchA = ((relA & 15) + nA + 15) >> 4, nchB = (((relA + dAB) & 15) + nB + 15) >> 4;
    const auto low32 = [](uint32_t k) -> uint64_t { return k >= 32 ? 0xFFFFFFFFull : (1ull << k) - 1; };
    const uint64_t in0 = low32((nchA + 1) >> 1) | (low32((nchB + 1) >> 1) << 32), in1 = low32(nchA >> 1) | (low32(nchB >> 1) << 32);
    const uint64_t badm = (ballot(ms0 != 0) & in0) | (ballot(ms1 != 0) & in1);
#else
    // (the first and last chunk also hold the neighbours' bytes: an invalid byte there defers the record for nothing -- harmless)
    const uint32_t nch = (a16 + n + 15) >> 4;                         // chunks the record touches: <= 64
    const uint64_t badm = ballot(((ms0 != 0) & (2 * u < nch)) | ((ms1 != 0) & (2 * u + 1 < nch)));
#endif
    uint32_t ok = elig & (((uint32_t)badm == 0 ? 1u : 0u) | ((uint32_t)(badm >> 32) == 0 ? 2u : 0u));
    // the record starts a16 bytes into its first chunk: removed on the packed words
    const uint32_t sh2 = 32 - 2 * a16;
    uint32_t W0 = lshr64(P0, P1, sh2), W1 = lshr64(P1, wave_shl1(P0), sh2);
    uint32_t* slot = slot0 + (hb ? (HASHING ? GH_STRIDE_DW : PAIR_SCRATCH_DW) : 0u);
    uint32_t* S = slot + (HASHING ? 64 : 4);
    S[2 * u] = W0; S[2 * u + 1] = W1;
    wave_sync();
    {   // periodic extension (pair_shape)
        const uint32_t* Se = S + sh.e0;
        const uint32_t x0 = Se[0], x1 = Se[1], x2 = Se[2];
        W0 = bfi_v(sh.m0, W0, alignbit(x0, x1, sh.sft));
        W1 = bfi_v(sh.m1, W1, alignbit(x1, x2, sh.sft));
    }
    wave_sync();
    S[2 * u] = W0; S[2 * u + 1] = W1;
    wave_sync();
    // reverse-complement strand from the extended forward words
    const uint32_t C0 = rc_word<2>(pair_window_at(S, sh.rc0)), C1 = rc_word<2>(pair_window_at(S, sh.rc1));
    const uint32_t W0n = wave_shl1(W0), C0n = wave_shl1(C0);          // word 2u + 2 (lane 31's: never a word of its record)
    // Level 1: 8-symbol prefixes (canon_fast.h word_min_key16); they decide the strand and the rotation whenever the strands'
    // minimal prefixes differ and the winner's is owned by one position (lib/src/canonicalize.rs:58-62: forward only if
    // strictly smaller).  The halves' minima come back as lane vectors: strand choice and tie are vector compares.
    uint32_t mF = pair_min16(W0, W1, W0n, sh), mC = pair_min16(C0, C1, C0n, sh), MF, MC;
    half_min2_bcast_u32(mF, mC, MF, MC);
    bool fwd = MF < MC;
    uint64_t tie = ballot(MF == MC);
    PairPos loc = pair_locate<true>(fwd ? W0 : C0, fwd ? W1 : C1, fwd ? W0n : C0n, fwd ? mF : mC, umin32(MF, MC), n, hb);
    if (((ok & 1) && ((uint32_t)tie != 0 || !loc.uA)) || ((ok & 2) && ((uint32_t)(tie >> 32) != 0 || !loc.uB))) {
        // Level 2 (rare): full 16-symbol keys, for both halves (its answer holds wherever level 1's did).  Equal keys on the two
        // strands, or a minimal key that is still not unique: the general kernel's.
        mF = pair_min32(W0, W1, W0n, sh); mC = pair_min32(C0, C1, C0n, sh);
        half_min2_bcast_u32(mF, mC, MF, MC);
        fwd = MF <= MC;
        tie = ballot(MF == MC);
        loc = pair_locate<false>(fwd ? W0 : C0, fwd ? W1 : C1, fwd ? W0n : C0n, fwd ? mF : mC, umin32(MF, MC), n, hb);
    }
    if ((uint32_t)tie != 0 || !loc.uA) ok &= ~1u;
    if ((uint32_t)(tie >> 32) != 0 || !loc.uB) ok &= ~2u;
    const uint32_t iA = (ok & 1) ? loc.posA : 0u, iB = (ok & 2) ? loc.posB : 0u;                 // (a failed half: in range, unused)
    const uint32_t idx = iA + (hb ? iB - iA : 0u);
    const uint64_t okm = ((ok & 1) ? 0xFFFFFFFFull : 0ull) | ((ok & 2) ? 0xFFFFFFFF00000000ull : 0ull);
    const bool hashing = HASHING && a.out_hash != nullptr;
    // the winning strand into the slot: the rotation fetch below and the merger's last stripe read it
    wave_sync();
    S[2 * u] = fwd ? W0 : C0; S[2 * u + 1] = fwd ? W1 : C1;
    if (HASHING && u == 0) { slot[128] = n; slot[129] = idx; slot[130] = lane_pred(okm) && hashing && n > 240 ? 1u : 0u; }   // (XXH3's long-input path; shorter: the xxh3 pass)
    wave_sync();
    uint64_t c0 = 0, c1 = 0;
    // secret words of the lane's two XXH3 cells (stripe (u >> 2) + 8k, pair u & 3): 16 bytes at offset 8 * stripe + 16 * pair of the
    // secret, read from its copy in LDS (sec; as lane constants they cost the build four registers it does not have)
    const uint64_t* kp = reinterpret_cast<const uint64_t*>(sec) + ((u >> 2) + 2 * (u & 3));
    uint8_t* const outA = a.out_bytes ? a.out_bytes + offA : nullptr;            // (scalar base + 32-bit lane offset)
    const uint32_t n16 = n - 16;
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {
        const uint32_t o = umin32(16 * u + 512 * k, n16);             // (the record's last cell is pulled back to end at n: full 16-byte stores only)
        uint32_t p = idx + o;
        p = umin32(p, p - n);                                         // p < 2n: p - n wraps to huge unless p >= n
        const u32x4 cell = fast_decode(lut, pair_window(S, p));
        if (outA != nullptr && lane_pred(okm & sh.in_rec[k])) store16(outA + (voff + o), cell);
        if (hashing && lane_pred(sh.on[k])) {
            // acc0 += d1 + lo(d0 ^ k0) * hi(d0 ^ k0), acc1 += d0 + lo(d1 ^ k1) * hi(d1 ^ k1) (XXH3's accumulate; the products go in
            // through the multiply-add's own addend)
            const uint64_t d0 = ((uint64_t)cell.y << 32) | cell.x, d1 = ((uint64_t)cell.w << 32) | cell.z;
            const uint64_t x0 = d0 ^ kp[8 * k], x1 = d1 ^ kp[8 * k + 1];
            c0 = (uint64_t)(uint32_t)x0 * (x0 >> 32) + (c0 + d1);
            c1 = (uint64_t)(uint32_t)x1 * (x1 >> 32) + (c1 + d0);
        }
    }
    if (hashing) {
        // sums over the stripes: the lane's two cells above, then lanes u, u+4, u+8, u+12 of a 16-lane row (DPP); the record's two
        // rows are added by the merger
        dpp_rowsum4_u64x2(c0, c1);
        if ((u & 15) >= 12) lds_store16(slot + (u >> 4) * 16 + (u & 3) * 4, u32x4{ (uint32_t)c0, (uint32_t)(c0 >> 32), (uint32_t)c1, (uint32_t)(c1 >> 32) });
        if (a.out_view && lane_pred(okm) && n <= 240 && u == 0) a.out_view[rec0 + hb] = fwd ? idx : (idx | 0x80000000u);   // hash not fused: the xxh3 pass reads the view
    }
    return ok;
}

}  // namespace ck
