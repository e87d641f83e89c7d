// canon_fast.h -- the streaming kernel of the canonicalize path.
//
// Handles the records that make up BASELINE's headline workload -- pure ACGT, 48..1008 bases -- with one
// packed word per lane held in REGISTERS (no LDS memory; cross-lane access by ds_bpermute / v_readlane,
// reductions by DPP), straight-line code, and a software pipeline that has the next record's bytes and
// the record after that's offsets in flight while the current one is computed.  Anything else (other
// alphabets, longer or shorter records, a repeated minimal key) is appended to a list for the general
// LDS kernel of canon_core.h.  Same reference functions as there (lib/src/canonicalize.rs:5-63).
#pragma once
#include "canon_core.h"

namespace ck {

constexpr uint32_t FAST_MIN_N = 48, FAST_MAX_N = 1008;

// the 16 symbols at cyclic symbol position p (lane-varying, p < 2n) of a one-word-per-lane strand
CK_DEV uint32_t reg_sym_word(uint32_t E, uint32_t p, uint32_t n)
{
    p = p >= n ? p - n : p;
    const uint32_t wi = p >> 4, sh = (p & 15) * 2;
    return lshr64(shfl(E, wi), shfl(E, wi + 1), 32 - sh);
}

// position of the minimal key M if exactly one valid position owns it.  shv = 32 - 2*(lane & 15).
CK_DEV bool fast2_locate(uint32_t E, uint32_t En, uint32_t m, uint32_t M, uint32_t n, uint32_t shv, uint32_t& pos)
{
    const uint32_t t = lane_id();
    uint64_t hm = ballot(m == M);
    if (popc64(hm) > 2) return false;
    uint32_t cnt = 0;
    while (hm) {        // 1 iteration; 2 when the minimum sits in the first 16 - n%16 positions (duplicate behind the end)
        const uint32_t l = (uint32_t)ffs64(hm);
        hm &= hm - 1;
        const uint32_t k = lshr64(readlane(E, l), readlane(En, l), shv);
        const uint32_t left = n - 16 * l;                                   // valid positions in word l
        const uint64_t pm = ballot(k == M && t < (left < 16 ? left : 16));
        cnt += (uint32_t)popc64(pm);
        if (pm) pos = 16 * l + (uint32_t)ffs64(pm);
    }
    return cnt == 1;
}

// Issue the LDS-DMA prefetch of a record: lane t's 16 input bytes land at buf + 16t.  Eligible record:
// lanes >= n/16 fetch the record's last 16 bytes (in bounds, n >= 48); their n%16 tail symbols are shifted
// up later.  Other records: a dummy fetch of the offsets array (always >= 16 readable bytes), so that every
// call issues exactly one vector-memory instruction -- the pipeline's vmcnt bookkeeping depends on it.
CK_DEV bool fast_issue(const CanonArgs& a, uint64_t off, uint64_t end, uint32_t* buf)
{
    const bool ok = end - off >= FAST_MIN_N && end - off <= FAST_MAX_N;
    const uint32_t t = lane_id(), n = (uint32_t)(end - off);
    const uint8_t* src = ok ? a.bytes + off + (t >= (n >> 4) ? n - 16 : 16 * t) : (const uint8_t*)a.offsets;
    glds16_async(buf, src);
    return ok;
}
CK_DEV u32x4 fast_fetch(const uint32_t* buf)
{
    const uint32_t* p = buf + 4 * lane_id();
    return u32x4{ p[0], p[1], p[2], p[3] };
}

// 16 ASCII bytes -> 16 two-bit codes (first byte in the top bits); `bad` = wave mask of lanes holding a byte
// that is not A/C/G/T.  Validity = the check LUT reproduces the dword; four ballots keep it at one v_cmp_ne
// per dword with the ORs on the scalar unit.
CK_DEV uint32_t fast_pack(u32x4 v, uint64_t& bad)
{
    const uint32_t d[4] = { v.x, v.y, v.z, v.w };
    uint32_t u[4];
    bad = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t sel = (d[k] >> 1) & HASH_MASK;
        bad |= ballot(perm(0u, CHK2_LO, sel) != d[k]);
        const uint32_t code = perm(0u, 0x02030100u, sel);
        const uint32_t t = code | (code << 10);
        u[k] = t | (t << 20);
    }
    return perm(u[0], u[1], 0x07030c0cu) | perm(u[2], u[3], 0x0c0c0703u);
}

// 256-entry LDS table: packed byte (4 symbols, first in the top bits) -> its 4 ASCII bytes.  Replaces ~6 VALU
// per output dword (spread the 2-bit fields into bytes, v_perm) by shift + mask + one ds_read_b32.
CK_DEV void fast_lut_init(uint32_t* lut, uint32_t tid, uint32_t nthreads)
{
    for (uint32_t x = tid; x < 256; x += nthreads) {
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) o |= ((0x54474341u >> (8 * ((x >> (6 - 2 * k)) & 3))) & 0xFFu) << (8 * k);
        lut[x] = o;
    }
}
CK_DEV u32x4 fast_decode(const uint32_t* lut, uint32_t w)
{
    return u32x4{ lut[w >> 24], lut[(w >> 16) & 0xFF], lut[(w >> 8) & 0xFF], lut[w & 0xFF] };
}

// returns false when the record must go to the general kernel
CK_DEV bool fast_process(const CanonArgs& a, const uint32_t* lut, uint64_t rec, uint64_t off, uint32_t n, u32x4 bytes)
{
    const uint32_t t = lane_id();
    const uint32_t nwf = n >> 4, r = n & 15, nwv = nwf + (r ? 1u : 0u);
    uint64_t bad;
    uint32_t F = fast_pack(bytes, bad);
    if (bad) return false;
    F <<= t >= nwf ? ((16 - r) & 15) * 2 : 0;
    // periodic extension (lanes >= nwf): E[nwf] = r tail symbols ++ head, E[nwv + e] = head shifted by r
    {
        const uint32_t A = shfl(F, t - nwv), B = shfl(F, t - nwv + 1);
        const uint32_t ext = lshr64(A, B, 32 - ((16 - r) & 15) * 2);
        const uint32_t fix = bfi(~(0xFFFFFFFFu >> (2 * r)), F, B >> (2 * r));
        F = t >= nwv ? ext : (t == nwf ? fix : F);        // r == 0: nwf == nwv, `fix` is never selected
    }
    const bool fwd_only = (a.flags & CK_FLAG_FWD_ONLY) != 0;
    // reverse-complement strand from the extended forward words:
    // rc word t = comp(reverse(forward symbols [n - 16(t+1), n - 16t) mod n))
    uint32_t C;
    {
        const int32_t p0 = (int32_t)n - 16 * (int32_t)(t + 1);
        const uint32_t p = (uint32_t)(p0 < 0 ? p0 + (int32_t)n : p0);
        const uint32_t g = ~lshr64(shfl(F, p >> 4), shfl(F, (p >> 4) + 1), 32 - (p & 15) * 2);
        const uint32_t v = bitrev(g);                       // reverses bits; swap the two bits of every symbol back
        C = bfi(0x55555555u, v >> 1, v << 1);
    }
    const uint32_t Fn = shfl(F, t + 1), Cn = shfl(C, t + 1);
    uint32_t mF = word_min_key<2>(F, Fn), mC = word_min_key<2>(C, Cn);
    mF = t < nwv ? mF : ~0u;
    mC = t < nwv ? mC : ~0u;
    uint32_t MF, MC;
    wave_min2_u32(mF, mC, MF, MC);
    const uint32_t shv = 32 - 2 * (t & 15);
    uint32_t iF = 0, iC = 0;
    if (!fast2_locate(F, Fn, mF, MF, n, shv, iF)) return false;
    if (!fwd_only && !fast2_locate(C, Cn, mC, MC, n, shv, iC)) return false;
    // lexicographic select (lib/src/canonicalize.rs:58-62).  The minimal keys ARE the first 16 symbols of
    // the two minimal rotations, so they decide unless equal; only then compare the full rotations.
    bool fwd = fwd_only || MF < MC;
    if (!fwd_only && MF == MC) {
        const uint32_t wa = reg_sym_word(F, iF + 16 * t, n), wb = reg_sym_word(C, iC + 16 * t, n);
        const uint32_t d = t < nwv ? (wa ^ wb) : 0u;
        const uint64_t bal = ballot(d != 0);
        if (bal) {
            const uint32_t l = (uint32_t)ffs64(bal);
            const uint32_t k = 16 * l + (uint32_t)clz32(readlane(d, l)) / 2;
            fwd = k < n && readlane(wa, l) < readlane(wb, l);
        }
    }
    if (a.out_bytes) {
        // every lane stores a full 16 bytes: the last lane's window is pulled back to end exactly at n, so it
        // overlaps its neighbour's with identical bytes -- one store instruction, no partial-store branches
        const uint32_t o = 16 * t + 16 <= n ? 16 * t : n - 16;
        const uint32_t w = reg_sym_word(fwd ? F : C, (fwd ? iF : iC) + o, n);
        if (t < nwv) store16(a.out_bytes + off + o, fast_decode(lut, w));
    }
    if (t == 0) {
        // unique minimum => period n; iC + iF < 2n
        if (a.out_index) a.out_index[rec] = fwd ? iF : (iC + iF >= n ? iC + iF - n : iC + iF);
        if (a.out_strand) a.out_strand[rec] = fwd ? 0 : 1;
    }
    return true;
}

// Software-pipelined grid-stride loop of one wave over two 1 KiB LDS buffers (lds[0..255], lds[256..511]):
// while record k is computed, the bytes of record k + stride are in flight (LDS-DMA) and so are the offsets
// of record k + 2*stride (scalar load, lgkmcnt).  vmcnt bookkeeping: between the DMA of a record and the
// point its bytes are needed, the only younger vector-memory instructions are the previous record's stores
// -- at least one when canonical bytes are written (the 16-byte store, or the defer-list store), possibly
// none otherwise -- so the wait is vmcnt(1) resp. vmcnt(0).
CK_DEV void canon_fast_wave_loop(const CanonArgs& a, const uint32_t* lut, uint32_t* lds, uint32_t wave_id, uint32_t n_waves)
{
    const uint64_t total = a.n_records, stride = n_waves;
    uint64_t rec = wave_id;
    if (rec >= total) return;
    uint32_t* bufA = lds;
    uint32_t* bufB = lds + 256;
    const bool stores = a.out_bytes != nullptr;
    uint64_t offA = a.offsets[rec], endA = a.offsets[rec + 1], offB = 0, endB = 0;
    bool okA = fast_issue(a, offA, endA, bufA), okB = false;
    vmem_wait<0>();
    ck_u32x4v sq = sload_u64x2(a.offsets + (rec + stride < total ? rec + stride : rec));
    for (;;) {
        // ---- A computes, B loads
        sload_wait(sq, offB, endB);
        const bool hasB = rec + stride < total;
        okB = fast_issue(a, offB, endB, bufB) && hasB;
        sq = sload_u64x2(a.offsets + (rec + 2 * stride < total ? rec + 2 * stride : rec));
        if (!(okA && fast_process(a, lut, rec, offA, (uint32_t)(endA - offA), fast_fetch(bufA))) && lane_id() == 0)
            a.defer_list[atomic_add_u32(a.defer_count, 1u)] = (uint32_t)rec;
        if (stores) vmem_wait<1>(); else vmem_wait<0>();
        rec += stride;
        if (!hasB) break;
        // ---- B computes, A loads
        sload_wait(sq, offA, endA);
        const bool hasA = rec + stride < total;
        okA = fast_issue(a, offA, endA, bufA) && hasA;
        sq = sload_u64x2(a.offsets + (rec + 2 * stride < total ? rec + 2 * stride : rec));
        if (!(okB && fast_process(a, lut, rec, offB, (uint32_t)(endB - offB), fast_fetch(bufB))) && lane_id() == 0)
            a.defer_list[atomic_add_u32(a.defer_count, 1u)] = (uint32_t)rec;
        if (stores) vmem_wait<1>(); else vmem_wait<0>();
        rec += stride;
        if (!hasA) break;
    }
}

}  // namespace ck
