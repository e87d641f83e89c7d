// xxh3_core.h -- XXH3-64 (seed 0, default secret) of one record by one wavefront.
//
// Replaces `xxhash_rust::xxh3::xxh3_64(canonicalized)` (reference src/uniq.rs:45; xxhash-rust 0.8.6
// implements the published XXH3 algorithm).  Written from the XXH3 specification.
//
// Long inputs (> 240 B): a 1024-byte block is 16 stripes x 4 accumulator pairs = 64 independent
// (stripe, pair) cells of 16 bytes -- exactly one cell per lane.  Between scrambles the accumulators
// are plain sums modulo 2^64, so the cells are combined with a 4-step xor-butterfly over the stripe
// bits of the lane id; lane t ends up holding accumulators (2j, 2j+1), j = t & 3.
#pragma once
#include "wave_prims.h"

namespace ck {

CK_CONST uint8_t XXH3_SECRET[192] = {
    0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c,
    0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f,
    0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21,
    0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6, 0x81, 0x3a, 0x26, 0x4c,
    0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65, 0x8b, 0x1b, 0x53, 0x2e, 0xa3,
    0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19, 0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8,
    0xa8, 0xfa, 0x76, 0x3f, 0xe3, 0x9c, 0x34, 0x3f, 0xf9, 0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d,
    0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31, 0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64,
    0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb, 0xc3, 0xc5, 0x81, 0xa0, 0xff, 0xfa, 0x13, 0x63, 0xeb,
    0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0, 0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e,
    0x2b, 0x16, 0xbe, 0x58, 0x7d, 0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce,
    0x45, 0xcb, 0x3a, 0x8f, 0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e,
};

constexpr uint64_t XP32_1 = 0x9E3779B1ull, XP32_2 = 0x85EBCA77ull, XP32_3 = 0xC2B2AE3Dull;
constexpr uint64_t XP64_1 = 0x9E3779B185EBCA87ull, XP64_2 = 0xC2B2AE3D27D4EB4Full, XP64_3 = 0x165667B19E3779F9ull,
                   XP64_4 = 0x85EBCA77C2B2AE63ull, XP64_5 = 0x27D4EB2F165667C5ull;
constexpr uint64_t XPMX1 = 0x165667919E3779F9ull, XPMX2 = 0x9FB21C651E98DF25ull;

CK_DEV uint64_t xsec64(uint32_t o)
{
    uint64_t v = 0;
#pragma unroll
    for (int i = 7; i >= 0; --i) v = (v << 8) | XXH3_SECRET[o + i];
    return v;
}
CK_DEV uint32_t xsec32(uint32_t o) { return (uint32_t)xsec64(o); }
CK_DEV uint64_t xrd64(const uint8_t* p) { return (uint64_t)load4(p) | ((uint64_t)load4(p + 4) << 32); }
CK_DEV uint64_t xrotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
CK_DEV uint64_t xswap64(uint64_t x)
{
    x = ((x & 0x00FF00FF00FF00FFull) << 8) | ((x >> 8) & 0x00FF00FF00FF00FFull);
    x = ((x & 0x0000FFFF0000FFFFull) << 16) | ((x >> 16) & 0x0000FFFF0000FFFFull);
    return (x << 32) | (x >> 32);
}
CK_DEV uint64_t xmulhi(uint64_t a, uint64_t b) { return mulhi64(a, b); }
CK_DEV uint64_t xfold(uint64_t a, uint64_t b) { return (a * b) ^ xmulhi(a, b); }
CK_DEV uint64_t xaval3(uint64_t h) { h ^= h >> 37; h *= XPMX1; return h ^ (h >> 32); }
CK_DEV uint64_t xaval64(uint64_t h)
{
    h ^= h >> 33; h *= XP64_2; h ^= h >> 29; h *= XP64_3; return h ^ (h >> 32);
}
// What is hashed: bytes in memory (XPlain), or -- hash-only batches, `uniq` without `--canonicalize` (src/uniq.rs:45,55-60)
// -- the canonical record as a VIEW of the input record (XView): rotation `rot` of the record itself or of its reverse
// complement (bio 1.3.1's table), never written out as bytes.  v[i] = s[(rot + i) mod n], or comp[s[n - 1 - (rot + i) mod n]].
struct XPlain { const uint8_t* p; };
struct XView { const uint8_t* s; uint32_t n, rot; bool rc; const uint8_t* comp; };
CK_DEV uint32_t xr8(const XPlain& r, uint32_t o) { return r.p[o]; }
CK_DEV uint32_t xr32(const XPlain& r, uint32_t o) { return load4(r.p + o); }
CK_DEV uint64_t xr64(const XPlain& r, uint32_t o) { return xrd64(r.p + o); }
CK_DEV uint32_t xr8(const XView& r, uint32_t o)
{
    uint32_t q = r.rot + o;
    q = q >= r.n ? q - r.n : q;
    return r.rc ? r.comp[r.s[r.n - 1 - q]] : r.s[q];
}
CK_DEV uint32_t xr32(const XView& r, uint32_t o) { return xr8(r, o) | (xr8(r, o + 1) << 8) | (xr8(r, o + 2) << 16) | (xr8(r, o + 3) << 24); }
CK_DEV uint64_t xr64(const XView& r, uint32_t o)
{
    uint32_t q = r.rot + o;
    q = q >= r.n ? q - r.n : q;
    if (q + 8 <= r.n) {                                     // the eight bytes do not wrap
        if (!r.rc) return xrd64(r.s + q);
        const uint64_t x = xrd64(r.s + (r.n - 8 - q));       // source bytes n-1-q-7 .. n-1-q: reversed, complemented
        uint64_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v |= (uint64_t)r.comp[(x >> (56 - 8 * i)) & 0xFF] << (8 * i);
        return v;
    }
    uint64_t v = 0;
    for (uint32_t i = 0; i < 8; ++i) v |= (uint64_t)xr8(r, o + i) << (8 * i);
    return v;
}

template <class R>
CK_DEV uint64_t xmix16(const R& in, uint32_t o, uint32_t so)
{
    return xfold(xr64(in, o) ^ xsec64(so), xr64(in, o + 8) ^ xsec64(so + 8));
}

// Lengths <= 240: every lane computes the same scalar recipe (addresses are wave-uniform).
template <class R>
CK_DEV uint64_t xxh3_short(const R& in, uint32_t len)
{
    if (len == 0) return xaval64(xsec64(56) ^ xsec64(64));
    if (len <= 3) {
        const uint32_t c1 = xr8(in, 0), c2 = xr8(in, len >> 1), c3 = xr8(in, len - 1);
        const uint32_t comb = (c1 << 16) | (c2 << 24) | c3 | (len << 8);
        return xaval64((uint64_t)comb ^ (uint64_t)(xsec32(0) ^ xsec32(4)));
    }
    if (len <= 8) {
        const uint64_t i1 = xr32(in, 0), i2 = xr32(in, len - 4);
        uint64_t h = (i2 + (i1 << 32)) ^ (xsec64(8) ^ xsec64(16));
        h ^= xrotl(h, 49) ^ xrotl(h, 24);
        h *= XPMX2; h ^= (h >> 35) + len; h *= XPMX2;
        return h ^ (h >> 28);
    }
    if (len <= 16) {
        const uint64_t lo = xr64(in, 0) ^ (xsec64(24) ^ xsec64(32)), hi = xr64(in, len - 8) ^ (xsec64(40) ^ xsec64(48));
        return xaval3((uint64_t)len + xswap64(lo) + hi + xfold(lo, hi));
    }
    if (len <= 128) {
        uint64_t acc = (uint64_t)len * XP64_1;
        if (len > 32) {
            if (len > 64) {
                if (len > 96) { acc += xmix16(in, 48, 96); acc += xmix16(in, len - 64, 112); }
                acc += xmix16(in, 32, 64); acc += xmix16(in, len - 48, 80);
            }
            acc += xmix16(in, 16, 32); acc += xmix16(in, len - 32, 48);
        }
        acc += xmix16(in, 0, 0); acc += xmix16(in, len - 16, 16);
        return xaval3(acc);
    }
    uint64_t acc = (uint64_t)len * XP64_1;
    const uint32_t rounds = len / 16;
    for (uint32_t i = 0; i < 8; ++i) acc += xmix16(in, 16 * i, 16 * i);
    acc = xaval3(acc);
    for (uint32_t i = 8; i < rounds; ++i) acc += xmix16(in, 16 * i, 16 * (i - 8) + 3);
    acc += xmix16(in, len - 16, 136 - 17);
    return xaval3(acc);
}

// One 16-byte cell: contributions to accumulators (2j, 2j+1) from data words d0,d1 with secret offset so.
CK_DEV void xcell(uint64_t d0, uint64_t d1, uint32_t so, uint64_t& a0, uint64_t& a1)
{
    const uint64_t k0 = d0 ^ xsec64(so), k1 = d1 ^ xsec64(so + 8);
    a0 += d1 + (uint64_t)(uint32_t)k0 * (k0 >> 32);
    a1 += d0 + (uint64_t)(uint32_t)k1 * (k1 >> 32);
}

CK_DEV uint64_t xsum_stripes(uint64_t v)   // sum over the 16 lanes that share (lane & 3)
{
#pragma unroll
    for (uint32_t m = 4; m <= 32; m <<= 1) {
        const uint32_t lo = shfl((uint32_t)v, lane_id() ^ m), hi = shfl((uint32_t)(v >> 32), lane_id() ^ m);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

// Per-lane constants of the long-input path (lane = stripe s = lane >> 2, accumulator pair j = lane & 3): secret
// words and initial accumulators.  A kernel that hashes many records computes them once per wave.
struct XWaveConst {
    uint64_t k0, k1;      // stripe cell:   secret at 8 s + 16 j, + 8
    uint64_t sc0, sc1;    // scramble:      secret at 128 + 16 j, + 8
    uint64_t l0, l1;      // last stripe:   secret at 121 + 16 j, + 8
    uint64_t m0, m1;      // final merge:   secret at 11 + 16 j, + 8
    uint64_t i0, i1;      // initial accumulators {P32_3, P64_1, P64_2, P64_3, P64_4, P32_2, P64_5, P32_1}[2j, 2j+1]
};
CK_DEV XWaveConst xwave_const()
{
    const uint32_t lane = lane_id(), j = lane & 3, s = lane >> 2;
    XWaveConst k;
    k.k0 = xsec64(8 * s + 16 * j); k.k1 = xsec64(8 * s + 16 * j + 8);
    k.sc0 = xsec64(128 + 16 * j); k.sc1 = xsec64(128 + 16 * j + 8);
    k.l0 = xsec64(121 + 16 * j); k.l1 = xsec64(121 + 16 * j + 8);
    k.m0 = xsec64(11 + 16 * j); k.m1 = xsec64(11 + 16 * j + 8);
    k.i0 = j == 0 ? XP32_3 : j == 1 ? XP64_2 : j == 2 ? XP64_4 : XP64_5;
    k.i1 = j == 0 ? XP64_1 : j == 1 ? XP64_3 : j == 2 ? XP32_2 : XP32_1;
    return k;
}
CK_DEV void xcell_k(uint64_t d0, uint64_t d1, uint64_t s0, uint64_t s1, uint64_t& a0, uint64_t& a1)
{
    const uint64_t k0 = d0 ^ s0, k1 = d1 ^ s1;
    a0 += d1 + (uint64_t)(uint32_t)k0 * (k0 >> 32);
    a1 += d0 + (uint64_t)(uint32_t)k1 * (k1 >> 32);
}

// XXH3-64 (seed 0) of in[0, len) computed by one wave; every lane returns the hash.  Long inputs: lane = (stripe,
// accumulator pair) cell of a 1024-byte block; the loads of up to four blocks and of the last stripe are all issued
// before the first is consumed (a 1-2 kb record is otherwise three dependent round trips).
template <class R>
CK_DEV uint64_t xxh3_64_wave_r(const R& in, uint32_t len, const XWaveConst& k)
{
    if (len <= 240) return xxh3_short(in, len);
    const uint32_t lane = lane_id(), j = lane & 3, s = lane >> 2;
    uint64_t a0 = k.i0, a1 = k.i1;
    const uint32_t nb = (len - 1) / 1024;                  // full blocks before the last (partial or full) one
    const uint32_t last_o = len - 64 + 16 * j;
    const uint64_t ld0 = xr64(in, last_o), ld1 = xr64(in, last_o + 8);   // last stripe, in flight with the blocks below
    constexpr uint32_t U = 4;
    for (uint32_t b0 = 0; b0 <= nb; b0 += U) {
        uint64_t d0[U], d1[U];
        bool on[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t b = b0 + u;
            const uint32_t stripes = b < nb ? 16u : ((len - 1) - 1024 * nb) / 64;
            on[u] = b <= nb && s < stripes;
            d0[u] = d1[u] = 0;
            if (on[u]) { const uint32_t o = 1024 * b + 16 * lane; d0[u] = xr64(in, o); d1[u] = xr64(in, o + 8); }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t b = b0 + u;
            if (b > nb) break;
            uint64_t c0 = 0, c1 = 0;
            if (on[u]) xcell_k(d0[u], d1[u], k.k0, k.k1, c0, c1);
            a0 += xsum_stripes(c0);
            a1 += xsum_stripes(c1);
            if (b < nb) {   // scramble with the last 64 secret bytes
                a0 = (a0 ^ (a0 >> 47) ^ k.sc0) * XP32_1;
                a1 = (a1 ^ (a1 >> 47) ^ k.sc1) * XP32_1;
            }
        }
    }
    xcell_k(ld0, ld1, k.l0, k.l1, a0, a1);                 // last stripe: the final 64 bytes, secret offset 192 - 64 - 7
    uint64_t t = xfold(a0 ^ k.m0, a1 ^ k.m1);
#pragma unroll
    for (uint32_t m = 1; m <= 2; m <<= 1) {
        const uint32_t lo = shfl((uint32_t)t, lane ^ m), hi = shfl((uint32_t)(t >> 32), lane ^ m);
        t += ((uint64_t)hi << 32) | lo;
    }
    return xaval3((uint64_t)len * XP64_1 + t);
}
CK_DEV uint64_t xxh3_64_wave(const uint8_t* in, uint32_t len, const XWaveConst& k) { return xxh3_64_wave_r(XPlain{ in }, len, k); }
CK_DEV uint64_t xxh3_64_wave(const uint8_t* in, uint32_t len) { return xxh3_64_wave(in, len, xwave_const()); }
// the canonical record as a view of the input record s[0, n): view = rotation | strand << 31 (CanonArgs::out_view)
CK_DEV uint64_t xxh3_64_wave_view(const uint8_t* s, uint32_t n, uint32_t view, const uint8_t* comp, const XWaveConst& k)
{
    // a rotation outside the record is not a view (a record no stage could take leaves none: its slot of the ctx's view array
    // is whatever an earlier batch wrote there): hashed as the record itself, never addressed with
    const uint32_t rot = view & 0x7FFFFFFFu;
    return xxh3_64_wave_r(XView{ s, n, rot < n ? rot : 0u, (view >> 31) != 0, comp }, n, k);
}

}  // namespace ck
