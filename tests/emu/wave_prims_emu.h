// wave_prims_emu.h -- CPU fiber implementation of the interface of circkit_amd/csrc/wave_prims.h (TEST INFRASTRUCTURE
// ONLY): every wavefront is 64 ucontext fibers scheduled round-robin, every collective a rendezvous of the 64 fibers
// (a lane that skips one deadlocks the wave, which is reported).  Included through CK_WAVE_PRIMS_OVERRIDE by
// tests/emu/emu.cpp; never part of the product build.
#pragma once
#include <string.h>
#define CK_DEV static inline
#define CK_DEV_NOINLINE static

namespace ck {

namespace emu {
uint32_t cur_lane();
uint32_t cur_wave();
void gather(uint64_t v, uint64_t out[64]);   // collective all-gather over the 64 lanes of the wave
void block_barrier();                        // rendezvous of every fiber of the workgroup
}

CK_DEV uint32_t lane_id() { return emu::cur_lane(); }
CK_DEV uint32_t wave_in_block() { return emu::cur_wave(); }
CK_DEV void block_barrier() { emu::block_barrier(); }
CK_DEV uint64_t ballot(bool p)
{
    uint64_t all[64]; emu::gather(p ? 1 : 0, all);
    uint64_t m = 0; for (int i = 0; i < 64; ++i) m |= (all[i] & 1) << i;
    return m;
}
CK_DEV uint32_t shfl(uint32_t v, uint32_t src) { uint64_t all[64]; emu::gather(v, all); return (uint32_t)all[src & 63]; }
CK_DEV uint32_t readlane(uint32_t v, uint32_t l) { uint64_t all[64]; emu::gather(v, all); return (uint32_t)all[l & 63]; }
CK_DEV uint32_t uniform(uint32_t v) { uint64_t all[64]; emu::gather(v, all); return (uint32_t)all[0]; }
CK_DEV uint64_t uniform64(uint64_t v) { uint64_t all[64]; emu::gather(v, all); return all[0]; }
CK_DEV void wave_sync() { uint64_t all[64]; emu::gather(0, all); }
CK_DEV uint32_t wave_min_u32(uint32_t v)
{
    uint64_t all[64]; emu::gather(v, all);
    uint32_t m = ~0u; for (int i = 0; i < 64; ++i) m = (uint32_t)all[i] < m ? (uint32_t)all[i] : m;
    return m;
}
CK_DEV uint64_t wave_sum_u64(uint64_t v)
{
    uint64_t all[64]; emu::gather(v, all);
    uint64_t s = 0; for (int i = 0; i < 64; ++i) s += all[i];
    return s;
}
template <int N>
CK_DEV uint32_t dpp_row_shr(uint32_t v)
{
    uint64_t all[64]; emu::gather(v, all);
    const uint32_t t = lane_id();
    return (t & 15) >= (uint32_t)N ? (uint32_t)all[t - N] : 0u;
}
CK_DEV uint32_t dpp_quad_xor1(uint32_t v) { uint64_t all[64]; emu::gather(v, all); return (uint32_t)all[lane_id() ^ 1]; }
CK_DEV uint32_t dpp_quad_xor2(uint32_t v) { uint64_t all[64]; emu::gather(v, all); return (uint32_t)all[lane_id() ^ 2]; }
CK_DEV void dpp_rowsum4_u64x2(uint64_t& a, uint64_t& b)
{
    const uint32_t t = lane_id();
    uint64_t all[64];
    for (int step = 4; step <= 8; step += 4) {
        emu::gather(a, all); a += (t & 15) >= (uint32_t)step ? all[t - step] : 0;
        emu::gather(b, all); b += (t & 15) >= (uint32_t)step ? all[t - step] : 0;
    }
}
CK_DEV uint64_t dpp_quadsum_u64(uint64_t a)
{
    uint64_t all[64];
    emu::gather(a, all); a += all[lane_id() ^ 1];
    emu::gather(a, all); a += all[lane_id() ^ 2];
    return a;
}
CK_DEV uint64_t add64_parts(uint64_t a, uint32_t lo, uint32_t hi) { return a + (((uint64_t)hi << 32) | lo); }
CK_DEV uint32_t wave_shl1(uint32_t v)
{
    uint64_t all[64]; emu::gather(v, all);
    return lane_id() < 63 ? (uint32_t)all[lane_id() + 1] : 0u;
}
CK_DEV void wave_min2_u32(uint32_t x, uint32_t y, uint32_t& mx, uint32_t& my)
{
    mx = wave_min_u32(x);
    my = wave_min_u32(y);
}
CK_DEV void half_min2_u32(uint32_t x, uint32_t y, uint32_t& xa, uint32_t& xb, uint32_t& ya, uint32_t& yb)
{
    uint64_t ax[64], ay[64];
    emu::gather(x, ax); emu::gather(y, ay);
    xa = xb = ya = yb = ~0u;
    for (int i = 0; i < 32; ++i) {
        xa = (uint32_t)ax[i] < xa ? (uint32_t)ax[i] : xa; xb = (uint32_t)ax[32 + i] < xb ? (uint32_t)ax[32 + i] : xb;
        ya = (uint32_t)ay[i] < ya ? (uint32_t)ay[i] : ya; yb = (uint32_t)ay[32 + i] < yb ? (uint32_t)ay[32 + i] : yb;
    }
}
CK_DEV void half_bcast(uint32_t x, uint32_t& lo, uint32_t& hi)
{
    uint64_t all[64]; emu::gather(x, all);
    lo = (uint32_t)all[lane_id() & 31]; hi = (uint32_t)all[32 + (lane_id() & 31)];
}
CK_DEV void half_min2_bcast_u32(uint32_t x, uint32_t y, uint32_t& mx, uint32_t& my)
{
    uint32_t xa, xb, ya, yb;
    half_min2_u32(x, y, xa, xb, ya, yb);
    mx = lane_id() < 32 ? xa : xb; my = lane_id() < 32 ? ya : yb;
}
CK_DEV bool lane_pred(uint64_t m) { return ((m >> lane_id()) & 1) != 0; }
CK_DEV uint32_t row_min16_u32(uint32_t v)
{
    uint64_t all[64]; emu::gather(v, all);
    const uint32_t base = lane_id() & ~15u;
    uint32_t m = ~0u; for (uint32_t i = base; i < base + 16; ++i) m = (uint32_t)all[i] < m ? (uint32_t)all[i] : m;
    return m;
}
CK_DEV uint32_t lshr64(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> s); }
CK_DEV uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (mask & a) | (~mask & b); }
CK_DEV uint32_t bfi_v(uint32_t mask, uint32_t a, uint32_t b) { return (mask & a) | (~mask & b); }
CK_DEV uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
    uint64_t src = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        uint32_t k = (sel >> (8 * i)) & 0xFF, b;
        if (k <= 7) b = (uint32_t)(src >> (8 * k)) & 0xFF;
        else if (k == 12) b = 0;            // v_perm_b32: 0x0c -> 0x00
        else if (k >= 13) b = 0xFF;         //             >= 0x0d -> 0xff
        else b = 0;                         // 8..11 (sign replication) are never used by the kernels
        r |= b << (8 * i);
    }
    return r;
}
CK_DEV uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh)
{
    return sh ? (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (32u - sh)) : hi;
}
CK_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (s & 31)); }
CK_DEV uint32_t bitrev(uint32_t v)
{
    uint32_t r = 0; for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
CK_DEV int ffs64(uint64_t v) { return __builtin_ctzll(v); }
CK_DEV int ffs32(uint32_t v) { return __builtin_ctz(v); }
CK_DEV int clz32(uint32_t v) { return __builtin_clz(v); }
CK_DEV int popc64(uint64_t v) { return __builtin_popcountll(v); }
CK_DEV int popc32(uint32_t v) { return __builtin_popcount(v); }

struct u32x4 { uint32_t x, y, z, w; };
CK_DEV u32x4 load16(const uint8_t* p) { u32x4 v; memcpy(&v, p, 16); return v; }
CK_DEV void store16(uint8_t* p, u32x4 v) { memcpy(p, &v, 16); }
CK_DEV void store8(uint8_t* p, uint32_t a, uint32_t b) { memcpy(p, &a, 4); memcpy(p + 4, &b, 4); }
CK_DEV void store4(uint8_t* p, uint32_t a) { memcpy(p, &a, 4); }
CK_DEV uint32_t load4(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
CK_DEV uint32_t atomic_add_u32(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
CK_DEV uint32_t lds_atomic_inc(uint32_t* p) { return (*p)++; }
CK_DEV void lds_atomic_or(uint32_t* p, uint32_t v) { *p |= v; }
CK_DEV void lds_atomic_min(uint32_t* p, uint32_t v) { if (v < *p) *p = v; }
CK_DEV void lds_atomic_add(uint32_t* p, uint32_t v) { *p += v; }
struct ck_u32x4v { uint32_t x, y, z, w; };
CK_DEV void glds16_async(uint32_t* lds_dst, const uint8_t* gsrc) { memcpy((uint8_t*)lds_dst + 16 * lane_id(), gsrc, 16); }
CK_DEV void glds16_async_s(uint32_t* lds_dst, const uint8_t* sbase, uint32_t voff) { memcpy((uint8_t*)lds_dst + 16 * lane_id(), sbase + voff, 16); }
CK_DEV void glds4_touch(uint32_t* lds_dump, const void* sbase, uint32_t voff) { memcpy((uint8_t*)lds_dump + 4 * lane_id(), (const uint8_t*)sbase + voff, 4); }
CK_DEV uint32_t sad_u8(uint32_t a, uint32_t b, uint32_t acc)
{
    for (int k = 0; k < 4; ++k) { int d = (int)((a >> (8 * k)) & 0xFF) - (int)((b >> (8 * k)) & 0xFF); acc += (uint32_t)(d < 0 ? -d : d); }
    return acc;
}
CK_DEV int ffs64_or_neg(uint64_t v) { return v ? __builtin_ctzll(v) : -1; }
CK_DEV int ffs32_or_neg(uint32_t v) { return v ? __builtin_ctz(v) : -1; }
CK_DEV uint32_t low_mask16(uint32_t left) { return (1u << (left < 16u ? left : 16u)) - 1u; }
template <int N>
CK_DEV void vmem_wait() {}
CK_DEV void sload_u64x2(const uint64_t* p, uint64_t& a, uint64_t& b) { a = p[0]; b = p[1]; }
CK_DEV void sload_2u64(const uint64_t* p0, const uint64_t* p1, uint64_t& a, uint64_t& b) { a = *p0; b = *p1; }
template <int SPAN>
CK_DEV void sload_group(const uint64_t* p, const uint64_t* q, uint64_t& s, uint64_t& e, uint64_t& o0, uint64_t& o1, uint64_t& o2)
{
    s = p[0]; e = p[SPAN]; o0 = q[0]; o1 = q[1]; o2 = q[2];
}
CK_DEV uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    const uint32_t lo = (a & 0xFFFF) < (b & 0xFFFF) ? (a & 0xFFFF) : (b & 0xFFFF), hi = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
    return (hi << 16) | lo;
}
#define CK_CONST static const
CK_DEV uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((__uint128_t)a * b) >> 64); }
CK_DEV uint32_t udot4(uint32_t a, uint32_t b, uint32_t c)
{
    for (int k = 0; k < 4; ++k) c += ((a >> (8 * k)) & 0xFF) * ((b >> (8 * k)) & 0xFF);
    return c;
}
CK_DEV u32x4 lds_load16(const uint32_t* p) { u32x4 v; memcpy(&v, p, 16); return v; }
CK_DEV void lds_store16(uint32_t* p, u32x4 v) { memcpy(p, &v, 16); }

}  // namespace ck
