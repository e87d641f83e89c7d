// wave_prims.h -- the handful of wavefront-level primitives the circkit kernels are written against: thin wrappers
// over CDNA4 intrinsics (64-lane ballot, ds_bpermute, v_perm_b32, v_alignbit_b32, DPP-based reductions).
//
// The kernels are written against this interface only, and keep one discipline: every collective (ballot / shfl /
// readlane / wave_min / wave_sync / DPP moves) is executed by all 64 lanes in wave-uniform control flow.  A test
// harness may therefore substitute its own implementation of the interface (CK_WAVE_PRIMS_OVERRIDE = path of its
// header; tests/emu/ runs the kernel source as 64 fibers per wave on the CPU and asserts that discipline).  Nothing of
// such a harness lives here or is linked into libcirckit_hip.so.
#pragma once
#include <stdint.h>

#ifdef CK_WAVE_PRIMS_OVERRIDE
#include CK_WAVE_PRIMS_OVERRIDE
#else
// ================================================================================================
// gfx950 device build
// ================================================================================================
#include <hip/hip_runtime.h>
#define CK_DEV __device__ __forceinline__
#define CK_DEV_NOINLINE __device__ __noinline__

namespace ck {

CK_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
CK_DEV uint32_t wave_in_block() { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
CK_DEV void block_barrier() { __syncthreads(); }
CK_DEV uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
CK_DEV uint32_t shfl(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v); }
CK_DEV uint32_t readlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
CK_DEV uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
CK_DEV uint64_t uniform64(uint64_t v)
{
    return ((uint64_t)uniform((uint32_t)(v >> 32)) << 32) | uniform((uint32_t)v);
}
// LDS traffic of one wave is issued and serviced in order; this only stops the compiler from
// moving LDS accesses across the point (no instruction besides a possible s_waitcnt).
CK_DEV void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// min over each aligned row of 16 lanes, in every lane of the row: 4 DPP steps (xor 1, xor 2, half mirror,
// mirror), no LDS crossbar traffic.
CK_DEV uint32_t row_min16_u32(uint32_t v)
{
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false); v = o < v ? o : v;   // quad_perm [1,0,3,2]
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false); v = o < v ? o : v;   // quad_perm [2,3,0,1]
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false); v = o < v ? o : v;  // row_half_mirror
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false); v = o < v ? o : v;  // row_mirror
    return v;
}
// wave-wide min, wave-uniform result (lands in SGPRs): row mins by DPP, then 4 readlanes + scalar mins.
CK_DEV uint32_t wave_min_u32(uint32_t v)
{
    v = row_min16_u32(v);
    const uint32_t a = readlane(v, 0), b = readlane(v, 16), c = readlane(v, 32), d = readlane(v, 48);
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
// lane i <- lane i+1 (lane 63 <- 0): DPP wave_shl:1, one VALU op instead of a ds_bpermute round trip.
CK_DEV uint32_t wave_shl1(uint32_t v)
{
    uint32_t o;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(o) : "v"(v));
    return o;
}
// value of lane i-N of the same 16-lane row, 0 where that lane does not exist (DPP row_shr:N, bound_ctrl)
template <int N>
CK_DEV uint32_t dpp_row_shr(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xF, 0xF, true);
}
// value of lane i^1 / i^2 (quad_perm)
CK_DEV uint32_t dpp_quad_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }
CK_DEV uint32_t dpp_quad_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); }
// 64-bit sums along DPP paths with the add fused into the DPP instruction (v_add_co / v_addc_co pairs, 2 VALU per
// 64-bit add; through update_dpp + uint64_t arithmetic hipcc emits 5).  Two accumulators are interleaved so that
// each pair fills the other's VALU->DPP wait states.
//   row sums: lane i += lane i-4, then += lane i-8 of its 16-lane row (lanes 12..15 end up with the sum over
//   i, i-4, i-8, i-12); lanes that do not exist contribute 0.
CK_DEV void dpp_rowsum4_u64x2(uint64_t& a, uint64_t& b)
{
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    asm volatile(
        "s_nop 1\n\t"
        "v_add_co_u32_dpp %0, vcc, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_add_co_u32_dpp %2, vcc, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_add_co_u32_dpp %0, vcc, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_add_co_u32_dpp %2, vcc, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1"
        : "+v"(al), "+v"(ah), "+v"(bl), "+v"(bh) :: "vcc");
    a = ((uint64_t)ah << 32) | al;
    b = ((uint64_t)bh << 32) | bl;
}
//   quad sum: every lane of a quad ends up with the sum over the quad
CK_DEV uint64_t dpp_quadsum_u64(uint64_t a)
{
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32);
    asm volatile(
        "s_nop 1\n\t"
        "v_add_co_u32_dpp %0, vcc, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_co_u32_dpp %0, vcc, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(al), "+v"(ah) :: "vcc");
    return ((uint64_t)ah << 32) | al;
}
// a + (hi:lo) with the carry chain spelled out (2 VALU)
CK_DEV uint64_t add64_parts(uint64_t a, uint32_t lo, uint32_t hi)
{
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32);
    asm("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(al), "+v"(ah) : "v"(lo), "v"(hi) : "vcc");
    return ((uint64_t)ah << 32) | al;
}
// Two independent wave-wide mins at once.  The two DPP chains are interleaved so each fills the other's
// VALU->DPP wait states (2 needed; one comes from the sibling instruction, one from s_nop 0) and the
// min is fused into the DPP instruction (hipcc emits v_mov_dpp + v_min otherwise).
CK_DEV void wave_min2_u32(uint32_t x, uint32_t y, uint32_t& mx, uint32_t& my)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x), "+v"(y));
    // rows 1,3 take min with lane 15 of rows 0,2; rows 2,3 take min with lane 31: lane 63 = wave min
    asm volatile(
        "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x), "+v"(y));
    mx = readlane(x, 63);
    my = readlane(y, 63);
}
// The same for the two HALVES of the wave apart (lanes 0..31 / 32..63: canon_pair.h runs one record in each): the row steps, then
// rows 1 and 3 take the minimum with lane 15 of rows 0 and 2 -- lanes 31 and 63 hold their half's minimum.
CK_DEV void half_min2_u32(uint32_t x, uint32_t y, uint32_t& xa, uint32_t& xb, uint32_t& ya, uint32_t& yb)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x), "+v"(y));
    xa = readlane(x, 31); xb = readlane(x, 63);
    ya = readlane(y, 31); yb = readlane(y, 63);
}
// lo = x of lane t & 31, hi = x of lane 32 + (t & 31): the lower / upper half of the wave shown to both halves (gfx950:
// v_permlane32_swap_b32 exchanges the upper 32 lanes of one register with the lower 32 of another -- here the same value twice)
CK_DEV void half_bcast(uint32_t x, uint32_t& lo, uint32_t& hi)
{
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    lo = r[0]; hi = r[1];
}
// ... and with the result in every lane of the half instead of in scalars: rows 1 and 3 hold it after the row_bcast step, and
// v_permlane16_swap_b32 (gfx950: swaps the odd rows of one register with the even rows of another) of the value with itself
// leaves rows 1, 1, 3, 3 in its second result.
CK_DEV void half_min2_bcast_u32(uint32_t x, uint32_t y, uint32_t& mx, uint32_t& my)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x), "+v"(y));
    mx = __builtin_amdgcn_permlane16_swap(x, x, false, false)[1];
    my = __builtin_amdgcn_permlane16_swap(y, y, false, false)[1];
}
// a wave-uniform 64-bit mask as a per-lane predicate (no instruction: the mask is used as the exec / select operand)
CK_DEV bool lane_pred(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// low 32 bits of (hi:lo) >> s, s in 0..63
CK_DEV uint32_t lshr64(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> s); }
CK_DEV uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (mask & a) | (~mask & b); }   // v_bfi_b32
// the same as ONE instruction whatever the compiler makes of the expression (with a mask that is a loop-invariant register it
// emits not / and / and / or)
CK_DEV uint32_t bfi_v(uint32_t mask, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}
CK_DEV uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane_id() ^ m) << 2), (int)(uint32_t)v);
        uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane_id() ^ m) << 2), (int)(uint32_t)(v >> 32));
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
CK_DEV uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
// funnel: the 32 bits starting `sh` bits (0..31) into the 64-bit value hi:lo, counted from the top
CK_DEV uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh)
{
    return sh ? __builtin_amdgcn_alignbit(hi, lo, 32u - sh) : hi;   // v_alignbit_b32: ({hi,lo} >> n)[31:0]
}
// low 32 bits of (hi:lo) >> (s & 31): ONE full-rate instruction (v_alignbit_b32; the 64-bit shift behind lshr64() issues at half
// rate and wants its operands in a register pair -- tools/microbench/valu_rate.hip)
CK_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbit(hi, lo, s); }
CK_DEV uint32_t bitrev(uint32_t v) { return __builtin_bitreverse32(v); }
CK_DEV int ffs64(uint64_t v) { return __builtin_ctzll(v); }   // v != 0
CK_DEV int ffs32(uint32_t v) { return __builtin_ctz(v); }     // v != 0
CK_DEV int clz32(uint32_t v) { return __builtin_clz(v); }     // v != 0
CK_DEV int popc64(uint64_t v) { return __builtin_popcountll(v); }
CK_DEV int popc32(uint32_t v) { return __builtin_popcount(v); }

struct u32x4 { uint32_t x, y, z, w; };
// Global memory on gfx950 runs in unaligned access mode: a dwordx4 load/store may start at any byte.
CK_DEV u32x4 load16(const uint8_t* p)
{
    typedef uint32_t v4 __attribute__((ext_vector_type(4), aligned(1)));
    v4 v = *reinterpret_cast<const v4*>(p);
    return u32x4{ v.x, v.y, v.z, v.w };
}
CK_DEV void store16(uint8_t* p, u32x4 v)
{
    typedef uint32_t v4 __attribute__((ext_vector_type(4), aligned(1)));
    v4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *reinterpret_cast<v4*>(p) = t;
}
CK_DEV void store8(uint8_t* p, uint32_t a, uint32_t b)
{
    typedef uint32_t v2 __attribute__((ext_vector_type(2), aligned(1)));
    v2 t; t.x = a; t.y = b;
    *reinterpret_cast<v2*>(p) = t;
}
CK_DEV void store4(uint8_t* p, uint32_t a)
{
    typedef uint32_t v1 __attribute__((aligned(1)));
    *reinterpret_cast<v1*>(p) = a;
}
CK_DEV uint32_t load4(const uint8_t* p)
{
    typedef uint32_t v1 __attribute__((aligned(1)));
    return *reinterpret_cast<const v1*>(p);
}
CK_DEV uint32_t atomic_add_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
CK_DEV uint32_t lds_atomic_inc(uint32_t* p) { return atomicAdd(p, 1u); }     // p in LDS: ds_add_rtn_u32
CK_DEV void lds_atomic_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
CK_DEV void lds_atomic_min(uint32_t* p, uint32_t v) { atomicMin(p, v); }
CK_DEV void lds_atomic_add(uint32_t* p, uint32_t v) { atomicAdd(p, v); }

// Two consecutive u64 (a CSR offset pair) through the scalar cache: s_load_dwordx4, tracked by lgkmcnt, so
// it never forces a vmcnt(0) that would drain the prefetched record bytes.  The wait is INSIDE the same asm
// statement: an asynchronous form (load now, wait a step later) is unsafe -- between the two statements the
// compiler considers the destination SGPRs ordinary values and may copy or reuse them while the load is
// still in flight (seen in the ISA: the registers were recycled for loop arithmetic; rare garbage offsets).
// p must be wave-uniform and the memory read-only for the kernel's lifetime.
// LDS-DMA: every lane copies 16 bytes from its own global address straight into LDS at
// lds_dst + 16*lane (global_load_lds_dwordx4; M0 carries the wave-uniform LDS base).  No VGPR destination,
// so the compiler cannot touch the data before it lands; completion is OUR bookkeeping: vmem_wait<N>(),
// N = vector-memory instructions issued after this one that may still be outstanding (vmcnt counts loads,
// stores and LDS-DMA together, in issue order).  Used by the software pipeline of canon_fast.h, where
// hipcc's own waitcnt insertion degrades to vmcnt(0) and would drain the prefetch.
CK_DEV void glds16_async(uint32_t* lds_dst, const uint8_t* gsrc)
{
    uint32_t keep;
    const uint32_t dst = (uint32_t)(uintptr_t)lds_dst;     // low 32 bits of an LDS generic address = LDS offset
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// same, source = scalar base + per-lane byte offset (no 64-bit vector address arithmetic); M0 is declared clobbered
// instead of saved and restored
CK_DEV void glds16_async_s(uint32_t* lds_dst, const uint8_t* sbase, uint32_t voff)
{
    const uint32_t dst = (uint32_t)(uintptr_t)lds_dst;
    // default cache policy: `nt` on the load and nt / sc1 on the output stores all measured slower (A/B on one box:
    // 4.32 ms plain, 4.39 nt loads, 4.43 nt stores, 4.98 sc1 stores)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(dst) : "memory", "m0");
}
// Touches global memory ahead of its use: every lane fetches 4 bytes at sbase + voff by LDS-DMA into a dump area of 256 bytes
// (nobody reads it).  No VGPR destination, so nothing ever waits for it but the hand-counted vmcnt of the caller's loop (one
// more vector-memory instruction in its order).  What it is for: the line is in the L2 when the scalar load of the next
// iteration asks for it.
CK_DEV void glds4_touch(uint32_t* lds_dump, const void* sbase, uint32_t voff)
{
    const uint32_t dst = (uint32_t)(uintptr_t)lds_dump;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" :: "v"(voff), "s"(sbase), "s"(dst) : "memory", "m0");
}
template <int N>
CK_DEV void vmem_wait()
{
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
CK_DEV uint32_t sad_u8(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }   // v_sad_u8
CK_DEV int ffs64_or_neg(uint64_t v) { return __builtin_ffsll((long long)v) - 1; }      // s_ff1_i32_b64: -1 for 0
// index of the lowest set bit of a wave-uniform word, -1 for 0: ONE s_ff1_i32_b32 (through __builtin_ffs the compiler adds a
// compare and two selects for the zero case the instruction already answers)
CK_DEV int ffs32_or_neg(uint32_t v)
{
    int r;
    asm("s_ff1_i32_b32 %0, %1" : "=s"(r) : "s"(v));
    return r;
}
// the low min(left, 16) bits set (left wave-uniform): s_min_u32 + s_bfm_b32
CK_DEV uint32_t low_mask16(uint32_t left)
{
    const uint32_t l16 = left < 16u ? left : 16u;
    return (1u << l16) - 1u;
}
typedef uint32_t ck_u32x4v __attribute__((ext_vector_type(4)));
CK_DEV void sload_u64x2(const uint64_t* p, uint64_t& a, uint64_t& b)
{
    ck_u32x4v r;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(p) : "memory");
    a = ((uint64_t)r.y << 32) | r.x;
    b = ((uint64_t)r.w << 32) | r.z;
}

// what one iteration of canon_stream.h needs from the offsets array, with a single wait: p[0] and p[SPAN] (first and
// one-past-last offset of a record group) and q[0..2] (the offsets of this wave's two records)
template <int SPAN>
CK_DEV void sload_group(const uint64_t* p, const uint64_t* q, uint64_t& s, uint64_t& e, uint64_t& o0, uint64_t& o1, uint64_t& o2)
{
    uint64_t x, y, z;
    ck_u32x4v r;
    asm volatile("s_load_dwordx2 %0, %4, 0x0\n\ts_load_dwordx2 %1, %4, %6\n\ts_load_dwordx4 %2, %5, 0x0\n\t"
                 "s_load_dwordx2 %3, %5, 0x10\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(x), "=&s"(y), "=&s"(r), "=&s"(z) : "s"(p), "s"(q), "n"(SPAN * 8) : "memory");
    s = x; e = y; o2 = z;
    o0 = ((uint64_t)r.y << 32) | r.x;
    o1 = ((uint64_t)r.w << 32) | r.z;
}
// v_dot4_u32_u8: sum of the four byte products + c.  Through the builtin, never inline asm: on gfx940+ a dot result
// may be read by another VALU instruction only 3 wait states later, and only the compiler's hazard recognizer
// inserts those (measured: the asm form returns stale registers).
CK_DEV uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
// 16 bytes from a 16-byte-aligned LDS address: one ds_read_b128 (conflict-free for consecutive lanes)
CK_DEV u32x4 lds_load16(const uint32_t* p)
{
    typedef uint32_t v4 __attribute__((ext_vector_type(4), aligned(16)));
    v4 v = *reinterpret_cast<const v4*>(p);
    return u32x4{ v.x, v.y, v.z, v.w };
}

CK_DEV void lds_store16(uint32_t* p, u32x4 v)      // 16-byte-aligned LDS address: one ds_write_b128
{
    typedef uint32_t v4 __attribute__((ext_vector_type(4), aligned(16)));
    v4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *reinterpret_cast<v4*>(p) = t;
}

// two 16-bit unsigned minima at once (v_pk_min_u16)
CK_DEV uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
#define CK_CONST __device__ __constant__
CK_DEV uint64_t mulhi64(uint64_t a, uint64_t b) { return __umul64hi(a, b); }

// two independent u64 (e.g. the first and last offset of a record group), one wait
CK_DEV void sload_2u64(const uint64_t* p0, const uint64_t* p1, uint64_t& a, uint64_t& b)
{
    uint64_t x, y;
    asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(x), "=&s"(y) : "s"(p0), "s"(p1) : "memory");
    a = x; b = y;
}

}  // namespace ck

#endif  // CK_WAVE_PRIMS_OVERRIDE
