// Instruction probes for gfx950 (run on the GPU box):
//  1. ds_read_b128 at byte-granular (unaligned) LDS addresses: correct? how much slower than aligned?
//  2. issue rate of candidate VALU instructions relative to v_alignbit_b32 (v_dot4_u32_u8, v_pk_min_u16, v_min3_u32,
//     v_perm_b32, v_bitop3_b32, v_lshl_or_b32, v_and_or_b32, v_mul_lo_u32).
// Build: hipcc --offload-arch=gfx950 -O3 -o instr_probe instr_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void lds_unaligned(unsigned* out, int align, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 1280 + 64];
    for (int i = threadIdx.x; i < 4 * 1280 + 64; i += 256) lds[i] = (unsigned char)(i * 7 + (i >> 8));
    __syncthreads();
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned addr = (unsigned)(uintptr_t)lds + wave * 1280 + align + 16 * lane;     // LDS byte address (low 32 bits of the flat pointer)
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        acc ^= v;
        addr ^= (i & 1) << 4;       // keep the loop from being hoisted
    }
    unsigned* o = out + (blockIdx.x * 256 + threadIdx.x) * 4;
    o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = acc.w;
}

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void rate(unsigned* out, int iters)
{
    unsigned v0 = threadIdx.x * 0x9E3779B9u, v1 = v0 + 0x1234567, v2 = v0 ^ 0x55aa55aa, v3 = v0 * 3;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(REP8("v_alignbit_b32 %0, %0, %1, 3\n v_alignbit_b32 %1, %1, %2, 5\n v_alignbit_b32 %2, %2, %3, 7\n v_alignbit_b32 %3, %3, %0, 9\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 1) asm volatile(REP8("v_dot4_u32_u8 %0, %0, %1, %2\n v_dot4_u32_u8 %1, %1, %2, %3\n v_dot4_u32_u8 %2, %2, %3, %0\n v_dot4_u32_u8 %3, %3, %0, %1\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 2) asm volatile(REP8("v_pk_min_u16 %0, %0, %1\n v_pk_min_u16 %1, %1, %2\n v_pk_min_u16 %2, %2, %3\n v_pk_min_u16 %3, %3, %0\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 3) asm volatile(REP8("v_min3_u32 %0, %0, %1, %2\n v_min3_u32 %1, %1, %2, %3\n v_min3_u32 %2, %2, %3, %0\n v_min3_u32 %3, %3, %0, %1\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 4) asm volatile(REP8("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 5) asm volatile(REP8("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, %0 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, %1 bitop3:0x96\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 6) asm volatile(REP8("v_lshl_or_b32 %0, %0, 10, %1\n v_lshl_or_b32 %1, %1, 10, %2\n v_lshl_or_b32 %2, %2, 10, %3\n v_lshl_or_b32 %3, %3, 10, %0\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 7) asm volatile(REP8("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 8) asm volatile(REP8("v_min_u32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_min_u32 %2, %2, %3\n v_min_u32 %3, %3, %0\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 9) asm volatile(REP8("v_pk_min_u16 %0, %0, %1\n v_alignbit_b32 %1, %1, %2, 5\n v_pk_min_u16 %2, %2, %3\n v_alignbit_b32 %3, %3, %0, 7\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 10) asm volatile(REP8("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 11) asm volatile(REP8("v_lshrrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 5, %1\n v_lshrrev_b64 %0, 7, %0\n v_lshrrev_b64 %1, 9, %1\n") : "+v"(*(unsigned long long*)&v0), "+v"(*(unsigned long long*)&v2));
    }
    out[blockIdx.x * 256 + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3;
}

template <int MODE>
float run_rate(unsigned* d, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(rate<MODE>, dim3(256 * 4), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(rate<MODE>, dim3(256 * 4), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main()
{
    unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 16);
    // 1. unaligned LDS reads
    std::vector<unsigned> h(256 * 4);
    for (int align = 0; align < 16; ++align) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(lds_unaligned, dim3(1), dim3(256), 0, 0, d, align, 1);
        hipMemcpy(h.data(), d, 256 * 16, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 256; ++t) {
            unsigned char exp[16];
            for (int k = 0; k < 16; ++k) { int i = (t >> 6) * 1280 + align + 16 * (t & 63) + k; exp[k] = (unsigned char)(i * 7 + (i >> 8)); }
            if (memcmp(exp, &h[t * 4], 16)) ++bad;
        }
        hipEventRecord(a);
        hipLaunchKernelGGL(lds_unaligned, dim3(256 * 4), dim3(256), 0, 0, d, align, 20000);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("ds_read_b128 align %2d: %s  %.3f ms / 20000 dependent reads, 4 waves/SIMD\n", align, bad ? "WRONG" : "ok", ms);
    }
    // 2. issue rates (4 waves/SIMD, 32 instr per iteration per wave)
    const int iters = 20000;
    const char* names[] = { "v_alignbit_b32", "v_dot4_u32_u8", "v_pk_min_u16", "v_min3_u32", "v_perm_b32", "v_bitop3_b32", "v_lshl_or_b32",
                            "v_mul_lo_u32", "v_min_u32", "pk_min+alignbit", "v_mad_u32_u24", "v_lshrrev_b64" };
    float ms[12] = { run_rate<0>(d, iters), run_rate<1>(d, iters), run_rate<2>(d, iters), run_rate<3>(d, iters), run_rate<4>(d, iters),
                     run_rate<5>(d, iters), run_rate<6>(d, iters), run_rate<7>(d, iters), run_rate<8>(d, iters), run_rate<9>(d, iters),
                     run_rate<10>(d, iters), run_rate<11>(d, iters) };
    for (int m = 0; m < 12; ++m) printf("%-16s %.3f ms  (%.2fx alignbit)\n", names[m], ms[m], ms[m] / ms[0]);
    return 0;
}
