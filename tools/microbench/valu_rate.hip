// Issue cost of the vector instructions the circkit kernels are made of, gfx950: cycles per instruction per SIMD with 8 waves
// per SIMD, eight independent chains per wave (so latency is hidden and the number is the pipe's rate).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters)
{
    unsigned v[8], w[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 7 + i; w[i] = threadIdx.x * 13 + i + 1; }
    unsigned sh = (threadIdx.x & 15) * 2 + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "v"(sh));
                if (MODE == 1) { unsigned long long x = ((unsigned long long)v[i] << 32) | w[i]; asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(x) : "v"(sh)); v[i] = (unsigned)x; w[i] = (unsigned)(x >> 32); }
                if (MODE == 2) { unsigned long long x = ((unsigned long long)v[i] << 32) | w[i]; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(v[i]), "v"(sh) : "vcc"); v[i] = (unsigned)x; w[i] = (unsigned)(x >> 32); }
                if (MODE == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "v"(sh));
                if (MODE == 4) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "v"(sh));
                if (MODE == 5) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                if (MODE == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(w[i]) : );
                if (MODE == 7) asm volatile("v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
                if (MODE == 8) asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v[i]) : "v"(w[i]));
                if (MODE == 9) { unsigned s; asm volatile("v_readlane_b32 %0, %1, 31" : "=s"(s) : "v"(v[i])); w[i] ^= s; }
                if (MODE == 10) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(w[i]));
                if (MODE == 11) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(v[i]) : "v"(w[i]));
                if (MODE == 12) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                if (MODE == 13) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(v[i]) : "v"(w[i]) : "vcc");
                if (MODE == 14) asm volatile("v_bfi_b32 %0, %2, %0, %1" : "+v"(v[i]) : "v"(w[i]), "v"(sh));
                if (MODE == 15) asm volatile("v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(v[i]) : "v"(w[i]));
            }
        }
    }
    unsigned acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= v[i] ^ w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// v_dot4 through the builtin (hazard recognizer)
__global__ __launch_bounds__(256) void kdot(unsigned* out, int iters)
{
    unsigned v[8], w[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 7 + i; w[i] = threadIdx.x * 13 + i + 1; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_udot4(v[i], w[i], v[(i + 3) & 7], false);
        }
    }
    unsigned acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= v[i] ^ w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
float run(unsigned* d, int blocks, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main()
{
    unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 4000, bpc = 8, blocks = 256 * bpc;
    const double clk = 2.4e6, per = iters * 64.0 * bpc;
    const char* names[] = { "v_alignbit_b32", "v_lshrrev_b64", "v_mad_u64_u32", "v_perm_b32", "v_sad_u8", "v_pk_min_u16", "v_cndmask_b32", "v_min_u32_dpp",
                            "v_mov_b32_dpp wave_shl", "v_readlane_b32 (+xor)", "v_permlane32_swap_b32", "v_lshl_or_b32", "v_mul_lo_u32", "v_add_co_u32", "v_bfi_b32", "v_min_u32_sdwa" };
    float ms[16];
    ms[0] = run<0>(d, blocks, iters); ms[1] = run<1>(d, blocks, iters); ms[2] = run<2>(d, blocks, iters); ms[3] = run<3>(d, blocks, iters);
    ms[4] = run<4>(d, blocks, iters); ms[5] = run<5>(d, blocks, iters); ms[6] = run<6>(d, blocks, iters); ms[7] = run<7>(d, blocks, iters);
    ms[8] = run<8>(d, blocks, iters); ms[9] = run<9>(d, blocks, iters); ms[10] = run<10>(d, blocks, iters); ms[11] = run<11>(d, blocks, iters);
    ms[12] = run<12>(d, blocks, iters); ms[13] = run<13>(d, blocks, iters); ms[14] = run<14>(d, blocks, iters); ms[15] = run<15>(d, blocks, iters);
    for (int i = 0; i < 16; ++i) printf("%-26s %.3f ms  %.2f cycles / instruction / SIMD\n", names[i], ms[i], ms[i] * clk / per);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kdot, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(kdot, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float m; hipEventElapsedTime(&m, a, b);
    printf("%-26s %.3f ms  %.2f cycles / instruction / SIMD\n", "v_dot4_u32_u8 (builtin)", m, m * clk / per);
    return 0;
}
