// Issue-rate microbenchmark for gfx950: does scalar (SALU) work issue for free beside vector (VALU) work?
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_bench issue_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters)
{
    unsigned v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    for (int i = 0; i < iters; ++i) {
#define V8 "v_alignbit_b32 %0, %0, %1, 3\n v_alignbit_b32 %1, %1, %2, 5\n v_alignbit_b32 %2, %2, %3, 7\n v_alignbit_b32 %3, %3, %4, 9\n" \
           "v_alignbit_b32 %4, %4, %5, 11\n v_alignbit_b32 %5, %5, %6, 13\n v_alignbit_b32 %6, %6, %7, 15\n v_alignbit_b32 %7, %7, %0, 17\n"
#define S8 "s_add_u32 %8, %8, %9\n s_xor_b32 %9, %9, %10\n s_add_u32 %10, %10, %11\n s_xor_b32 %11, %11, %8\n" \
           "s_add_u32 %8, %8, %10\n s_xor_b32 %9, %9, %11\n s_add_u32 %10, %10, %8\n s_xor_b32 %11, %11, %9\n"
#define VS8 "v_alignbit_b32 %0, %0, %1, 3\n s_add_u32 %8, %8, %9\n v_alignbit_b32 %1, %1, %2, 5\n s_xor_b32 %9, %9, %10\n" \
            "v_alignbit_b32 %2, %2, %3, 7\n s_add_u32 %10, %10, %11\n v_alignbit_b32 %3, %3, %4, 9\n s_xor_b32 %11, %11, %8\n" \
            "v_alignbit_b32 %4, %4, %5, 11\n s_add_u32 %8, %8, %10\n v_alignbit_b32 %5, %5, %6, 13\n s_xor_b32 %9, %9, %11\n" \
            "v_alignbit_b32 %6, %6, %7, 15\n s_add_u32 %10, %10, %8\n v_alignbit_b32 %7, %7, %0, 17\n s_xor_b32 %11, %11, %9\n"
        if (MODE == 0) asm volatile(V8 V8 V8 V8 V8 V8 V8 V8 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
        if (MODE == 1) asm volatile(S8 S8 S8 S8 S8 S8 S8 S8 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
        if (MODE == 2) asm volatile(VS8 VS8 VS8 VS8 VS8 VS8 VS8 VS8 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7 ^ s0 ^ s1 ^ s2 ^ s3;
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
    const int iters = 20000;
    for (int bpc : {1, 2, 4, 8}) {           // blocks of 4 waves per CU -> 1, 2, 4, 8 waves per SIMD
        const int blocks = 256 * bpc;
        const float v = run<0>(d, blocks, iters), s = run<1>(d, blocks, iters), vs = run<2>(d, blocks, iters);
        // per wave: 64 instr of each kind per iteration
        const double clk = 2.4e6;            // cycles per ms at 2.4 GHz
        printf("waves/SIMD=%d  VALU-only %.2f ms (%.2f cyc/instr/SIMD)  SALU-only %.2f ms (%.2f)  interleaved V+S %.2f ms (%.2f per pair)\n",
               bpc, v, v * clk / (iters * 64.0 * bpc), s, s * clk / (iters * 64.0 * bpc), vs, vs * clk / (iters * 64.0 * bpc));
    }
    return 0;
}
