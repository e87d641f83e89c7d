// Ceilings of this pool for 10 GB streams: read-only (plain loads / LDS-DMA), write-only, and copies through both
// paths, over grid shapes.  Build: hipcc --offload-arch=gfx950 -O3 -o ceiling_bench ceiling_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v4 __attribute__((ext_vector_type(4)));

// MODE 0: read-only plain loads (sum to defeat DCE), 1: write-only, 2: copy plain, 3: read-only LDS-DMA, 4: copy LDS-DMA -> ds_read -> store
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void k(const v4* __restrict__ in, v4* __restrict__ out, size_t n, unsigned* sink)
{
    __shared__ __attribute__((aligned(16))) v4 lds[256 * UNROLL];
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    v4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i + 256 * (UNROLL - 1) < n; i += stride) {
        if (MODE == 0) {
            v4 r[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) r[u] = in[i + u * 256];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= r[u];
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) out[i + u * 256] = v4{ (unsigned)i, 1, 2, 3 };
        } else if (MODE == 2) {
            v4 r[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) r[u] = in[i + u * 256];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) out[i + u * 256] = r[u];
        } else {
            const unsigned wave = threadIdx.x >> 6;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds + u * 256 + wave * 64));
                const v4* src = in + i + u * 256;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory", "m0");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 3) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) acc ^= lds[u * 256 + threadIdx.x];
            } else {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) out[i + u * 256] = lds[u * 256 + threadIdx.x];
            }
        }
    }
    if ((MODE == 0 || MODE == 3) && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}
template <class F> float timeit(F f)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); f();
    (void)hipEventRecord(a);
    for (int i = 0; i < 10; ++i) f();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main()
{
    const size_t bytes = 10000000000ull, n = bytes / 16;
    v4 *a, *b; unsigned* sink;
    (void)hipMalloc(&a, bytes + 65536); (void)hipMalloc(&b, bytes + 65536); (void)hipMalloc(&sink, 4); (void)hipMemset(a, 1, bytes);
    const char* names[] = { "read plain", "write", "copy plain", "read LDS-DMA", "copy LDS-DMA" };
#define RUN(MODE, UNROLL, BLOCKS) { float ms = timeit([&] { hipLaunchKernelGGL((k<MODE, UNROLL>), dim3(BLOCKS), dim3(256), 0, 0, a, b, n, sink); }); \
        printf("%-14s unroll=%d blocks=%6d: %.3f ms  %.2f TB/s\n", names[MODE], UNROLL, BLOCKS, ms, ((MODE == 2 || MODE == 4) ? 2.0 : 1.0) * bytes / ms / 1e9); }
    RUN(0, 4, 2048) RUN(0, 8, 2048) RUN(0, 8, 8192) RUN(0, 4, 32768)
    RUN(1, 4, 2048) RUN(1, 8, 2048) RUN(1, 4, 32768)
    RUN(2, 4, 2048) RUN(2, 8, 2048) RUN(2, 8, 1024) RUN(2, 4, 8192) RUN(2, 4, 32768) RUN(2, 2, 65536)
    RUN(3, 4, 2048) RUN(3, 8, 2048) RUN(3, 4, 8192)
    RUN(4, 4, 2048) RUN(4, 8, 2048) RUN(4, 4, 8192) RUN(4, 2, 32768)
    return 0;
}
