// Copy-bandwidth ceiling on this MI355X: what does a plain 16 B/lane streaming copy reach, and do
// non-temporal loads/stores or the grid shape matter?  (20 GB of traffic per run.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v4 __attribute__((ext_vector_type(4)));

template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void copyk(const v4* __restrict__ in, v4* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n; i += stride) {
        v4 r[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (i + u * 256 < n) r[u] = (MODE & 1) ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (i + u * 256 < n) { if (MODE & 2) __builtin_nontemporal_store(r[u], out + i + u * 256); else out[i + u * 256] = r[u]; }
    }
}
// one wave copies one 1000-byte "record" per step (16 B per lane at an 8-byte-aligned address), grid-stride over records
template <int MODE>
__global__ __launch_bounds__(256) void reck(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, size_t nrec)
{
    typedef unsigned u4 __attribute__((ext_vector_type(4), aligned(1)));
    const unsigned t = threadIdx.x & 63;
    const size_t nw = (size_t)gridDim.x * 4;
    for (size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < nrec; r += nw) {
        const size_t o = r * 1000 + (t < 62 ? 16 * t : 984);
        if (t < 63) {
            u4 v = (MODE & 1) ? __builtin_nontemporal_load((const u4*)(in + o)) : *(const u4*)(in + o);
            if (MODE & 2) __builtin_nontemporal_store(v, (u4*)(out + o)); else *(u4*)(out + o) = v;
        }
    }
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
    v4 *a, *b; (void)hipMalloc(&a, bytes + 64); (void)hipMalloc(&b, bytes + 64); (void)hipMemset(a, 1, bytes);
#define RUN(MODE, UNROLL, BLOCKS) { float ms = timeit([&] { hipLaunchKernelGGL((copyk<MODE, UNROLL>), dim3(BLOCKS), dim3(256), 0, 0, a, b, n); }); \
        printf("copy mode=%d unroll=%d blocks=%d: %.3f ms %.2f TB/s\n", MODE, UNROLL, BLOCKS, ms, 2.0 * bytes / ms / 1e9); }
    RUN(0, 1, 2048) RUN(0, 4, 2048) RUN(0, 4, 4096) RUN(0, 8, 2048) RUN(1, 4, 2048) RUN(2, 4, 2048) RUN(3, 4, 2048) RUN(3, 4, 8192) RUN(3, 1, 65536)
#define RUNR(MODE, BLOCKS) { float ms = timeit([&] { hipLaunchKernelGGL((reck<MODE>), dim3(BLOCKS), dim3(256), 0, 0, (const unsigned char*)a, (unsigned char*)b, (size_t)10000000); }); \
        printf("record-copy mode=%d blocks=%d: %.3f ms %.2f TB/s\n", MODE, BLOCKS, ms, 2.0 * bytes / ms / 1e9); }
    RUNR(0, 2048) RUNR(3, 2048) RUNR(2, 2048) RUNR(0, 1536) RUNR(3, 4096)
    return 0;
}
