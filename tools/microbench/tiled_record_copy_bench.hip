// Record-per-wave copy (as record_copy_bench) but each workgroup walks a contiguous TILE of records instead of
// jumping by the grid stride every step: does contiguity of a workgroup's successive requests matter?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4 __attribute__((ext_vector_type(4), aligned(1)));

template <int TILE, int UNROLL>
__global__ __launch_bounds__(256) void tiled(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, size_t nrec)
{
    const unsigned t = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned lo = t < 62 ? 16 * t : 984;
    const size_t ntiles = (nrec + TILE - 1) / TILE;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (unsigned s = 0; s < TILE / 4; s += UNROLL) {
            u4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const size_t r = tile * TILE + 4 * (s + u) + wave;
                if (t < 63 && r < nrec) v[u] = *(const u4*)(in + r * 1000 + lo);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const size_t r = tile * TILE + 4 * (s + u) + wave;
                if (t < 63 && r < nrec) *(u4*)(out + r * 1000 + lo) = v[u];
            }
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
    const size_t nrec = 10000000, bytes = nrec * 1000;
    unsigned char *a, *b; (void)hipMalloc(&a, bytes + 64); (void)hipMalloc(&b, bytes + 64); (void)hipMemset(a, 1, bytes);
#define RUN(TL, UN, BL) { float ms = timeit([&] { hipLaunchKernelGGL((tiled<TL, UN>), dim3(BL), dim3(256), 0, 0, a, b, nrec); }); \
        printf("tiled record copy tile=%d unroll=%d blocks=%d: %.3f ms %.2f TB/s\n", TL, UN, BL, ms, 2.0 * bytes / ms / 1e9); }
    RUN(16, 1, 2048) RUN(16, 4, 2048) RUN(64, 1, 2048) RUN(64, 2, 2048) RUN(64, 4, 2048) RUN(64, 4, 16384) RUN(256, 4, 2048) RUN(256, 2, 16384) RUN(1024, 4, 2048)
    return 0;
}
