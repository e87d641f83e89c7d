// Which feature of the record-per-wave access pattern costs bandwidth?  Pure copies, 10M records.
//   RECLEN 1000 (8-byte aligned records) vs 1024 (16-byte aligned, full waves)
//   UNROLL: records in flight per wave (loads issued before the stores)
//   blocks per CU
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4 __attribute__((ext_vector_type(4), aligned(1)));

template <int RECLEN, int UNROLL>
__global__ __launch_bounds__(256) void reck(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, size_t nrec)
{
    const unsigned t = threadIdx.x & 63;
    const size_t nw = (size_t)gridDim.x * 4;
    constexpr unsigned full = RECLEN / 16, tail = RECLEN % 16 ? RECLEN - 16 : 0;
    const unsigned lane_off = t < full ? 16 * t : tail;
    const bool active = t < full + (RECLEN % 16 ? 1 : 0);
    for (size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < nrec; r += nw * UNROLL) {
        u4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t rr = r + u * nw;
            if (active && rr < nrec) v[u] = *(const u4*)(in + rr * RECLEN + lane_off);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t rr = r + u * nw;
            if (active && rr < nrec) *(u4*)(out + rr * RECLEN + lane_off) = v[u];
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
    const size_t nrec = 10000000;
    unsigned char *a, *b; (void)hipMalloc(&a, nrec * 1024 + 64); (void)hipMalloc(&b, nrec * 1024 + 64); (void)hipMemset(a, 1, nrec * 1024);
#define RUN(RL, UN, BL) { float ms = timeit([&] { hipLaunchKernelGGL((reck<RL, UN>), dim3(BL), dim3(256), 0, 0, a, b, nrec); }); \
        printf("reclen=%d unroll=%d blocks=%d: %.3f ms %.2f TB/s\n", RL, UN, BL, ms, 2.0 * nrec * RL / ms / 1e9); }
    RUN(1000, 1, 2048) RUN(1000, 2, 2048) RUN(1000, 4, 2048) RUN(1000, 8, 2048) RUN(1000, 4, 4096) RUN(1000, 2, 1536)
    RUN(1024, 1, 2048) RUN(1024, 2, 2048) RUN(1024, 4, 2048) RUN(1024, 8, 2048)
    RUN(1008, 1, 2048) RUN(1008, 4, 2048)
    return 0;
}
