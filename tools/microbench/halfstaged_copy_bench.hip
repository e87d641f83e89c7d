// Variant of staged_copy_bench: loads stay record-per-wave (16 B/lane at the record's own 8-byte alignment, as the
// canonicalize kernel's LDS-DMA ring does); only the STORES are staged: the four waves of a workgroup drop their
// records into a shared 4000-byte LDS image and the workgroup stores it flat (full 32-B sectors).  10M x 1000 B.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v4 __attribute__((ext_vector_type(4)));
typedef unsigned v2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4), aligned(1)));

template <int UNROLL>
__global__ __launch_bounds__(256) void halfstaged(const unsigned char* __restrict__ in, v4* __restrict__ out, size_t nrec)
{
    __shared__ __attribute__((aligned(16))) unsigned char img[2][4096];
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned lo = lane < 62 ? 16 * lane : 984;
    int buf = 0;
    for (size_t g = blockIdx.x; g * 4 < nrec; g += gridDim.x) {
        const size_t r = g * 4 + wave;
        if (lane < 63 && r < nrec) {
            u4 v = *(const u4*)(in + r * 1000 + lo);
            const unsigned o = 1000 * wave + lo;
            *(v2*)(img[buf] + o) = v2{ v.x, v.y }; *(v2*)(img[buf] + o + 8) = v2{ v.z, v.w };
        }
        __syncthreads();
        if (tid < 250) out[g * 250 + tid] = *(const v4*)(img[buf] + 16 * tid);
        buf ^= 1;
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
    unsigned char* a; v4* b; (void)hipMalloc(&a, bytes + 64); (void)hipMalloc(&b, bytes + 64); (void)hipMemset(a, 1, bytes);
#define RUN(BL) { float ms = timeit([&] { hipLaunchKernelGGL((halfstaged<1>), dim3(BL), dim3(256), 0, 0, a, b, nrec); }); \
        printf("half-staged copy blocks=%d: %.3f ms %.2f TB/s\n", BL, ms, 2.0 * bytes / ms / 1e9); }
    RUN(1536) RUN(2048) RUN(4096) RUN(16384) RUN(65536)
    return 0;
}
