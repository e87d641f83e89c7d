// Third variant: only the LOADS are staged (flat, 16-byte aligned group loads into LDS, double buffered); each wave
// then stores its record straight to global memory, 16 B per lane at the record's own 8-byte alignment.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v4 __attribute__((ext_vector_type(4)));
typedef unsigned v2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4), aligned(1)));

template <int GR>   // records per group
__global__ __launch_bounds__(256) void loadstaged(const v4* __restrict__ in, unsigned char* __restrict__ out, size_t ngroups)
{
    constexpr unsigned CH = GR * 1000 / 16;     // 16-byte chunks per group (GR multiple of 2)
    __shared__ __attribute__((aligned(16))) unsigned char img[2][GR * 1000 + 16];
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned lo = lane < 62 ? 16 * lane : 984;
    size_t g = blockIdx.x;
    if (g < ngroups) for (unsigned i = tid; i < CH; i += 256) *(v4*)(img[0] + 16 * i) = in[g * CH + i];
    int buf = 0;
    for (; g < ngroups; g += gridDim.x) {
        const size_t gn = g + gridDim.x;
        if (gn < ngroups) for (unsigned i = tid; i < CH; i += 256) *(v4*)(img[buf ^ 1] + 16 * i) = in[gn * CH + i];
        __syncthreads();
        for (unsigned k = wave; k < GR; k += 4) {
            if (lane < 63) {
                const unsigned o = 1000 * k + lo;
                v2 a = *(const v2*)(img[buf] + o), b = *(const v2*)(img[buf] + o + 8);
                u4 v; v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y;
                *(u4*)(out + (g * GR + k) * 1000 + lo) = v;
            }
        }
        buf ^= 1;
        __syncthreads();
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
    v4* a; unsigned char* b; (void)hipMalloc(&a, bytes + 64); (void)hipMalloc(&b, bytes + 64); (void)hipMemset(a, 1, bytes);
#define RUN(GR, BL) { float ms = timeit([&] { hipLaunchKernelGGL((loadstaged<GR>), dim3(BL), dim3(256), 0, 0, a, b, nrec / GR); }); \
        printf("load-staged copy group=%d blocks=%d: %.3f ms %.2f TB/s\n", GR, BL, ms, 2.0 * bytes / ms / 1e9); }
    RUN(8, 1536) RUN(8, 4096) RUN(8, 16384) RUN(16, 1536) RUN(16, 4096) RUN(4, 2048) RUN(4, 16384)
    return 0;
}
