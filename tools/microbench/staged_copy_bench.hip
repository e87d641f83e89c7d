// Would staging records through LDS so that global loads AND stores are flat, line-aligned 16 B/lane accesses
// beat the record-per-wave pattern?  Pure copy, 10M x 1000 B.  A workgroup takes 16 consecutive records
// (16000 B = 125 full 128-B lines): flat load -> LDS, per-wave record copy LDS -> LDS (8-byte aligned, the shape the
// canonicalize kernel would use), flat store.  Compare with record_copy_bench (4.4 ms) and copy_bench (3.7 ms).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v4 __attribute__((ext_vector_type(4)));
typedef unsigned v2 __attribute__((ext_vector_type(2)));

template <int DB>
__global__ __launch_bounds__(256) void staged(const v4* __restrict__ in, v4* __restrict__ out, size_t ngroups)
{
    __shared__ __attribute__((aligned(16))) unsigned char lin[DB][16000], lout[16000];
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    size_t g = blockIdx.x;
    if (DB == 2 && g < ngroups)
        for (unsigned i = tid; i < 1000; i += 256) *(v4*)(lin[0] + 16 * i) = in[g * 1000 + i];
    int buf = 0;
    for (; g < ngroups; g += gridDim.x) {
        if (DB == 1) {
            for (unsigned i = tid; i < 1000; i += 256) *(v4*)(lin[0] + 16 * i) = in[g * 1000 + i];
        } else {
            const size_t gn = g + gridDim.x;
            if (gn < ngroups)
                for (unsigned i = tid; i < 1000; i += 256) *(v4*)(lin[buf ^ 1] + 16 * i) = in[gn * 1000 + i];
        }
        __syncthreads();
        for (unsigned k = wave; k < 16; k += 4) {              // the "compute": record k, 16 B per lane, 8-byte aligned
            const unsigned o = 1000 * k + (lane < 62 ? 16 * lane : 984);
            if (lane < 63) {
                v2 a = *(const v2*)(lin[DB == 2 ? buf : 0] + o), b = *(const v2*)(lin[DB == 2 ? buf : 0] + o + 8);
                *(v2*)(lout + o) = a; *(v2*)(lout + o + 8) = b;
            }
        }
        __syncthreads();
        for (unsigned i = tid; i < 1000; i += 256) out[g * 1000 + i] = *(const v4*)(lout + 16 * i);
        if (DB == 2) buf ^= 1;
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
    const size_t nrec = 10000000, ngroups = nrec / 16, bytes = nrec * 1000;
    v4 *a, *b; (void)hipMalloc(&a, bytes + 64); (void)hipMalloc(&b, bytes + 64); (void)hipMemset(a, 1, bytes);
#define RUN(DB, BL) { float ms = timeit([&] { hipLaunchKernelGGL((staged<DB>), dim3(BL), dim3(256), 0, 0, a, b, ngroups); }); \
        printf("staged copy dbuf=%d blocks=%d: %.3f ms %.2f TB/s\n", DB, BL, ms, 2.0 * bytes / ms / 1e9); }
    RUN(1, 1024) RUN(1, 2048) RUN(1, 4096) RUN(1, 16384) RUN(2, 768) RUN(2, 1536) RUN(2, 4096) RUN(2, 16384)
    return 0;
}
