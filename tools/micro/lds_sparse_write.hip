// Micro-benchmark: what an LDS store costs the CU when few lanes are active (the eviction path of K1 writes 48-byte records
// from two or three lanes of a wave): ds_write_b128 / b64 / b32 with 64, 8, 2 active lanes, 16 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ void k(unsigned long long* out, int iters, int active)
{
    __shared__ __attribute__((aligned(16))) uint32_t buf[16 * 64 * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* base = buf + (wave * 64 + lane) * 16;
    typedef uint32_t u4 __attribute__((ext_vector_type(4), aligned(16)));
    typedef uint32_t u2 __attribute__((ext_vector_type(2), aligned(8)));
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (KIND == 0) *reinterpret_cast<volatile u4*>(base + 4 * (u & 3)) = u4{ (uint32_t)i, (uint32_t)u, 3u, 4u };
                else if (KIND == 1) *reinterpret_cast<volatile u2*>(base + 2 * (u & 7)) = u2{ (uint32_t)i, (uint32_t)u };
                else *reinterpret_cast<volatile uint32_t*>(base + (u & 15)) = (uint32_t)(i + u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = buf[5];
}
template <int KIND>
void run(const char* name, int active)
{
    unsigned long long* out; hipMalloc(&out, 256 * 8);
    const int iters = 2000, waves = 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(waves * 64), 0, 0, out, iters, active);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-14s active lanes %2d: %.1f cycles per wave-instruction and CU (2.4 GHz)\n", name, active, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * waves));
    hipFree(out);
}
int main()
{
    for (int a : {64, 8, 2}) { run<0>("ds_write_b128", a); run<1>("ds_write_b64", a); run<2>("ds_write_b32", a); }
    return 0;
}
