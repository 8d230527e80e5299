// Micro-benchmark: LDS atomic add throughput per CU (gfx950): u64 vs u32, returning or not, distinct / shared addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ void k(unsigned long long* out, int iters, int stride_elems, int active_lanes)
{
    __shared__ unsigned long long tab[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) tab[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int idx = ((threadIdx.x * stride_elems) & 4095);
    unsigned int acc = 0;
    if (lane < active_lanes) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                if (KIND == 0) atomicAdd(&tab[idx + u], (unsigned long long)(i + u));
                else if (KIND == 1) atomicAdd(reinterpret_cast<unsigned int*>(tab) + idx * 2 + u, (unsigned int)(i + u));
                else if (KIND == 2) acc += atomicAdd(reinterpret_cast<unsigned int*>(tab) + idx * 2 + u, (unsigned int)(i + u));
                else if (KIND == 3) acc += (unsigned int)atomicAdd(&tab[idx + u], (unsigned long long)(i + u));
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tab[5] + acc;
}
template <int KIND>
void run(const char* name, int waves, int stride, int active)
{
    unsigned long long* out; hipMalloc(&out, 256 * 8);
    const int iters = 500;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(waves * 64), 0, 0, out, iters, stride, active);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)iters * 10 * waves;   // wave-instructions per CU
    printf("%-14s waves=%2d stride=%2d lanes=%2d  %.3f ms  %.1f cycles per wave-instruction per CU (2.4 GHz)\n", name, waves, stride,
           active, ms, ms * 1e-3 * 2.4e9 / winst);
    hipFree(out);
}
int main()
{
    for (int active : {64, 32, 8}) {
        run<0>("ds_add_u64", 8, 10, active); run<1>("ds_add_u32", 8, 10, active);
        run<2>("ds_add_rtn_u32", 8, 10, active); run<3>("ds_add_rtn_u64", 8, 10, active);
    }
    run<0>("ds_add_u64", 8, 0, 64); run<1>("ds_add_u32", 8, 0, 64);   // all lanes one address
    run<0>("ds_add_u64", 8, 1, 64); run<1>("ds_add_u32", 8, 1, 64);   // consecutive
    run<0>("ds_add_u64", 16, 10, 64); run<1>("ds_add_u32", 16, 10, 64);
    return 0;
}
