// Micro-benchmark: latency of DEPENDENT global loads under the occupancy of Path B's first half
// (1024 workgroups of 4 waves, 4 per CU): (a) a 34-lane contiguous row of 8-byte words out of a 578 KB table, the next row
// chosen by the data just read; (b) a 64-lane gather of 8-byte entries out of a 2.8 MB table.  Prints cycles per trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned long long* tab, int rows, int w64, int entries, int iters, unsigned long long* out, int lds_pad)
{
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lds_pad > 0 && threadIdx.x == 0) smem[lds_pad - 1] = 1;
    uint32_t r = (blockIdx.x * 4 + wave) * 2654435761u;
    unsigned long long acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        unsigned long long v = 0;
        if (MODE == 0) {
            const uint32_t row = r % (uint32_t)rows;
            if (lane < w64) v = tab[(size_t)row * w64 + lane];
        } else {
            const uint32_t e = (r + lane * 40503u) % (uint32_t)entries;
            v = tab[e];
        }
        // the next address depends on what came back
        unsigned long long s = v;
        s ^= __shfl_xor(s, 1); s ^= __shfl_xor(s, 2);
        acc += s;
        r = r * 1664525u + 1013904223u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { atomicAdd(out, t1 - t0); atomicAdd(out + 1, acc); }
}
int main()
{
    const int rows = 2124, w64 = 34, entries = 352082, iters = 200;
    std::vector<unsigned long long> h((size_t)entries);
    for (size_t i = 0; i < h.size(); ++i) h[i] = i * 0x9E3779B97F4A7C15ull;
    unsigned long long *tab, *out;
    hipMalloc(&tab, h.size() * 8); hipMemcpy(tab, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&out, 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    for (int blocks : {1, 256, 1024}) {
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(out, 0, 16);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 39 * 1024, 0, tab, rows, w64, entries, iters, out, 39 * 1024);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 39 * 1024, 0, tab, rows, w64, entries, iters, out, 39 * 1024);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long r[2]; hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
                if (rep) printf("%-28s workgroups %5d: %8.0f clock ticks per dependent trip, %7.1f ns per trip by the wall clock\n",
                                mode == 0 ? "row of 34 words (578 KB)" : "64-lane gather (2.8 MB)", blocks,
                                (double)r[0] / ((double)blocks * 4 * iters), ms * 1e6 / iters);
            }
        }
    }
    return 0;
}
