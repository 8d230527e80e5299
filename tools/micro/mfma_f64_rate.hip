// mfma_f64_rate.hip -- issue rate of v_mfma_f64_16x16x4_f64 on one SIMD: independent chains, 1..4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-mfma-vgpr-form -o mfma_f64_rate mfma_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void rate(double* out, int iters, double a0, double b0)
{
    double4_t acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = double4_t{ 0.0, 0.0, 0.0, 0.0 };
    double a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0.0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
void run(int waves_per_simd, double* out)
{
    const int iters = 20000, cus = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid(cus), block(256 * waves_per_simd);     // one workgroup per CU, waves_per_simd waves on each SIMD
    hipLaunchKernelGGL(rate<CHAINS>, grid, block, 0, 0, out, 100, 1.0, 1e-3);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate<CHAINS>, grid, block, 0, 0, out, iters, 1.0, 1e-3);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * CHAINS * waves_per_simd;
    std::printf("chains %d waves/SIMD %d: %.3f ms -> %.1f cycles per MFMA per SIMD at 2.4 GHz, %.1f TFLOP/s\n", CHAINS, waves_per_simd, ms,
                ms * 1e-3 * 2.4e9 / per_simd, per_simd * 1024.0 * 2048.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    double* out;
    hipMalloc(&out, 256 * 1024 * 8);
    run<1>(1, out); run<4>(1, out); run<4>(2, out); run<4>(4, out); run<8>(1, out);
    return 0;
}
