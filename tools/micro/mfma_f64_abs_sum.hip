// mfma_f64_abs_sum.hip -- the inner loop of the L1 pass (Path B): d = A x B (v_mfma_f64_16x16x4_f64, C = 0), acc += |d|,
// four B operands per A.  How many cycles per matrix instruction one wave reaches by itself, and 2 / 4 / 5 waves per SIMD,
// for three instruction orders: as the compiler schedules the plain loop (V0), the four products first and the sixteen
// additions after them (V1), the additions of a product issued behind the NEXT product (V2).
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o mfma_f64_abs_sum mfma_f64_abs_sum.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define MFMA(a, b) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, zero4, 0, 0, 0)
#define ADD4(acc, d) { acc[0] += __builtin_fabs(d[0]); acc[1] += __builtin_fabs(d[1]); acc[2] += __builtin_fabs(d[2]); acc[3] += __builtin_fabs(d[3]); }

template <int V>
__global__ void loop(double* out, const double* in, int iters)
{
    const double4_t zero4 = { 0.0, 0.0, 0.0, 0.0 };
    double4_t acc[4] = { zero4, zero4, zero4, zero4 };
    double b[4];
    for (int g = 0; g < 4; ++g) b[g] = in[threadIdx.x + 64 * g];
    double a = in[threadIdx.x + 256];
    if (V == 0) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { const double4_t d = MFMA(a, b[g]); ADD4(acc[g], d); }
            a += 1.0;
        }
    } else if (V == 1) {
        for (int i = 0; i < iters; ++i) {
            const double4_t d0 = MFMA(a, b[0]), d1 = MFMA(a, b[1]), d2 = MFMA(a, b[2]), d3 = MFMA(a, b[3]);
            __builtin_amdgcn_sched_barrier(0);
            ADD4(acc[0], d0); ADD4(acc[1], d1); ADD4(acc[2], d2); ADD4(acc[3], d3);
            __builtin_amdgcn_sched_barrier(0);
            a += 1.0;
        }
    } else {
        double4_t d3 = zero4;
        for (int i = 0; i < iters; ++i) {
            const double4_t d0 = MFMA(a, b[0]);
            __builtin_amdgcn_sched_barrier(0);
            ADD4(acc[3], d3);
            __builtin_amdgcn_sched_barrier(0);
            const double4_t d1 = MFMA(a, b[1]);
            __builtin_amdgcn_sched_barrier(0);
            ADD4(acc[0], d0);
            __builtin_amdgcn_sched_barrier(0);
            const double4_t d2 = MFMA(a, b[2]);
            __builtin_amdgcn_sched_barrier(0);
            ADD4(acc[1], d1);
            __builtin_amdgcn_sched_barrier(0);
            d3 = MFMA(a, b[3]);
            __builtin_amdgcn_sched_barrier(0);
            ADD4(acc[2], d2);
            __builtin_amdgcn_sched_barrier(0);
            a += 1.0;
        }
        ADD4(acc[3], d3);
    }
    double s = 0.0;
    for (int g = 0; g < 4; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
void run(int waves_per_simd, double* out, const double* in)
{
    const int iters = 5000, cus = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid(cus), block(256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd);
    const dim3 g2(waves_per_simd > 4 ? cus * 2 : cus), b2(waves_per_simd > 4 ? 256 * waves_per_simd / 2 : block.x);
    hipLaunchKernelGGL(loop<V>, g2, b2, 0, 0, out, in, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(loop<V>, g2, b2, 0, 0, out, in, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 4 * waves_per_simd;
    std::printf("order V%d, waves/SIMD %d: %.3f ms -> %.1f cycles per matrix instruction and SIMD at 2.4 GHz (%.1f per wave)\n", V, waves_per_simd, ms,
                ms * 1e-3 * 2.4e9 / per_simd, ms * 1e-3 * 2.4e9 / (iters * 4.0));
}

int main()
{
    double *out, *in;
    hipMalloc(&out, 512 * 2048 * 8); hipMalloc(&in, 4096 * 8);
    hipMemset(in, 0, 4096 * 8);
    for (int w : { 1, 2, 4, 6 }) { run<0>(w, out, in); run<1>(w, out, in); run<2>(w, out, in); }
    return 0;
}
