// wave_sum11_f64 against eleven wave_sum_f64, bit for bit, on random doubles (one wave per block, 256 blocks).
#include "../../mcmc_amd/csrc/device_math.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
using namespace mc;
__global__ void k(const double* in, unsigned long long* out)
{
    const int lane = threadIdx.x, b = blockIdx.x;
    double v[11], w[11];
    for (int i = 0; i < 11; ++i) { v[i] = in[((size_t)b * 11 + i) * 64 + lane]; w[i] = wave_sum_f64(v[i]); }
    wave_sum11_f64(v, lane);
    for (int i = 0; i < 11; ++i) {
        out[(((size_t)b * 11 + i) * 64 + lane) * 2] = d2u(v[i]);
        out[(((size_t)b * 11 + i) * 64 + lane) * 2 + 1] = d2u(w[i]);
    }
}
int main()
{
    const int B = 256; const size_t N = (size_t)B * 11 * 64;
    double* h = (double*)malloc(N * 8);
    srand(7);
    for (size_t i = 0; i < N; ++i) {
        const double m = (double)rand() / RAND_MAX - 0.5;
        const int e = rand() % 40 - 20;
        h[i] = (i % 5 == 0) ? 0.0 : ldexp(m, e);
    }
    double* d; unsigned long long* o;
    if (hipMalloc(&d, N * 8) != hipSuccess || hipMalloc(&o, N * 16) != hipSuccess) return 2;
    if (hipMemcpy(d, h, N * 8, hipMemcpyHostToDevice) != hipSuccess) return 2;
    hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, d, o);
    unsigned long long* r = (unsigned long long*)malloc(N * 16);
    if (hipMemcpy(r, o, N * 16, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    size_t bad = 0;
    for (size_t i = 0; i < N; ++i) if (r[2 * i] != r[2 * i + 1]) { if (bad < 5) printf("mismatch at %zu: %016llx vs %016llx\n", i, r[2 * i], r[2 * i + 1]); ++bad; }
    printf("%zu of %zu results differ\n", bad, N);
    return bad ? 1 : 0;
}
