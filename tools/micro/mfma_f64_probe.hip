// mfma_f64_probe.hip -- what v_mfma_f64_16x16x4_f64 computes, bit for bit: register layout and rounding order.
// One wave per trial; lane l supplies a = A[l % 16][l / 16], b = B[l / 16][l % 16], c[r] = C[4 * (l / 16) + r][l % 16]
// (the assumed layout; the analysis script checks it) and the raw result registers are written out.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_f64_probe mfma_f64_probe.hip ; run: ./mfma_f64_probe out.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void probe(const double* A, const double* B, const double* C, double* D)
{
    const int t = blockIdx.x, l = threadIdx.x;
    const double a = A[t * 64 + (l % 16) * 4 + l / 16];
    const double b = B[t * 64 + (l / 16) * 16 + l % 16];
    double4_t c;
    for (int r = 0; r < 4; ++r) c[r] = C[t * 256 + (4 * (l / 16) + r) * 16 + l % 16];
    const double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[t * 256 + l * 4 + r] = d[r];
}

int main(int argc, char** argv)
{
    const int T = 64;
    std::vector<double> A(T * 64), B(T * 64), C(T * 256), D(T * 256);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto val = [&](int spread) {
        const double m = (double)(rnd() >> 11) / 9007199254740992.0 + 0.5;       // [0.5, 1.5)
        const int e = (int)(rnd() % (2 * spread + 1)) - spread;
        return ((rnd() & 1) ? -m : m) * std::ldexp(1.0, e);
    };
    for (int t = 0; t < T; ++t) {
        const int spread = t < 16 ? 0 : (t < 32 ? 3 : (t < 48 ? 20 : 40));
        for (int k = 0; k < 64; ++k) { A[t * 64 + k] = val(spread); B[t * 64 + k] = val(spread); }
        for (int k = 0; k < 256; ++k) C[t * 256 + k] = (t % 4 == 0) ? 0.0 : val(spread);
    }
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(T), dim3(64), 0, 0, dA, dB, dC, dD);
    if (hipDeviceSynchronize() != hipSuccess) { std::fprintf(stderr, "kernel failed\n"); return 1; }
    hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
    FILE* f = std::fopen(argc > 1 ? argv[1] : "mfma_probe.bin", "wb");
    const int hdr[2] = { T, 0 };
    std::fwrite(hdr, sizeof hdr, 1, f);
    std::fwrite(A.data(), 8, A.size(), f); std::fwrite(B.data(), 8, B.size(), f);
    std::fwrite(C.data(), 8, C.size(), f); std::fwrite(D.data(), 8, D.size(), f);
    std::fclose(f);
    std::printf("wrote %d trials\n", T);
    return 0;
}
