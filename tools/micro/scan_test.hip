// checks the wave64 DPP scans used by the chain kernel against a serial scan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
template <int CTRL, int ROW_MASK>
__device__ uint32_t dpp_u32(uint32_t identity, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false); }
__device__ uint32_t scan_add(uint32_t v)
{
    v += dpp_u32<0x111, 0xf>(0u, v); v += dpp_u32<0x112, 0xf>(0u, v); v += dpp_u32<0x114, 0xf>(0u, v);
    v += dpp_u32<0x118, 0xf>(0u, v); v += dpp_u32<0x142, 0xa>(0u, v); v += dpp_u32<0x143, 0xc>(0u, v);
    return v;
}
__device__ uint32_t scan_max(uint32_t v)
{
    v = max(v, dpp_u32<0x111, 0xf>(0u, v)); v = max(v, dpp_u32<0x112, 0xf>(0u, v)); v = max(v, dpp_u32<0x114, 0xf>(0u, v));
    v = max(v, dpp_u32<0x118, 0xf>(0u, v)); v = max(v, dpp_u32<0x142, 0xa>(0u, v)); v = max(v, dpp_u32<0x143, 0xc>(0u, v));
    return v;
}
__global__ void k(const uint32_t* in, uint32_t* oa, uint32_t* om)
{
    const uint32_t v = in[blockIdx.x * 64 + threadIdx.x];
    oa[blockIdx.x * 64 + threadIdx.x] = scan_add(v);
    om[blockIdx.x * 64 + threadIdx.x] = scan_max(v);
}
int main()
{
    const int N = 64 * 200;
    uint32_t *h = (uint32_t*)malloc(N * 4), *a = (uint32_t*)malloc(N * 4), *m = (uint32_t*)malloc(N * 4);
    srand(1);
    for (int i = 0; i < N; ++i) h[i] = (rand() % 3 == 0) ? 0 : rand() % 1000;
    uint32_t *di, *da, *dm; hipMalloc(&di, N * 4); hipMalloc(&da, N * 4); hipMalloc(&dm, N * 4);
    hipMemcpy(di, h, N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(N / 64), dim3(64), 0, 0, di, da, dm);
    hipMemcpy(a, da, N * 4, hipMemcpyDeviceToHost); hipMemcpy(m, dm, N * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < N / 64; ++b) {
        uint32_t s = 0, mx = 0;
        for (int l = 0; l < 64; ++l) {
            s += h[b * 64 + l]; mx = mx > h[b * 64 + l] ? mx : h[b * 64 + l];
            if (a[b * 64 + l] != s || m[b * 64 + l] != mx) { if (bad < 5) printf("block %d lane %d: add %u want %u, max %u want %u\n", b, l, a[b * 64 + l], s, m[b * 64 + l], mx); bad++; }
        }
    }
    printf("scan test: %d mismatches\n", bad);
    return bad != 0;
}
