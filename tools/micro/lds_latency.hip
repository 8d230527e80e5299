// Dependent-operation latencies on gfx950, one wave per CU: LDS pointer chase (ds_read_b32, random
// per-lane addresses), ds_bpermute chain, DPP scan, LDS atomic-min + read back, ballot -> branch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

constexpr int N = 4096;   // dwords of LDS
constexpr int ITER = 2048;

template <int CTRL, int ROW_MASK>
__device__ uint32_t dpp_u32(uint32_t identity, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false); }
__device__ uint32_t scan_add(uint32_t v)
{
    v += dpp_u32<0x111, 0xf>(0u, v); v += dpp_u32<0x112, 0xf>(0u, v); v += dpp_u32<0x114, 0xf>(0u, v);
    v += dpp_u32<0x118, 0xf>(0u, v); v += dpp_u32<0x142, 0xa>(0u, v); v += dpp_u32<0x143, 0xc>(0u, v);
    return v;
}

__global__ __launch_bounds__(64) void k(const uint32_t* perm, unsigned long long* out, uint32_t* sink)
{
    __shared__ uint32_t lds[N];
    __shared__ uint16_t lds16[N];
    __shared__ uint8_t lds8[N];
    const int lane = threadIdx.x;
    for (int i = lane; i < N; i += 64) { lds[i] = perm[i]; lds16[i] = (uint16_t)perm[i]; lds8[i] = (uint8_t)(perm[i] & 63); }
    __syncthreads();
    unsigned long long t0, t1;
    uint32_t x = lane * 61 % N;
    // 1. dependent ds_read_b32
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) x = lds[x];
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[0] = t1 - t0;
    // 2. dependent ds_read_u16
    x = x % N;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) x = lds16[x];
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[1] = t1 - t0;
    // 3. dependent ds_bpermute
    uint32_t y = x + lane;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((y & 63u) * 4u), (int)(y * 5u + 1u));
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[2] = t1 - t0;
    // 4. dependent DPP scans
    uint32_t z = y;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) z = scan_add(z ^ (uint32_t)i);
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[3] = t1 - t0;
    // 5. LDS atomic min, fence, read back (dependent)
    uint32_t w = (z + lane) % N;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) {
        atomicMin(&lds[w], (uint32_t)lane);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        w = (lds[w] + w * 7u + 1u) % N;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[4] = t1 - t0;
    // 6. ballot -> uniform branch -> VALU
    uint32_t v = w;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) {
        const uint64_t m = __ballot((v & 1u) != 0);
        if (m & 0x10) v = v * 3u + 1u; else v = v + (uint32_t)__popcll(m);
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[5] = t1 - t0;
    // 7. dependent ds_read_u8 + ds_write_b8 to another address (store does not block)
    uint32_t u = v % N;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) { u = (lds8[u] * 61u + u + 1u) % N; lds8[(u * 3u) % N] = (uint8_t)(u & 63u); }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[6] = t1 - t0;
    // 8. f64 divide chain
    double d = 1.0 + (double)u;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) d = 4294967295.0 / (d + 1.0);
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[7] = t1 - t0;
    // 9. readlane chain (v_readlane -> s -> v_mov)
    uint32_t q = u;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) q = q + (uint32_t)__builtin_amdgcn_readlane((int)q, (int)(__builtin_amdgcn_readfirstlane((int)q) & 63));
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[8] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = x + y + z + w + v + u + (uint32_t)d + q;
}

int main()
{
    uint32_t* h = (uint32_t*)malloc(N * 4);
    srand(3);
    for (int i = 0; i < N; ++i) h[i] = (uint32_t)(rand() % N);
    uint32_t *dp, *ds; unsigned long long* dout;
    hipMalloc(&dp, N * 4); hipMalloc(&ds, 64 * 4 * 256); hipMalloc(&dout, 16 * 8);
    hipMemcpy(dp, h, N * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dp, dout, ds);
    unsigned long long o[16];
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    const char* names[] = { "ds_read_b32 chase", "ds_read_u16 chase", "ds_bpermute chain", "DPP scan_add (6 steps)", "atomicMin+read back",
                            "ballot+branch", "ds_read_u8 chase + store", "f64 divide", "readfirstlane+readlane chain" };
    for (int i = 0; i < 9; ++i) printf("%-32s %7.1f cycles per iteration\n", names[i], (double)o[i] / ITER);
    return 0;
}
