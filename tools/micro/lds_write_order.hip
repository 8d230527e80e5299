// Which lane's store lands when several lanes of one wave store to the same LDS address in one instruction?
// (Undefined by the ISA; the chain kernel's claim arbitration is correct either way and only its retry rate depends on it.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int mode)
{
    __shared__ unsigned char b[256];
    __shared__ unsigned int w[64];
    const int lane = threadIdx.x;
    b[lane] = 0xEE; w[lane & 63] = 0xEEEEEEEE;
    __syncthreads();
    // groups of lanes store to one address: group g = lane / mode... here: address = lane % mode
    const int a = lane % mode;
    b[a] = (unsigned char)lane;
    w[a] = (unsigned int)lane;
    __syncthreads();
    out[lane] = b[a];
    out[64 + lane] = (int)w[a];
}
int main()
{
    int* d; hipMalloc(&d, 128 * sizeof(int));
    int h[128];
    for (int mode : {1, 2, 8, 16, 32}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("address = lane %% %2d: byte store winner per address:", mode);
        for (int a = 0; a < (mode < 8 ? mode : 8); ++a) printf(" %d", h[a]);
        printf("   dword store winner:");
        for (int a = 0; a < (mode < 8 ? mode : 8); ++a) printf(" %d", h[64 + a]);
        printf("\n");
    }
    return 0;
}
