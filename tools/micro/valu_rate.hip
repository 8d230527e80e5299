// Micro-benchmark: issue rate of the integer VALU instructions K1 is made of (wave64, gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ void k(uint32_t* out, int iters)
{
    uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x12345678u, c = a + 77u, d = b + 99u;
    uint32_t e = a * 3, f = b * 5, g = c * 7, h = d * 11;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) {  // v_dot2_i32_i16 (VOP3P), 8 independent chains
                asm volatile("v_dot2_i32_i16 %0, %1, %2, %0\n v_dot2_i32_i16 %3, %1, %2, %3\n v_dot2_i32_i16 %4, %1, %2, %4\n v_dot2_i32_i16 %5, %1, %2, %5\n"
                             "v_dot2_i32_i16 %6, %1, %2, %6\n v_dot2_i32_i16 %7, %1, %2, %7\n v_dot2_i32_i16 %8, %1, %2, %8\n v_dot2_i32_i16 %9, %1, %2, %9"
                             : "+v"(c), "+v"(a), "+v"(b), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a), "+v"(b));
            } else if (KIND == 1) {  // v_pk_mul_lo_u16
                asm volatile("v_pk_mul_lo_u16 %0, %0, %1\n v_pk_mul_lo_u16 %2, %2, %1\n v_pk_mul_lo_u16 %3, %3, %1\n v_pk_mul_lo_u16 %4, %4, %1\n"
                             "v_pk_mul_lo_u16 %5, %5, %1\n v_pk_mul_lo_u16 %6, %6, %1\n v_pk_mul_lo_u16 %7, %7, %1\n v_pk_mul_lo_u16 %8, %8, %1"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a));
            } else if (KIND == 2) {  // v_add_u32
                asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n"
                             "v_add_u32 %5, %5, %1\n v_add_u32 %6, %6, %1\n v_add_u32 %7, %7, %1\n v_add_u32 %8, %8, %1"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a));
            } else if (KIND == 3) {  // v_and_b32 / v_xor mix
                asm volatile("v_xor_b32 %0, %0, %1\n v_and_b32 %2, %2, %1\n v_xor_b32 %3, %3, %1\n v_or_b32 %4, %4, %1\n"
                             "v_xor_b32 %5, %5, %1\n v_and_b32 %6, %6, %1\n v_xor_b32 %7, %7, %1\n v_or_b32 %8, %8, %1"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a));
            } else if (KIND == 4) {  // v_mad_i32_i24
                asm volatile("v_mad_i32_i24 %0, %0, %1, %2\n v_mad_i32_i24 %3, %3, %1, %2\n v_mad_i32_i24 %4, %4, %1, %2\n v_mad_i32_i24 %5, %5, %1, %2\n"
                             "v_mad_i32_i24 %6, %6, %1, %2\n v_mad_i32_i24 %7, %7, %1, %2\n v_mad_i32_i24 %8, %8, %1, %2\n v_mad_i32_i24 %9, %9, %1, %2"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a), "+v"(b));
            } else if (KIND == 5) {  // v_pk_min_u16 / v_pk_sub_u16
                asm volatile("v_pk_min_u16 %0, %0, %1\n v_pk_sub_u16 %2, %2, %1\n v_pk_min_u16 %3, %3, %1\n v_pk_sub_u16 %4, %4, %1\n"
                             "v_pk_min_u16 %5, %5, %1\n v_pk_sub_u16 %6, %6, %1\n v_pk_min_u16 %7, %7, %1\n v_pk_sub_u16 %8, %8, %1"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a));
            } else if (KIND == 7) {  // v_cmp_ne_u32 to vcc (independent)
                asm volatile("v_cmp_ne_u32 vcc, %0, %1\n v_cmp_ne_u32 vcc, %2, %1\n v_cmp_ne_u32 vcc, %3, %1\n v_cmp_ne_u32 vcc, %4, %1\n"
                             "v_cmp_ne_u32 vcc, %5, %1\n v_cmp_ne_u32 vcc, %6, %1\n v_cmp_ne_u32 vcc, %7, %1\n v_cmp_ne_u32 vcc, %8, %1"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a) :: "vcc");
            } else if (KIND == 8) {  // 64-bit add: v_add_co_u32 + v_addc_co_u32
                asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %1, vcc\n v_add_co_u32 %3, vcc, %3, %1\n v_addc_co_u32 %4, vcc, %4, %1, vcc\n"
                             "v_add_co_u32 %5, vcc, %5, %1\n v_addc_co_u32 %6, vcc, %6, %1, vcc\n v_add_co_u32 %7, vcc, %7, %1\n v_addc_co_u32 %8, vcc, %8, %1, vcc"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a) :: "vcc");
            } else if (KIND == 9) {  // v_cndmask with an SGPR pair mask (VOP3)
                asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %2, %2, %1, s[20:21]\n v_cndmask_b32 %3, %3, %1, s[20:21]\n v_cndmask_b32 %4, %4, %1, s[20:21]\n"
                             "v_cndmask_b32 %5, %5, %1, s[20:21]\n v_cndmask_b32 %6, %6, %1, s[20:21]\n v_cndmask_b32 %7, %7, %1, s[20:21]\n v_cndmask_b32 %8, %8, %1, s[20:21]"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a) :: "s20", "s21");
            } else if (KIND == 10) {  // v_add3_u32 / v_lshl_or_b32 / v_and_or_b32 / v_bfi_b32
                asm volatile("v_add3_u32 %0, %0, %1, %2\n v_lshl_or_b32 %3, %3, 16, %1\n v_and_or_b32 %4, %4, %1, %2\n v_bfi_b32 %5, %1, %5, %2\n"
                             "v_add3_u32 %6, %6, %1, %2\n v_lshl_or_b32 %7, %7, 16, %1\n v_and_or_b32 %8, %8, %1, %2\n v_bfi_b32 %9, %1, %9, %2"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a), "+v"(b));
            } else if (KIND == 11) {  // v_mad_u32_u24 / v_mul_u32_u24
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mul_u32_u24 %3, %3, %1\n v_mad_u32_u24 %4, %4, %1, %2\n v_mul_u32_u24 %5, %5, %1\n"
                             "v_mad_u32_u24 %6, %6, %1, %2\n v_mul_u32_u24 %7, %7, %1\n v_mad_u32_u24 %8, %8, %1, %2\n v_mul_u32_u24 %9, %9, %1"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a), "+v"(b));
            } else if (KIND == 12) {  // v_dot2c_i32_i16 (VOP2 accumulate in place)
                asm volatile("v_dot2c_i32_i16 %0, %1, %2\n v_dot2c_i32_i16 %3, %1, %2\n v_dot2c_i32_i16 %4, %1, %2\n v_dot2c_i32_i16 %5, %1, %2\n"
                             "v_dot2c_i32_i16 %6, %1, %2\n v_dot2c_i32_i16 %7, %1, %2\n v_dot2c_i32_i16 %8, %1, %2\n v_dot2c_i32_i16 %9, %1, %2"
                             : "+v"(c), "+v"(a), "+v"(b), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a), "+v"(b));
            } else if (KIND == 13) {  // v_dot4_i32_i8
                asm volatile("v_dot4_i32_i8 %0, %1, %2, %0\n v_dot4_i32_i8 %3, %1, %2, %3\n v_dot4_i32_i8 %4, %1, %2, %4\n v_dot4_i32_i8 %5, %1, %2, %5\n"
                             "v_dot4_i32_i8 %6, %1, %2, %6\n v_dot4_i32_i8 %7, %1, %2, %7\n v_dot4_i32_i8 %8, %1, %2, %8\n v_dot4_i32_i8 %9, %1, %2, %9"
                             : "+v"(c), "+v"(a), "+v"(b), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a), "+v"(b));
            } else if (KIND == 14) {  // v_pk_add_u16 / v_pk_lshlrev / v_perm_b32
                asm volatile("v_pk_add_u16 %0, %0, %1\n v_perm_b32 %2, %2, %1, %3\n v_pk_add_u16 %4, %4, %1\n v_perm_b32 %5, %5, %1, %3\n"
                             "v_pk_add_u16 %6, %6, %1\n v_perm_b32 %7, %7, %1, %3\n v_pk_add_u16 %8, %8, %1\n v_perm_b32 %9, %9, %1, %3"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a), "+v"(b));
            } else if (KIND == 6) {  // v_cndmask_b32
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n v_cndmask_b32 %4, %4, %1, vcc\n"
                             "v_cndmask_b32 %5, %5, %1, vcc\n v_cndmask_b32 %6, %6, %1, vcc\n v_cndmask_b32 %7, %7, %1, vcc\n v_cndmask_b32 %8, %8, %1, vcc"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(a) :: "vcc");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
template <int KIND>
void run(const char* name, int wpb)
{
    uint32_t* out; hipMalloc(&out, 256 * 16 * 1024 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(wpb * 64), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_wave = (double)iters * 16 * 8;
    const double waves_per_simd = wpb / 4.0;
    // cycles per instruction per SIMD at 2.4 GHz nominal
    printf("%-22s waves/CU=%2d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, wpb, ms,
           ms * 1e-3 * 2.4e9 / (inst_per_wave * waves_per_simd));
    hipFree(out);
}
int main()
{
    for (int wpb : {16}) {
        run<0>("v_dot2_i32_i16", wpb); run<1>("v_pk_mul_lo_u16", wpb); run<2>("v_add_u32", wpb);
        run<3>("v_xor/and/or", wpb); run<4>("v_mad_i32_i24", wpb); run<5>("v_pk_min/sub_u16", wpb); run<6>("v_cndmask_b32 vcc", wpb);
        run<7>("v_cmp_ne_u32 vcc", wpb); run<8>("v_add_co/addc", wpb); run<9>("v_cndmask sgpr", wpb);
        run<10>("add3/lshl_or/and_or/bfi", wpb); run<11>("v_mad/mul_u32_u24", wpb); run<12>("v_dot2c_i32_i16", wpb);
        run<13>("v_dot4_i32_i8", wpb); run<14>("v_pk_add_u16/v_perm", wpb);
    }
    return 0;
}
