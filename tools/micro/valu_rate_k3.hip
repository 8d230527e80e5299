// Micro-benchmark: issue rate of the instructions the chain kernel (K3) is made of (wave64, gfx950): the Philox
// multiply, f64 arithmetic and conversions, lane moves.  Eight independent chains per kind; 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ void k(uint32_t* out, int iters)
{
    uint32_t a = threadIdx.x * 2654435761u + 1u, b = a ^ 0x12345678u, c = a + 77u, d = b + 99u;
    double x0 = 1.0 + a * 1e-9, x1 = 1.0 + b * 1e-9, x2 = 1.0 + c * 1e-9, x3 = 1.0 + d * 1e-9;
    double x4 = x0 + 0.5, x5 = x1 + 0.5, x6 = x2 + 0.5, x7 = x3 + 0.5, y = 1.0000001;
    unsigned long long p0 = a, p1 = b, p2 = c, p3 = d, p4 = a + 1, p5 = b + 1, p6 = c + 1, p7 = d + 1;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) {        // v_mad_u64_u32 (the Philox multiply)
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, 0\n v_mad_u64_u32 %1, vcc, %8, %9, 0\n v_mad_u64_u32 %2, vcc, %8, %9, 0\n v_mad_u64_u32 %3, vcc, %8, %9, 0\n"
                             "v_mad_u64_u32 %4, vcc, %8, %9, 0\n v_mad_u64_u32 %5, vcc, %8, %9, 0\n v_mad_u64_u32 %6, vcc, %8, %9, 0\n v_mad_u64_u32 %7, vcc, %8, %9, 0"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 1) { // v_mul_lo_u32 / v_mul_hi_u32
                asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                             "v_mul_lo_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(a | 1u));
            } else if (KIND == 2) { // v_add_f64
                asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                             "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
            } else if (KIND == 3) { // v_mul_f64
                asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                             "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
            } else if (KIND == 4) { // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                             "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
            } else if (KIND == 5) { // v_cvt_f64_u32
                asm volatile("v_cvt_f64_u32 %0, %8\n v_cvt_f64_u32 %1, %8\n v_cvt_f64_u32 %2, %8\n v_cvt_f64_u32 %3, %8\n"
                             "v_cvt_f64_u32 %4, %8\n v_cvt_f64_u32 %5, %8\n v_cvt_f64_u32 %6, %8\n v_cvt_f64_u32 %7, %8"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (KIND == 6) { // v_rcp_f64
                asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                             "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (KIND == 7) { // v_readlane_b32 to distinct SGPRs
                asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n"
                             "v_readlane_b32 s24, %0, 5\n v_readlane_b32 s25, %1, 5\n v_readlane_b32 s26, %2, 5\n v_readlane_b32 s27, %3, 5"
                             :: "v"(a), "v"(b), "v"(c), "v"(d) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            } else if (KIND == 8) { // v_readlane + dependent VALU use (the spill reload pattern, with the hazard nops)
                asm volatile("v_readlane_b32 s20, %0, 3\n s_nop 1\n v_xor_b32 %1, s20, %1\n v_readlane_b32 s21, %0, 4\n s_nop 1\n v_xor_b32 %2, s21, %2\n"
                             "v_readlane_b32 s22, %0, 5\n s_nop 1\n v_xor_b32 %3, s22, %3\n v_readlane_b32 s23, %0, 6\n s_nop 1\n v_xor_b32 %1, s23, %1"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "s20", "s21", "s22", "s23");
            } else if (KIND == 9) { // v_cmp_lt_f64 to an SGPR pair + v_cndmask on it
                asm volatile("v_cmp_lt_f64 s[20:21], %0, %4\n v_cndmask_b32 %5, %5, %6, s[20:21]\n v_cmp_lt_f64 s[22:23], %1, %4\n v_cndmask_b32 %5, %5, %6, s[22:23]\n"
                             "v_cmp_lt_f64 s[24:25], %2, %4\n v_cndmask_b32 %5, %5, %6, s[24:25]\n v_cmp_lt_f64 s[26:27], %3, %4\n v_cndmask_b32 %5, %5, %6, s[26:27]"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y), "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            } else if (KIND == 10) { // DPP row_shr add (the wave scans / reductions)
                asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                             "v_add_u32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                             "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                             "v_add_u32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (KIND == 11) { // SALU: s_and_b64 / s_add_u32 mix
                asm volatile("s_add_u32 s20, s20, 1\n s_and_b64 s[22:23], s[22:23], exec\n s_add_u32 s21, s21, 1\n s_or_b64 s[24:25], s[24:25], exec\n"
                             "s_add_u32 s20, s20, 1\n s_and_b64 s[22:23], s[22:23], exec\n s_add_u32 s21, s21, 1\n s_or_b64 s[24:25], s[24:25], exec"
                             ::: "s20", "s21", "s22", "s23", "s24", "s25", "scc");
            } else if (KIND == 12) { // s_and_saveexec / s_or exec pair around one VALU (a predicated statement)
                asm volatile("s_mov_b64 vcc, exec\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %2, %2, %1\n s_or_b64 exec, exec, s[20:21]\n"
                             "s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %3, %3, %1\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "s20", "s21", "vcc", "scc");   // scc: the loop's own compare must not straddle this
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)(p0 ^ p1 ^ p2 ^ p3 ^ p4 ^ p5 ^ p6 ^ p7) ^
                                                 (uint32_t)__double_as_longlong(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
}
template <int KIND>
void run(const char* name, int wpb, int per_group)
{
    uint32_t* out; hipMalloc(&out, 256 * 16 * 1024 * 4);
    const int iters = 1000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(wpb * 64), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double groups_per_wave = (double)iters * 16;
    const double waves_per_simd = wpb / 4.0;
    printf("%-44s waves/SIMD=%d  %.3f ms  -> %.2f cycles per %s per SIMD (2.4 GHz)\n", name, wpb / 4, ms,
           ms * 1e-3 * 2.4e9 / (groups_per_wave * per_group * waves_per_simd), per_group == 8 ? "instruction" : "statement");
    fflush(stdout);
    hipFree(out);
}
int main()
{
    for (int wpb : {8, 16}) {
        run<0>("v_mad_u64_u32", wpb, 8); run<1>("v_mul_lo_u32 / v_mul_hi_u32", wpb, 8); run<2>("v_add_f64", wpb, 8);
        run<3>("v_mul_f64", wpb, 8); run<4>("v_fma_f64", wpb, 8); run<5>("v_cvt_f64_u32", wpb, 8); run<6>("v_rcp_f64", wpb, 8);
        run<7>("v_readlane_b32", wpb, 8); run<8>("v_readlane + s_nop 1 + v_xor (statement)", wpb, 4);
        run<9>("v_cmp_lt_f64 sgpr + v_cndmask (statement)", wpb, 4); run<10>("v_add_u32_dpp + s_nop 1 (statement)", wpb, 8);
        run<11>("s_add_u32 / s_and_b64", wpb, 8); run<12>("saveexec + v_add + restore (statement)", wpb, 4);
    }
    return 0;
}
