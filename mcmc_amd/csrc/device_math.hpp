// device_math.hpp -- deterministic f64 arithmetic shared by every kernel.
//
// Bit-exact accept/reject between the GPU and a CPU run needs log/exp that are
// built from IEEE + - * / only (OCML and libm differ in the last ulp), 3x3
// solves with a fixed evaluation order, a counter-based RNG and fixed-shape
// wavefront reductions.  Compile with -ffp-contract=off: no FMA contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define MC_HD __host__ __device__ __forceinline__
#define MC_D __device__ __forceinline__

namespace mc {

constexpr int kWave = 64;
constexpr int kMomStride = 10;
constexpr int kFitStride = 12;
constexpr uint16_t kNoLabel = 0xFFFFu;
constexpr double kLog2Pi = 1.8378770664093453;

// ---- Philox4x32-10 ---------------------------------------------------------
constexpr uint32_t kPhiloxM0 = 0xD2511F53u, kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kPhiloxW0 = 0x9E3779B9u, kPhiloxW1 = 0xBB67AE85u;
constexpr uint32_t kPhiloxTag = 0x4D434D43u;

struct u32x4 { uint32_t x, y, z, w; };

MC_HD u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)kPhiloxM0 * c.x;
        uint64_t p1 = (uint64_t)kPhiloxM1 * c.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += kPhiloxW0; k1 += kPhiloxW1;
    }
    return c;
}

// ---- bit moves -------------------------------------------------------------
MC_HD uint64_t d2u(double x) { return __builtin_bit_cast(uint64_t, x); }
MC_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

// ---- det_log / det_exp -----------------------------------------------------
MC_HD double det_log(double x)
{
    if (x != x) return x;
    if (x < 0.0) return u2d(0x7FF8000000000000ull);
    if (x == 0.0) return u2d(0xFFF0000000000000ull);
    uint64_t u = d2u(x);
    if (u == 0x7FF0000000000000ull) return x;
    int e = 0;
    if ((u >> 52) == 0) { x = x * 18014398509481984.0; u = d2u(x); e = -54; }
    e += (int)(u >> 52) - 1023;
    double m = u2d((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > 1.41421356237309514547e+00) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    double lm = 2.0 * s + (2.0 * s) * (z * p);
    double de = (double)e;
    return de * 6.93147180369123816490e-01 + (lm + de * 1.90821492927058770002e-10);
}

MC_HD double det_exp(double x)
{
    if (x != x) return x;
    if (x > 709.782712893384) return u2d(0x7FF0000000000000ull);
    if (x < -745.2) return 0.0;
    double t = x * 1.44269504088896338700e+00;
    long long k = (long long)(t + (t < 0.0 ? -0.5 : 0.5));
    double dk = (double)k;
    double r = (x - dk * 6.93147180369123816490e-01) - dk * 1.90821492927058770002e-10;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    long long k1 = k / 2, k2 = k - k1;
    double s1 = u2d((uint64_t)(k1 + 1023) << 52);
    double s2 = u2d((uint64_t)(k2 + 1023) << 52);
    return (p * s1) * s2;
}

// ---- 3x3 SPD solve (LDL^T, no pivoting, fixed order) -----------------------
// a = {a11 a12 a22 a13 a23 a33}; unknown order (a, b, c) for rows [u v 1].
struct Sym3 { double a11, a12, a22, a13, a23, a33; };
struct Vec3 { double x, y, z; };

MC_HD void ldl3(const Sym3& a, const Vec3& r, Vec3& x, Vec3& d)
{
    double d1 = a.a11;
    double l21 = a.a12 / d1;
    double l31 = a.a13 / d1;
    double d2 = a.a22 - l21 * a.a12;
    double t32 = a.a23 - l31 * a.a12;
    double l32 = t32 / d2;
    double d3 = (a.a33 - l31 * a.a13) - l32 * t32;
    double y1 = r.x;
    double y2 = r.y - l21 * y1;
    double y3 = (r.z - l31 * y1) - l32 * y2;
    double z1 = y1 / d1, z2 = y2 / d2, z3 = y3 / d3;
    double x3 = z3;
    double x2 = z2 - l32 * x3;
    double x1 = (z1 - l21 * x2) - l31 * x3;
    x.x = x1; x.y = x2; x.z = x3;
    d.x = d1; d.y = d2; d.z = d3;
}

// Residual of one member against the fused plane: (theta-theta_i)' M_i (theta-theta_i).
// f points at the member's 12-double fit record.
template <typename P>
MC_HD double member_quad(const P f, const Vec3& th)
{
    double da = th.x - f[9], db = th.y - f[10], dc = th.z - f[11];
    return da * ((f[0] * da + f[1] * db) + f[3] * dc)
         + db * ((f[1] * da + f[2] * db) + f[4] * dc)
         + dc * ((f[3] * da + f[4] * db) + f[5] * dc);
}

// probabilityL_ from the error sum (already divided by 2 var) and the LDL pivots.
MC_HD double segment_logprob(double err, const Vec3& d, double log_var, int* inf_flag)
{
    double logdet = ((det_log(d.x) + det_log(d.y)) + det_log(d.z)) - 3.0 * log_var;
    double p = -err + 0.5 * (3.0 * kLog2Pi - logdet);
    if (p != p) p = 0.75;                 // Graph.cpp:134
    if (p - p != 0.0) *inf_flag = 1;      // Graph.cpp:137
    return p;
}

MC_HD uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// ---- wavefront helpers (64 lanes) ------------------------------------------
MC_D int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
MC_D int mbcnt(uint64_t m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
MC_D uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
MC_D int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Fixed-shape xor butterfly, offsets 1,2,4,8,16,32: every lane ends with the
// same bits, and a CPU loop over 64 partials reproduces them exactly.
MC_D double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) v = v + __shfl_xor(v, off, kWave);
    return v;
}
MC_D double wave_prod_f64(double v)
{
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) v = v * __shfl_xor(v, off, kWave);
    return v;
}
MC_D uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, off, kWave);
    return v;
}
MC_D int wave_sum_i32(int v)
{
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) v += __shfl_xor(v, off, kWave);
    return v;
}
MC_D int wave_min_i32(int v)
{
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) { int o = __shfl_xor(v, off, kWave); v = o < v ? o : v; }
    return v;
}

}  // namespace mc
