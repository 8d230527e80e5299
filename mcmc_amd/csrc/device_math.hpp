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
constexpr int kFitStride = 16;   // Suu Suv Svv Su Sv N | Sub Svb Sb | theta a b c | Sbb ssr 0 0
constexpr int kSumStride = 10;   // Suu Suv Svv Su Sv N Sub Svb Sb Sbb (exact, integer valued)
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

// ---- det_sqrt: Newton iteration from IEEE + * / only, identical on CPU and GPU ----
MC_HD double det_sqrt(double x)
{
    if (x != x) return x;
    if (x < 0.0) return u2d(0x7FF8000000000000ull);
    if (x == 0.0) return x;
    uint64_t u = d2u(x);
    if (u == 0x7FF0000000000000ull) return x;
    int e = 0;
    if ((u >> 52) == 0) { x = x * 18014398509481984.0 * 18014398509481984.0; u = d2u(x); e = -108; }
    e += (int)(u >> 52) - 1023;
    const int odd = e & 1;
    e -= odd;
    const double m = u2d((u & 0x000FFFFFFFFFFFFFull) | ((uint64_t)(1023 + odd) << 52));
    double y = (m + 2.0) / 3.0;
    // six Newton steps by definition; a step that returns its argument has reached the iteration's fixed point and the
    // remaining steps would return it again, so they are not run (same bits, four or five divisions instead of six)
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
        const double yn = 0.5 * (y + m / y);
        if (yn == y) break;
        y = yn;
    }
    return y * u2d((uint64_t)(e / 2 + 1023) << 52);
}

// ---- det_rsqrt: 1 / sqrt(x) for a normal positive x from IEEE + - * only (no division: a division is some
// 150 dependent cycles on the device, this is about as much in all), identical on CPU and GPU.  x = m 2^e with
// m in [1, 2) and e even or odd: a quadratic seed for 1 / sqrt(m) (relative error 3.6e-3), three Newton steps
// r <- r (1.5 - (m / 2) r^2) (error 1.5 e^2 each: 2e-5, 6e-10, below rounding), 1 / sqrt(2) for an odd e, and the
// power of two.  Good to a few units in the last place; used by the Jacobi rotations of the plane fit alone.
MC_HD double det_rsqrt(double x)
{
    const uint64_t u = d2u(x);
    int e = (int)(u >> 52) - 1023;
    const int odd = e & 1;
    e -= odd;
    const double m = u2d((u & 0x000FFFFFFFFFFFFFull) | (1023ull << 52));
    const double hm = 0.5 * m;
    double r = (0.14496475 * m - 0.72223657) * m + 1.57368075;
    r = r * (1.5 - hm * (r * r));
    r = r * (1.5 - hm * (r * r));
    r = r * (1.5 - hm * (r * r));
    if (odd) r = r * 0.70710678118654752440;
    return r * u2d((uint64_t)(1023 - e / 2) << 52);
}

// ---- 3x3 SPD solve (LDL^T, no pivoting, fixed order) -----------------------
// a = {a11 a12 a22 a13 a23 a33}; unknown order (a, b, c) for rows [u v 1].
struct Sym3 { double a11, a12, a22, a13, a23, a33; };
struct Vec3 { double x, y, z; };

// Returns r' A^-1 r from the factorisation (sum y_k^2 / d_k): the residual of the fit is b'b minus it.
MC_HD double ldl3(const Sym3& a, const Vec3& r, Vec3& x, Vec3& d)
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
    return (y1 * z1 + y2 * z2) + y3 * z3;
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

// probabilityL_ of a segment from its sums: m[10] exact (kSumStride order), R = sum of the members' ssr.
// 2 var * totalError = (Sbb - m' M^-1 m) - R: the members' factor errors at the fused mean
// (MyMath.h:247-252) written as residual at theta minus the members' own residuals.
template <typename P>
MC_HD double segment_from_sums(const P m, double R, double var, double log_var, Vec3& th, double& err, int* inf_flag)
{
    Sym3 A = { m[0], m[1], m[2], m[3], m[4], m[5] };
    Vec3 r = { m[6], m[7], m[8] }, d;
    const double quad = ldl3(A, r, th, d);
    err = ((m[9] - quad) - R) / (2.0 * var);
    return segment_logprob(err, d, log_var, inf_flag);
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
// The exchanges run on the DPP crossbar instead of ds_bpermute: after the xor-1
// and xor-2 steps (quad_perm) every lane of a quad holds the same value, so the
// xor-4 partner may be any lane of the other quad of the 8-group (row_half_mirror)
// and the xor-8 partner any lane of the other 8-group of the row (row_mirror);
// IEEE + and * are commutative, so the bits are those of the xor butterfly.  The
// last two steps combine the four row values: (r0 op r1) op (r2 op r3).
template <int CTRL>
MC_D double dpp_f64(double v)
{
    const uint64_t u = d2u(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, true);
    return u2d(((uint64_t)hi << 32) | lo);
}
MC_D double readlane_f64(double v, int src)
{
    const uint64_t u = d2u(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), src);
    return u2d(((uint64_t)hi << 32) | lo);
}
MC_D double wave_sum_f64(double v)
{
    v = v + dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v = v + dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v = v + dpp_f64<0x141>(v);   // row_half_mirror
    v = v + dpp_f64<0x140>(v);   // row_mirror
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return (r0 + r1) + (r2 + r3);
}
MC_D double wave_prod_f64(double v)
{
    v = v * dpp_f64<0xB1>(v);
    v = v * dpp_f64<0x4E>(v);
    v = v * dpp_f64<0x141>(v);
    v = v * dpp_f64<0x140>(v);
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return (r0 * r1) * (r2 * r3);
}
// ---- several sums at once ----
// wave_sum_f64 of up to sixteen values, bit for bit, for about half the instructions: a stage of the butterfly halves
// the number of registers instead of repeating itself in every lane.  Values a and b meet at a stage whose partner
// differs in lane bit k: lanes with the bit clear keep a and hand b over, lanes with it set keep b and hand a over, and
// own + received is the same addition the plain butterfly does in that lane (its partner holds the mirror image).  After
// the stages xor 1, 2, 4, 8 lane p of every row holds the row's sum of value p; two exchanges across rows give
// (r0 + r1) + (r2 + r3) everywhere, and value k is read from lane k.  The exchanges must pair lane l with l ^ 2^k exactly
// (the mirrors of wave_sum_f64 will not do: the partner has to hold the same value), so xor 4 is two rotations with
// complementary bank masks, xor 8 a rotation by 8, and the rows use ds_bpermute.
template <int CTRL, int BANK>
MC_D double dpp_f64_banks(double old, double v)
{
    const uint64_t u = d2u(v), o = d2u(old);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)o, (int)(uint32_t)u, CTRL, 0xf, BANK, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(o >> 32), (int)(uint32_t)(u >> 32), CTRL, 0xf, BANK, false);
    return u2d(((uint64_t)hi << 32) | lo);
}
template <int STAGE>
MC_D double xchg_f64(double v)
{
    if constexpr (STAGE == 0) return dpp_f64<0xB1>(v);                  // l ^ 1: quad_perm [1,0,3,2]
    else if constexpr (STAGE == 1) return dpp_f64<0x4E>(v);             // l ^ 2: quad_perm [2,3,0,1]
    else if constexpr (STAGE == 2)                                      // l ^ 4: lanes 4-7, 12-15 read l - 4 (row_ror:4),
        return dpp_f64_banks<0x12C, 0x5>(dpp_f64_banks<0x124, 0xA>(v, v), v);   // lanes 0-3, 8-11 read l + 4 (row_ror:12)
    else return dpp_f64<0x128>(v);                                      // l ^ 8: row_ror:8
}
// one stage: the pair (a, b) becomes one register
template <int STAGE>
MC_D double fold_f64(double a, double b, int lane)
{
    const bool hi = (lane >> STAGE) & 1;
    const double keep = hi ? b : a, send = hi ? a : b;
    return keep + xchg_f64<STAGE>(send);
}
// v[0..10] -> their wave sums, one per lane: lane l of every row of 16 lanes returns the sum of v[l & 15] (0 for 11..15)
MC_D double wave_sum11_lanes_f64(const double (&v)[11], int lane)
{
    const double w0 = fold_f64<0>(v[0], v[1], lane), w1 = fold_f64<0>(v[2], v[3], lane), w2 = fold_f64<0>(v[4], v[5], lane),
                 w3 = fold_f64<0>(v[6], v[7], lane), w4 = fold_f64<0>(v[8], v[9], lane), w5 = fold_f64<0>(v[10], 0.0, lane);
    const double u0 = fold_f64<1>(w0, w1, lane), u1 = fold_f64<1>(w2, w3, lane), u2 = fold_f64<1>(w4, w5, lane);
    const double t0 = fold_f64<2>(u0, u1, lane), t1 = fold_f64<2>(u2, 0.0, lane);
    double r = fold_f64<3>(t0, t1, lane);
    r = r + __shfl_xor(r, 16, kWave);
    r = r + __shfl_xor(r, 32, kWave);
    return r;
}
// v[0..10] -> their wave sums (every lane gets all eleven)
MC_D void wave_sum11_f64(double (&v)[11], int lane)
{
    const double r = wave_sum11_lanes_f64(v, lane);
#pragma unroll
    for (int k = 0; k < 11; ++k) v[k] = readlane_f64(r, k);
}
// every lane of a half of the wave (lanes 0..31, lanes 32..63) gets the value of lane K of its half: the LDS crossbar's
// swizzle in bit-mask mode (and 0, or K, xor 0), no memory access and no vector-pipe slot
template <int K>
MC_D double bcast_half_f64(double v)
{
    const uint64_t u = d2u(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(uint32_t)u, (K << 5));
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(uint32_t)(u >> 32), (K << 5));
    return u2d(((uint64_t)hi << 32) | lo);
}
MC_D uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, off, kWave);
    return v;
}
MC_D int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
           (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
MC_D int wave_min_i32(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

}  // namespace mc
