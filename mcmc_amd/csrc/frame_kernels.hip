// frame_kernels.hip -- per-frame kernels of the proposal-evaluation path.
//
//  K1  sp_moments_kernel   pixels -> per-superpixel int64 moment sums   (HBM-bound)
//  K1b local_fit_kernel    moments -> centred fit record + theta_i + A1 filter
//  K2  edge_prob_kernel    per CSR entry q_ij
//      segment_prob_kernel Segment(superPixels) for an explicit member list
//
// Reference behaviour restated (not code): OverSegmentation.cpp:145-162,198-222,
// MyMath.h:120-170,189-275,284-296, Graph.cpp:69-138, Problem.cpp:372-445.
#include "device_math.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace mc {

// ---------------------------------------------------------------------------
// K1: moment accumulation.
//
// HBM layout: disp int16[F][H][W], label uint16[F][H][W]; 4 bytes per pixel are
// read exactly once with 16-byte loads (8 pixels per lane per plane).  Threads
// of a workgroup sweep a band of rows; when 8*blockDim is a multiple of W every
// lane stays on the same 8-pixel-wide column strip from row to row, where the
// superpixel label changes only every ~25 rows.  Each lane therefore keeps two
// register-resident accumulator slots keyed by label (two, because a strip that
// straddles a vertical superpixel boundary alternates between two labels) and
// only touches LDS when a third label evicts one.  LDS holds one int64[S][10]
// table per workgroup (ds_add_u64); at the end the non-zero rows are added to
// HBM with 64-bit integer atomics.  All sums are integers, so the result is
// bit-identical for any band count, block size or arrival order.
// ---------------------------------------------------------------------------
struct Slot {
    uint32_t lab;
    int32_t n, su, sv, sb;
    int64_t suu, suv, svv, sub, svb, sbb;
};

MC_D void slot_reset(Slot& s, uint32_t lab)
{
    s.lab = lab; s.n = 0; s.su = 0; s.sv = 0; s.sb = 0;
    s.suu = 0; s.suv = 0; s.svv = 0; s.sub = 0; s.svb = 0; s.sbb = 0;
}

template <bool USE_LDS>
MC_D void slot_flush(const Slot& s, unsigned long long* tab, unsigned long long* gmom)
{
    if (s.n == 0) return;
    unsigned long long* t = (USE_LDS ? tab : gmom) + (size_t)s.lab * kMomStride;
    atomicAdd(t + 0, (unsigned long long)(long long)s.n);
    atomicAdd(t + 1, (unsigned long long)(long long)s.su);
    atomicAdd(t + 2, (unsigned long long)(long long)s.sv);
    atomicAdd(t + 3, (unsigned long long)s.suu);
    atomicAdd(t + 4, (unsigned long long)s.suv);
    atomicAdd(t + 5, (unsigned long long)s.svv);
    atomicAdd(t + 6, (unsigned long long)(long long)s.sb);
    atomicAdd(t + 7, (unsigned long long)s.sub);
    atomicAdd(t + 8, (unsigned long long)s.svb);
    atomicAdd(t + 9, (unsigned long long)s.sbb);
}

// 8 consecutive pixels of one row, all in superpixel s.lab
MC_D void slot_add8(Slot& s, int u0, int v, int s1, int sk, uint64_t s2)
{
    int g = 8 * u0 + 28;
    s.n += 8;
    s.su += g;
    s.sv += 8 * v;
    s.suu += (int64_t)(8 * u0 * u0 + 56 * u0 + 140);
    s.suv += (int64_t)(v * g);
    s.svv += (int64_t)(8 * v * v);
    s.sb += s1;
    s.sub += (int64_t)(u0 * s1 + sk);
    s.svb += (int64_t)(v * s1);
    s.sbb += (int64_t)s2;
}

MC_D void slot_add1(Slot& s, int u, int v, int b)
{
    s.n += 1;
    s.su += u;
    s.sv += v;
    s.suu += (int64_t)(u * u);
    s.suv += (int64_t)(u * v);
    s.svv += (int64_t)(v * v);
    s.sb += b;
    s.sub += (int64_t)(u * b);
    s.svb += (int64_t)(v * b);
    s.sbb += (int64_t)(b * b);
}

template <bool USE_LDS>
__global__ __launch_bounds__(1024) void sp_moments_kernel(MomentArgs a)
{
    extern __shared__ unsigned long long tab[];
    const int frame = a.first_frame + (int)blockIdx.y;
    const int band = (int)blockIdx.x;
    const int S = a.S[frame];
    const int T = (int)blockDim.x, tid = (int)threadIdx.x;
    const int W = a.W, H = a.H;
    unsigned long long* gmom = a.mom + (size_t)frame * a.s_max * kMomStride;

    if (USE_LDS) {
        for (int i = tid; i < S * kMomStride; i += T) tab[i] = 0ull;
        __syncthreads();
    }
    const int rows_per_band = (H + a.bands - 1) / a.bands;
    const int r0 = band * rows_per_band;
    const int r1 = min(H, r0 + rows_per_band);
    if (r0 < r1) {
        const size_t base = (size_t)frame * a.frame_stride + (size_t)r0 * W;
        const uint4* dp = reinterpret_cast<const uint4*>(a.disp + base);
        const uint4* lp = reinterpret_cast<const uint4*>(a.label + base);
        const int nchunks = (r1 - r0) * (W / 8);
        int c = tid;
        int v = r0 + (c * 8) / W;
        int u0 = (c * 8) % W;
        const int dv = (T * 8) / W, du = (T * 8) % W;

        Slot A, B;
        slot_reset(A, 0xFFFFFFFFu);
        slot_reset(B, 0xFFFFFFFFu);
        int lru = 0;  // slot to evict next: 0 = A, 1 = B

        for (; c < nchunks; c += T) {
            const uint4 d4 = dp[c];
            const uint4 l4 = lp[c];
            const uint32_t first = l4.x & 0xFFFFu;
            const uint32_t pat = first | (first << 16);
            const bool same8 = (l4.x == pat) & (l4.y == pat) & (l4.z == pat) & (l4.w == pat);
            if (same8) {
                if (first < (uint32_t)S) {
                    const int b0 = (int)(short)(d4.x & 0xFFFFu), b1 = (int)d4.x >> 16;
                    const int b2 = (int)(short)(d4.y & 0xFFFFu), b3 = (int)d4.y >> 16;
                    const int b4 = (int)(short)(d4.z & 0xFFFFu), b5 = (int)d4.z >> 16;
                    const int b6 = (int)(short)(d4.w & 0xFFFFu), b7 = (int)d4.w >> 16;
                    const int s1 = ((b0 + b1) + (b2 + b3)) + ((b4 + b5) + (b6 + b7));
                    const int sk = (b1 + 2 * b2 + 3 * b3) + (4 * b4 + 5 * b5) + (6 * b6 + 7 * b7);
                    // b*b <= 2^30 (b = -32768): pair sums fit u32, the total needs 64 bits
                    const uint32_t p01 = (uint32_t)(b0 * b0) + (uint32_t)(b1 * b1), p23 = (uint32_t)(b2 * b2) + (uint32_t)(b3 * b3);
                    const uint32_t p45 = (uint32_t)(b4 * b4) + (uint32_t)(b5 * b5), p67 = (uint32_t)(b6 * b6) + (uint32_t)(b7 * b7);
                    const uint64_t s2 = ((uint64_t)p01 + (uint64_t)p23) + ((uint64_t)p45 + (uint64_t)p67);
                    if (first == A.lab) { slot_add8(A, u0, v, s1, sk, s2); lru = 1; }
                    else if (first == B.lab) { slot_add8(B, u0, v, s1, sk, s2); lru = 0; }
                    else if (lru == 0) {
                        slot_flush<USE_LDS>(A, tab, gmom); slot_reset(A, first);
                        slot_add8(A, u0, v, s1, sk, s2); lru = 1;
                    } else {
                        slot_flush<USE_LDS>(B, tab, gmom); slot_reset(B, first);
                        slot_add8(B, u0, v, s1, sk, s2); lru = 0;
                    }
                }
            } else {
                const uint32_t lw[4] = { l4.x, l4.y, l4.z, l4.w };
                const uint32_t dw[4] = { d4.x, d4.y, d4.z, d4.w };
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t lk = (k & 1) ? (lw[k >> 1] >> 16) : (lw[k >> 1] & 0xFFFFu);
                    const int bk = (k & 1) ? ((int)dw[k >> 1] >> 16) : (int)(short)(dw[k >> 1] & 0xFFFFu);
                    if (lk < (uint32_t)S) {
                        if (lk == A.lab) { slot_add1(A, u0 + k, v, bk); lru = 1; }
                        else if (lk == B.lab) { slot_add1(B, u0 + k, v, bk); lru = 0; }
                        else if (lru == 0) {
                            slot_flush<USE_LDS>(A, tab, gmom); slot_reset(A, lk);
                            slot_add1(A, u0 + k, v, bk); lru = 1;
                        } else {
                            slot_flush<USE_LDS>(B, tab, gmom); slot_reset(B, lk);
                            slot_add1(B, u0 + k, v, bk); lru = 0;
                        }
                    }
                }
            }
            u0 += du; v += dv;
            if (u0 >= W) { u0 -= W; v += 1; }
        }
        slot_flush<USE_LDS>(A, tab, gmom);
        slot_flush<USE_LDS>(B, tab, gmom);
    }
    if (USE_LDS) {
        __syncthreads();
        if (a.bands == 1) {
            for (int i = tid; i < S * kMomStride; i += T) gmom[i] = tab[i];
        } else {
            for (int i = tid; i < S * kMomStride; i += T) {
                unsigned long long x = tab[i];
                if (x) atomicAdd(gmom + i, x);
            }
        }
    }
}

// Generic fallback for widths that are not a multiple of 8: one pixel per
// thread-iteration, global atomics per pixel run of one lane.  Correct for any
// shape; not the measured path.
__global__ void sp_moments_generic_kernel(MomentArgs a)
{
    const int frame = a.first_frame + (int)blockIdx.y;
    const int S = a.S[frame];
    const int W = a.W, H = a.H;
    unsigned long long* gmom = a.mom + (size_t)frame * a.s_max * kMomStride;
    const size_t base = (size_t)frame * a.frame_stride;
    const int rows_per_band = (H + a.bands - 1) / a.bands;
    const int r0 = (int)blockIdx.x * rows_per_band, r1 = min(H, r0 + rows_per_band);
    for (int v = r0; v < r1; ++v) {
        // each thread owns a contiguous span of the row
        const int span = (W + (int)blockDim.x - 1) / (int)blockDim.x;
        const int ub = (int)threadIdx.x * span, ue = min(W, ub + span);
        Slot A; slot_reset(A, 0xFFFFFFFFu);
        for (int u = ub; u < ue; ++u) {
            const uint32_t l = a.label[base + (size_t)v * W + u];
            const int b = a.disp[base + (size_t)v * W + u];
            if (l >= (uint32_t)S) continue;
            if (l != A.lab) { slot_flush<false>(A, nullptr, gmom); slot_reset(A, l); }
            slot_add1(A, u, v, b);
        }
        slot_flush<false>(A, nullptr, gmom);
    }
}

// ---------------------------------------------------------------------------
// K1b: local fit.  One thread per (frame, superpixel).
// ---------------------------------------------------------------------------
__global__ void local_fit_kernel(FitArgs a)
{
    const int frame = a.first_frame + (int)blockIdx.y;
    const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int S = a.S[frame];
    if (s >= S) return;
    const long long* m = reinterpret_cast<const long long*>(a.mom) + ((size_t)frame * a.s_max + s) * kMomStride;
    double* f = a.fit + ((size_t)frame * a.s_max + s) * kFitStride;
    const long long cx = a.W / 2, cy = a.H / 2;
    const long long n = m[0];
    const long long su = m[1] - cx * n, sv = m[2] - cy * n;
    const long long suu = m[3] - 2 * cx * m[1] + cx * cx * n;
    const long long suv = m[4] - cx * m[2] - cy * m[1] + cx * cy * n;
    const long long svv = m[5] - 2 * cy * m[2] + cy * cy * n;
    const long long sb = m[6];
    const long long sub = m[7] - cx * m[6];
    const long long svb = m[8] - cy * m[6];
    const __int128 du = (__int128)n * suu - (__int128)su * su;
    const __int128 dvv = (__int128)n * svv - (__int128)sv * sv;
    const bool ok = (n >= 10) && (du != 0) && (dvv != 0) && (sb != 0);
    a.valid[(size_t)frame * a.s_max + s] = ok ? 1 : 0;
    f[0] = (double)suu; f[1] = (double)suv; f[2] = (double)svv;
    f[3] = (double)su;  f[4] = (double)sv;  f[5] = (double)n;
    f[6] = (double)sub; f[7] = (double)svb; f[8] = (double)sb;
    double le = 0.0;
    Vec3 x = { 0.0, 0.0, 0.0 };
    if (ok) {
        Sym3 A = { f[0], f[1], f[2], f[3], f[4], f[5] };
        Vec3 r = { f[6], f[7], f[8] }, d;
        ldl3(A, r, x, d);
        double sse = (double)m[9] - ((x.x * r.x + x.y * r.y) + x.z * r.z);
        le = sse / (2.0 * a.var);
    }
    f[9] = x.x; f[10] = x.y; f[11] = x.z;
    a.local_err[(size_t)frame * a.s_max + s] = le;
}

// ---------------------------------------------------------------------------
// K2: edge probabilities.  One thread per directed CSR entry.
// ---------------------------------------------------------------------------
__global__ void edge_prob_kernel(EdgeArgs a)
{
    const int frame = a.first_frame + (int)blockIdx.y;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const int S = a.S[frame];
    const uint32_t E2 = a.E2[frame];
    if (e >= E2) return;
    const uint32_t* rowp = a.rowp + (size_t)frame * (a.s_max + 1);
    const uint32_t j = a.col[(size_t)frame * a.e2_max + e];
    // row of entry e: largest i with rowp[i] <= e
    int lo = 0, hi = S;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (rowp[mid] <= e) lo = mid; else hi = mid; }
    const int i = lo;
    const uint8_t* valid = a.valid + (size_t)frame * a.s_max;
    double* qo = a.q + (size_t)frame * a.e2_max + e;
    if (j >= (uint32_t)S || (int)j == i || !valid[i] || !valid[j]) { *qo = -1.0; return; }
    const double* fi = a.fit + ((size_t)frame * a.s_max + i) * kFitStride;
    const double* fj = a.fit + ((size_t)frame * a.s_max + j) * kFitStride;
    const double log_var = det_log(a.var);
    int inf_flag = 0;
    // together {i, j}
    Sym3 A = { fi[0] + fj[0], fi[1] + fj[1], fi[2] + fj[2], fi[3] + fj[3], fi[4] + fj[4], fi[5] + fj[5] };
    Vec3 r = { fi[6] + fj[6], fi[7] + fj[7], fi[8] + fj[8] }, x, d;
    ldl3(A, r, x, d);
    const double ei = member_quad(fi, x), ej = member_quad(fj, x);
    // wave-order sum of two members (same lane or butterfly of two non-zeros among zeros):
    // one IEEE addition either way, and a + b == b + a.
    const double esum = ei + ej;
    const double T = segment_logprob(esum / (2.0 * a.var), d, log_var, &inf_flag);
    // separate {i}, {j}: residual is exactly zero
    Sym3 Ai = { fi[0], fi[1], fi[2], fi[3], fi[4], fi[5] };
    Vec3 ri = { fi[6], fi[7], fi[8] }, xi, di;
    ldl3(Ai, ri, xi, di);
    Sym3 Aj = { fj[0], fj[1], fj[2], fj[3], fj[4], fj[5] };
    Vec3 rj = { fj[6], fj[7], fj[8] }, xj, dj;
    ldl3(Aj, rj, xj, dj);
    const double pi = segment_logprob(member_quad(fi, xi) / (2.0 * a.var), di, log_var, &inf_flag);
    const double pj = segment_logprob(member_quad(fj, xj) / (2.0 * a.var), dj, log_var, &inf_flag);  // quad == 0
    const double Sp = pi + pj;
    double lB, lS;
    if (T > Sp) { lB = T; lS = Sp; } else { lS = T; lB = Sp; }
    const double sum = lB + det_log(1.0 + det_exp(lS - lB));
    const double ratio = det_exp(T - sum);
    if (ratio != ratio) atomicOr(a.status + frame, 1);
    if (inf_flag) atomicOr(a.status + frame, 2);
    *qo = ratio;
}

// ---------------------------------------------------------------------------
// Segment(superPixels): one wavefront evaluates one explicit member set.
// mem[i] != 0 marks members (device byte array of S entries).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void segment_prob_kernel(SegArgs a)
{
    const int lane = lane_id();
    const int S = a.S;
    const double* fit = a.fit;
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    int cnt = 0;
    for (int i = lane; i < S; i += kWave) {
        if (!a.mem[i]) continue;
        const double* f = fit + (size_t)i * kFitStride;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] += f[k];
        cnt++;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = wave_sum_f64(acc[k]);  // exact integers: order-free
    cnt = wave_sum_i32(cnt);
    Sym3 A = { acc[0], acc[1], acc[2], acc[3], acc[4], acc[5] };
    Vec3 r = { acc[6], acc[7], acc[8] }, x = { 0, 0, 0 }, d = { 1, 1, 1 };
    double p = 0.0, err = 0.0;
    int inf_flag = 0;
    if (cnt > 0) {
        ldl3(A, r, x, d);
        double part = 0.0;
        for (int i = lane; i < S; i += kWave) {
            if (!a.mem[i]) continue;
            part += member_quad(fit + (size_t)i * kFitStride, x);
        }
        err = wave_sum_f64(part) / (2.0 * a.var);
        p = segment_logprob(err, d, det_log(a.var), &inf_flag);
    }
    if (lane == 0) {
        a.out[0] = p; a.out[1] = err; a.out[2] = x.x; a.out[3] = x.y; a.out[4] = x.z;
        a.out[5] = inf_flag ? 1.0 : 0.0;
    }
}

// ---------------------------------------------------------------------------
// Device self-tests of the shared math (parity of det_log/det_exp/division/Philox).
// ---------------------------------------------------------------------------
__global__ void selftest_math_kernel(const double* x, const double* y, int n, double* ol, double* oe, double* od)
{
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    ol[i] = det_log(x[i]);
    oe[i] = det_exp(x[i]);
    od[i] = x[i] / y[i];
}
__global__ void selftest_philox_kernel(const uint32_t* ck, uint32_t* out)
{
    u32x4 c = { ck[0], ck[1], ck[2], ck[3] };
    u32x4 r = philox4x32_10(c, ck[4], ck[5]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_moments(const MomentArgs& a, int n_frames, int threads, size_t lds_bytes, bool generic, hipStream_t st)
{
    dim3 grid(a.bands, n_frames);
    if (generic) {
        hipLaunchKernelGGL(sp_moments_generic_kernel, grid, dim3(256), 0, st, a);
    } else if (lds_bytes) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sp_moments_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        hipLaunchKernelGGL(sp_moments_kernel<true>, grid, dim3(threads), lds_bytes, st, a);
    } else {
        hipLaunchKernelGGL(sp_moments_kernel<false>, grid, dim3(threads), 0, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_local_fit(const FitArgs& a, int n_frames, hipStream_t st)
{
    dim3 grid((a.s_max + 127) / 128, n_frames);
    hipLaunchKernelGGL(local_fit_kernel, grid, dim3(128), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_edge_prob(const EdgeArgs& a, int n_frames, hipStream_t st)
{
    if (a.e2_max == 0) return hipSuccess;
    dim3 grid((a.e2_max + 127) / 128, n_frames);
    hipLaunchKernelGGL(edge_prob_kernel, grid, dim3(128), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_segment_prob(const SegArgs& a, hipStream_t st)
{
    hipLaunchKernelGGL(segment_prob_kernel, dim3(1), dim3(64), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_selftest_math(const double* x, const double* y, int n, double* ol, double* oe, double* od, hipStream_t st)
{
    hipLaunchKernelGGL(selftest_math_kernel, dim3((n + 127) / 128), dim3(128), 0, st, x, y, n, ol, oe, od);
    return hipGetLastError();
}
hipError_t launch_selftest_philox(const uint32_t* ck, uint32_t* out, hipStream_t st)
{
    hipLaunchKernelGGL(selftest_philox_kernel, dim3(1), dim3(1), 0, st, ck, out);
    return hipGetLastError();
}

}  // namespace mc
