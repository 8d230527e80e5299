// multiview_kernels.hip -- K4: Path B ("Price multiview") proposal evaluation.
//
//  mv_vertex_moments_kernel  per-vertex N, sum p, sum p p^T of the XYZ sub-cloud (one wave per vertex)
//  mv_fit_segment_kernel     fitPlaneSVD + computeProbability of an explicit vertex set
//  mv_chain_init_kernel      MultiviewSegmentation ctor: Bernoulli edges, connected components, fits
//  mv_chain_run_kernel       PatternedMatcher step loop: DFS growth, split / transfer proposal,
//                            two refits, four segment probabilities, accept decision
//
// Restates (behaviour, not code): MathUtils.cpp:17-81, SPGraph.cpp:89-104,
// MultiviewSegment.cpp:33-96, MultiviewSegmentation.cpp:11-274,
// PatternedMatcher.cpp:58-130.  The plane fit works on pooled second moments
// (3x3 symmetric eigen-solve in registers) instead of concatenating every point
// into a 3xN SVD; the L1 residual is not a moment function and takes one pass
// over the points of the affected segments (16 B per point, PCL PointXYZ stride).
//
// One workgroup (4 waves) per chain: the data-dependent DFS is serial per chain,
// the refits and the point pass use all 256 lanes.  All f64 results are produced
// by IEEE operations in a fixed, documented order
// (wave-order sums, det_exp / det_sqrt from + - * / only, Jacobi sweeps with one stated stopping rule).
#include <cstdlib>
#include "device_math.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace mc {

constexpr uint32_t kPhiloxTagMv = 0x4D564D43u;  // "MVMC"
constexpr int kMvWaves = 4;
constexpr int kMvThreads = kMvWaves * 64;
constexpr int kMvWavesFirstHalf = 4;            // waves per chain in the first half of a split iteration (4 or 8).  Measured at C4: eight waves halve a chain's
                                                // growth time when it has a CU to itself (88 k against 207 k cycles), but with four chains on a CU
                                                // the iteration takes as long as with four waves (2.25 against 2.15 ms per 10 steps): the CU's
                                                // throughput, not the chain's latency, is the limit
#ifndef MC_MV_GROWTH_WAVES
#define MC_MV_GROWTH_WAVES 8
#endif
constexpr int kMvGrowthWaves = MC_MV_GROWTH_WAVES;   // waves per chain of the growth-only kernel (MODE 4)
constexpr int kMvMaxV = 64 * 64;                // a bitmap row is one 64-bit word per lane (the fused kernel's LDS allows ~3.4 k vertices)

struct Sym33 { double a00, a01, a02, a11, a12, a22; };
struct Mat33 { double m[3][3]; };

// one Jacobi rotation on the (p, q) plane; k is the third index.  Left out when the off-diagonal element is below
// rounding against the diagonal (apq^2 <= 2^-106 |app aqq|); an angle too small to show in 1 + t^2 takes the short
// form (the CPU restatement used by the tests makes the same tests in the same order: same bits)
#define MV_ROT(APP, AQQ, APQ, AKP, AKQ, P, Q)                                          \
    do {                                                                                \
        const double apq = APQ;                                                         \
        const double dd = APP * AQQ;                                                    \
        if (!(apq * apq <= 0x1p-106 * (dd < 0.0 ? -dd : dd))) {                         \
            rotated = true;                                                             \
            const double theta = (AQQ - APP) / (2.0 * apq);                             \
            const double at = theta < 0.0 ? -theta : theta;                             \
            double t, c;                                                                \
            if (at >= 0x1p27) { t = 1.0 / (2.0 * at); c = 1.0; }                        \
            else {                                                                      \
                const double h2 = at * at + 1.0;                                        \
                t = 1.0 / (at + h2 * det_rsqrt(h2));                                    \
                c = det_rsqrt(t * t + 1.0);                                             \
            }                                                                           \
            if (theta < 0.0) t = -t;                                                    \
            const double s = t * c;                                                     \
            const double akp = AKP, akq = AKQ;                                          \
            APP = APP - t * apq;                                                        \
            AQQ = AQQ + t * apq;                                                        \
            APQ = 0.0;                                                                  \
            AKP = c * akp - s * akq;                                                    \
            AKQ = s * akp + c * akq;                                                    \
            _Pragma("unroll") for (int i = 0; i < 3; ++i) {                             \
                const double vip = V.m[i][P], viq = V.m[i][Q];                          \
                V.m[i][P] = c * vip - s * viq;                                          \
                V.m[i][Q] = s * vip + c * viq;                                          \
            }                                                                           \
        }                                                                               \
    } while (0)

// fitPlaneSVD on pooled moments m[10] -> plane[5] = nx ny nz d det(cov)
MC_HD void plane_from_moments(const double* m, double* plane)
{
    const double N = m[0];
    const double cx = m[1] / N, cy = m[2] / N, cz = m[3] / N;
    Sym33 A;
    A.a00 = m[4] - cx * m[1]; A.a01 = m[5] - cx * m[2]; A.a02 = m[6] - cx * m[3];
    A.a11 = m[7] - cy * m[2]; A.a12 = m[8] - cy * m[3]; A.a22 = m[9] - cz * m[3];
    Mat33 V;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V.m[i][j] = i == j ? 1.0 : 0.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 8; ++sweep) {
        bool rotated = false;
        MV_ROT(A.a00, A.a11, A.a01, A.a02, A.a12, 0, 1);   // (p,q) = (0,1), k = 2
        MV_ROT(A.a00, A.a22, A.a02, A.a01, A.a12, 0, 2);   // (0,2), k = 1
        MV_ROT(A.a11, A.a22, A.a12, A.a01, A.a02, 1, 2);   // (1,2), k = 0
        if (!rotated) break;                                // the first sweep that rotates nothing ends them
    }
    int k = 0;
    double best = A.a00;
    if (A.a11 < best) { k = 1; best = A.a11; }
    if (A.a22 < best) { k = 2; best = A.a22; }
    const double nx = k == 0 ? V.m[0][0] : (k == 1 ? V.m[0][1] : V.m[0][2]);
    const double ny = k == 0 ? V.m[1][0] : (k == 1 ? V.m[1][1] : V.m[1][2]);
    const double nz = k == 0 ? V.m[2][0] : (k == 1 ? V.m[2][1] : V.m[2][2]);
    plane[0] = nx; plane[1] = ny; plane[2] = nz;
    plane[3] = (nx * cx + ny * cy) + nz * cz;                 // + n.centroid (sic, MathUtils.cpp:41)
    plane[4] = ((A.a00 * A.a11) * A.a22) / 8.0;               // det(U S^2 U^T / 2), MathUtils.cpp:45
}

MC_HD double prob_from_error(double err, double det)
{
    double p = det_exp(-err / 2.0) * det_sqrt((2.0 * 3.14159265358979323846) * det);
    if (p != p) p = 0.0;                                      // MultiviewSegment.cpp:86-91
    return p;
}

// mean |n.p + d| over the points of vertex v, wave order (one wave, all lanes return it)
// The residual of one point: the chain fma(d, 1, fma(nz, z, fma(ny, y, fma(nx, x, 0)))), bit for bit what
// v_mfma_f64_16x16x4_f64 makes of a row of four products added to C = 0 (tools/micro/mfma_f64_probe.hip), so
// that the matrix-core form of the L1 pass (mv_errors_kernel) and this one agree.
MC_D double point_residual(double nx, double ny, double nz, double d0, double x, double y, double z)
{
    double d = __builtin_fma(x, nx, 0.0);
    d = __builtin_fma(y, ny, d);
    d = __builtin_fma(z, nz, d);
    return __builtin_fma(1.0, d0, d);
}
// 16 partial sums m = 4 r + g (point index mod 16), one per lane of a row of 16 lanes, combined over r, then g (see
// mv_errors_kernel: r sits in the registers of a lane there, g across the rows of the wave); every lane of the row
// ends with the same bits
MC_D double vertex_tree_sum(double v)
{
    v = v + __shfl_xor(v, 4, kWave);
    v = v + __shfl_xor(v, 8, kWave);
    v = v + __shfl_xor(v, 1, kWave);
    v = v + __shfl_xor(v, 2, kWave);
    return v;
}
// one vertex, one wave: the first 16 lanes work (the partial sums are defined by point index mod 16), every lane returns the mean
MC_D double vertex_error(const float4* pts, uint32_t b, uint32_t e, const double* pl, int lane)
{
    const double nx = pl[0], ny = pl[1], nz = pl[2], d0 = pl[3];
    double part = 0.0;
    if (lane < 16)
        for (uint32_t i = b + (uint32_t)lane; i < e; i += 16u) {
            const float4 p = pts[i];
            part += __builtin_fabs(point_residual(nx, ny, nz, d0, (double)p.x, (double)p.y, (double)p.z));
        }
    const double t = vertex_tree_sum(part);
    return readlane_f64(t, 0) / (double)(e - b);
}

__global__ __launch_bounds__(256) void mv_vertex_moments_kernel(MvGraph g, double* out)
{
    const int lane = (int)threadIdx.x & 63;
    const int v = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (v >= g.V) return;
    const uint32_t b = g.pt_ptr[v], e = g.pt_ptr[v + 1];
    double a[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) a[k] = 0.0;
    for (uint32_t i = b + (uint32_t)lane; i < e; i += kWave) {
        const float4 p = g.pts[i];
        const double x = p.x, y = p.y, z = p.z;
        a[0] += 1.0; a[1] += x; a[2] += y; a[3] += z;
        a[4] += x * x; a[5] += x * y; a[6] += x * z; a[7] += y * y; a[8] += y * z; a[9] += z * z;
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) a[k] = wave_sum_f64(a[k]);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 10; ++k) out[(size_t)v * 10 + k] = a[k];
}

// pooled moments of the vertices selected by pred(v), wave order by vertex id (one wave)
template <typename Pred>
MC_D int pool_moments(const double* vm, int V, int lane, Pred pred, double* m)
{
    double a[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) a[k] = 0.0;
    int n = 0;
    // two rows per trip, fetched whether or not they are members (the table is L2-resident): the loads of a
    // trip then do not wait for the membership test, and only members are added, in vertex order
    for (int v0 = lane; v0 < V; v0 += 2 * kWave) {
        const int v1 = v0 + kWave;
        const bool in0 = pred(v0), in1 = v1 < V && pred(v1);
        if (!__ballot(in0 || in1)) continue;
        const double* r0 = vm + (size_t)v0 * 10;
        const double* r1 = vm + (size_t)(v1 < V ? v1 : v0) * 10;
        double t0[10], t1[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) { t0[k] = r0[k]; t1[k] = r1[k]; }
        if (in0) {
            n++;
#pragma unroll
            for (int k = 0; k < 10; ++k) a[k] += t0[k];
        }
        if (in1) {
            n++;
#pragma unroll
            for (int k = 0; k < 10; ++k) a[k] += t1[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) m[k] = wave_sum_f64(a[k]);
    return wave_sum_i32(n);
}

// Segment(members): mem[v] != 0 marks members.  One workgroup of 4 waves.
__global__ __launch_bounds__(kMvThreads) void mv_fit_segment_kernel(MvGraph g, const uint8_t* mem, double* out /*7*/)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* errv = reinterpret_cast<double*>(smem);
    double* pl = errv + g.V;
    // wave index as a scalar: the loops over "this wave's" vertices then get scalar control flow
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    if (wave == 0) {
        double m[10], p5[5];
        pool_moments(g.vm, g.V, lane, [&](int v) { return mem[v] != 0; }, m);
        plane_from_moments(m, p5);
        if (lane == 0) for (int k = 0; k < 5; ++k) pl[k] = p5[k];
    }
    __syncthreads();
    for (int v = wave; v < g.V; v += kMvWaves) {
        if (!mem[v]) continue;
        const double ev = vertex_error(g.pts, g.pt_ptr[v], g.pt_ptr[v + 1], pl, lane);
        if (lane == 0) errv[v] = ev;
    }
    __syncthreads();
    if (wave == 0) {
        double part = 0.0;
        for (int v = lane; v < g.V; v += kWave) if (mem[v]) part += errv[v];
        const double err = wave_sum_f64(part);
        if (lane == 0) {
            for (int k = 0; k < 5; ++k) out[k] = pl[k];
            out[5] = prob_from_error(err, pl[4]);
            out[6] = err;
        }
    }
}

// ---------------------------------------------------------------------------
// chains
// ---------------------------------------------------------------------------
struct MvShared {
    uint32_t rbuf[256];
    double pl[4][5];
    double P[4];
    int have[4];        // P[k] came from the cache
    uint32_t start, orig, target, nE, nd;
    int kind;           // 0 transfer, 1 split, 2 skipped (no neighbour segment), 3 segment table full
    int nn;
    double a;
};

struct MvRng { uint32_t k0, k1, lo, hi; };

// draw d of the step, serial use by one thread: a block's four words are produced at its
// first use (d & 255 < 64) and kept in LDS for draws d + 64, d + 128, d + 192
MC_D uint32_t mv_draw(const MvRng& g, uint32_t d, uint32_t* rbuf)
{
    const uint32_t i = d & 255u;
    if (i < 64u) {
        u32x4 c = { g.lo, g.hi, (d >> 8) * 64u + i, kPhiloxTagMv };
        const u32x4 r = philox4x32_10(c, g.k0, g.k1);
        rbuf[i] = r.x; rbuf[64 + i] = r.y; rbuf[128 + i] = r.z; rbuf[192 + i] = r.w;
        return r.x;
    }
    return rbuf[i];
}
// random access (no LDS): any draw index
MC_D uint32_t mv_draw_at(const MvRng& g, uint32_t d)
{
    u32x4 c = { g.lo, g.hi, (d >> 8) * 64u + (d & 63u), kPhiloxTagMv };
    const u32x4 r = philox4x32_10(c, g.k0, g.k1);
    const uint32_t w = (d >> 6) & 3u;
    return w == 0 ? r.x : (w == 1 ? r.y : (w == 2 ? r.z : r.w));
}
MC_D double mv_unit(uint32_t x) { return (double)x / 4294967295.0; }
// lanes of one wave exchanging data through LDS: stops the compiler from forwarding or reordering across it
MC_D void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

// candidate buffers: 1 KB per wave.  Kept at four waves' worth unless the first half really runs eight: with 8 KB the
// per-chain state at C4 is 39 KB and only three chains fit a CU (first half 194 us instead of 132 us)
constexpr size_t kMvCandBytes = (kMvWavesFirstHalf > kMvWaves ? kMvWavesFirstHalf : kMvWaves) * 1024;
struct MvLds {
    uint16_t *cur, *own, *elist, *sv;
    uint8_t *inE, *nb;
    uint32_t* se;       // [V] claim words of the growth (0 = free)
    uint32_t* cand;     // [waves][256] words: candidate edges per wave (init: 128 entry indices; run: 128 (entry, neighbour) pairs)
    double* errv;       // [4][V]
    MvShared* sh;
};

// with_err: the [4][V] residual table of the fused kernel (the split halves keep it in HBM, which lets five
// workgroups share a CU instead of one)
__host__ __device__ inline size_t mv_lds_bytes(int V, bool with_err, size_t cand_bytes = kMvCandBytes)
{
    auto A = [](size_t x) { return (x + 15) & ~(size_t)15; };
    return A(sizeof(MvShared)) + cand_bytes + (with_err ? A((size_t)V * 32) : 0) + A((size_t)V * 4) + 4 * A((size_t)V * 2) + 2 * A((size_t)V);
}

MC_D MvLds mv_carve(unsigned char* smem, int V, bool with_err, size_t cand_bytes = kMvCandBytes)
{
    auto A = [](size_t x) { return (x + 15) & ~(size_t)15; };
    MvLds L;
    size_t o = 0;
    L.sh = reinterpret_cast<MvShared*>(smem + o); o += A(sizeof(MvShared));
    L.cand = reinterpret_cast<uint32_t*>(smem + o); o += cand_bytes;
    L.errv = reinterpret_cast<double*>(smem + o); if (with_err) o += A((size_t)V * 32);
    L.se = reinterpret_cast<uint32_t*>(smem + o); o += A((size_t)V * 4);
    L.cur = reinterpret_cast<uint16_t*>(smem + o); o += A((size_t)V * 2);
    L.own = reinterpret_cast<uint16_t*>(smem + o); o += A((size_t)V * 2);
    L.elist = reinterpret_cast<uint16_t*>(smem + o); o += A((size_t)V * 2);
    L.sv = reinterpret_cast<uint16_t*>(smem + o); o += A((size_t)V * 2);
    L.inE = smem + o; o += A((size_t)V);
    L.nb = smem + o;
    return L;
}

// MultiviewSegmentation ctor (MultiviewSegmentation.cpp:11-69)
__global__ __launch_bounds__(kMvThreads) void mv_chain_init_kernel(MvGraph g, MvChainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int V = g.V;
    MvLds L = mv_carve(smem, V, false);
    const int chain = (int)blockIdx.x;
    // wave index as a scalar: the loops over "this wave's" vertices then get scalar control flow
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    // the claim words' space holds the bitmap of the vertices that have a component (as in the chain step)
    const int W64 = g.W64;
    unsigned long long* LB = reinterpret_cast<unsigned long long*>(L.se);
    for (int v = tid; v < V; v += kMvThreads) L.cur[v] = 0xFFFFu;
    for (int w = tid; w < W64; w += kMvThreads) LB[w] = 0ull;
    __syncthreads();
    {
        // Bernoulli(q) per undirected edge (draw index = rank of the edge, so both directions agree and the
        // order of the search does not matter), then connected components numbered by their smallest vertex.
        // A component is grown breadth first by the four waves, as in the chain step: a wave takes frontier
        // vertices in turn; lane w looks at word w of the vertex's bitmap row (MvGraph::abm), candidates =
        // row & ~(vertices that have a component); they are collected and tried 64 at a time (draw <= threshold);
        // a vertex is claimed with an atomic OR on the bitmap.
        MvRng rg = { a.seed, a.chain_offset + (uint32_t)chain, 0xFFFFFFFFu, 0xFFFFFFFFu };
        uint2* cb = reinterpret_cast<uint2*>(L.cand + wave * 256);
        int ncomp = 0;
        for (int s0 = 0; s0 < V; ++s0) {
            if (L.cur[s0] != 0xFFFFu) continue;      // uniform over the workgroup (LDS value, read after a barrier)
            __syncthreads();                          // every wave has read cur[s0] before it is written
            if (tid == 0) { L.cur[s0] = (uint16_t)ncomp; LB[s0 >> 6] |= 1ull << (s0 & 63); L.elist[0] = (uint16_t)s0; L.sh->nE = 1u; }
            __syncthreads();
            auto try_edges = [&](uint2 ej, bool valid) {
                if (!valid) return;
                const uint2 t = g.et[ej.x];
                const uint32_t j = ej.y;
                const unsigned long long bit = 1ull << (j & 63u);
                if (LB[j >> 6] & bit) return;
                if (mv_draw_at(rg, t.x) <= t.y && (atomicOr(&LB[j >> 6], bit) & bit) == 0ull) {   // u <= q, inclusive (:23), as draw <= threshold
                    const uint32_t idx = atomicAdd(&L.sh->nE, 1u);
                    L.elist[idx] = (uint16_t)j; L.cur[j] = (uint16_t)ncomp;
                }
            };
            uint32_t head = 0, tail = 1;
            while (head < tail) {
                uint32_t cn = 0;
                uint32_t vi = head + (uint32_t)wave;
                unsigned long long aw = 0ull, naw = 0ull;
                uint32_t wb = 0u, nwb = 0u;
                auto fetch = [&](uint32_t at, unsigned long long& ra, uint32_t& rb) {
                    const size_t r = (size_t)uni((uint32_t)L.elist[at]) * (size_t)W64;
                    ra = 0ull; rb = 0u;
                    if (lane < W64) { ra = g.abm[r + lane]; rb = g.wbase[r + lane]; }
                };
                if (vi < tail) fetch(vi, aw, wb);
                while (vi < tail) {
                    const uint32_t vn = vi + (uint32_t)kMvWaves;
                    if (vn < tail) fetch(vn, naw, nwb);
                    unsigned long long c = aw;
                    if (c) c &= ~LB[lane];
                    for (;;) {
                        const uint64_t m = __ballot(c != 0ull);
                        if (m == 0) break;
                        if (c) {
                            const int b = __ffsll((unsigned long long)c) - 1;
                            cb[cn + (uint32_t)mbcnt(m)] = make_uint2(wb + (uint32_t)__popcll(aw & ((1ull << b) - 1ull)), (uint32_t)(lane * 64 + b));
                            c &= c - 1ull;
                        }
                        cn += (uint32_t)__popcll((unsigned long long)m);
                        wave_lds_fence();
                        if (cn >= (uint32_t)kWave) {
                            try_edges(cb[lane], true);
                            const uint32_t rest = cn - (uint32_t)kWave;
                            const uint2 keep = (uint32_t)lane < rest ? cb[kWave + lane] : make_uint2(0u, 0u);
                            wave_lds_fence();
                            if ((uint32_t)lane < rest) cb[lane] = keep;
                            cn = rest;
                            wave_lds_fence();
                        }
                    }
                    vi = vn; aw = naw; wb = nwb;
                }
                if (cn) try_edges(cb[(uint32_t)lane < cn ? lane : 0], (uint32_t)lane < cn);
                __syncthreads();
                head = tail; tail = uni(L.sh->nE);
                __syncthreads();
            }
            ncomp++;
        }
        if (tid == 0) L.sh->nn = ncomp;
    }
    __syncthreads();
    const int nseg = uni(L.sh->nn);
    uint16_t* gcur = a.cur + (size_t)chain * V;
    uint16_t* gown = a.own + (size_t)chain * V;
    for (int v = tid; v < V; v += kMvThreads) { gcur[v] = L.cur[v]; gown[v] = L.cur[v]; }
    double* planes = a.plane + (size_t)chain * a.seg_cap * 5;
    uint8_t* pvalid = a.segP_valid + (size_t)chain * a.seg_cap;
    for (int s = tid; s < a.seg_cap; s += kMvThreads) pvalid[s] = 0;
    // A wave pools the moments of its segments (wave, wave + 4, ...) one after the other -- the pooled sums are defined
    // by the wave's lane order -- and hands segment i's to lane i; the eigen-solves, a long chain of dependent f64
    // operations each, then run once for up to 64 segments, one per lane (the same scalar arithmetic per segment).
    for (int s0 = wave; s0 < nseg; s0 += kMvWaves * kWave) {
        double mine[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) mine[k] = k == 0 ? 1.0 : 0.0;      // lanes without a segment solve a harmless one
        for (int i = 0; i < kWave; ++i) {
            const int s = s0 + i * kMvWaves;
            if (s >= nseg) break;
            double m[10];
            pool_moments(g.vm, V, lane, [&](int v) { return L.cur[v] == (uint16_t)s; }, m);
            if (lane == i) {
#pragma unroll
                for (int k = 0; k < 10; ++k) mine[k] = m[k];
            }
        }
        double p5[5];
        plane_from_moments(mine, p5);
        const int s = s0 + lane * kMvWaves;
        if (s < nseg) for (int k = 0; k < 5; ++k) planes[(size_t)s * 5 + k] = p5[k];
    }
    if (tid == 0) {
        a.nseg[chain] = nseg; a.status[chain] = 0;
        a.n_accept[chain] = 0ull; a.n_split[chain] = 0ull; a.n_skip[chain] = 0ull; a.trace_hash[chain] = 0ull;
    }
}

// Exchange between the three kernels of a split step (MODE 1 -> mv_errors_kernel -> MODE 2)
struct MvExchange {
    MvShared* sh;       // [C]
    uint8_t* inE;       // [C][V] the grown set
    uint8_t* need;      // [V][C] bit k: the L1 residual of vertex v under plane k of this chain is wanted; zero between
                        // iterations (the first half writes the non-zero codes, mv_err_pairs_kernel clears them again)
    double* err;        // [C][4][V]
    uint16_t* plist;    // [queue places][64] the pairs (chain in block << 2 | plane) of an item, beside the item's place in the queue
    uint4* queue;       // [chain blocks * V * 64] items of 64 pairs: .x vertex | chain block << 16, .y item | pairs << 8,
                        // .z first point of the vertex, .w its point count
    unsigned long long* prof;   // option "profile": the L1 kernel's counters [16..23] (mcmc_mv_chains_get_profile), else null
    uint32_t* counters; // [0] queue length: counted up by mv_err_pairs_kernel, cleared by the second half
    // growth ahead (MODE 3 / 6): the set's flags [C][V] and the shared blocks [C] that a growth kernel (MODE 4) left for THIS
    // step -- the growth of a step needs nothing of the steps before it (the trials are tied to the edges and the step,
    // the segments it walks are the initial ones), so it runs a step ahead on a stream of its own; null: x.inE / x.sh
    const uint8_t* g_inE;
    const MvShared* g_sh;
};

// PatternedMatcher.cpp:58-130.  MODE 0: `steps` whole iterations per launch, one workgroup per chain, the L1
// pass included (every chain streams the points of its four segments through L2 by itself).
// MODE 1 / MODE 2: the first / second half of ONE iteration; between them mv_errors_kernel evaluates the
// L1 residuals of all chains vertex by vertex, so a vertex's points are read once for every chain that
// needs them instead of once per chain.  Same arithmetic in the same order, hence the same bits.
// MODE 4 + MODE 3: the first half as two kernels -- 4 grows the component and picks the move with NW = kMvGrowthWaves
// waves per chain inside 64 VGPRs (a chain's latency is what few chains wait for), 3 takes the set's flags and the
// shared block from it, moves the vertices of a split proposal, fits the planes and writes the chain back.
// NW: waves per chain (kMvWavesFirstHalf for the first half of a split iteration); the segment fits use waves 0..3
// whatever NW is.
template <int MODE, int NW = kMvWaves>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 8 : 4)))   // four chains (of four or eight waves) per CU
void mv_chain_run_kernel(MvGraph g, MvChainArgs a, MvExchange x)
{
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int V = g.V;
    MvLds L = mv_carve(smem, V, MODE == 0, (size_t)(NW > kMvWaves ? NW : kMvWaves) * 1024);
    MvShared* sh = L.sh;
    const int chain = (int)blockIdx.x;
    // wave index as a scalar: the loops over "this wave's" vertices then get scalar control flow
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    uint16_t* gcur = a.cur + (size_t)chain * V;
    uint16_t* gown = a.own + (size_t)chain * V;
    double* planes = a.plane + (size_t)chain * a.seg_cap * 5;
    double* segP = a.segP + (size_t)chain * a.seg_cap;
    uint8_t* pvalid = a.segP_valid + (size_t)chain * a.seg_cap;
    // the claim words' space holds two bitmaps here: the grown set and the union of the grown vertices' rows
    const int W64 = g.W64;
    unsigned long long* EB = reinterpret_cast<unsigned long long*>(L.se);
    unsigned long long* NBU = EB + W64;
    unsigned long long* NBS = reinterpret_cast<unsigned long long*>(L.nb);  // the neighbouring segments of the grown set, a bitmap by initial id (W64 words of the V bytes)
    unsigned long long* SM = reinterpret_cast<unsigned long long*>(L.sv);   // start's segment as a bitmap, per step
    for (int v = tid; v < V; v += NT) { L.cur[v] = gcur[v]; L.own[v] = gown[v]; L.inE[v] = 0; }
    for (int w = tid; w < 2 * W64; w += NT) EB[w] = 0ull;
    for (int w = tid; w < W64; w += NT) NBS[w] = 0ull;
    if (MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6) {
        // second half (2) / plane fits after a growth kernel (3): the grown set and the shared block come from the kernel
        // before; the second half also gets the residuals of the error kernel
        const uint8_t* in_inE = (MODE == 3 && x.g_inE) ? x.g_inE : x.inE;
        const MvShared* in_sh = (MODE == 3 && x.g_sh) ? x.g_sh : x.sh;
        for (int v = tid; v < V; v += NT) L.inE[v] = in_inE[(size_t)chain * V + v];
        const uint32_t* src = reinterpret_cast<const uint32_t*>(in_sh + chain);
        uint32_t* dst = reinterpret_cast<uint32_t*>(sh);
        for (int k = tid; k < (int)(sizeof(MvShared) / 4); k += NT) dst[k] = src[k];
        if (MODE == 2 || MODE == 5 || MODE == 6) {
            L.errv = x.err + (size_t)chain * 4 * V;
            if (chain == 0 && tid == 0) x.counters[0] = 0u;     // the error kernel has drained the queue: empty for the next iteration
        }
    }
    int nseg = a.nseg[chain], status = a.status[chain];
    unsigned long long n_accept = a.n_accept[chain], n_split = a.n_split[chain], n_skip = a.n_skip[chain];
    unsigned long long trace_hash = a.trace_hash[chain];
    unsigned long long ptime = (MODE == 1 && a.prof) ? __builtin_readcyclecounter() : 0ull;
    auto stamp = [&](int k) {
        if (MODE == 1 && a.prof && tid == 0) {
            const unsigned long long t1 = __builtin_readcyclecounter();
            atomicAdd(a.prof + k, t1 - ptime);
            ptime = t1;
        }
    };
    __syncthreads();
    stamp(0);

    for (unsigned long long ts = 0; ts < a.steps; ++ts) {
        const unsigned long long t = a.step0 + ts;
        // which half this pass is: the kernel's own, or -- MODE 5, two passes -- the second half of step step0, then
        // the first half of step step0 + 1
        // (MODE 6: ... then the plane fits of step step0 + 1 on the set a growth kernel has left)
        const int em = MODE == 5 ? (ts == 0 ? 2 : 1) : (MODE == 6 ? (ts == 0 ? 2 : 3) : MODE);
        if (MODE == 6 && ts == 1) {
            for (int v = tid; v < V; v += NT) L.inE[v] = x.g_inE[(size_t)chain * V + v];
            const uint32_t* src = reinterpret_cast<const uint32_t*>(x.g_sh + chain);
            uint32_t* dst = reinterpret_cast<uint32_t*>(sh);
            for (int k = tid; k < (int)(sizeof(MvShared) / 4); k += NT) dst[k] = src[k];
            __syncthreads();
        }
        if (MODE == 5 && ts == 1) {
            // the set of the step just finished came with the kernel's prologue: the first half starts from cleared flags
            for (int v = tid; v < V; v += NT) L.inE[v] = 0;
            __syncthreads();
        }
        if (status != 0) {
            if (em != 2 && em != 4 && tid == 0 && a.trace) { a.trace[(size_t)chain * a.trace_stride + a.trace_first + ts] = 0xFF; if (a.trace_a) a.trace_a[(size_t)chain * a.trace_stride + a.trace_first + ts] = 0.0; }
            continue;
        }
        MvRng rg = { a.seed, a.chain_offset + (uint32_t)chain, (uint32_t)t, (uint32_t)(t >> 32) };
        // ---- phase 1a: start vertex, growth, move type ----
        // getNewConnectedSet (MultiviewSegmentation.cpp:207-257) is a recursive walk with one trial per
        // same-segment neighbour that is not yet in the set.  The trial of an undirected edge owns draw
        // 16 + rank(edge) of the step, so the set is the component of `start` among the same-segment edges
        // whose trial succeeds whatever the order of the walk: the four waves grow it breadth first, each
        // wave taking frontier vertices in turn, collecting the candidate edges of 64 out-edges at a time and
        // trying them 64 at once (the Philox call is the expensive part; candidates are ~10 % of the edges).
        if (em != 2 && em != 3) {
            const uint32_t start = mv_draw_at(rg, 0u) % (uint32_t)V;
            const uint16_t cs = L.cur[start];
            if (tid == 0) { L.inE[start] = 1; EB[start >> 6] = 1ull << (start & 63u); L.elist[0] = (uint16_t)start; sh->nE = 1u; }
            // The scan works on bitmap rows (MvGraph::abm): lane w of a wave holds word w of "same initial segment
            // as start" (S) and looks at word w of a frontier vertex's row, so one trip of a wave covers all 64 x 64
            // possible neighbours: candidates = row & S & ~set.  The words of the rows also pile up in U, from
            // which getNeighborSegments is answered after the growth.
            for (int bw = wave; bw < W64; bw += NW) {
                const int j = bw * 64 + lane;
                const unsigned long long m = __ballot(j < V && L.cur[j < V ? j : 0] == cs);
                if (lane == 0) SM[bw] = m;
            }
            __syncthreads();
            const unsigned long long S = lane < W64 ? SM[lane] : 0ull;
            unsigned long long U = 0ull;
            uint2* cb = reinterpret_cast<uint2*>(L.cand + wave * 256);
            auto try_edges = [&](uint2 ej, bool valid) {
                if (!valid) return;
                const uint2 t = g.et[ej.x];          // the trial's draw index and threshold: the one global load of a trial
                const uint32_t j = ej.y;
                if (L.inE[j]) return;
                // u <= q (inclusive, :23) as draw <= threshold; the first to set j's bit appends it
                if (mv_draw_at(rg, 16u + t.x) <= t.y) {
                    const unsigned long long bit = 1ull << (j & 63u);
                    if ((atomicOr(&EB[j >> 6], bit) & bit) == 0ull) {
                        const uint32_t idx = atomicAdd(&sh->nE, 1u);
                        L.elist[idx] = (uint16_t)j; L.inE[j] = 1;
                    }
                }
            };
            uint32_t head = 0, tail = 1;
            stamp(1);
            // coarse timers of wave 0 (option "profile"): a handful of clock reads per level and per batch of trials -- a
            // clock read is a scalar-memory round trip, so nothing shorter than that is bracketed
            const bool gp = MODE == 1 && a.prof != nullptr;
            unsigned long long g_scan = 0, g_batch = 0, g_last = 0, g_bar = 0, g_nb = 0, g_trips = 0;
            const unsigned long long tg0 = gp ? __builtin_readcyclecounter() : 0ull;
            while (head < tail) {
                if (MODE == 1 && a.prof && tid == 0) atomicAdd(a.prof + 8, 1ull);
                const unsigned long long tl0 = gp ? __builtin_readcyclecounter() : 0ull;
                uint32_t cn = 0;
                // A wave's frontier vertices in turn, kChunk at a time: the rows of a chunk are all asked for before the first
                // is scanned, so the wave waits for one round trip per chunk instead of one per vertex.  Every lane loads
                // every time (vertex and word clamped, masked at the use): a load under a condition is merged with its
                // "else" value at once, and that merge waits for it.
#ifndef MC_MV_CHUNK
#define MC_MV_CHUNK 4
#endif
                constexpr int kChunk = MC_MV_CHUNK;
                const uint32_t lw = (uint32_t)(lane < W64 ? lane : 0);
                for (uint32_t v0 = head + (uint32_t)wave; v0 < tail; v0 += (uint32_t)(NW * kChunk)) {
                    unsigned long long ra[kChunk];
                    uint32_t rb[kChunk];
#pragma unroll
                    for (int k = 0; k < kChunk; ++k) {
                        const uint32_t at = min(v0 + (uint32_t)(k * NW), tail - 1u);
                        const size_t r = (size_t)uni((uint32_t)L.elist[at]) * (size_t)W64 + lw;
                        ra[k] = g.abm[r]; rb[k] = g.wbase[r];
                    }
#pragma unroll
                    for (int k = 0; k < kChunk; ++k) {
                        if (v0 + (uint32_t)(k * NW) < tail) {      // wave-uniform
                            const unsigned long long aw = lane < W64 ? ra[k] : 0ull;
                            const uint32_t wb = rb[k];
                            U |= aw;
                            unsigned long long c = aw & S;
                            if (c) c &= ~EB[lane];
                            // one candidate per lane and round: its entry index is the word's first plus the bits below it
                            for (;;) {
                                const uint64_t m = __ballot(c != 0ull);
                                if (m == 0) break;
                                if (c) {
                                    const int b = __ffsll((unsigned long long)c) - 1;
                                    cb[cn + (uint32_t)mbcnt(m)] = make_uint2(wb + (uint32_t)__popcll(aw & ((1ull << b) - 1ull)), (uint32_t)(lane * 64 + b));
                                    c &= c - 1ull;
                                }
                                cn += (uint32_t)__popcll((unsigned long long)m);
                                wave_lds_fence();
                                if (cn >= (uint32_t)kWave) {
                                    const unsigned long long tb0 = gp ? __builtin_readcyclecounter() : 0ull;
                                    try_edges(cb[lane], true);
                                    if (gp) { g_batch += __builtin_readcyclecounter() - tb0; g_nb++; }
                                    const uint32_t rest = cn - (uint32_t)kWave;
                                    const uint2 keep = (uint32_t)lane < rest ? cb[kWave + lane] : make_uint2(0u, 0u);
                                    wave_lds_fence();
                                    if ((uint32_t)lane < rest) cb[lane] = keep;
                                    cn = rest;
                                    wave_lds_fence();
                                }
                            }
                            if (gp) g_trips++;
                        }
                    }
                }
                const unsigned long long tl1 = gp ? __builtin_readcyclecounter() : 0ull;
                if (cn) try_edges(cb[(uint32_t)lane < cn ? lane : 0], (uint32_t)lane < cn);
                const unsigned long long tl2 = gp ? __builtin_readcyclecounter() : 0ull;
                __syncthreads();
                head = tail; tail = uni(sh->nE);
                __syncthreads();
                if (gp) { g_scan += tl1 - tl0; g_last += tl2 - tl1; g_bar += __builtin_readcyclecounter() - tl2; }
            }
            if (gp && tid == 0) {
                const unsigned long long tot = __builtin_readcyclecounter() - tg0;
                atomicAdd(a.prof + 24, g_scan); atomicAdd(a.prof + 25, g_batch); atomicAdd(a.prof + 26, g_last); atomicAdd(a.prof + 27, g_bar);
                atomicAdd(a.prof + 28, g_nb); atomicAdd(a.prof + 29, g_trips); atomicAdd(a.prof + 30, tot); atomicMax(a.prof + 31, tot);
            }

            // the neighbours of the grown set, all waves together (read by wave 0 after the next barrier)
            if (U & ~S) atomicOr(&NBU[lane], U & ~S);
            stamp(2);
            if (MODE == 1 && a.prof && tid == 0) atomicAdd(a.prof + 9, (unsigned long long)sh->nE);
            if (tid == 0) {
                const double moveType = mv_unit(mv_draw_at(rg, 1u));
                sh->start = start; sh->orig = L.own[start]; sh->nd = 0;
                int kind = 0;
                if (moveType < a.split_prob) {
                    kind = n_split >= (unsigned long long)(a.seg_cap - V) ? 3 : 1;   // max_new_segments used up
                    sh->target = (uint32_t)nseg;
                }
                sh->kind = kind; sh->a = 0.0;
            }
        }
        if (em == 3 && tid == 0) {
            // what the growth kernel could not know when it ran ahead: the segment of the start vertex now, and, for a
            // split proposal, whether there is room and which id the new segment gets
            sh->orig = L.own[sh->start];
            if (sh->kind == 1 || sh->kind == 3) {
                sh->kind = n_split >= (unsigned long long)(a.seg_cap - V) ? 3 : 1;
                sh->target = (uint32_t)nseg;
            }
        }
        __syncthreads();
        const uint32_t nE = sh->nE, orig = sh->orig;
        int kind = sh->kind;
        if (kind == 3 && em != 4) {
            if (em == 2) break;                  // frozen and traced by the first half
            status = -6;
            if (em != 3) {
                for (uint32_t i = (uint32_t)tid; i < nE; i += NT) L.inE[L.elist[i]] = 0;
                for (int w = tid; w < 2 * W64; w += NT) EB[w] = 0ull;
            }
            if (tid == 0 && a.trace) { a.trace[(size_t)chain * a.trace_stride + a.trace_first + ts] = 0xFF; if (a.trace_a) a.trace_a[(size_t)chain * a.trace_stride + a.trace_first + ts] = 0.0; }
            __syncthreads();
            continue;
        }
        if (em == 2) {
            // nothing: the proposal was made by the first half
        } else if (kind == 1 || kind == 3) {
            // addNewSegment (MultiviewSegmentation.cpp:85-110): the vertices really move (the growth kernel leaves that
            // to the kernel that fits the planes, which finds the set in its flags instead of the list)
            const uint32_t tg = sh->target;
            if (em == 3) {
                for (int v = tid; v < V; v += NT)
                    if (L.inE[v]) { pvalid[L.own[v]] = 0; L.own[v] = (uint16_t)tg; }
            } else if (em != 4) {
                for (uint32_t i = (uint32_t)tid; i < nE; i += NT) {
                    const uint32_t v = L.elist[i];
                    pvalid[L.own[v]] = 0;
                    L.own[v] = (uint16_t)tg;
                }
            }
            if (em != 4) {
                if (tid == 0) pvalid[tg] = 0;
                nseg++; n_split++;
            }
        } else if (em != 3) {
            // getNeighborSegments (MultiviewSegmentation.cpp:112-142) on the initial ids, from the rows the growth
            // scan has seen (every grown vertex's row exactly once)
            if (wave == 0) {
                // the initial ids of the neighbouring vertices (NBU: every neighbour of the set outside start's segment) as a
                // bitmap of segment ids, word w in lane w: the count is a sum of popcounts and the idx-th neighbour segment
                // (ascending id) a prefix sum away -- a flag byte per segment cost two scans of V / 64 LDS trips each
                for (int w0 = 0; w0 < W64; w0 += kWave) {
                    const int w = w0 + lane;
                    unsigned long long u = w < W64 ? NBU[w] : 0ull;
                    while (u) {
                        const int b = __ffsll((unsigned long long)u) - 1;
                        const uint32_t c = L.cur[w * 64 + b];
                        atomicOr(&NBS[c >> 6], 1ull << (c & 63u));
                        u &= u - 1ull;
                    }
                }
                wave_lds_fence();
                const unsigned long long nbw = lane < W64 ? NBS[lane] : 0ull;      // W64 <= 64 (kMvMaxV)
                const int pc = __popcll(nbw);
                int incl = pc;
#pragma unroll
                for (int off = 1; off < kWave; off <<= 1) {
                    const int t = __shfl_up(incl, off, kWave);
                    if (lane >= off) incl += t;
                }
                const int nn = __shfl(incl, kWave - 1, kWave);
                uint32_t target = 0;
                if (nn >= 1) {
                    // round(randbetween(0, n-1)) (PatternedMatcher.cpp:93)
                    const double x = (double)(nn - 1) * mv_unit(mv_draw_at(rg, 2u));
                    const int idx = (int)(x + 0.5);
                    const bool mine = idx >= incl - pc && idx < incl;
                    unsigned long long mm = nbw;
                    if (mine) for (int r = 0; r < idx - (incl - pc); ++r) mm &= mm - 1ull;
                    const uint32_t cand = (uint32_t)(lane * 64 + (mine ? __ffsll((unsigned long long)mm) - 1 : 0));
                    const uint64_t who = __ballot(mine);
                    target = (uint32_t)__builtin_amdgcn_readlane((int)cand, who ? __ffsll((unsigned long long)who) - 1 : 0);
                }
                if (lane == 0) { sh->nn = nn; sh->target = target; if (nn < 1) sh->kind = 2; }
            }
        }
        __syncthreads();
        stamp(3);
        if (em == 4) {
            // growth kernel done: the set's flags and the shared block go to the kernel that fits the planes
            for (int v = tid; v < V; v += NT) x.inE[(size_t)chain * V + v] = L.inE[v];
            const uint32_t* src = reinterpret_cast<const uint32_t*>(sh);
            uint32_t* dst = reinterpret_cast<uint32_t*>(x.sh + chain);
            for (int k = tid; k < (int)(sizeof(MvShared) / 4); k += NT) dst[k] = src[k];
            continue;
        }
        kind = sh->kind;
        const uint32_t target = sh->target;
        if (kind != 2) {
            // ---- phase 2: planes of SA1, SA2, SB1, SB2 (one wave each) ----
            auto member = [&](int k, int v) -> bool {
                const uint32_t o = L.own[v];
                if (k == 0) return o == orig;
                if (k == 1) return o == target;
                if (k == 2) return o == orig && !L.inE[v];
                return o == target || L.inE[v];
            };
            if (em != 2) {
                double m[10], p5[5];
                int cached = 0;
                if (wave == 0) {
                    for (int k = 0; k < 5; ++k) p5[k] = planes[(size_t)orig * 5 + k];
                    cached = pvalid[orig];
                    if (MODE == 5 || MODE == 3 || MODE == 6) {
                        // Fused kernel (and the plane-fit kernel that follows a growth kernel): this wave has no plane to fit,
                        // so it does the hand-over while the others solve theirs
                        // -- which residuals the L1 pass owes this chain (the codes need the sets and whether SA1 / SA2
                        // have a cached probability, not the planes), the set's flags, and the vertex sets if a split
                        // moved vertices (they change nowhere else, so the kernel's end has nothing left to write).
                        const int hh0 = cached, hh1 = kind == 1 ? 0 : (int)pvalid[target];
                        for (int v = lane; v < V; v += kWave) {
                            const uint32_t o = L.own[v];
                            const bool ie = L.inE[v] != 0;
                            const uint32_t nb4 = (uint32_t)(!hh0 && o == orig) | ((uint32_t)(!hh1 && o == target) << 1) |
                                                 ((uint32_t)(o == orig && !ie) << 2) | ((uint32_t)(o == target || ie) << 3);
                            if (nb4) x.need[(size_t)v * a.n_chains + chain] = (uint8_t)nb4;
                            x.inE[(size_t)chain * V + v] = L.inE[v];
                            if (kind == 1) gown[v] = (uint16_t)o;
                        }
                    }
                } else if (wave == 1) {
                    if (kind == 1) {
                        pool_moments(g.vm, V, lane, [&](int v) { return L.inE[v] != 0; }, m);
                        plane_from_moments(m, p5);
                        if (lane == 0) for (int k = 0; k < 5; ++k) planes[(size_t)target * 5 + k] = p5[k];
                    } else {
                        for (int k = 0; k < 5; ++k) p5[k] = planes[(size_t)target * 5 + k];
                        cached = pvalid[target];
                    }
                } else if (wave == 2) {
                    const int n = pool_moments(g.vm, V, lane, [&](int v) { return member(2, v); }, m);
                    if (n > 0) plane_from_moments(m, p5);
                    else for (int k = 0; k < 5; ++k) p5[k] = planes[(size_t)orig * 5 + k];   // fit skipped (:37)
                } else if (wave == 3) {
                    pool_moments(g.vm, V, lane, [&](int v) { return member(3, v); }, m);
                    plane_from_moments(m, p5);
                }
                if (lane == 0 && wave < 4) {
                    for (int k = 0; k < 5; ++k) sh->pl[wave][k] = p5[k];
                    sh->have[wave] = cached;
                    if (cached) sh->P[wave] = segP[wave == 0 ? orig : target];
                }
            }
            __syncthreads();
            stamp(4);
            // ---- phase 3: L1 residual pass over the points of the four sets ----
            const int h0 = sh->have[0], h1 = sh->have[1];
            if (em == 1 && MODE != 5) {
                for (int v = tid; v < V; v += NT) {
                    const uint32_t o = L.own[v];
                    const bool ie = L.inE[v] != 0;
                    const uint32_t nb4 = (uint32_t)(!h0 && o == orig) | ((uint32_t)(!h1 && o == target) << 1) |
                                         ((uint32_t)(o == orig && !ie) << 2) | ((uint32_t)(o == target || ie) << 3);
                    // the table is all zero between iterations (mv_err_pairs_kernel clears what it reads), so only the
                    // vertices of the sets are written: [V][C] makes these byte stores one cache line each
                    if (nb4) x.need[(size_t)v * a.n_chains + chain] = (uint8_t)nb4;
                    x.inE[(size_t)chain * V + v] = L.inE[v];
                }
                if (a.prof) {
                    // the profile's own counting, outside the timed phases (slot 7 takes its time): points of SB1 u SB2
                    // (the algorithmic-bytes figure of SURVEY 8d), evaluations and residuals |n.p + d| asked for
                    stamp(5);
                    int prof_pts = 0, prof_ev = 0, prof_pe = 0;
                    for (int v = tid; v < V; v += NT) {
                        const uint32_t o = L.own[v];
                        const bool ie = L.inE[v] != 0;
                        const uint32_t nb4 = (uint32_t)(!h0 && o == orig) | ((uint32_t)(!h1 && o == target) << 1) |
                                             ((uint32_t)(o == orig && !ie) << 2) | ((uint32_t)(o == target || ie) << 3);
                        if (o == orig || o == target || ie) {
                            prof_pts += (int)(g.pt_ptr[v + 1] - g.pt_ptr[v]);
                            prof_ev += __popc(nb4);
                            prof_pe += __popc(nb4) * (int)(g.pt_ptr[v + 1] - g.pt_ptr[v]);
                        }
                    }
                    prof_pts = wave_sum_i32(prof_pts); prof_ev = wave_sum_i32(prof_ev); prof_pe = wave_sum_i32(prof_pe);
                    if (lane == 0) {
                        atomicAdd(a.prof + 10, (unsigned long long)prof_pts); atomicAdd(a.prof + 11, (unsigned long long)prof_ev);
                        atomicAdd(a.prof + 12, (unsigned long long)prof_pe);
                    }
                    stamp(7);
                }
            }
            if (em == 0) {
                // A vertex's 16 partial sums (point index mod 16) live in one row of 16 lanes, so a wave takes four
                // vertices at a time, one per row; plane k is evaluated for the rows that want it, the tree runs inside
                // a row (its exchanges are xor 4, 8, 1, 2).
                const int vrow = lane >> 4, m16 = lane & 15;
                for (int vb = wave * 4; vb < V; vb += NW * 4) {
                    const int v = vb + vrow;
                    uint32_t nb4 = 0u, b = 0u, e = 0u;
                    if (v < V) {
                        const uint32_t o = L.own[v];
                        const bool ie = L.inE[v] != 0;
                        nb4 = (uint32_t)(!h0 && o == orig) | ((uint32_t)(!h1 && o == target) << 1) |
                              ((uint32_t)(o == orig && !ie) << 2) | ((uint32_t)(o == target || ie) << 3);
                        if (nb4) { b = g.pt_ptr[v]; e = g.pt_ptr[v + 1]; }
                    }
                    if (!__ballot(nb4 != 0u)) continue;
#pragma unroll 1
                    for (int k = 0; k < 4; ++k) {
                        const bool want = ((nb4 >> k) & 1u) != 0u;
                        if (!__ballot(want)) continue;
                        const double nx = sh->pl[k][0], ny = sh->pl[k][1], nz = sh->pl[k][2], d0 = sh->pl[k][3];
                        double part = 0.0;
                        if (want)
                            for (uint32_t i = b + (uint32_t)m16; i < e; i += 16u) {
                                const float4 p = g.pts[i];
                                part += __builtin_fabs(point_residual(nx, ny, nz, d0, (double)p.x, (double)p.y, (double)p.z));
                            }
                        const double t = vertex_tree_sum(part);
                        if (want && m16 == 0) L.errv[(size_t)k * V + v] = t / (double)(e - b);
                    }
                }
            }
            __syncthreads();
            // ---- phase 4: segment probabilities (MultiviewSegment.cpp:71-96), one wave each ----
            if (em != 1 && em != 3 && wave < 4 && !sh->have[wave]) {
                // four rows per trip: the membership tests first (LDS only), then the members' fetches together (in the
                // split iteration the table is behind L2 -- 68 KB per chain, 70 MB for 1024 chains -- and a lane that
                // is no member fetches nothing, so only the lines that hold a member's residual move); only members
                // are added, in vertex order
                double part = 0.0;
                for (int v0 = lane; v0 < V; v0 += 4 * kWave) {
                    double tv[4]; bool in[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int v = v0 + u * kWave;
                        in[u] = v < V && member(wave, v);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        tv[u] = 0.0;
                        if (in[u]) tv[u] = L.errv[(size_t)wave * V + (v0 + u * kWave)];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (in[u]) part += tv[u];
                }
                const double err = wave_sum_f64(part);
                const double p = prob_from_error(err, sh->pl[wave][4]);
                if (lane == 0) {
                    sh->P[wave] = p;
                    if (wave < 2) { const uint32_t sg = wave == 0 ? orig : target; segP[sg] = p; pvalid[sg] = 1; }
                }
            }
            __syncthreads();
        }
        if (em == 1 || em == 3) {
            // first half done: hand the shared block over (a skipped proposal wants no residuals)
            const uint32_t* src = reinterpret_cast<const uint32_t*>(sh);
            uint32_t* dst = reinterpret_cast<uint32_t*>(x.sh + chain);
            for (int k = tid; k < (int)(sizeof(MvShared) / 4); k += NT) dst[k] = src[k];
            stamp(5);
            continue;
        }
        // ---- phase 5: computeProposalRatio + accept (MultiviewSegmentation.cpp:270-274; PatternedMatcher.cpp:111-112) ----
        uint8_t flags = kind == 1 ? 2 : (kind == 2 ? 4 : 0);
        double av = 0.0;
        if (kind != 2) {
            av = (sh->P[2] / sh->P[0]) * (sh->P[3] / sh->P[1]);
            const double u = mv_unit(mv_draw_at(rg, 3u));
            if (u < av) { flags |= 1; n_accept++; }
        } else {
            n_skip++;
        }
        trace_hash = mix64(mix64(trace_hash ^ ((uint64_t)flags | ((uint64_t)sh->start << 8) | ((uint64_t)nE << 32))) ^ d2u(av));
        if (tid == 0 && a.trace) {
            a.trace[(size_t)chain * a.trace_stride + a.trace_first + ts] = flags;
            if (a.trace_a) a.trace_a[(size_t)chain * a.trace_stride + a.trace_first + ts] = av;
        }
        // clear the step's flags (the halves of a split step start from cleared LDS instead)
        if (em == 0) {
            for (uint32_t i = (uint32_t)tid; i < nE; i += NT) L.inE[L.elist[i]] = 0;
            for (int w = tid; w < 2 * W64; w += NT) EB[w] = 0ull;
            for (int w = tid; w < W64; w += NT) NBS[w] = 0ull;
        }
        __syncthreads();
    }
    if (MODE != 2 && MODE != 4 && MODE != 5 && MODE != 6) for (int v = tid; v < V; v += NT) gown[v] = L.own[v];   // MODE 5: written when a split moved vertices
    stamp(6);
    if (tid == 0) {
        if (MODE != 2 && MODE != 4) { a.nseg[chain] = nseg; a.status[chain] = status; a.n_split[chain] = n_split; }
        if (MODE != 1 && MODE != 3 && MODE != 4) { a.n_accept[chain] = n_accept; a.n_skip[chain] = n_skip; a.trace_hash[chain] = trace_hash; }
    }
}

// L1 residuals of one split step, two kernels.
//
// mv_err_pairs_kernel: a wave takes a vertex, packs the (chain, plane) pairs that want the vertex's residual into a
// list and cuts it into items of up to 64 pairs; the items of a workgroup's 16 vertices go to a queue with one
// atomic (the vertices come by descending cost of their items, MvGraph::vorder, so the queue is roughly longest first).
//
// mv_errors_kernel: a wave takes an item.  The residuals of 16 points under 16 planes are one
// v_mfma_f64_16x16x4_f64: A = the points (row = point, k = x y z 1), B = the planes (k = nx ny nz d, column = pair),
// C = 0; the instruction adds the four products of a row as a chain of FMAs in k order, which is the definition of
// the residual (point_residual above, and the oracle).  Lane l holds column l % 16 (a pair) and rows 4 r + l / 16
// (points) in its four result registers r, so |.| is added with plain vector adds, lane by lane, into the 16 partial
// sums by point index mod 16: m = 4 r + g with r the register and g = l / 16.  The tree over r needs no data
// movement, the one over g is two cross-row exchanges per group of 16 pairs and item.  Up to four groups of 16
// pairs share the points of an item.  Measured: 64 cycles per instruction and SIMD (tools/micro/mfma_f64_rate.hip).
constexpr int kErrChainsPerBlock = 1024;
constexpr int kErrItemsPerBlock = kErrChainsPerBlock * 4 / 16;         // at most 256 items per (vertex, chain block)
constexpr int kErrVertsPerGroup = 16;                                  // vertices per workgroup of mv_err_pairs_kernel
typedef double double4_t __attribute__((ext_vector_type(4)));
#ifndef MC_MV_ERR_RING
#define MC_MV_ERR_RING 4
#endif
// blocks of 16 points asked for ahead of the one multiplied: MC_MV_ERR_RING of them for an item of 64 pairs, more for
// an item of fewer pairs, whose blocks take less time (a block is asked for 16 matrix instructions ahead either way)
MC_D constexpr int err_ring(int G) { return G == 1 ? 4 * MC_MV_ERR_RING : G == 2 ? 2 * MC_MV_ERR_RING : G == 3 ? (4 * MC_MV_ERR_RING + 2) / 3 : MC_MV_ERR_RING; }


#ifdef MC_MV_ERR_CHECK
// diagnostic build only: out-of-range indices of the L1 kernels are counted and clamped instead of used
__device__ unsigned long long g_err_chk[16];
extern "C" int mcmc_mv_debug_check(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_err_chk), sizeof(unsigned long long) * 16);
}
#define MC_CHK(slot, cond, val) do { if (!(cond)) { atomicAdd(&g_err_chk[slot], 1ull); g_err_chk[8 + slot] = (unsigned long long)(val); } } while (0)
#else
#define MC_CHK(slot, cond, val) do { } while (0)
#endif

__global__ __launch_bounds__(kErrVertsPerGroup * kWave) void mv_err_pairs_kernel(MvGraph g, int n_chains, MvExchange x)
{
    __shared__ uint32_t n_items_of[kErrVertsPerGroup], q_base;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int c0 = (int)blockIdx.y * kErrChainsPerBlock, c1 = min(n_chains, c0 + kErrChainsPerBlock);
    const int slot = (int)blockIdx.x * kErrVertsPerGroup + wave;       // one vertex per wave
    uint32_t v = 0, np = 0, pb = 0, pcnt = 0, ppi = kWave, at = 0;
    uint32_t code[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) code[k] = 0u;
    if (slot < g.V) {
        uint32_t mine = 0;
        v = uni(g.vorder[slot]);                                       // vertices with the longest items first
        pb = uni(g.pt_ptr[v]); pcnt = uni(g.pt_ptr[v + 1]) - pb;
        ppi = 16u * mv_err_groups_per_item(pcnt);                         // pairs per item of this vertex
        // lane l looks at chains c0 + 16 l .. + 15: all sixteen codes are fetched before any is used
        uint8_t* row = x.need + (size_t)v * n_chains;
        if ((n_chains & 15) == 0 && (reinterpret_cast<uintptr_t>(x.need) & 15u) == 0) {
            // the lane's sixteen codes are one aligned 16-byte word (rows and chain blocks start at multiples of 16)
            uint4 q = make_uint4(0u, 0u, 0u, 0u);
            uint4* qp = reinterpret_cast<uint4*>(row + c0 + 16 * lane);
            if (c0 + 16 * lane < c1) q = *qp;
            const uint32_t qq[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
            for (int k = 0; k < 16; ++k) code[k] = (qq[k >> 2] >> (8 * (k & 3))) & 0xFFu;
            mine = (uint32_t)(__popc(q.x) + __popc(q.y) + __popc(q.z) + __popc(q.w));      // codes are four bits wide
            // every code is read exactly once, here: cleared for the next iteration
            if (q.x | q.y | q.z | q.w) *qp = make_uint4(0u, 0u, 0u, 0u);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int cc = c0 + 16 * lane + k;
                code[k] = cc < c1 ? (uint32_t)row[cc] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) mine += (uint32_t)__popc(code[k]);
#pragma unroll
            for (int k = 0; k < 16; ++k) if (code[k]) row[c0 + 16 * lane + k] = 0;
        }
        // exclusive prefix over the lanes
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, off, kWave);
            if (lane >= off) incl += t;
        }
        np = (uint32_t)__shfl((int)incl, kWave - 1, kWave);
        at = incl - mine;
    }
    const uint32_t items = (np + ppi - 1u) / ppi;                      // at most 256
    if (lane == 0) n_items_of[wave] = items;
    __syncthreads();
    uint32_t my_first = 0, total = 0;
    for (int s = 0; s < kErrVertsPerGroup; ++s) {
        if (s == wave) my_first = total;
        total += n_items_of[s];
    }
    if (!total) return;
    if (tid == 0) q_base = atomicAdd(x.counters, total);
    __syncthreads();
#ifdef MC_MV_ERR_CHECK
    const size_t qcap = (size_t)gridDim.y * g.V * kErrItemsPerBlock;
    MC_CHK(0, (size_t)q_base + my_first + items <= qcap, q_base + my_first + items);
    if ((size_t)q_base + my_first + items > qcap) return;
#endif
    for (uint32_t k = (uint32_t)lane; k < items; k += kWave)
        x.queue[q_base + my_first + k] = make_uint4(v | ((uint32_t)blockIdx.y << 16), k | (np << 8), pb, pcnt);
    // the pairs of an item sit where the item's place in the queue says (kWave places each), so that the wave that
    // takes the item asks for them together with the queue record
    uint16_t* list = x.plist + (size_t)(q_base + my_first) * kWave;
    uint32_t ik = at / ppi, ipos = at - ik * ppi;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint32_t cbits = code[k];
        while (cbits) {
            const uint32_t pl = (uint32_t)__ffs((int)cbits) - 1u;
            cbits &= cbits - 1u;
            MC_CHK(1, ik < items && ipos < ppi, ((unsigned long long)ik << 32) | ipos);
#ifdef MC_MV_ERR_CHECK
            if (ik >= items) continue;
#endif
            list[ik * kWave + ipos] = (uint16_t)(((uint32_t)(16 * lane + k) << 2) | pl);
            if (++ipos == ppi) { ipos = 0u; ++ik; }
        }
    }
}

template <int G>
MC_D void err_item(const MvGraph& g, const MvExchange& x, const uint4 rec, uint32_t here, const uint32_t (&l4)[4], uint32_t lw,
                   int lane, unsigned long long t_item, int x_n_chains)
{
    MC_CHK(4, (rec.x & 0xFFFFu) < (uint32_t)g.V && rec.z + rec.w <= g.pt_ptr[g.V], ((unsigned long long)rec.x << 32) | rec.z);
#ifdef MC_MV_ERR_CHECK
    if (!((rec.x & 0xFFFFu) < (uint32_t)g.V && rec.z + rec.w <= g.pt_ptr[g.V])) return;
#endif
    constexpr int kErrRing = err_ring(G);
    unsigned long long* const prof = t_item ? x.prof : nullptr;      // one item in 16 is timed (the counters are shared)
    const int v = (int)(rec.x & 0xFFFFu), c0 = (int)(rec.x >> 16) * kErrChainsPerBlock;
    const uint32_t b = rec.z, cnt = rec.w;
    const int jl = lane & 15, kl = lane >> 4;
    // B: component kl (nx ny nz d) of the plane of pair 16 gi + jl; places past the item's pairs evaluate a zero plane
    // (every lane loads a plane -- one that exists where it has no pair -- see the note on conditional loads below)
    const uint32_t wpr = (uint32_t)lane < here ? lw : 0xFFFFFFFFu;      // the pair this lane writes
    double bp[G];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        const bool has = (uint32_t)(16 * gi + jl) < here;
        uint32_t pr = has ? l4[gi] : 0u;                                 // the block's first chain, plane 0: memory that exists
        MC_CHK(2, c0 + (int)(pr >> 2) < x_n_chains, ((unsigned long long)pr << 32) | (unsigned)c0);
#ifdef MC_MV_ERR_CHECK
        if (c0 + (int)(pr >> 2) >= x_n_chains) pr = 0;
#endif
        const double t = x.sh[c0 + (int)(pr >> 2)].pl[pr & 3u][kl];
        bp[gi] = has ? t : 0.0;
    }
    // A: component kl (x y z 1) of point 16 t + jl; kErrRing blocks are on their way while one is multiplied.  Every
    // lane issues every load (the block index clamped to the vertex's last block; ptsd ends with 16 points of padding):
    // a load under a condition is merged with its "else" value at once, which makes every wait a wait for all loads.
    const double* pbase = g.ptsd + (size_t)b * 4;             // uniform; the lane's part of the address is a 32-bit offset
    const uint32_t loff = (uint32_t)(jl * 4 + kl);
    const uint32_t nblk = (cnt + 15u) >> 4, nfull = cnt >> 4, tail = cnt & 15u;
    const uint32_t lastb = (nblk ? nblk : 1u) - 1u;
    const double4_t zero4 = { 0.0, 0.0, 0.0, 0.0 };
    double4_t acc[G];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) acc[gi] = zero4;
    double ring[kErrRing];
#pragma unroll
    for (int w = 0; w < kErrRing; ++w) ring[w] = (pbase + (size_t)min((uint32_t)w, lastb) * 64)[loff];
    unsigned long long t_loop = 0ull;
    if (prof) {      // the planes have arrived: everything before the first block
        t_loop = __builtin_readcyclecounter() + (unsigned long long)(d2u(bp[0]) & 1ull) * 0ull;
        if (lane == 0) atomicAdd(prof + 16, t_loop - t_item);
    }
    // complete blocks, kErrRing at a time: nothing but the load, the products and the additions
    uint32_t t0 = 0u;
    for (; t0 + (uint32_t)kErrRing <= nfull; t0 += (uint32_t)kErrRing) {
#pragma unroll
        for (int w = 0; w < kErrRing; ++w) {
            const double a = ring[w];
            ring[w] = (pbase + (size_t)min(t0 + (uint32_t)(w + kErrRing), lastb) * 64)[loff];
#pragma unroll
            for (int gi = 0; gi < G; ++gi) {
                const double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[gi], zero4, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[gi][r] += __builtin_fabs(d[r]);
            }
        }
    }
    // what is left: fewer than kErrRing complete blocks and the one with rows past the vertex's points
#pragma unroll
    for (int w = 0; w < kErrRing; ++w) {
        const uint32_t t = t0 + (uint32_t)w;
        if (t >= nblk) break;
        const double a = ring[w];
        const bool cut = t >= nfull;
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[gi], zero4, 0, 0, 0);
            if (cut) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if ((uint32_t)(4 * r + kl) >= tail) d[r] = 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[gi][r] += __builtin_fabs(d[r]);
        }
    }
    unsigned long long t_end = 0ull;
    if (prof) {
        t_end = __builtin_readcyclecounter() + (unsigned long long)(d2u(acc[0][0]) & 1ull) * 0ull;
        if (lane == 0) { atomicAdd(prof + 17, t_end - t_loop); atomicAdd(prof + 20, (unsigned long long)nblk); atomicAdd(prof + 21, (unsigned long long)nblk * G); }
    }
    // the tree: over r in the lane, over g across the four rows of the wave
    double res = 0.0;
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        double t = (acc[gi][0] + acc[gi][1]) + (acc[gi][2] + acc[gi][3]);
        t = t + __shfl_xor(t, 16, kWave);
        t = t + __shfl_xor(t, 32, kWave);
        if (kl == gi) res = t;
    }
    // lane l writes pair l of the item: the mean, one division per pair
    MC_CHK(3, wpr == 0xFFFFFFFFu || (c0 + (int)(wpr >> 2) < x_n_chains && v < g.V), ((unsigned long long)wpr << 32) | (unsigned)v);
#ifdef MC_MV_ERR_CHECK
    if (wpr != 0xFFFFFFFFu && !(c0 + (int)(wpr >> 2) < x_n_chains && v < g.V)) return;
#endif
    if (wpr != 0xFFFFFFFFu) x.err[(size_t)(c0 + (int)(wpr >> 2)) * 4 * g.V + (size_t)(wpr & 3u) * g.V + v] = res / (double)cnt;
    if (prof && lane == 0) { atomicAdd(prof + 18, __builtin_readcyclecounter() - t_end); atomicAdd(prof + 19, 1ull); }
}

#ifdef MC_MV_ERR_TIMELINE
// diagnostic build only (tools/mv_timeline.py): when and where every item of the last launch ran
__device__ unsigned long long g_err_tl[16384 * 4];
extern "C" int mcmc_mv_debug_timeline(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_err_tl), sizeof(unsigned long long) * 16384 * 4);
}
#endif

__global__ __launch_bounds__(kWave) void mv_errors_kernel(MvGraph g, int n_chains, MvExchange x)
{
    const int lane = (int)threadIdx.x;
    // Workgroup b runs on XCD b % 8 (round-robin dispatch).  The items of a vertex are neighbours in the queue and read
    // the same points, so runs of 8 items go to one XCD and meet in its L2: wave i of XCD k takes item 8 (8 (i / 8) + k) + i % 8.
    const uint32_t xcd = blockIdx.x & 7u, wi = blockIdx.x >> 3;
    const uint32_t slot = ((wi >> 3) * 8u + xcd) * 8u + (wi & 7u);
    // The record and the pairs of the wave's first item are asked for together with the length of the queue (the
    // queue and the pair lists are as long as the launch at least, so the place exists whether or not an item is in
    // it): an item costs its wave two round trips before the first product, this one and the planes + first points.
    uint4 rec = x.queue[slot];
    uint32_t l4[4], lw;
    {
        const uint16_t* list = x.plist + (size_t)slot * kWave;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) l4[gi] = (uint32_t)list[16 * gi + (lane & 15)];
        lw = (uint32_t)list[lane];
    }
    const uint32_t n_items = uni(x.counters[0]);               // cleared by the second half of the iteration
    if (slot >= n_items) return;
    const unsigned long long t_wave = x.prof ? __builtin_readcyclecounter() : 0ull;
    for (uint32_t item = slot;;) {
        const unsigned long long t_item = (x.prof && (item & 15u) == 0u) ? __builtin_readcyclecounter() : 0ull;
        rec.x = uni(rec.x); rec.y = uni(rec.y); rec.z = uni(rec.z); rec.w = uni(rec.w);
        const uint32_t ppi = 16u * mv_err_groups_per_item(rec.w);
        const uint32_t here = min((rec.y >> 8) - (rec.y & 0xFFu) * ppi, ppi);
#ifdef MC_MV_ERR_TIMELINE
        const unsigned long long tl0 = wall_clock64();
#endif
        switch ((here + 15u) >> 4) {
            case 1: err_item<1>(g, x, rec, here, l4, lw, lane, t_item, n_chains); break;
            case 2: err_item<2>(g, x, rec, here, l4, lw, lane, t_item, n_chains); break;
            case 3: err_item<3>(g, x, rec, here, l4, lw, lane, t_item, n_chains); break;
            default: err_item<4>(g, x, rec, here, l4, lw, lane, t_item, n_chains); break;
        }
#ifdef MC_MV_ERR_TIMELINE
        if (lane == 0 && item < 16384u) {
            g_err_tl[item * 4 + 0] = tl0; g_err_tl[item * 4 + 1] = wall_clock64();
            g_err_tl[item * 4 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
            g_err_tl[item * 4 + 3] = (unsigned long long)((rec.w + 15u) >> 4) * ((here + 15u) >> 4);
        }
#endif
        item += gridDim.x;                                     // queues longer than the launch: rare (16384 waves)
        if (item >= n_items) break;
        rec = x.queue[item];
        const uint16_t* list = x.plist + (size_t)item * kWave;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) l4[gi] = (uint32_t)list[16 * gi + (lane & 15)];
        lw = (uint32_t)list[lane];
    }
    if (x.prof && lane == 0 && (slot & 15u) == 0u) { atomicAdd(x.prof + 22, 1ull); atomicAdd(x.prof + 23, __builtin_readcyclecounter() - t_wave); }
}

// pMerge (GraphUtils.cpp:11-20): exp(-|w (.) |t2 - t1||^2 T / 2); one thread per pair
__global__ void mv_pmerge_kernel(const float* t1, const float* t2, const float* w, int dim, int n, double T, double* out)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    double s = 0.0;
    for (int k = 0; k < dim; ++k) {
        double d = (double)t2[(size_t)i * dim + k] - (double)t1[(size_t)i * dim + k];
        if (d < 0.0) d = -d;
        const double e = (double)w[k] * d;
        s += e * e;
    }
    out[i] = det_exp(-s * T / 2.0);
}

// ---- world graph (GraphUtils.cpp:52-138 getPlanarAdjacencyGraph, :140-202 mergeNewScanGraph) ----
MC_D double pmerge_pair(const float* th, const float* w, int dim, int i, int j, double T)
{
    // theta1 = the lower id: |t2 - t1| is symmetric, so both directions get the same bits
    const int a = i < j ? i : j, b = i < j ? j : i;
    double s = 0.0;
    for (int k = 0; k < dim; ++k) {
        double d = (double)th[(size_t)b * dim + k] - (double)th[(size_t)a * dim + k];
        if (d < 0.0) d = -d;
        const double e = (double)w[k] * d;
        s += e * e;
    }
    return det_exp(-s * T / 2.0);
}

// every pair with a vertex of a later scan: one thread per (row, 32 columns)
__global__ void mv_world_pairs_kernel(const float* th, const float* w, int dim, int V, int n0, int words, double thr,
                                      double T, uint32_t* bits)
{
    const int wd = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int i = (int)blockIdx.y;
    if (wd >= words) return;
    uint32_t m = 0;
    for (int b = 0; b < 32; ++b) {
        const int j = wd * 32 + b;
        if (j >= V || j == i) continue;                       // no self loops (:166)
        if ((i > j ? i : j) < n0) continue;                   // both in the first scan: in-view edges only
        if (pmerge_pair(th, w, dim, i, j, T) > thr) m |= 1u << b;
    }
    if (m) atomicOr(bits + (size_t)i * words + wd, m);
}

// in-view neighbours (the superpixel graph of each scan): one thread per directed entry
__global__ void mv_world_inview_kernel(const float* th, const float* w, int dim, int V, const uint32_t* rowp,
                                       const uint32_t* col, uint32_t e2, int words, double thr, double T, uint32_t* bits)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= e2) return;
    int lo = 0, hi = V;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rowp[mid] <= e) lo = mid; else hi = mid; }
    const int i = lo, j = (int)col[e];
    if (j >= V || j == i) return;
    if (pmerge_pair(th, w, dim, i, j, T) > thr) {
        atomicOr(bits + (size_t)i * words + (j >> 5), 1u << (j & 31));
        atomicOr(bits + (size_t)j * words + (i >> 5), 1u << (i & 31));
    }
}

__global__ void mv_world_degree_kernel(const uint32_t* bits, int V, int words, uint32_t* deg)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= V) return;
    uint32_t d = 0;
    for (int k = 0; k < words; ++k) d += (uint32_t)__popc(bits[(size_t)i * words + k]);
    deg[i] = d;
}

__global__ void mv_world_fill_kernel(const float* th, const float* w, int dim, int V, int words, const uint32_t* bits,
                                     const uint32_t* rowp, double T, uint32_t* col, double* prob)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= V) return;
    uint32_t e = rowp[i];
    for (int k = 0; k < words; ++k) {
        uint32_t m = bits[(size_t)i * words + k];
        while (m) {
            const int j = k * 32 + __ffs((int)m) - 1;
            col[e] = (uint32_t)j; prob[e] = pmerge_pair(th, w, dim, i, j, T);
            ++e; m &= m - 1u;
        }
    }
}

hipError_t launch_mv_world_bits(const float* th, const float* w, int dim, int V, int n0, const uint32_t* in_rowp,
                                const uint32_t* in_col, uint32_t in_e2, double in_thr, double thr, double T, int words,
                                uint32_t* bits, uint32_t* deg, hipStream_t st)
{
    hipLaunchKernelGGL(mv_world_pairs_kernel, dim3((unsigned)((words + 63) / 64), (unsigned)V), dim3(64), 0, st, th, w, dim, V, n0,
                       words, thr, T, bits);
    if (in_e2)
        hipLaunchKernelGGL(mv_world_inview_kernel, dim3((in_e2 + 127) / 128), dim3(128), 0, st, th, w, dim, V, in_rowp, in_col,
                           in_e2, words, in_thr, T, bits);
    hipLaunchKernelGGL(mv_world_degree_kernel, dim3((unsigned)((V + 127) / 128)), dim3(128), 0, st, bits, V, words, deg);
    return hipGetLastError();
}

hipError_t launch_mv_world_fill(const float* th, const float* w, int dim, int V, int words, const uint32_t* bits,
                                const uint32_t* rowp, double T, uint32_t* col, double* prob, hipStream_t st)
{
    hipLaunchKernelGGL(mv_world_fill_kernel, dim3((unsigned)((V + 127) / 128)), dim3(128), 0, st, th, w, dim, V, words, bits, rowp,
                       T, col, prob);
    return hipGetLastError();
}

hipError_t launch_mv_pmerge(const float* t1, const float* t2, const float* w, int dim, int n, double T, double* out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(mv_pmerge_kernel, dim3((n + 127) / 128), dim3(128), 0, st, t1, t2, w, dim, n, T, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_mv_vertex_moments(const MvGraph& g, double* out, hipStream_t st)
{
    if (g.V <= 0) return hipSuccess;
    hipLaunchKernelGGL(mv_vertex_moments_kernel, dim3((g.V + 3) / 4), dim3(256), 0, st, g, out);
    return hipGetLastError();
}

size_t mv_chain_lds_bytes(int V) { return mv_lds_bytes(V, true); }

static hipError_t mv_set_lds_attr()
{
    static DeviceOnce once;
    bool* done = once.slot();
    if (done && *done) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mv_chain_run_kernel<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mv_chain_run_kernel<1, kMvWavesFirstHalf>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mv_chain_run_kernel<2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mv_chain_run_kernel<5, kMvWavesFirstHalf>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mv_chain_run_kernel<4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mv_chain_run_kernel<6>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mv_chain_run_kernel<3>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mv_chain_run_kernel<4, kMvGrowthWaves>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mv_chain_init_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mv_fit_segment_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) return e;
    if (done) *done = true;
    return hipSuccess;
}

hipError_t launch_mv_fit_segment(const MvGraph& g, const uint8_t* mem, double* out, hipStream_t st)
{
    hipError_t e = mv_set_lds_attr();
    if (e != hipSuccess) return e;
    const size_t bytes = ((size_t)g.V + 8) * 8;
    if (bytes > 160 * 1024 - 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mv_fit_segment_kernel, dim3(1), dim3(kMvThreads), bytes, st, g, mem, out);
    return hipGetLastError();
}

hipError_t launch_mv_chain_init(const MvGraph& g, const MvChainArgs& a, hipStream_t st)
{
    if (g.V > kMvMaxV) return hipErrorInvalidValue;
    hipError_t e = mv_set_lds_attr();
    if (e != hipSuccess) return e;
    const size_t bytes = mv_lds_bytes(g.V, false);
    if (bytes > 160 * 1024 - 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mv_chain_init_kernel, dim3(a.n_chains), dim3(kMvThreads), bytes, st, g, a);
    return hipGetLastError();
}

hipError_t launch_mv_chain_run(const MvGraph& g, const MvChainArgs& a, hipStream_t st)
{
    hipError_t e = mv_set_lds_attr();
    if (e != hipSuccess) return e;
    if (g.V > kMvMaxV) return hipErrorInvalidValue;
    const size_t bytes = mv_lds_bytes(g.V, true);
    if (bytes > 160 * 1024 - 64) return hipErrorInvalidValue;
    MvExchange x{};
    hipLaunchKernelGGL(mv_chain_run_kernel<0>, dim3(a.n_chains), dim3(kMvThreads), bytes, st, g, a, x);
    return hipGetLastError();
}

size_t mv_exchange_shared_bytes() { return sizeof(MvShared); }

// the same `steps` iterations as four kernels per iteration (see mv_chain_run_kernel)
size_t mv_split_plist_entries(int V, int n_chains) { return mv_split_queue_entries(V, n_chains) * kWave; }      // kWave places per place in the queue
// kWave places more than items can be: the L1 kernel reads the place of every wave of its launch, a multiple of kWave
size_t mv_split_queue_entries(int V, int n_chains) { return (size_t)((n_chains + kErrChainsPerBlock - 1) / kErrChainsPerBlock) * V * kErrItemsPerBlock + kWave; }

hipError_t launch_mv_chain_run_split(const MvGraph& g, const MvChainArgs& a0, const MvSplitBufs& bufs, hipStream_t st, bool split_growth,
                                     bool fuse_halves, bool opens, bool closes)
{
    if (g.V > kMvMaxV) return hipErrorInvalidValue;
    hipError_t e = mv_set_lds_attr();
    if (e != hipSuccess) return e;
    const size_t bytes = mv_lds_bytes(g.V, false);
    const size_t gbytes = mv_lds_bytes(g.V, false, (size_t)(kMvGrowthWaves > kMvWaves ? kMvGrowthWaves : kMvWaves) * 1024);
    if (bytes > 160 * 1024 - 64 || gbytes > 160 * 1024 - 64) return hipErrorInvalidValue;
    MvExchange x{ reinterpret_cast<MvShared*>(bufs.sh), bufs.inE, bufs.need, bufs.err, bufs.plist, bufs.queue, a0.prof, bufs.counters };
    const dim3 pgrid((unsigned)((g.V + kErrVertsPerGroup - 1) / kErrVertsPerGroup), (unsigned)((a0.n_chains + kErrChainsPerBlock - 1) / kErrChainsPerBlock));
    if (pgrid.y > 65535u) return hipErrorInvalidValue;      // > 64 M chains per launch: shard the chains
    // error kernel: a fixed number of waves that share the queue; more than the items of a typical iteration would only idle
    const size_t qmax = mv_split_queue_entries(g.V, a0.n_chains) - kWave;
    unsigned ewaves = (unsigned)std::min<size_t>((qmax + 63) & ~(size_t)63, 16384);      // a multiple of 64: see the item mapping
    if (const char* ev = std::getenv("MCMC_MV_EWAVES")) { const long n = std::atol(ev); if (n > 0) ewaves = (unsigned)std::min<size_t>(((size_t)n + 63) & ~(size_t)63, (qmax + 63) & ~(size_t)63); }   // tuning knob
    size_t elds = 0;
    if (const char* ev = std::getenv("MCMC_MV_EWPS")) { const long n = std::atol(ev); if (n > 0) elds = (size_t)(160 * 1024 / (4 * n)) - 256; }   // tuning knob: waves per SIMD
    for (unsigned long long ts = 0; ts < a0.steps; ++ts) {
        MvChainArgs a = a0;
        a.step0 = a0.step0 + ts; a.steps = 1; a.trace_first = a0.trace_first + ts;
        // fuse_halves: the second half of the step before and this step's first half are one kernel (MODE 5) -- a chain
        // goes on to its next proposal as soon as its own residuals are summed, and its state is loaded once; the
        // first step of a run (opens) has no step before it, the last (closes) gets its second half as a kernel
        const bool fused = fuse_halves && !split_growth && !(opens && ts == 0);
        if (fused) {
            MvChainArgs af = a;
            af.step0 = a.step0 - 1; af.steps = 2; af.trace_first = a.trace_first - 1;
            hipLaunchKernelGGL((mv_chain_run_kernel<5, kMvWavesFirstHalf>), dim3(a.n_chains), dim3(kMvWavesFirstHalf * 64), bytes, st, g, af, x);
        } else if (split_growth) {
            // the growth in a kernel of its own (eight waves per chain within 64 VGPRs), then the plane fits
            hipLaunchKernelGGL((mv_chain_run_kernel<4, kMvGrowthWaves>), dim3(a.n_chains), dim3(kMvGrowthWaves * 64), gbytes, st, g, a, x);
            hipLaunchKernelGGL(mv_chain_run_kernel<3>, dim3(a.n_chains), dim3(kMvThreads), bytes, st, g, a, x);
        } else
            hipLaunchKernelGGL((mv_chain_run_kernel<1, kMvWavesFirstHalf>), dim3(a.n_chains), dim3(kMvWavesFirstHalf * 64), bytes, st, g, a, x);
        hipLaunchKernelGGL(mv_err_pairs_kernel, pgrid, dim3(kErrVertsPerGroup * kWave), 0, st, g, a.n_chains, x);
        hipLaunchKernelGGL(mv_errors_kernel, dim3(ewaves), dim3(kWave), elds, st, g, a.n_chains, x);
        if (!(fuse_halves && !split_growth) || (closes && ts + 1 == a0.steps))
            hipLaunchKernelGGL(mv_chain_run_kernel<2>, dim3(a.n_chains), dim3(kMvThreads), bytes, st, g, a, x);
    }
    return hipGetLastError();
}

// The same iterations with the growth a step ahead: growth(t + 1) (MODE 4, stream gst, into slot (t + 1) & 1 of the
// two hand-over buffers) runs while the main stream does [second half of t - 1 + plane fits of t] (MODE 6), the pair
// lists and the L1 kernel of step t.  ev_g[s]: growth into slot s done; ev_c[s]: the kernel that read slot s done.
// Slots go by the parity of the absolute step, so that a run can leave the growth of the step after its last one in its
// slot (grow_next) and the next run of the same chains start from it (have_first) instead of waiting for a growth
// that nothing runs beside.
hipError_t launch_mv_chain_run_ahead(const MvGraph& g, const MvChainArgs& a0, const MvSplitBufs& bufs, const MvAheadBufs& ab,
                                     hipStream_t st, hipStream_t gst, int growth_waves, bool have_first, bool grow_next)
{
    if (g.V > kMvMaxV) return hipErrorInvalidValue;
    hipError_t e = mv_set_lds_attr();
    if (e != hipSuccess) return e;
    const size_t bytes = mv_lds_bytes(g.V, false);
    const size_t gbytes = mv_lds_bytes(g.V, false, (size_t)(kMvGrowthWaves > kMvWaves ? kMvGrowthWaves : kMvWaves) * 1024);
    if (bytes > 160 * 1024 - 64 || gbytes > 160 * 1024 - 64) return hipErrorInvalidValue;
    const dim3 pgrid((unsigned)((g.V + kErrVertsPerGroup - 1) / kErrVertsPerGroup), (unsigned)((a0.n_chains + kErrChainsPerBlock - 1) / kErrChainsPerBlock));
    if (pgrid.y > 65535u) return hipErrorInvalidValue;
    const size_t qmax = mv_split_queue_entries(g.V, a0.n_chains) - kWave;
    const unsigned ewaves = (unsigned)std::min<size_t>((qmax + 63) & ~(size_t)63, 16384);
    hipError_t first = hipSuccess;                               // the first failure of an event call (the launches report through hipGetLastError)
    auto keep = [&](hipError_t r) { if (first == hipSuccess && r != hipSuccess) first = r; };
    auto growth = [&](unsigned long long ts) {
        // into slot ts & 1, once the kernel that read that slot two steps ago is done
        const int slot = (int)((a0.step0 + ts) & 1ull);
        if (ts >= 2) keep(hipStreamWaitEvent(gst, ab.ev_c[slot], 0));
        MvChainArgs a = a0;
        a.step0 = a0.step0 + ts; a.steps = 1; a.trace_first = a0.trace_first + ts;
        MvExchange gx{};
        gx.sh = reinterpret_cast<MvShared*>(ab.sh[slot]); gx.inE = ab.inE[slot];
        if (growth_waves == kMvGrowthWaves)
            hipLaunchKernelGGL((mv_chain_run_kernel<4, kMvGrowthWaves>), dim3(a.n_chains), dim3(kMvGrowthWaves * 64), gbytes, gst, g, a, gx);
        else
            hipLaunchKernelGGL(mv_chain_run_kernel<4>, dim3(a.n_chains), dim3(kMvThreads), bytes, gst, g, a, gx);
        keep(hipEventRecord(ab.ev_g[slot], gst));
    };
    // the state the growth kernels read (labels; status) is ready when the main stream gets here
    keep(hipEventRecord(ab.ev_start, st));
    keep(hipStreamWaitEvent(gst, ab.ev_start, 0));
    if (a0.steps && !have_first) growth(0);
    for (unsigned long long ts = 0; ts < a0.steps; ++ts) {
        const int slot = (int)((a0.step0 + ts) & 1ull);
        if (ts + 1 < a0.steps || grow_next) growth(ts + 1);    // queued before this step's kernels: it runs beside them
        MvExchange x{ reinterpret_cast<MvShared*>(bufs.sh), bufs.inE, bufs.need, bufs.err, bufs.plist, bufs.queue, a0.prof, bufs.counters,
                      ab.inE[slot], reinterpret_cast<const MvShared*>(ab.sh[slot]) };
        MvChainArgs a = a0;
        a.step0 = a0.step0 + ts; a.steps = 1; a.trace_first = a0.trace_first + ts;
        keep(hipStreamWaitEvent(st, ab.ev_g[slot], 0));
        if (ts == 0)
            hipLaunchKernelGGL(mv_chain_run_kernel<3>, dim3(a.n_chains), dim3(kMvThreads), bytes, st, g, a, x);
        else {
            MvChainArgs af = a;
            af.step0 = a.step0 - 1; af.steps = 2; af.trace_first = a.trace_first - 1;
            hipLaunchKernelGGL(mv_chain_run_kernel<6>, dim3(a.n_chains), dim3(kMvThreads), bytes, st, g, af, x);
        }
        keep(hipEventRecord(ab.ev_c[slot], st));
        hipLaunchKernelGGL(mv_err_pairs_kernel, pgrid, dim3(kErrVertsPerGroup * kWave), 0, st, g, a.n_chains, x);
        hipLaunchKernelGGL(mv_errors_kernel, dim3(ewaves), dim3(kWave), 0, st, g, a.n_chains, x);
        if (ts + 1 == a0.steps)
            hipLaunchKernelGGL(mv_chain_run_kernel<2>, dim3(a.n_chains), dim3(kMvThreads), bytes, st, g, a, x);
    }
    // nothing of the growth stream may outlive the call's work on the main stream
    keep(hipEventRecord(ab.ev_start, gst));
    keep(hipStreamWaitEvent(st, ab.ev_start, 0));
    const hipError_t launched = hipGetLastError();
    return launched != hipSuccess ? launched : first;
}

}  // namespace mc
