// capi_overseg.hip -- over-segmentation front end: what turns a colour +
// disparity frame into the label map and superpixel graph the hot path consumes.
//
//   bilateral_kernel       OverSegmentation::bilateralFilter   src/libraries/OverSegmentation.cpp:304-391
//   edge_weight_kernel     createEdges / computeWeight          :236-301
//   (host) segment_graph   include/Segmentation/segment-graph.h:47-111, disjoint-set.h:42-79
//   label / adjacency      createSuperPixels                    :89-194
//
// The filter and the edge weights are stencils over the frame (one 16x16 tile per workgroup, the tile
// of disparities and colours in LDS, the disparity Gaussian from an LDS table because disparities are
// integers).  The reference sorts the edges with std::sort on the weight alone, so the order of equal weights is
// whatever that standard library's introsort leaves: the default ("tie_order" = 1) runs that very call on the host
// over the edges in creation order.  Option "tie_order" = 0 sorts on the device instead (stable radix sort of the
// weight bits by rocPRIM, a library sort: equal weights stay in creation order), which is faster but is a layout
// of this library's own wherever weights tie.  The union-find pass is a sequential scan over the sorted edges and
// runs on the host, as in the reference; labels and the neighbour relation are built on the device from the roots
// (rank of root pixels, S x S bit matrix, CSR by row).
// Restates behaviour, not code; every quirk kept is listed in DESIGN.md.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mcmc_amd.h"
#include "device_math.hpp"

#pragma clang fp contract(off)

using namespace mc;

namespace {

constexpr int kTile = 16;
constexpr int kMaxWindow = 32;

MC_HD double gauss2(double x, double y, double stdev)   // OverSegmentation.h:75-79
{
    const double norm = 1.0 / det_sqrt(2.0 * 3.14159265358979323846 * stdev * stdev);
    const double exponent = (x * x + y * y) / (2 * stdev * stdev);
    return norm * det_exp(-exponent);
}

// One thread per filtered pixel.  sk: normalised spatial kernel [i][j]; gtab[d] = Gaussian of an integer
// disparity difference d (odd windows; beyond n_tab the Gaussian is exactly norm * 0).
template <bool EVEN>
__global__ __launch_bounds__(kTile* kTile) void bilateral_kernel(const int16_t* disp, const uint8_t* colors, int W, int H,
                                                                int w, const double* sk, const double* gtab, int n_tab,
                                                                double dstd, uint8_t* smoothed, int* status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int T = kTile + w;
    double* s_sk = reinterpret_cast<double*>(sm);                  // w*w
    double* s_g = s_sk + w * w;                                     // n_tab
    int16_t* s_d = reinterpret_cast<int16_t*>(s_g + n_tab);         // T*T
    uint8_t* s_c = reinterpret_cast<uint8_t*>(s_d + ((T * T + 3) & ~3));   // T*T*3
    const int tid = (int)threadIdx.x;
    const int bx = (int)blockIdx.x * kTile, by = (int)blockIdx.y * kTile;
    for (int k = tid; k < w * w; k += kTile * kTile) s_sk[k] = sk[k];
    for (int k = tid; k < n_tab; k += kTile * kTile) s_g[k] = gtab[k];
    for (int k = tid; k < T * T; k += kTile * kTile) {
        const int ty = k / T, tx = k - ty * T;
        const int gx = min(bx + tx, W - 1), gy = min(by + ty, H - 1);
        s_d[k] = disp[(size_t)gy * W + gx];
        const uint8_t* c = colors + ((size_t)gy * W + gx) * 3;
        s_c[k * 3 + 0] = c[0]; s_c[k * 3 + 1] = c[1]; s_c[k * 3 + 2] = c[2];
    }
    __syncthreads();
    const int lx = tid % kTile, ly = tid / kTile;
    const int x = bx + lx, y = by + ly;
    if (x + w >= W || y + w >= H) return;            // :326-327, the right and bottom bands stay 0
    const int c0 = (w + 1) / 2;                      // ceil(w / 2.0), :335
    double md = (double)s_d[(ly + c0) * T + lx + c0];
    if (EVEN)
        md = (s_d[(ly + c0) * T + lx + c0] + s_d[(ly + c0) * T + lx + c0 + 1] + s_d[(ly + c0 + 1) * T + lx + c0] +
              s_d[(ly + c0 + 1) * T + lx + c0 + 1]) / 4.0;
    auto joint = [&](int i, int j) -> double {
        const double dd = md - (double)s_d[(ly + j) * T + lx + i];
        double g;
        if (EVEN) g = gauss2(dd, 0.0, dstd);
        else {
            const int a = (int)(dd < 0.0 ? -dd : dd);
            g = s_g[a < n_tab ? a : n_tab - 1];
        }
        return s_sk[i * w + j] * g;
    };
    double sum = 0.0;
    for (int j = 0; j < w; ++j)
        for (int i = 0; i < w; ++i) sum += joint(i, j);
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    for (int j = 0; j < w; ++j)
        for (int i = 0; i < w; ++i) {
            const double kv = joint(i, j) / sum;
            const uint8_t* c = s_c + ((ly + j) * T + lx + i) * 3;
            t0 += (double)c[0] * kv; t1 += (double)c[1] * kv; t2 += (double)c[2] * kv;
        }
    const long long v0 = (long long)t0, v1 = (long long)t1, v2 = (long long)t2;   // truncation, :381
    // a kernel sum of 0 (every Gaussian underflowed) makes the means NaN
    if (!(v0 >= 0 && v0 < 256 && v1 >= 0 && v1 < 256 && v2 >= 0 && v2 < 256) || t0 != t0 || t1 != t1 || t2 != t2) atomicOr(status, 1);
    uint8_t* o = smoothed + ((size_t)y * W + x) * 3;
    o[0] = (uint8_t)v0; o[1] = (uint8_t)v1; o[2] = (uint8_t)v2;
}

// Correctly rounded float square root (what sqrtf is on the host): start from the f64 root and move to a
// neighbouring float while x lies beyond the midpoint between them (midpoints have 25 significant bits,
// so their squares are exact in f64).
__device__ float sqrt_f32_rn(float x)
{
    if (!(x > 0.f)) return x == 0.f ? 0.f : __builtin_nanf("");
    float y = (float)__builtin_sqrt((double)x);
    for (int it = 0; it < 2; ++it) {
        const float up = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, y) + 1u);
        const float dn = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, y) - 1u);
        const double mu = ((double)y + (double)up) * 0.5, md = ((double)y + (double)dn) * 0.5;
        if ((double)x > mu * mu) y = up;
        else if ((double)x < md * md) y = dn;
    }
    return y;
}

// One thread per pixel, four slots per pixel in the reference's direction order; a slot whose
// neighbour is outside the frame gets a = -1 and is dropped when the host packs the list.
__global__ void edge_weight_kernel(const int16_t* disp, const uint8_t* smoothed, int W, int H, double cf, double df,
                                   double distf, int32_t* ea, int32_t* eb, float* ew)
{
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= W * H) return;
    const int y = k / W, x = k - y * W;
    const int dx[4] = { 1, 0, 1, 1 }, dy[4] = { 0, 1, 1, -1 };   // RIGHT, BOTTOM, BOTTOM-RIGHT, TOP-RIGHT
    const uint8_t* c1 = smoothed + (size_t)k * 3;
    const float d1 = (float)disp[k];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x2 = x + dx[i], y2 = y + dy[i];
        int32_t a = -1, b = -1;
        float wgt = 0.f;
        if (x2 < W && y2 < H && y2 >= 0) {
            const int k2 = y2 * W + x2;
            const uint8_t* c2 = smoothed + (size_t)k2 * 3;
            const float dr = (float)c1[2] - (float)c2[2];
            const float dg = (float)c1[1] - (float)c2[1];
            const float db = (float)c1[0] - (float)c2[0];
            const float colorv = sqrt_f32_rn(dr * dr + dg * dg + db * db);
            const float d2 = (float)disp[k2];
            const float dd = d1 - d2;
            const float dispv = dd < 0.f ? -dd : dd;
            const float distv = (dx[i] != 0 && dy[i] != 0) ? (float)1.4142135623730951 : 1.0f;
            wgt = (float)((cf * (double)colorv + df * (double)dispv) + distf * (double)distv);
            a = k; b = k2;
        }
        ea[(size_t)k * 4 + i] = a; eb[(size_t)k * 4 + i] = b; ew[(size_t)k * 4 + i] = wgt;
    }
}

// Sort key of an edge slot: the weight's bits in an order-preserving form (sign flip for >= 0, complement for < 0),
// all ones for a slot without an edge, so that those end up last; the value is the slot index (creation order).
__global__ void edge_key_kernel(const int32_t* ea, const float* ew, uint32_t n, uint32_t* key, uint32_t* idx, int* n_valid)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = false;
    if (s < n) {
        valid = ea[s] >= 0;
        const uint32_t bits = __float_as_uint(ew[s]);
        key[s] = valid ? ((bits & 0x80000000u) ? ~bits : (bits | 0x80000000u)) : 0xFFFFFFFFu;
        idx[s] = s;
    }
    const uint64_t m = __ballot(valid);
    if ((threadIdx.x & 63u) == 0u && m) atomicAdd(n_valid, (int)__popcll((unsigned long long)m));
}

// ---- labels from roots: rank of the root pixels in ascending pixel order ----
// single workgroup, 1024 threads, contiguous chunks (N is a frame: a few 100k)
__global__ __launch_bounds__(1024) void root_rank_kernel(const int32_t* root, int N, int32_t* rank, int* S_out)
{
    __shared__ int part[1024];
    const int tid = (int)threadIdx.x;
    const int chunk = (N + 1023) / 1024;
    const int b = tid * chunk, e = min(N, b + chunk);
    int cnt = 0;
    for (int k = b; k < e; ++k) cnt += root[k] == k;
    part[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - cnt;
    for (int k = b; k < e; ++k) rank[k] = (root[k] == k) ? run++ : -1;
    if (tid == 1023) *S_out = part[1023];
}

__global__ void label_kernel(const int32_t* root, const int32_t* rank, int N, uint16_t* label)
{
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k < N) label[k] = (uint16_t)rank[root[k]];
}

// neighbour relation: LEFT, TOP, TOP-RIGHT contacts (the reference's TOP-LEFT compare reads the pixel's own
// id out of its ring buffer and never registers, OverSegmentation.cpp:174-186), both directions
__global__ void adjacency_kernel(const uint16_t* label, int W, int H, int S, int words, uint32_t* bits)
{
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= W * H) return;
    const int y = k / W, x = k - y * W;
    const uint32_t id = label[k];
    const int dx[3] = { -1, 0, 1 }, dy[3] = { 0, -1, -1 };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int x2 = x + dx[i], y2 = y + dy[i];
        if (x2 < 0 || y2 < 0 || x2 >= W) continue;
        const uint32_t id2 = label[y2 * W + x2];
        if (id2 != id) {
            atomicOr(bits + (size_t)id * words + (id2 >> 5), 1u << (id2 & 31u));
            atomicOr(bits + (size_t)id2 * words + (id >> 5), 1u << (id & 31u));
        }
    }
    (void)S;
}

__global__ void degree_kernel(const uint32_t* bits, int S, int words, uint32_t* deg)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= S) return;
    uint32_t d = 0;
    for (int w = 0; w < words; ++w) d += (uint32_t)__popc(bits[(size_t)i * words + w]);
    deg[i] = d;
}

__global__ void fill_col_kernel(const uint32_t* bits, int S, int words, const uint32_t* row_ptr, uint32_t* col)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= S) return;
    uint32_t e = row_ptr[i];
    for (int w = 0; w < words; ++w) {
        uint32_t m = bits[(size_t)i * words + w];
        while (m) { const int b = __ffs((int)m) - 1; col[e++] = (uint32_t)(w * 32 + b); m &= m - 1u; }
    }
}

// segment_graph on the host (segment-graph.h:47-111; disjoint-set.h:42-79): one scan over edges that are already
// in order.  Same sets and same roots as the reference's universe (union by rank, ties to the second root,
// disjoint-set.h:68-79): find() here halves the path instead of re-linking only the queried element (:58-64), which
// changes how fast a root is reached, not which; a set's threshold, size and rank sit in one record at its root.
struct WEdge { float w; int32_t a, b; };
struct FhNode { float thr; int32_t size, rank; };
// edge(e, a, b, w): endpoints and weight of the e-th edge of the scan
// first(e): first endpoint of the e-th edge (prefetch of its parent link, 16 edges ahead)
template <class EdgeAt, class First>
int fh_scan(int nv, size_t ne, float c, int32_t* root, EdgeAt edge, First first)
{
    std::vector<int32_t> p((size_t)nv);
    std::vector<FhNode> nd((size_t)nv, FhNode{ c / 1, 1, 0 });
    for (int i = 0; i < nv; ++i) p[(size_t)i] = i;
    auto find = [&](int x) { while (p[(size_t)x] != x) { const int g = p[(size_t)p[(size_t)x]]; p[(size_t)x] = g; x = g; } return x; };
    int sets = nv;
    for (size_t e = 0; e < ne; ++e) {
        if (e + 16 < ne) __builtin_prefetch(&p[(size_t)first(e + 16)]);
        int a, b; float w;
        edge(e, a, b, w);
        const int ra = find(a), rb = find(b);
        if (ra == rb) continue;
        FhNode& na = nd[(size_t)ra];
        FhNode& nb = nd[(size_t)rb];
        if (w <= na.thr && w <= nb.thr) {
            --sets;
            int r;
            if (na.rank > nb.rank) { p[(size_t)rb] = ra; na.size += nb.size; r = ra; }
            else { p[(size_t)ra] = rb; nb.size += na.size; if (na.rank == nb.rank) nb.rank++; r = rb; }
            nd[(size_t)r].thr = w + c / (float)(size_t)nd[(size_t)r].size;
        }
    }
    for (int k = 0; k < nv; ++k) root[k] = find(k);
    return sets;
}

// Edges sorted on the device: slot s is the edge of pixel s / 4 in direction s % 4 (RIGHT, BOTTOM, BOTTOM-RIGHT,
// TOP-RIGHT), its weight comes back out of the sort key.
int segment_graph_sorted(int nv, int W, const uint32_t* slot, const uint32_t* key, size_t ne, float c, int32_t* root, long long* n_ties)
{
    long long ties = 0;
    for (size_t e = 1; e < ne; ++e) ties += key[e] == key[e - 1];
    if (n_ties) *n_ties = ties;
    const int off[4] = { 1, W, W + 1, 1 - W };
    return fh_scan(nv, ne, c, root, [&](size_t e, int& a, int& b, float& w) {
        a = (int)(slot[e] >> 2); b = a + off[slot[e] & 3u];
        const uint32_t kb = key[e];
        const uint32_t bits = (kb & 0x80000000u) ? (kb & 0x7FFFFFFFu) : ~kb;
        std::memcpy(&w, &bits, 4);
    }, [&](size_t e) { return (int)(slot[e] >> 2); });
}

// The reference's own order: std::sort on the weight alone (segment-graph.h:35-37,53) over the edges in creation order.
int segment_graph_host(int nv, std::vector<WEdge>& edges, float c, int32_t* root, long long* n_ties)
{
    std::sort(edges.begin(), edges.end(), [](const WEdge& x, const WEdge& y) { return x.w < y.w; });
    long long ties = 0;
    for (size_t e = 1; e < edges.size(); ++e) ties += edges[e].w == edges[e - 1].w;
    if (n_ties) *n_ties = ties;
    return fh_scan(nv, edges.size(), c, root, [&](size_t e, int& a, int& b, float& w) { a = edges[e].a; b = edges[e].b; w = edges[e].w; },
                   [&](size_t e) { return (int)edges[e].a; });
}

}  // namespace

struct mcmc_overseg {
    int device = 0, W = 0, H = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int16_t* d_disp = nullptr; uint8_t* d_colors = nullptr; uint8_t* d_smoothed = nullptr;
    int32_t *d_ea = nullptr, *d_eb = nullptr, *d_root = nullptr, *d_rank = nullptr;
    float* d_ew = nullptr;
    uint16_t* d_label = nullptr;
    double *d_sk = nullptr, *d_gtab = nullptr;
    int* d_status = nullptr;          // [0] filter range flag, [1] S
    uint32_t *d_bits = nullptr, *d_deg = nullptr, *d_rowp = nullptr, *d_col = nullptr;
    size_t bits_cap = 0, col_cap = 0, row_cap = 0;
    // the edge sort: keys and slot indices in and out, rocprim's scratch; the sorted lists and the roots cross the
    // bus through pinned host memory
    uint32_t *d_key = nullptr, *d_key2 = nullptr, *d_idx = nullptr, *d_idx2 = nullptr;
    void* d_sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
    uint32_t *p_key = nullptr, *p_idx = nullptr; int32_t* p_root = nullptr;
    hipEvent_t ev[6] = {};
    int opt_tie_order = 1;            // 1: std::sort on the host, the reference's own call (default); 0: device sort, equal weights in creation order
    long long n_ties = 0;             // edges of the last run whose weight equals their predecessor's in scan order
    // results of the last run (host)
    std::vector<uint32_t> h_rowp, h_col;
    int S = 0; uint32_t e2 = 0;
    bool have = false;
    float ms_filter = 0.f, ms_edges = 0.f, ms_fh = 0.f, ms_graph = 0.f, ms_sort = 0.f, ms_scan = 0.f;
};

namespace {
int ofail(mcmc_overseg* o, int code, const std::string& msg) { if (o) o->err = msg; return code; }
#define OCHK(o, expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return ofail((o), MCMC_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
template <typename T> hipError_t oalloc(T** p, size_t n) { *p = nullptr; return hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T)); }
template <typename T> void ofree(T*& p) { if (p) (void)hipFree(p); p = nullptr; }
}  // namespace

extern "C" {

int mcmc_overseg_create(int device, int width, int height, mcmc_overseg** out)
{
    if (!out) return MCMC_E_BADARG;
    *out = nullptr;
    if (width <= 2 || height <= 2 || width > 4096 || height > 4096) return MCMC_E_BADARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return MCMC_E_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MCMC_E_DEVICE;
    mcmc_overseg* o = new mcmc_overseg();
    o->device = device; o->W = width; o->H = height;
    const size_t N = (size_t)width * height;
    bool ok = hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && oalloc(&o->d_disp, N) == hipSuccess && oalloc(&o->d_colors, N * 3) == hipSuccess &&
         oalloc(&o->d_smoothed, N * 3) == hipSuccess && oalloc(&o->d_ea, N * 4) == hipSuccess &&
         oalloc(&o->d_eb, N * 4) == hipSuccess && oalloc(&o->d_ew, N * 4) == hipSuccess &&
         oalloc(&o->d_root, N) == hipSuccess && oalloc(&o->d_rank, N) == hipSuccess && oalloc(&o->d_label, N) == hipSuccess &&
         oalloc(&o->d_sk, (size_t)kMaxWindow * kMaxWindow) == hipSuccess && oalloc(&o->d_gtab, 4096) == hipSuccess &&
         oalloc(&o->d_status, 4) == hipSuccess &&
         oalloc(&o->d_key, N * 4) == hipSuccess && oalloc(&o->d_key2, N * 4) == hipSuccess &&
         oalloc(&o->d_idx, N * 4) == hipSuccess && oalloc(&o->d_idx2, N * 4) == hipSuccess;
    if (ok) {
        ok = rocprim::radix_sort_pairs(nullptr, o->sort_tmp_bytes, o->d_key, o->d_key2, o->d_idx, o->d_idx2, N * 4, 0, 32, o->stream) == hipSuccess &&
             hipMalloc(&o->d_sort_tmp, std::max<size_t>(o->sort_tmp_bytes, 16)) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void**>(&o->p_key), N * 16, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void**>(&o->p_idx), N * 16, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void**>(&o->p_root), N * 4, hipHostMallocDefault) == hipSuccess;
        for (int k = 0; ok && k < 6; ++k) ok = hipEventCreate(&o->ev[k]) == hipSuccess;
    }
    if (!ok) { mcmc_overseg_destroy(o); return MCMC_E_DEVICE; }
    *out = o;
    return MCMC_OK;
}

int mcmc_overseg_destroy(mcmc_overseg* o)
{
    if (!o) return MCMC_E_BADARG;
    (void)hipSetDevice(o->device);
    if (o->stream) (void)hipStreamSynchronize(o->stream);
    ofree(o->d_disp); ofree(o->d_colors); ofree(o->d_smoothed); ofree(o->d_ea); ofree(o->d_eb); ofree(o->d_ew);
    ofree(o->d_root); ofree(o->d_rank); ofree(o->d_label); ofree(o->d_sk); ofree(o->d_gtab); ofree(o->d_status);
    ofree(o->d_bits); ofree(o->d_deg); ofree(o->d_rowp); ofree(o->d_col);
    ofree(o->d_key); ofree(o->d_key2); ofree(o->d_idx); ofree(o->d_idx2);
    if (o->d_sort_tmp) (void)hipFree(o->d_sort_tmp);
    if (o->p_key) (void)hipHostFree(o->p_key);
    if (o->p_idx) (void)hipHostFree(o->p_idx);
    if (o->p_root) (void)hipHostFree(o->p_root);
    for (int k = 0; k < 6; ++k) if (o->ev[k]) (void)hipEventDestroy(o->ev[k]);
    if (o->stream) (void)hipStreamDestroy(o->stream);
    delete o;
    return MCMC_OK;
}

const char* mcmc_overseg_last_error(const mcmc_overseg* o) { return o ? o->err.c_str() : "null handle"; }

void mcmc_overseg_default_params(mcmc_overseg_params* p)
{
    if (!p) return;
    // output/overSegmentationParams.txt of the reference
    p->window_size = 9; p->spatial_stdev = 5.0; p->disparity_stdev = 3.0; p->weight_threshold = 300;
    p->color_factor = 1.0; p->disparity_factor = 1.0; p->distance_factor = 1.0;
}

int mcmc_overseg_set_option(mcmc_overseg* o, const char* name, int value)
{
    if (!o || !name) return ofail(o, MCMC_E_BADARG, "overseg_set_option: null argument");
    if (std::string(name) == "tie_order" && (value == 0 || value == 1)) { o->opt_tie_order = value; return MCMC_OK; }
    return ofail(o, MCMC_E_BADARG, std::string("overseg_set_option: unknown option or value: ") + name);
}

int mcmc_overseg_run(mcmc_overseg* o, const uint8_t* colors, const int16_t* disp, const mcmc_overseg_params* pp,
                     int* n_superpixels, uint32_t* n_directed_edges)
{
    if (!o || !colors || !disp || !pp) return ofail(o, MCMC_E_BADARG, "overseg_run: null argument");
    const mcmc_overseg_params p = *pp;
    // the reference throws for non-positive deviations (OverSegmentation.cpp:309-310)
    if (!(p.spatial_stdev > 0.0) || !(p.disparity_stdev > 0.0)) return ofail(o, MCMC_E_BADARG, "overseg_run: non-positive standard deviation");
    if (p.window_size < 0 || p.window_size > kMaxWindow || p.window_size >= o->W || p.window_size >= o->H || p.weight_threshold < 0)
        return ofail(o, MCMC_E_BADARG, "overseg_run: bad window size or threshold");
    OCHK(o, hipSetDevice(o->device));
    const int W = o->W, H = o->H, w = p.window_size;
    const size_t N = (size_t)W * H;
    o->have = false;
    hipEvent_t e0 = o->ev[0], e1 = o->ev[1], e2 = o->ev[2], e3 = o->ev[3], e4 = o->ev[4], e5 = o->ev[5];
    OCHK(o, hipMemcpyAsync(o->d_disp, disp, N * 2, hipMemcpyHostToDevice, o->stream));
    OCHK(o, hipMemcpyAsync(o->d_colors, colors, N * 3, hipMemcpyHostToDevice, o->stream));
    OCHK(o, hipMemsetAsync(o->d_status, 0, 4 * sizeof(int), o->stream));
    OCHK(o, hipEventRecord(e0, o->stream));
    // ---- 1. cross-bilateral filter
    if (w < 2) {
        OCHK(o, hipMemcpyAsync(o->d_smoothed, o->d_colors, N * 3, hipMemcpyDeviceToDevice, o->stream));   // :313-316
    } else {
        std::vector<double> sk((size_t)w * w);
        double sum = 0.0;
        const double mid = (1.0 + w) / 2.0;
        for (int i = 1; i <= w; ++i)
            for (int j = 1; j <= w; ++j) { const double v = gauss2(mid - i, mid - j, p.spatial_stdev); sk[(size_t)(i - 1) * w + (j - 1)] = v; sum += v; }
        for (double& v : sk) v = v / sum;
        // Gaussian of integer disparity differences until it is exactly norm * 0
        std::vector<double> gt;
        for (int d = 0; d < 4095; ++d) { const double g = gauss2((double)d, 0.0, p.disparity_stdev); gt.push_back(g); if (g == 0.0) break; }
        if (gt.back() != 0.0) gt.push_back(gauss2(70000.0, 0.0, p.disparity_stdev));
        if (gt.back() != 0.0 && (w % 2) == 1) return ofail(o, MCMC_E_TOO_LARGE, "overseg_run: disparity deviation too large for the table");
        OCHK(o, hipMemcpyAsync(o->d_sk, sk.data(), sk.size() * 8, hipMemcpyHostToDevice, o->stream));
        OCHK(o, hipMemcpyAsync(o->d_gtab, gt.data(), gt.size() * 8, hipMemcpyHostToDevice, o->stream));
        OCHK(o, hipStreamSynchronize(o->stream));   // sk / gt are locals
        OCHK(o, hipMemsetAsync(o->d_smoothed, 0, N * 3, o->stream));
        const int T = kTile + w;
        const size_t lds = ((size_t)w * w + gt.size()) * 8 + (size_t)((T * T + 3) & ~3) * 2 + (size_t)T * T * 3;
        const dim3 grid((unsigned)((W + kTile - 1) / kTile), (unsigned)((H + kTile - 1) / kTile));
        if (w % 2 == 0)
            hipLaunchKernelGGL(bilateral_kernel<true>, grid, dim3(kTile * kTile), lds, o->stream, o->d_disp, o->d_colors, W, H, w,
                               o->d_sk, o->d_gtab, (int)gt.size(), p.disparity_stdev, o->d_smoothed, o->d_status);
        else
            hipLaunchKernelGGL(bilateral_kernel<false>, grid, dim3(kTile * kTile), lds, o->stream, o->d_disp, o->d_colors, W, H, w,
                               o->d_sk, o->d_gtab, (int)gt.size(), p.disparity_stdev, o->d_smoothed, o->d_status);
        OCHK(o, hipGetLastError());
    }
    OCHK(o, hipEventRecord(e1, o->stream));
    // ---- 2. edge weights
    hipLaunchKernelGGL(edge_weight_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, o->stream, o->d_disp, o->d_smoothed, W, H,
                       p.color_factor, p.disparity_factor, p.distance_factor, o->d_ea, o->d_eb, o->d_ew);
    OCHK(o, hipGetLastError());
    OCHK(o, hipEventRecord(e2, o->stream));
    int st[4] = { 0, 0, 0, 0 };
    const auto t0 = std::chrono::steady_clock::now();
    if (o->opt_tie_order == 0) {
        // ---- 3. sort on the device (equal weights stay in creation order), union-find scan on the host
        hipLaunchKernelGGL(edge_key_kernel, dim3((unsigned)((N * 4 + 255) / 256)), dim3(256), 0, o->stream, o->d_ea, o->d_ew,
                           (uint32_t)(N * 4), o->d_key, o->d_idx, o->d_status + 2);
        OCHK(o, hipGetLastError());
        size_t tmp = o->sort_tmp_bytes;
        OCHK(o, rocprim::radix_sort_pairs(o->d_sort_tmp, tmp, o->d_key, o->d_key2, o->d_idx, o->d_idx2, N * 4, 0, 32, o->stream));
        OCHK(o, hipEventRecord(e5, o->stream));
        OCHK(o, hipMemcpyAsync(st, o->d_status, sizeof st, hipMemcpyDeviceToHost, o->stream));
        OCHK(o, hipMemcpyAsync(o->p_key, o->d_key2, N * 16, hipMemcpyDeviceToHost, o->stream));
        OCHK(o, hipMemcpyAsync(o->p_idx, o->d_idx2, N * 16, hipMemcpyDeviceToHost, o->stream));
        OCHK(o, hipStreamSynchronize(o->stream));
        if (st[0]) return ofail(o, MCMC_E_RANGE, "overseg_run: filtered channel value outside 0..255");   // "Weird channel value."
        (void)hipEventElapsedTime(&o->ms_sort, e2, e5);
        const auto t1 = std::chrono::steady_clock::now();
        segment_graph_sorted((int)N, W, o->p_idx, o->p_key, (size_t)st[2], (float)(size_t)p.weight_threshold, o->p_root, &o->n_ties);
        o->ms_scan = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t1).count();
    } else {
        // ---- 3'. the reference's own order of equal weights: std::sort on the host over the edges in creation order
        std::vector<int32_t> ta(N * 4), tb(N * 4);
        std::vector<float> tw(N * 4);
        OCHK(o, hipMemcpyAsync(ta.data(), o->d_ea, N * 16, hipMemcpyDeviceToHost, o->stream));
        OCHK(o, hipMemcpyAsync(tb.data(), o->d_eb, N * 16, hipMemcpyDeviceToHost, o->stream));
        OCHK(o, hipMemcpyAsync(tw.data(), o->d_ew, N * 16, hipMemcpyDeviceToHost, o->stream));
        OCHK(o, hipMemcpyAsync(st, o->d_status, sizeof st, hipMemcpyDeviceToHost, o->stream));
        OCHK(o, hipStreamSynchronize(o->stream));
        if (st[0]) return ofail(o, MCMC_E_RANGE, "overseg_run: filtered channel value outside 0..255");   // "Weird channel value."
        std::vector<WEdge> edges;
        edges.reserve(N * 4);
        for (size_t s = 0; s < N * 4; ++s)
            if (ta[s] >= 0) edges.push_back(WEdge{ tw[s], ta[s], tb[s] });
        const auto t1 = std::chrono::steady_clock::now();
        segment_graph_host((int)N, edges, (float)(size_t)p.weight_threshold, o->p_root, &o->n_ties);
        o->ms_scan = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t1).count();
        o->ms_sort = 0.f;
    }
    o->ms_fh = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // ---- 4. labels and neighbours on the device
    OCHK(o, hipMemcpyAsync(o->d_root, o->p_root, N * 4, hipMemcpyHostToDevice, o->stream));
    OCHK(o, hipEventRecord(e3, o->stream));
    hipLaunchKernelGGL(root_rank_kernel, dim3(1), dim3(1024), 0, o->stream, o->d_root, (int)N, o->d_rank, o->d_status + 1);
    OCHK(o, hipGetLastError());
    OCHK(o, hipMemcpyAsync(st, o->d_status, sizeof st, hipMemcpyDeviceToHost, o->stream));
    OCHK(o, hipStreamSynchronize(o->stream));
    const int S = st[1];
    if (S <= 0 || S >= 0xFFFF) return ofail(o, MCMC_E_TOO_LARGE, "overseg_run: " + std::to_string(S) + " superpixels do not fit 16-bit labels");
    hipLaunchKernelGGL(label_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, o->stream, o->d_root, o->d_rank, (int)N, o->d_label);
    OCHK(o, hipGetLastError());
    const int words = (S + 31) / 32;
    if ((size_t)S * words > o->bits_cap) { ofree(o->d_bits); OCHK(o, oalloc(&o->d_bits, (size_t)S * words)); o->bits_cap = (size_t)S * words; }
    if ((size_t)S + 1 > o->row_cap) { ofree(o->d_deg); ofree(o->d_rowp); OCHK(o, oalloc(&o->d_deg, (size_t)S + 1)); OCHK(o, oalloc(&o->d_rowp, (size_t)S + 1)); o->row_cap = (size_t)S + 1; }
    OCHK(o, hipMemsetAsync(o->d_bits, 0, (size_t)S * words * 4, o->stream));
    hipLaunchKernelGGL(adjacency_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, o->stream, o->d_label, W, H, S, words, o->d_bits);
    OCHK(o, hipGetLastError());
    hipLaunchKernelGGL(degree_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, o->stream, o->d_bits, S, words, o->d_deg);
    OCHK(o, hipGetLastError());
    std::vector<uint32_t> deg((size_t)S);
    OCHK(o, hipMemcpyAsync(deg.data(), o->d_deg, (size_t)S * 4, hipMemcpyDeviceToHost, o->stream));
    OCHK(o, hipStreamSynchronize(o->stream));
    o->h_rowp.assign((size_t)S + 1, 0);
    for (int i = 0; i < S; ++i) o->h_rowp[(size_t)i + 1] = o->h_rowp[(size_t)i] + deg[(size_t)i];
    const uint32_t E2 = o->h_rowp[(size_t)S];
    if ((size_t)E2 > o->col_cap) { ofree(o->d_col); OCHK(o, oalloc(&o->d_col, (size_t)E2)); o->col_cap = E2; }
    OCHK(o, hipMemcpyAsync(o->d_rowp, o->h_rowp.data(), ((size_t)S + 1) * 4, hipMemcpyHostToDevice, o->stream));
    hipLaunchKernelGGL(fill_col_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, o->stream, o->d_bits, S, words, o->d_rowp, o->d_col);
    OCHK(o, hipGetLastError());
    o->h_col.assign((size_t)E2, 0);
    if (E2) OCHK(o, hipMemcpyAsync(o->h_col.data(), o->d_col, (size_t)E2 * 4, hipMemcpyDeviceToHost, o->stream));
    OCHK(o, hipEventRecord(e4, o->stream));
    OCHK(o, hipStreamSynchronize(o->stream));
    (void)hipEventElapsedTime(&o->ms_filter, e0, e1);
    (void)hipEventElapsedTime(&o->ms_edges, e1, e2);
    (void)hipEventElapsedTime(&o->ms_graph, e3, e4);
    o->S = S; o->e2 = E2; o->have = true;
    if (n_superpixels) *n_superpixels = S;
    if (n_directed_edges) *n_directed_edges = E2;
    return MCMC_OK;
}

int mcmc_overseg_get_smoothed(mcmc_overseg* o, uint8_t* out)
{
    if (!o || !out || !o->have) return ofail(o, MCMC_E_STATE, "overseg_get_smoothed: run first");
    OCHK(o, hipSetDevice(o->device));
    OCHK(o, hipMemcpy(out, o->d_smoothed, (size_t)o->W * o->H * 3, hipMemcpyDeviceToHost));
    return MCMC_OK;
}

long long mcmc_overseg_get_edges(mcmc_overseg* o, int32_t* a, int32_t* b, float* w)
{
    if (!o || !o->have) return ofail(o, MCMC_E_STATE, "overseg_get_edges: run first");
    // the slots stay on the device; packed to the reference's creation order only when somebody asks
    const size_t N4 = (size_t)o->W * o->H * 4;
    std::vector<int32_t> ta(N4), tb(N4);
    std::vector<float> tw(N4);
    if (hipSetDevice(o->device) != hipSuccess || hipMemcpy(ta.data(), o->d_ea, N4 * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(tb.data(), o->d_eb, N4 * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(tw.data(), o->d_ew, N4 * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return ofail(o, MCMC_E_DEVICE, "overseg_get_edges: copy failed");
    size_t n = 0;
    for (size_t s = 0; s < N4; ++s)
        if (ta[s] >= 0) {
            if (a) a[n] = ta[s];
            if (b) b[n] = tb[s];
            if (w) w[n] = tw[s];
            ++n;
        }
    return (long long)n;
}

int mcmc_overseg_get_roots(mcmc_overseg* o, int32_t* out)
{
    if (!o || !out || !o->have) return ofail(o, MCMC_E_STATE, "overseg_get_roots: run first");
    std::memcpy(out, o->p_root, (size_t)o->W * o->H * 4);
    return MCMC_OK;
}

int mcmc_overseg_get_labels(mcmc_overseg* o, uint16_t* out)
{
    if (!o || !out || !o->have) return ofail(o, MCMC_E_STATE, "overseg_get_labels: run first");
    OCHK(o, hipSetDevice(o->device));
    OCHK(o, hipMemcpy(out, o->d_label, (size_t)o->W * o->H * 2, hipMemcpyDeviceToHost));
    return MCMC_OK;
}

int mcmc_overseg_get_graph(mcmc_overseg* o, uint32_t* row_ptr, uint32_t* col)
{
    if (!o || !o->have) return ofail(o, MCMC_E_STATE, "overseg_get_graph: run first");
    if (row_ptr) std::memcpy(row_ptr, o->h_rowp.data(), o->h_rowp.size() * 4);
    if (col) std::memcpy(col, o->h_col.data(), o->h_col.size() * 4);
    return MCMC_OK;
}

int mcmc_overseg_device_buffers(mcmc_overseg* o, const void** d_disp, const void** d_label)
{
    if (!o || !o->have) return ofail(o, MCMC_E_STATE, "overseg_device_buffers: run first");
    if (d_disp) *d_disp = o->d_disp;
    if (d_label) *d_label = o->d_label;
    return MCMC_OK;
}

int mcmc_overseg_last_ms(mcmc_overseg* o, float* filter_ms, float* edges_ms, float* union_find_ms, float* graph_ms)
{
    if (!o || !o->have) return ofail(o, MCMC_E_STATE, "overseg_last_ms: run first");
    if (filter_ms) *filter_ms = o->ms_filter;
    if (edges_ms) *edges_ms = o->ms_edges;
    if (union_find_ms) *union_find_ms = o->ms_fh;
    if (graph_ms) *graph_ms = o->ms_graph;
    return MCMC_OK;
}

long long mcmc_overseg_last_tie_count(mcmc_overseg* o)
{
    if (!o || !o->have) return ofail(o, MCMC_E_STATE, "overseg_last_tie_count: run first");
    return o->n_ties;
}

int mcmc_overseg_last_sort_ms(mcmc_overseg* o, float* device_sort_ms, float* host_scan_ms)
{
    if (!o || !o->have) return ofail(o, MCMC_E_STATE, "overseg_last_sort_ms: run first");
    if (device_sort_ms) *device_sort_ms = o->ms_sort;
    if (host_scan_ms) *host_scan_ms = o->ms_scan;
    return MCMC_OK;
}

// ---------------------------------------------------------------------------
// A batch of frames: OverSegmentation::overSegment is per frame and its sort + union-find scan run on a host thread,
// so a C5-sized job (256 frames) wants one frame per host core.  The batch owns n_workers handles (each with its own
// stream and buffers) and as many threads; results of every frame are kept until the next run: the device planes in
// one allocation (they go to mcmc_frame_upload_device as they are), the neighbour CSR on the host.
// ---------------------------------------------------------------------------
struct mcmc_overseg_batch {
    int device = 0, W = 0, H = 0;
    std::vector<mcmc_overseg*> workers;
    std::string err;
    int16_t* d_disp = nullptr; uint16_t* d_label = nullptr; size_t cap_frames = 0;
    std::vector<std::vector<uint32_t>> rowp, col;
    std::vector<int> S;
    std::vector<long long> ties;
    int n_frames = 0;
    float wall_ms = 0.f;
};

namespace {
int bfail(mcmc_overseg_batch* b, int code, const std::string& msg) { if (b) b->err = msg; return code; }
}

int mcmc_overseg_batch_create(int device, int width, int height, int n_workers, mcmc_overseg_batch** out)
{
    if (!out || n_workers < 1 || n_workers > 64) return MCMC_E_BADARG;
    *out = nullptr;
    mcmc_overseg_batch* b = new mcmc_overseg_batch();
    b->device = device; b->W = width; b->H = height;
    for (int k = 0; k < n_workers; ++k) {
        mcmc_overseg* w = nullptr;
        const int rc = mcmc_overseg_create(device, width, height, &w);
        if (rc != MCMC_OK) { for (mcmc_overseg* x : b->workers) mcmc_overseg_destroy(x); delete b; return rc; }
        b->workers.push_back(w);
    }
    *out = b;
    return MCMC_OK;
}

int mcmc_overseg_batch_destroy(mcmc_overseg_batch* b)
{
    if (!b) return MCMC_E_BADARG;
    (void)hipSetDevice(b->device);
    for (mcmc_overseg* w : b->workers) mcmc_overseg_destroy(w);
    ofree(b->d_disp); ofree(b->d_label);
    delete b;
    return MCMC_OK;
}

const char* mcmc_overseg_batch_last_error(const mcmc_overseg_batch* b) { return b ? b->err.c_str() : "null handle"; }

int mcmc_overseg_batch_set_option(mcmc_overseg_batch* b, const char* name, int value)
{
    if (!b) return MCMC_E_BADARG;
    for (mcmc_overseg* w : b->workers) {
        const int rc = mcmc_overseg_set_option(w, name, value);
        if (rc != MCMC_OK) return bfail(b, rc, mcmc_overseg_last_error(w));
    }
    return MCMC_OK;
}

int mcmc_overseg_batch_run(mcmc_overseg_batch* b, int n_frames, const uint8_t* colors, const int16_t* disp,
                           const mcmc_overseg_params* p, int* n_superpixels, uint32_t* n_directed_edges)
{
    if (!b || n_frames < 1 || !colors || !disp || !p) return bfail(b, MCMC_E_BADARG, "overseg_batch_run: bad argument");
    if (hipSetDevice(b->device) != hipSuccess) return bfail(b, MCMC_E_DEVICE, "overseg_batch_run: hipSetDevice failed");
    const size_t N = (size_t)b->W * b->H;
    b->n_frames = 0;
    if ((size_t)n_frames > b->cap_frames) {
        ofree(b->d_disp); ofree(b->d_label); b->cap_frames = 0;
        if (oalloc(&b->d_disp, N * n_frames) != hipSuccess || oalloc(&b->d_label, N * n_frames) != hipSuccess) {
            ofree(b->d_disp); ofree(b->d_label);
            return bfail(b, MCMC_E_DEVICE, "overseg_batch_run: no device memory for the frames' planes");
        }
        b->cap_frames = (size_t)n_frames;
    }
    b->rowp.assign((size_t)n_frames, {}); b->col.assign((size_t)n_frames, {});
    b->S.assign((size_t)n_frames, 0); b->ties.assign((size_t)n_frames, 0);
    std::atomic<int> next(0), failed(0);
    std::mutex mu;
    int first_rc = MCMC_OK;
    std::string first_err;
    const auto t0 = std::chrono::steady_clock::now();
    auto work = [&](mcmc_overseg* w) {
        (void)hipSetDevice(b->device);
        for (;;) {
            const int f = next.fetch_add(1);
            if (f >= n_frames || failed.load()) return;
            int S = 0; uint32_t e2 = 0;
            int rc = mcmc_overseg_run(w, colors + (size_t)f * N * 3, disp + (size_t)f * N, p, &S, &e2);
            std::string msg = rc == MCMC_OK ? std::string() : std::string(mcmc_overseg_last_error(w));
            if (rc == MCMC_OK) {
                b->S[(size_t)f] = S; b->ties[(size_t)f] = w->n_ties;
                b->rowp[(size_t)f] = w->h_rowp; b->col[(size_t)f] = w->h_col;
                hipError_t e = hipMemcpyAsync(b->d_disp + (size_t)f * N, w->d_disp, N * 2, hipMemcpyDeviceToDevice, w->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(b->d_label + (size_t)f * N, w->d_label, N * 2, hipMemcpyDeviceToDevice, w->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
                if (e != hipSuccess) { rc = MCMC_E_DEVICE; msg = std::string("overseg_batch_run: ") + hipGetErrorString(e); }
            }
            if (rc != MCMC_OK) {
                std::lock_guard<std::mutex> g(mu);
                if (first_rc == MCMC_OK) { first_rc = rc; first_err = "frame " + std::to_string(f) + ": " + msg; }
                failed.store(1);
                return;
            }
        }
    };
    const int nt = std::min<int>((int)b->workers.size(), n_frames);
    std::vector<std::thread> th;
    for (int k = 1; k < nt; ++k) th.emplace_back(work, b->workers[(size_t)k]);
    work(b->workers[0]);                       // the calling thread is the first worker
    for (std::thread& t : th) t.join();
    b->wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (first_rc != MCMC_OK) return bfail(b, first_rc, first_err);
    b->n_frames = n_frames;
    for (int f = 0; f < n_frames; ++f) {
        if (n_superpixels) n_superpixels[f] = b->S[(size_t)f];
        if (n_directed_edges) n_directed_edges[f] = (uint32_t)b->col[(size_t)f].size();
    }
    return MCMC_OK;
}

int mcmc_overseg_batch_get_graph(mcmc_overseg_batch* b, int frame, uint32_t* row_ptr, uint32_t* col)
{
    if (!b || frame < 0 || frame >= b->n_frames) return bfail(b, MCMC_E_STATE, "overseg_batch_get_graph: no such frame in the last run");
    if (row_ptr) std::memcpy(row_ptr, b->rowp[(size_t)frame].data(), b->rowp[(size_t)frame].size() * 4);
    if (col) std::memcpy(col, b->col[(size_t)frame].data(), b->col[(size_t)frame].size() * 4);
    return MCMC_OK;
}

int mcmc_overseg_batch_device_buffers(mcmc_overseg_batch* b, int frame, const void** d_disp, const void** d_label)
{
    if (!b || frame < 0 || frame >= b->n_frames) return bfail(b, MCMC_E_STATE, "overseg_batch_device_buffers: no such frame in the last run");
    const size_t N = (size_t)b->W * b->H;
    if (d_disp) *d_disp = b->d_disp + (size_t)frame * N;
    if (d_label) *d_label = b->d_label + (size_t)frame * N;
    return MCMC_OK;
}

int mcmc_overseg_batch_get_labels(mcmc_overseg_batch* b, int frame, uint16_t* out)
{
    if (!b || !out || frame < 0 || frame >= b->n_frames) return bfail(b, MCMC_E_STATE, "overseg_batch_get_labels: no such frame in the last run");
    const size_t N = (size_t)b->W * b->H;
    if (hipSetDevice(b->device) != hipSuccess || hipMemcpy(out, b->d_label + (size_t)frame * N, N * 2, hipMemcpyDeviceToHost) != hipSuccess)
        return bfail(b, MCMC_E_DEVICE, "overseg_batch_get_labels: copy failed");
    return MCMC_OK;
}

long long mcmc_overseg_batch_tie_count(mcmc_overseg_batch* b, int frame)
{
    if (!b || frame < 0 || frame >= b->n_frames) return bfail(b, MCMC_E_STATE, "overseg_batch_tie_count: no such frame in the last run");
    return b->ties[(size_t)frame];
}

float mcmc_overseg_batch_last_wall_ms(const mcmc_overseg_batch* b) { return b ? b->wall_ms : -1.f; }

}  // extern "C"
