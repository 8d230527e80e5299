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
// integers); the union-find pass is a sequential scan over sorted edges and runs on the host, as in the
// reference (std::sort on weight alone, so equal weights fall as this standard library leaves them,
// which is what the reference's result depends on too); labels and the neighbour relation are built on
// the device from the roots (rank of root pixels, S x S bit matrix, CSR by row).
// Restates behaviour, not code; every quirk kept is listed in DESIGN.md.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
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

// segment_graph on the host (segment-graph.h:47-111; disjoint-set.h:42-79)
struct WEdge { float w; int32_t a, b; };
struct Forest {
    std::vector<int32_t> parent, rank, size;
    explicit Forest(int n) : parent((size_t)n), rank((size_t)n, 0), size((size_t)n, 1) { for (int i = 0; i < n; ++i) parent[(size_t)i] = i; }
    int find(int x) { int y = x; while (y != parent[(size_t)y]) y = parent[(size_t)y]; parent[(size_t)x] = y; return y; }
    void join(int x, int y)
    {
        if (rank[(size_t)x] > rank[(size_t)y]) { parent[(size_t)y] = x; size[(size_t)x] += size[(size_t)y]; }
        else { parent[(size_t)x] = y; size[(size_t)y] += size[(size_t)x]; if (rank[(size_t)x] == rank[(size_t)y]) rank[(size_t)y]++; }
    }
};

int segment_graph_host(int nv, std::vector<WEdge>& edges, float c, std::vector<int32_t>& root)
{
    std::sort(edges.begin(), edges.end(), [](const WEdge& x, const WEdge& y) { return x.w < y.w; });
    Forest f(nv);
    std::vector<float> thr((size_t)nv, c / 1);
    int sets = nv;
    for (const WEdge& e : edges) {
        const int ra = f.find(e.a), rb = f.find(e.b);
        if (ra == rb) continue;
        if (e.w <= thr[(size_t)ra] && e.w <= thr[(size_t)rb]) {
            f.join(ra, rb);
            --sets;
            const int r = f.find(ra);
            thr[(size_t)r] = e.w + c / (float)(size_t)f.size[(size_t)r];
        }
    }
    root.resize((size_t)nv);
    for (int k = 0; k < nv; ++k) root[(size_t)k] = f.find(k);
    return sets;
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
    // results of the last run (host)
    std::vector<int32_t> h_ea, h_eb, h_root;
    std::vector<float> h_ew;
    std::vector<uint32_t> h_rowp, h_col;
    int S = 0; uint32_t e2 = 0;
    bool have = false;
    float ms_filter = 0.f, ms_edges = 0.f, ms_fh = 0.f, ms_graph = 0.f;
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
         oalloc(&o->d_status, 4) == hipSuccess;
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
    hipEvent_t e0, e1, e2, e3, e4;
    OCHK(o, hipEventCreate(&e0)); OCHK(o, hipEventCreate(&e1)); OCHK(o, hipEventCreate(&e2));
    OCHK(o, hipEventCreate(&e3)); OCHK(o, hipEventCreate(&e4));
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
    std::vector<int32_t> ta(N * 4), tb(N * 4);
    std::vector<float> tw(N * 4);
    OCHK(o, hipMemcpyAsync(ta.data(), o->d_ea, N * 16, hipMemcpyDeviceToHost, o->stream));
    OCHK(o, hipMemcpyAsync(tb.data(), o->d_eb, N * 16, hipMemcpyDeviceToHost, o->stream));
    OCHK(o, hipMemcpyAsync(tw.data(), o->d_ew, N * 16, hipMemcpyDeviceToHost, o->stream));
    int st[4] = { 0, 0, 0, 0 };
    OCHK(o, hipMemcpyAsync(st, o->d_status, sizeof st, hipMemcpyDeviceToHost, o->stream));
    OCHK(o, hipStreamSynchronize(o->stream));
    if (st[0]) return ofail(o, MCMC_E_RANGE, "overseg_run: filtered channel value outside 0..255");   // "Weird channel value."
    // ---- 3. union-find on the host, edges in the reference's creation order
    std::vector<WEdge> edges;
    edges.reserve(N * 4);
    o->h_ea.clear(); o->h_eb.clear(); o->h_ew.clear();
    for (size_t s = 0; s < N * 4; ++s)
        if (ta[s] >= 0) { edges.push_back(WEdge{ tw[s], ta[s], tb[s] }); o->h_ea.push_back(ta[s]); o->h_eb.push_back(tb[s]); o->h_ew.push_back(tw[s]); }
    const auto t0 = std::chrono::steady_clock::now();
    segment_graph_host((int)N, edges, (float)(size_t)p.weight_threshold, o->h_root);
    o->ms_fh = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // ---- 4. labels and neighbours on the device
    OCHK(o, hipMemcpyAsync(o->d_root, o->h_root.data(), N * 4, hipMemcpyHostToDevice, o->stream));
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
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); (void)hipEventDestroy(e3); (void)hipEventDestroy(e4);
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
    const size_t n = o->h_ea.size();
    if (a) std::memcpy(a, o->h_ea.data(), n * 4);
    if (b) std::memcpy(b, o->h_eb.data(), n * 4);
    if (w) std::memcpy(w, o->h_ew.data(), n * 4);
    return (long long)n;
}

int mcmc_overseg_get_roots(mcmc_overseg* o, int32_t* out)
{
    if (!o || !out || !o->have) return ofail(o, MCMC_E_STATE, "overseg_get_roots: run first");
    std::memcpy(out, o->h_root.data(), o->h_root.size() * 4);
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

}  // extern "C"
