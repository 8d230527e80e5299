// capi_multiview.hip -- C ABI of Path B (include/mcmc_amd.h, mcmc_mv_*): the SPGraph resident
// in HBM (points, per-vertex moments, CSR edges) and per-chain segmentation state.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/mcmc_amd.h"
#include "kernels.hpp"

using namespace mc;

struct mcmc_mv {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int V = 0; uint32_t E2 = 0; size_t P = 0;
    float4* d_pts = nullptr; uint32_t* d_ptptr = nullptr; double* d_vm = nullptr;
    uint32_t *d_rowp = nullptr, *d_col = nullptr; uint2* d_et = nullptr; double* d_q = nullptr;
    unsigned long long* d_abm = nullptr; uint32_t* d_wbase = nullptr;   // bitmap form of the adjacency
    double* d_ptsd = nullptr; uint32_t* d_vorder = nullptr;                               // points as doubles in wave order
    uint8_t* d_mem = nullptr; double* d_out = nullptr;
    bool have_graph = false, have_vm = false;
    // chains
    int n = 0, seg_cap = 0; uint32_t seed = 0, offset = 0; double split_prob = 0.10;
    unsigned long long step = 0, last_steps = 0, trace_cap = 0;
    uint16_t *cur = nullptr, *own = nullptr; double *plane = nullptr, *segP = nullptr; uint8_t* pvalid = nullptr;
    int *nseg = nullptr, *status = nullptr;
    unsigned long long *n_accept = nullptr, *n_split = nullptr, *n_skip = nullptr, *trace_hash = nullptr;
    uint8_t* trace = nullptr; double* trace_a = nullptr;
    // split step: exchange buffers between the two halves of an iteration and the error kernel
    void* ex_sh = nullptr; uint8_t *ex_inE = nullptr, *ex_need = nullptr; double* ex_err = nullptr;
    uint16_t* ex_plist = nullptr; uint4* ex_queue = nullptr; uint32_t* ex_counters = nullptr;
    int opt_split = -1;   // -1 automatic (split from 32 chains on), 0 fused, 1 split
    int opt_fuse_halves = -1;   // second half of a step + first half of the next as one kernel: 1 yes, 0 no, -1 unless the growth kernel runs apart
    int opt_split_growth = -1;  // the first half as a growth kernel (eight waves per chain) + a plane-fit kernel: 1 yes, 0 no, -1 when at most 256 chains run (from there on the fused halves win)
                                // (C4, iteration: 92 -> 83 us for one chain, 146 -> 135 at 256, 159 -> 156 at 512; first half 120 -> 130 at 1024, where the CUs are full either way)
    int opt_groups = 0;   // groups of chains, each running its iterations on a stream of its own: 0 automatic, 1..kMaxGroups
    int ex_groups = 0;    // the group count the exchange buffers were carved for
    static constexpr int kMaxGroups = 8;
    hipStream_t gstream[kMaxGroups] = {};
    hipEvent_t gev_start = nullptr, gev_done[kMaxGroups] = {};
    // growth a step ahead (launch_mv_chain_run_ahead): a stream for the growth kernels, two hand-over slots, events
    int opt_ahead = -1;   // 1 on, 0 off, -1 automatic: up to 3072 chains, unless "groups", "split_growth" or "fuse_halves" are set by hand
    int opt_ahead_waves = 0;   // waves per chain of the growth kernel that runs ahead: 4, 8, 0 automatic
    hipStream_t ahead_stream = nullptr;
    MvAheadBufs ahead{};
    int ahead_n = 0;      // the chain count the slots were sized for
    long long ahead_ready_step = -1;   // the step whose growth the last run left in its slot for the next run (-1: none)
    int opt_prof = 0;
    unsigned long long* d_prof = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr; bool ev_valid = false;
};

namespace {

int mfail(mcmc_mv* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }

#define MVCHK(c, expr)                                                                           \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return mfail((c), MCMC_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

template <typename T> hipError_t mvalloc(T** p, size_t n)
{
    *p = nullptr;
    return hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T));
}
template <typename T> void mvfree(T*& p) { if (p) (void)hipFree(p); p = nullptr; }

void free_graph(mcmc_mv* c)
{
    mvfree(c->d_pts); mvfree(c->d_ptptr); mvfree(c->d_vm); mvfree(c->d_rowp); mvfree(c->d_col);
    mvfree(c->d_et); mvfree(c->d_q); mvfree(c->d_mem); mvfree(c->d_abm); mvfree(c->d_wbase);
    mvfree(c->d_ptsd); mvfree(c->d_vorder);
    c->have_graph = c->have_vm = false;
}
void free_chains(mcmc_mv* c)
{
    mvfree(c->cur); mvfree(c->own); mvfree(c->plane); mvfree(c->segP); mvfree(c->pvalid); mvfree(c->nseg);
    mvfree(c->status); mvfree(c->n_accept); mvfree(c->n_split); mvfree(c->n_skip); mvfree(c->trace_hash);
    mvfree(c->trace); mvfree(c->trace_a);
    { uint8_t* p = static_cast<uint8_t*>(c->ex_sh); mvfree(p); c->ex_sh = nullptr; }
    mvfree(c->ex_inE); mvfree(c->ex_need); mvfree(c->ex_err); mvfree(c->d_prof);
    mvfree(c->ex_plist); mvfree(c->ex_queue); mvfree(c->ex_counters);
    c->ex_groups = 0;
    for (int k = 0; k < 2; ++k) { mvfree(c->ahead.inE[k]); uint8_t* p = static_cast<uint8_t*>(c->ahead.sh[k]); mvfree(p); c->ahead.sh[k] = nullptr; }
    c->ahead_n = 0; c->ahead_ready_step = -1;
    c->n = 0; c->trace_cap = 0; c->last_steps = 0;
}
MvGraph graph_of(const mcmc_mv* c)
{
    MvGraph g{};
    g.V = c->V; g.pts = c->d_pts; g.pt_ptr = c->d_ptptr; g.vm = c->d_vm; g.rowp = c->d_rowp; g.col = c->d_col;
    g.q = c->d_q; g.et = c->d_et;
    g.W64 = (c->V + 63) / 64; g.abm = c->d_abm; g.wbase = c->d_wbase;
    g.ptsd = c->d_ptsd; g.vorder = c->d_vorder;
    return g;
}
MvChainArgs chain_args_of(const mcmc_mv* c)
{
    MvChainArgs a{};
    a.n_chains = c->n; a.seg_cap = c->seg_cap; a.seed = c->seed; a.chain_offset = c->offset;
    a.split_prob = c->split_prob; a.cur = c->cur; a.own = c->own; a.plane = c->plane; a.segP = c->segP;
    a.segP_valid = c->pvalid; a.nseg = c->nseg; a.status = c->status; a.n_accept = c->n_accept;
    a.n_split = c->n_split; a.n_skip = c->n_skip; a.trace_hash = c->trace_hash;
    return a;
}

}  // namespace

extern "C" {

int mcmc_mv_create(int device, mcmc_mv** out)
{
    if (!out) return MCMC_E_BADARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return MCMC_E_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MCMC_E_DEVICE;
    mcmc_mv* c = new mcmc_mv();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        mvalloc(&c->d_out, 8) != hipSuccess) {
        delete c;
        return MCMC_E_DEVICE;
    }
    *out = c;
    return MCMC_OK;
}

int mcmc_mv_destroy(mcmc_mv* c)
{
    if (!c) return MCMC_E_BADARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_chains(c); free_graph(c); mvfree(c->d_out);
    (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1);
    if (c->gev_start) (void)hipEventDestroy(c->gev_start);
    if (c->ahead_stream) { (void)hipStreamSynchronize(c->ahead_stream); (void)hipStreamDestroy(c->ahead_stream); }
    for (int k = 0; k < 2; ++k) {
        if (c->ahead.ev_g[k]) (void)hipEventDestroy(c->ahead.ev_g[k]);
        if (c->ahead.ev_c[k]) (void)hipEventDestroy(c->ahead.ev_c[k]);
    }
    if (c->ahead.ev_start) (void)hipEventDestroy(c->ahead.ev_start);
    for (int k = 0; k < mcmc_mv::kMaxGroups; ++k) {
        if (c->gev_done[k]) (void)hipEventDestroy(c->gev_done[k]);
        if (c->gstream[k]) { (void)hipStreamSynchronize(c->gstream[k]); (void)hipStreamDestroy(c->gstream[k]); }
    }
    (void)hipStreamDestroy(c->stream);
    delete c;
    return MCMC_OK;
}

const char* mcmc_mv_last_error(const mcmc_mv* c) { return c ? c->err.c_str() : "null context"; }

int mcmc_mv_synchronize(mcmc_mv* c)
{
    if (!c) return MCMC_E_BADARG;
    MVCHK(c, hipSetDevice(c->device));
    MVCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_mv_set_graph(mcmc_mv* c, int V, const uint32_t* pt_ptr, const float* pts, const uint32_t* rowp,
                      const uint32_t* col, const double* q, uint32_t e2)
{
    if (!c || V <= 0 || V >= 0xFFFF || !pt_ptr || !pts || !rowp || (e2 && (!col || !q)))
        return mfail(c, MCMC_E_BADARG, "mv_set_graph: bad argument");
    if (pt_ptr[0] != 0 || rowp[0] != 0 || rowp[V] != e2) return mfail(c, MCMC_E_BADARG, "mv_set_graph: offsets do not span the arrays");
    // the whole of row_ptr first: col is only read once every row is known to lie inside [0, e2]
    for (int v = 0; v < V; ++v) {
        if (pt_ptr[v + 1] <= pt_ptr[v]) return mfail(c, MCMC_E_BADARG, "mv_set_graph: every vertex needs at least one point");
        if (rowp[v + 1] < rowp[v] || rowp[v + 1] > e2) return mfail(c, MCMC_E_BADARG, "mv_set_graph: row_ptr not monotone");
    }
    for (int v = 0; v < V; ++v) {
        for (uint32_t e = rowp[v]; e < rowp[v + 1]; ++e) {
            if (col[e] >= (uint32_t)V || col[e] == (uint32_t)v) return mfail(c, MCMC_E_BADARG, "mv_set_graph: bad neighbour id");
            if (e > rowp[v] && col[e] <= col[e - 1]) return mfail(c, MCMC_E_BADARG, "mv_set_graph: neighbours must ascend");
            if (!(q[e] >= 0.0)) return mfail(c, MCMC_E_BADARG, "mv_set_graph: BernoulliProbability must be a probability");
        }
    }
    if (mv_chain_lds_bytes(V) > 160 * 1024 - 64) return mfail(c, MCMC_E_TOO_LARGE, "mv_set_graph: per-chain state does not fit 160 KB of LDS");
    // rank of every undirected edge among the (i < j) entries; the mirror entry shares it
    std::vector<uint32_t> erank(std::max<uint32_t>(e2, 1));
    uint32_t r = 0;
    for (int i = 0; i < V; ++i)
        for (uint32_t e = rowp[i]; e < rowp[i + 1]; ++e)
            if ((uint32_t)i < col[e]) erank[e] = r++;
    for (int i = 0; i < V; ++i)
        for (uint32_t e = rowp[i]; e < rowp[i + 1]; ++e)
            if ((uint32_t)i > col[e]) {
                const uint32_t j = col[e];
                const uint32_t* b = col + rowp[j];
                const uint32_t* en = col + rowp[j + 1];
                const uint32_t* it = std::lower_bound(b, en, (uint32_t)i);
                if (it == en || *it != (uint32_t)i) return mfail(c, MCMC_E_BADARG, "mv_set_graph: adjacency is not symmetric");
                erank[e] = erank[(size_t)(it - col)];
            }
    // the growth scan's view of the graph: a bitmap row per vertex, the entry index at which each 64-bit word's
    // edges start (an edge's index is that plus the number of lower bits in its word: columns ascend), and the
    // Bernoulli trial as an integer threshold: x / 4294967295.0 is monotone in x, so `u <= q`
    // (MultiviewSegmentation.cpp:23) holds exactly for the draws x <= thr
    const size_t W64 = ((size_t)V + 63) / 64;
    std::vector<unsigned long long> abm((size_t)V * W64, 0ull);
    std::vector<uint32_t> wbase((size_t)V * W64, 0u);
    std::vector<uint2> et(std::max<uint32_t>(e2, 1));
    for (int v = 0; v < V; ++v) {
        uint32_t e = rowp[v];
        for (size_t w = 0; w < W64; ++w) {
            wbase[(size_t)v * W64 + w] = e;
            while (e < rowp[v + 1] && (col[e] >> 6) == w) { abm[(size_t)v * W64 + w] |= 1ull << (col[e] & 63u); ++e; }
        }
    }
    for (uint32_t e = 0; e < e2; ++e) {
        const double qe = q[e];
        uint32_t x = 0xFFFFFFFFu;
        if (qe < 1.0) {
            x = (uint32_t)(qe * 4294967295.0);
            while (x > 0u && (double)x / 4294967295.0 > qe) --x;
            while (x < 0xFFFFFFFFu && (double)(x + 1u) / 4294967295.0 <= qe) ++x;
        }
        et[e].x = erank[e]; et[e].y = x;
    }
    // the L1 pass's view of the points (mv_errors_kernel): doubles (x, y, z, 1), and the vertices in the order their
    // items are queued
    const size_t P = pt_ptr[V];
    std::vector<double> ptsd((P + 16) * 4, 0.0);
    for (size_t i = 0; i < P; ++i) {
        ptsd[i * 4] = (double)pts[i * 4]; ptsd[i * 4 + 1] = (double)pts[i * 4 + 1]; ptsd[i * 4 + 2] = (double)pts[i * 4 + 2];
        ptsd[i * 4 + 3] = 1.0;
    }
    std::vector<uint32_t> vorder((size_t)V);
    for (int v = 0; v < V; ++v) vorder[(size_t)v] = (uint32_t)v;
    // by descending cost of a full item (blocks x groups of pairs, mv_err_groups_per_item), then by point count: the L1
    // kernel's waves take the queue in this order, longest items first
    auto item_cost = [&](uint32_t v) { const uint32_t n = pt_ptr[v + 1] - pt_ptr[v]; return ((n + 15u) >> 4) * mv_err_groups_per_item(n); };
    std::stable_sort(vorder.begin(), vorder.end(), [&](uint32_t a, uint32_t b2) {
        const uint32_t ca = item_cost(a), cb = item_cost(b2);
        return ca != cb ? ca > cb : pt_ptr[a + 1] - pt_ptr[a] > pt_ptr[b2 + 1] - pt_ptr[b2]; });
    MVCHK(c, hipSetDevice(c->device));
    MVCHK(c, hipStreamSynchronize(c->stream));
    free_chains(c); free_graph(c);
    MVCHK(c, mvalloc(&c->d_ptsd, (P + 16) * 4)); MVCHK(c, mvalloc(&c->d_vorder, (size_t)V));
    MVCHK(c, hipMemcpyAsync(c->d_ptsd, ptsd.data(), (P + 16) * 32, hipMemcpyHostToDevice, c->stream));
    MVCHK(c, hipMemcpyAsync(c->d_vorder, vorder.data(), (size_t)V * 4, hipMemcpyHostToDevice, c->stream));
    MVCHK(c, mvalloc(&c->d_abm, (size_t)V * W64)); MVCHK(c, mvalloc(&c->d_wbase, (size_t)V * W64)); MVCHK(c, mvalloc(&c->d_et, e2));
    MVCHK(c, hipMemcpyAsync(c->d_abm, abm.data(), (size_t)V * W64 * 8, hipMemcpyHostToDevice, c->stream));
    MVCHK(c, hipMemcpyAsync(c->d_wbase, wbase.data(), (size_t)V * W64 * 4, hipMemcpyHostToDevice, c->stream));
    if (e2) MVCHK(c, hipMemcpyAsync(c->d_et, et.data(), (size_t)e2 * 8, hipMemcpyHostToDevice, c->stream));
    MVCHK(c, mvalloc(&c->d_pts, P)); MVCHK(c, mvalloc(&c->d_ptptr, (size_t)V + 1)); MVCHK(c, mvalloc(&c->d_vm, (size_t)V * 10));
    MVCHK(c, mvalloc(&c->d_rowp, (size_t)V + 1)); MVCHK(c, mvalloc(&c->d_col, e2));
    MVCHK(c, mvalloc(&c->d_q, e2)); MVCHK(c, mvalloc(&c->d_mem, (size_t)V));
    MVCHK(c, hipMemcpyAsync(c->d_pts, pts, P * 16, hipMemcpyHostToDevice, c->stream));
    MVCHK(c, hipMemcpyAsync(c->d_ptptr, pt_ptr, ((size_t)V + 1) * 4, hipMemcpyHostToDevice, c->stream));
    MVCHK(c, hipMemcpyAsync(c->d_rowp, rowp, ((size_t)V + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (e2) {
        MVCHK(c, hipMemcpyAsync(c->d_col, col, (size_t)e2 * 4, hipMemcpyHostToDevice, c->stream));
        MVCHK(c, hipMemcpyAsync(c->d_q, q, (size_t)e2 * 8, hipMemcpyHostToDevice, c->stream));
    }
    MVCHK(c, hipStreamSynchronize(c->stream));
    c->V = V; c->E2 = e2; c->P = P; c->have_graph = true;
    return MCMC_OK;
}

int mcmc_mv_vertex_moments(mcmc_mv* c)
{
    if (!c || !c->have_graph) return mfail(c, MCMC_E_STATE, "mv_vertex_moments: set the graph first");
    MVCHK(c, hipSetDevice(c->device));
    MVCHK(c, launch_mv_vertex_moments(graph_of(c), c->d_vm, c->stream));
    c->have_vm = true;
    return MCMC_OK;
}

int mcmc_mv_get_vertex_moments(mcmc_mv* c, double* out)
{
    if (!c || !out) return mfail(c, MCMC_E_BADARG, "mv_get_vertex_moments: bad argument");
    if (!c->have_vm) return mfail(c, MCMC_E_STATE, "mv_get_vertex_moments: moments not computed");
    MVCHK(c, hipSetDevice(c->device));
    MVCHK(c, hipMemcpyAsync(out, c->d_vm, (size_t)c->V * 80, hipMemcpyDeviceToHost, c->stream));
    MVCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_mv_fit_segment(mcmc_mv* c, const uint32_t* members, int n, double* plane5, double* prob, double* lerr)
{
    if (!c || !members || n <= 0) return mfail(c, MCMC_E_BADARG, "mv_fit_segment: bad argument");
    if (!c->have_vm) return mfail(c, MCMC_E_STATE, "mv_fit_segment: vertex moments not computed");
    std::vector<uint8_t> mem((size_t)c->V, 0);
    for (int k = 0; k < n; ++k) {
        if (members[k] >= (uint32_t)c->V) return mfail(c, MCMC_E_BADARG, "mv_fit_segment: vertex id out of range");
        mem[members[k]] = 1;
    }
    MVCHK(c, hipSetDevice(c->device));
    MVCHK(c, hipMemcpyAsync(c->d_mem, mem.data(), (size_t)c->V, hipMemcpyHostToDevice, c->stream));
    hipError_t e = launch_mv_fit_segment(graph_of(c), c->d_mem, c->d_out, c->stream);
    if (e == hipErrorInvalidValue) return mfail(c, MCMC_E_TOO_LARGE, "mv_fit_segment: graph too large");
    MVCHK(c, e);
    double out[7];
    MVCHK(c, hipMemcpyAsync(out, c->d_out, sizeof out, hipMemcpyDeviceToHost, c->stream));
    MVCHK(c, hipStreamSynchronize(c->stream));
    if (plane5) std::memcpy(plane5, out, 5 * sizeof(double));
    if (prob) *prob = out[5];
    if (lerr) *lerr = out[6];
    return MCMC_OK;
}

int mcmc_mv_build_world_graph(mcmc_mv* c, int V, int dim, const float* theta, const float* weights, int n_first_scan,
                              const uint32_t* in_row_ptr, const uint32_t* in_col, double in_view_threshold,
                              double merge_threshold, double temperature, uint32_t* row_ptr_out, uint32_t* col_out,
                              double* prob_out, uint32_t capacity, uint32_t* e2_out)
{
    if (!c || !theta || !weights || !row_ptr_out || !e2_out || V <= 0 || V > 65535 || dim <= 0 || n_first_scan < 0 || n_first_scan > V)
        return mfail(c, MCMC_E_BADARG, "mv_build_world_graph: bad argument");
    const uint32_t in_e2 = in_row_ptr ? in_row_ptr[V] : 0u;
    if (in_row_ptr) {
        if (in_row_ptr[0] != 0 || (in_e2 && !in_col)) return mfail(c, MCMC_E_BADARG, "mv_build_world_graph: bad in-view graph");
        for (int i = 0; i < V; ++i)
            if (in_row_ptr[i + 1] < in_row_ptr[i]) return mfail(c, MCMC_E_BADARG, "mv_build_world_graph: in-view row_ptr not monotone");
        for (uint32_t e = 0; e < in_e2; ++e)
            if (in_col[e] >= (uint32_t)V) return mfail(c, MCMC_E_BADARG, "mv_build_world_graph: in-view neighbour id out of range");
    }
    MVCHK(c, hipSetDevice(c->device));
    const int words = (V + 31) / 32;
    float *dth = nullptr, *dw = nullptr; uint32_t *dbits = nullptr, *ddeg = nullptr, *drp = nullptr, *dcol = nullptr, *dirp = nullptr, *dic = nullptr;
    double* dp = nullptr;
    auto cleanup = [&]() { mvfree(dth); mvfree(dw); mvfree(dbits); mvfree(ddeg); mvfree(drp); mvfree(dcol); mvfree(dirp); mvfree(dic); mvfree(dp); };
    hipError_t e = mvalloc(&dth, (size_t)V * dim);
    if (e == hipSuccess) e = mvalloc(&dw, (size_t)dim);
    if (e == hipSuccess) e = mvalloc(&dbits, (size_t)V * words);
    if (e == hipSuccess) e = mvalloc(&ddeg, (size_t)V);
    if (e == hipSuccess) e = mvalloc(&drp, (size_t)V + 1);
    if (e == hipSuccess && in_e2) { e = mvalloc(&dirp, (size_t)V + 1); if (e == hipSuccess) e = mvalloc(&dic, (size_t)in_e2); }
    if (e == hipSuccess) e = hipMemcpyAsync(dth, theta, (size_t)V * dim * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dw, weights, (size_t)dim * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && in_e2) {
        e = hipMemcpyAsync(dirp, in_row_ptr, ((size_t)V + 1) * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dic, in_col, (size_t)in_e2 * 4, hipMemcpyHostToDevice, c->stream);
    }
    if (e == hipSuccess) e = hipMemsetAsync(dbits, 0, (size_t)V * words * 4, c->stream);
    if (e == hipSuccess) e = launch_mv_world_bits(dth, dw, dim, V, n_first_scan, dirp, dic, in_e2, in_view_threshold, merge_threshold,
                                                  temperature, words, dbits, ddeg, c->stream);
    std::vector<uint32_t> deg((size_t)V);
    if (e == hipSuccess) e = hipMemcpyAsync(deg.data(), ddeg, (size_t)V * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { cleanup(); return mfail(c, MCMC_E_DEVICE, std::string("mv_build_world_graph: ") + hipGetErrorString(e)); }
    row_ptr_out[0] = 0;
    for (int i = 0; i < V; ++i) row_ptr_out[i + 1] = row_ptr_out[i] + deg[(size_t)i];
    const uint32_t E2 = row_ptr_out[V];
    *e2_out = E2;
    if (E2 > capacity || (E2 && (!col_out || !prob_out))) { cleanup(); return mfail(c, MCMC_E_TOO_LARGE, "mv_build_world_graph: " + std::to_string(E2) + " directed edges exceed the capacity given"); }
    if (E2) {
        e = mvalloc(&dcol, (size_t)E2);
        if (e == hipSuccess) e = mvalloc(&dp, (size_t)E2);
        if (e == hipSuccess) e = hipMemcpyAsync(drp, row_ptr_out, ((size_t)V + 1) * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = launch_mv_world_fill(dth, dw, dim, V, words, dbits, drp, temperature, dcol, dp, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(col_out, dcol, (size_t)E2 * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(prob_out, dp, (size_t)E2 * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    cleanup();
    if (e != hipSuccess) return mfail(c, MCMC_E_DEVICE, std::string("mv_build_world_graph: ") + hipGetErrorString(e));
    return MCMC_OK;
}

int mcmc_mv_pmerge(mcmc_mv* c, const float* t1, const float* t2, const float* w, int dim, int n, double T, double* out)
{
    if (!c || !t1 || !t2 || !w || !out || dim <= 0 || n <= 0) return mfail(c, MCMC_E_BADARG, "mv_pmerge: bad argument");
    MVCHK(c, hipSetDevice(c->device));
    float *d1 = nullptr, *d2 = nullptr, *dw = nullptr; double* dout = nullptr;
    const size_t N = (size_t)n * dim;
    hipError_t e = mvalloc(&d1, N);
    if (e == hipSuccess) e = mvalloc(&d2, N);
    if (e == hipSuccess) e = mvalloc(&dw, (size_t)dim);
    if (e == hipSuccess) e = mvalloc(&dout, (size_t)n);
    if (e == hipSuccess) e = hipMemcpyAsync(d1, t1, N * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d2, t2, N * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dw, w, (size_t)dim * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_mv_pmerge(d1, d2, dw, dim, n, T, dout, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    mvfree(d1); mvfree(d2); mvfree(dw); mvfree(dout);
    MVCHK(c, e);
    return MCMC_OK;
}

int mcmc_mv_chains_init(mcmc_mv* c, int n_chains, uint32_t seed, uint32_t offset, int max_new, double split_prob)
{
    if (!c || n_chains <= 0 || max_new < 0 || !(split_prob >= 0.0 && split_prob <= 1.0))
        return mfail(c, MCMC_E_BADARG, "mv_chains_init: bad argument");
    if (!c->have_vm) return mfail(c, MCMC_E_STATE, "mv_chains_init: vertex moments not computed");
    if ((long long)c->V + max_new >= 0xFFFF) return mfail(c, MCMC_E_TOO_LARGE, "mv_chains_init: segment ids are 16 bit");
    MVCHK(c, hipSetDevice(c->device));
    MVCHK(c, hipStreamSynchronize(c->stream));
    free_chains(c);
    c->n = n_chains; c->seg_cap = c->V + max_new; c->seed = seed; c->offset = offset; c->split_prob = split_prob; c->step = 0;
    const size_t N = (size_t)n_chains, NV = N * c->V, NS = N * c->seg_cap;
    MVCHK(c, mvalloc(&c->cur, NV)); MVCHK(c, mvalloc(&c->own, NV)); MVCHK(c, mvalloc(&c->plane, NS * 5));
    MVCHK(c, mvalloc(&c->segP, NS)); MVCHK(c, mvalloc(&c->pvalid, NS)); MVCHK(c, mvalloc(&c->nseg, N));
    MVCHK(c, mvalloc(&c->status, N)); MVCHK(c, mvalloc(&c->n_accept, N)); MVCHK(c, mvalloc(&c->n_split, N));
    MVCHK(c, mvalloc(&c->n_skip, N)); MVCHK(c, mvalloc(&c->trace_hash, N));
    hipError_t e = launch_mv_chain_init(graph_of(c), chain_args_of(c), c->stream);
    if (e == hipErrorInvalidValue) return mfail(c, MCMC_E_TOO_LARGE, "mv_chains_init: graph too large for LDS");
    MVCHK(c, e);
    MVCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_mv_set_option(mcmc_mv* c, const char* name, int value)
{
    if (!c || !name) return MCMC_E_BADARG;
    c->ahead_ready_step = -1;             // whatever changes, the next run grows its own first step
    if (std::string(name) == "split_step") c->opt_split = value;
    else if (std::string(name) == "split_growth") c->opt_split_growth = value;
    else if (std::string(name) == "fuse_halves") c->opt_fuse_halves = value;
    else if (std::string(name) == "growth_ahead") c->opt_ahead = value;
    else if (std::string(name) == "growth_ahead_waves" && (value == 0 || value == 4 || value == 8)) c->opt_ahead_waves = value;
    else if (std::string(name) == "groups" && value >= 0 && value <= mcmc_mv::kMaxGroups) c->opt_groups = value;
    else if (std::string(name) == "profile") c->opt_prof = value;
    else return mfail(c, MCMC_E_BADARG, std::string("mv_set_option: unknown option ") + name);
    return MCMC_OK;
}

int mcmc_mv_chains_set_trace(mcmc_mv* c, uint64_t cap)
{
    if (!c || c->n == 0) return mfail(c, MCMC_E_STATE, "mv_chains_set_trace: no chains");
    MVCHK(c, hipSetDevice(c->device));
    MVCHK(c, hipStreamSynchronize(c->stream));
    mvfree(c->trace); mvfree(c->trace_a); c->trace_cap = 0;
    if (cap) {
        MVCHK(c, mvalloc(&c->trace, (size_t)c->n * cap));
        MVCHK(c, mvalloc(&c->trace_a, (size_t)c->n * cap));
        c->trace_cap = cap;
    }
    return MCMC_OK;
}

int mcmc_mv_chains_run(mcmc_mv* c, uint64_t steps)
{
    if (!c || c->n == 0) return mfail(c, MCMC_E_STATE, "mv_chains_run: mv_chains_init first");
    MVCHK(c, hipSetDevice(c->device));
    MvChainArgs a = chain_args_of(c);
    a.step0 = c->step; a.steps = steps;
    if (c->trace && steps <= c->trace_cap) { a.trace = c->trace; a.trace_a = c->trace_a; }
    a.trace_stride = steps; a.trace_first = 0;
    // the split iteration (L1 pass on the matrix cores) is the default at every chain count: at C4 it takes 111 us per
    // iteration for one chain against 605 us of the fused kernel (tools/mv_fused.py); "split_step" = 0 keeps the fused one
    const bool split = c->opt_split != 0;
    // Groups.  A launch lasts as long as its slowest chain, and the four kernels of an iteration wait for one another;
    // with the chains in G groups, each running its iterations on a stream of its own, a group only waits for ITS
    // slowest chain and the groups' kernels fill each other's tails.  Chains do not interact, so the grouping changes no
    // result.  Each group gets its share of the exchange buffers and sees itself as a whole job (per-chain pointers
    // advanced to its first chain, chain_offset likewise).  Measured at C4 (tools/mv_l1_time.py, MVG=): 4096 chains 7.23 ->
    // 7.79 M proposals/s with two groups (three the same; from four on the streams outnumber the hardware queues and
    // it is slower); 1024 chains 5.76 -> 5.95 (with the halves of an iteration apart, round 3's first form, two groups
    // gained nothing there), 768 and 512 chains nothing.  Automatic: two groups from 1024 chains on.
    // Growth a step ahead.  Measured at C4 (tools/mv_l1_time.py, GROWTH_AHEAD=), us per iteration without / with: 1 chain
    // 67.9 / 50.6, 64: 108.4 / 72.5, 256: 118.4 / 84.7, 512: 136.5 / 101.0, 768: 154.1 / 135.6, 1024: 165.7 / 160.9,
    // 1280: 190.8 / 196.0, 1536: 223.5 / 224.0, 1792: 246.5 / 252.9, 2048: 267.6 / 283.1 (from there the device is full
    // either way and the extra kernel costs).  Eight growth waves per chain are best where the CUs have room for them,
    // four from 384 chains on (512 chains: 101.0 against 109.9, 1024: 161.3 against 166.5).
    const bool ahead = split && !c->opt_prof &&
        (c->opt_ahead > 0 || (c->opt_ahead < 0 && c->n <= 3072 && c->opt_groups == 0 && c->opt_split_growth < 0 && c->opt_fuse_halves < 0));
    const int ahead_waves = c->opt_ahead_waves ? c->opt_ahead_waves : (c->n > 384 ? 4 : 8);
    int G = 1;
    if (split && !c->opt_prof && !ahead) {
        G = c->opt_groups > 0 ? c->opt_groups : (c->n >= 1024 ? 2 : 1);
        G = std::min(G, std::max(1, c->n / 64));
    }
    // The device has few hardware queues (four per process by default): a handle keeps only the streams of the mode it
    // runs in, or the groups' streams end up sharing a queue with an idle growth stream (2048 chains after a run of 1024
    // on the same handle: 349 us per iteration instead of 267).
    if (!ahead && c->ahead_stream) {
        MVCHK(c, hipStreamSynchronize(c->ahead_stream)); MVCHK(c, hipStreamDestroy(c->ahead_stream)); c->ahead_stream = nullptr;
    }
    if ((ahead || G == 1) && c->gstream[0]) {
        for (int k = 0; k < mcmc_mv::kMaxGroups; ++k)
            if (c->gstream[k]) { MVCHK(c, hipStreamSynchronize(c->gstream[k])); MVCHK(c, hipStreamDestroy(c->gstream[k])); c->gstream[k] = nullptr; }
    }
    if (ahead) {
        // A stream of ordinary priority: a low-priority stream (hipStreamCreateWithPriority) gained 2 % at 2048 chains, and a
        // process that holds five or more of them -- one per mcmc_mv handle -- stalls for a minute and a half at its first
        // run (tools/mv_streams_probe.py: the device has few hardware queues of that class); ordinary streams do not, with
        // ten handles and two dozen other streams in the process.
        if (!c->ahead_stream) MVCHK(c, hipStreamCreateWithFlags(&c->ahead_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            if (!c->ahead.ev_g[k]) MVCHK(c, hipEventCreateWithFlags(&c->ahead.ev_g[k], hipEventDisableTiming));
            if (!c->ahead.ev_c[k]) MVCHK(c, hipEventCreateWithFlags(&c->ahead.ev_c[k], hipEventDisableTiming));
        }
        if (!c->ahead.ev_start) MVCHK(c, hipEventCreateWithFlags(&c->ahead.ev_start, hipEventDisableTiming));
        if (c->ahead_n != c->n) {
            for (int k = 0; k < 2; ++k) {
                mvfree(c->ahead.inE[k]); { uint8_t* p8 = static_cast<uint8_t*>(c->ahead.sh[k]); mvfree(p8); c->ahead.sh[k] = nullptr; }
                uint8_t* shp = nullptr;
                MVCHK(c, mvalloc(&c->ahead.inE[k], (size_t)c->n * c->V));
                MVCHK(c, mvalloc(&shp, (size_t)c->n * mv_exchange_shared_bytes())); c->ahead.sh[k] = shp;
            }
            c->ahead_n = c->n; c->ahead_ready_step = -1;
        }
    }
    // A run leaves the growth of the step after its last one behind (it runs beside the last step's kernels); the next
    // run of the same chains starts from it.  Any other kind of run, new chains or an option set in between discard it.
    const bool have_first = ahead && steps && c->ahead_ready_step == (long long)c->step;
    if (steps) c->ahead_ready_step = -1;                         // a run of no steps launches nothing and leaves it
    if (split && c->ex_groups != G) {
        { uint8_t* p8 = static_cast<uint8_t*>(c->ex_sh); mvfree(p8); c->ex_sh = nullptr; }
        mvfree(c->ex_inE); mvfree(c->ex_need); mvfree(c->ex_err); mvfree(c->ex_plist); mvfree(c->ex_queue); mvfree(c->ex_counters);
        const size_t N = (size_t)c->n, NV = N * c->V;
        size_t pl = 0, qu = 0;
        for (int gidx = 0; gidx < G; ++gidx) {
            const int n0 = (int)((long long)c->n * gidx / G), n1 = (int)((long long)c->n * (gidx + 1) / G);
            pl += mv_split_plist_entries(c->V, n1 - n0); qu += mv_split_queue_entries(c->V, n1 - n0);
        }
        uint8_t* shp = nullptr;
        MVCHK(c, mvalloc(&shp, N * mv_exchange_shared_bytes())); c->ex_sh = shp;
        MVCHK(c, mvalloc(&c->ex_inE, NV)); MVCHK(c, mvalloc(&c->ex_need, NV)); MVCHK(c, mvalloc(&c->ex_err, NV * 4));
        MVCHK(c, mvalloc(&c->ex_plist, pl)); MVCHK(c, mvalloc(&c->ex_queue, qu));
        MVCHK(c, mvalloc(&c->ex_counters, kMvExchangeCounters * (size_t)G));
        MVCHK(c, hipMemsetAsync(c->ex_counters, 0, kMvExchangeCounters * (size_t)G * sizeof(uint32_t), c->stream));
        MVCHK(c, hipMemsetAsync(c->ex_need, 0, NV, c->stream));      // all zero between iterations (MvExchange::need)
        c->ex_groups = G;
    }
    if (c->opt_prof) {
        if (!c->d_prof) MVCHK(c, mvalloc(&c->d_prof, 32));
        MVCHK(c, hipMemsetAsync(c->d_prof, 0, 32 * 8, c->stream));
        a.prof = c->d_prof;
    }
    MVCHK(c, hipEventRecord(c->ev0, c->stream));
    if (ahead) {
        const MvSplitBufs bufs{ c->ex_sh, c->ex_inE, c->ex_need, c->ex_err, c->ex_plist, c->ex_queue, c->ex_counters };
        MVCHK(c, launch_mv_chain_run_ahead(graph_of(c), a, bufs, c->ahead, c->stream, c->ahead_stream, ahead_waves, have_first, true));
        if (steps) c->ahead_ready_step = (long long)(c->step + steps);
    } else if (split && G == 1) {
        const MvSplitBufs bufs{ c->ex_sh, c->ex_inE, c->ex_need, c->ex_err, c->ex_plist, c->ex_queue, c->ex_counters };
        const bool sg = !c->opt_prof && (c->opt_split_growth > 0 || (c->opt_split_growth < 0 && c->n <= 256));
        MVCHK(c, launch_mv_chain_run_split(graph_of(c), a, bufs, c->stream, sg, !c->opt_prof && c->opt_fuse_halves != 0));
    } else if (split) {
        if (!c->gev_start) MVCHK(c, hipEventCreateWithFlags(&c->gev_start, hipEventDisableTiming));
        for (int gidx = 0; gidx < G; ++gidx) {
            if (!c->gstream[gidx]) MVCHK(c, hipStreamCreateWithFlags(&c->gstream[gidx], hipStreamNonBlocking));
            if (!c->gev_done[gidx]) MVCHK(c, hipEventCreateWithFlags(&c->gev_done[gidx], hipEventDisableTiming));
        }
        MVCHK(c, hipEventRecord(c->gev_start, c->stream));
        MvChainArgs ag[mcmc_mv::kMaxGroups];
        MvSplitBufs bg[mcmc_mv::kMaxGroups];
        size_t pl = 0, qu = 0;
        const size_t V = (size_t)c->V, shb = mv_exchange_shared_bytes();
        for (int gidx = 0; gidx < G; ++gidx) {
            const size_t n0 = (size_t)((long long)c->n * gidx / G), n1 = (size_t)((long long)c->n * (gidx + 1) / G);
            MvChainArgs x = a;
            x.n_chains = (int)(n1 - n0); x.chain_offset = a.chain_offset + (uint32_t)n0;
            x.cur += n0 * V; x.own += n0 * V; x.plane += n0 * (size_t)c->seg_cap * 5; x.segP += n0 * (size_t)c->seg_cap;
            x.segP_valid += n0 * (size_t)c->seg_cap; x.nseg += n0; x.status += n0; x.n_accept += n0; x.n_split += n0;
            x.n_skip += n0; x.trace_hash += n0;
            if (x.trace) { x.trace += n0 * a.trace_stride; x.trace_a += n0 * a.trace_stride; }
            ag[gidx] = x;
            bg[gidx] = MvSplitBufs{ static_cast<uint8_t*>(c->ex_sh) + n0 * shb, c->ex_inE + n0 * V, c->ex_need + n0 * V,
                                    c->ex_err + n0 * V * 4, c->ex_plist + pl, c->ex_queue + qu, c->ex_counters + kMvExchangeCounters * gidx };
            pl += mv_split_plist_entries(c->V, (int)(n1 - n0)); qu += mv_split_queue_entries(c->V, (int)(n1 - n0));
            MVCHK(c, hipStreamWaitEvent(c->gstream[gidx], c->gev_start, 0));
        }
        // iteration by iteration across the groups, so that the streams' queues fill side by side
        for (unsigned long long ts = 0; ts < steps; ++ts) {
            for (int gidx = 0; gidx < G; ++gidx) {
                MvChainArgs x = ag[gidx];
                x.step0 = a.step0 + ts; x.steps = 1; x.trace_first = ts;
                MVCHK(c, launch_mv_chain_run_split(graph_of(c), x, bg[gidx], c->gstream[gidx], c->opt_split_growth > 0,
                                                   c->opt_fuse_halves != 0, ts == 0, ts + 1 == steps));
            }
        }
        for (int gidx = 0; gidx < G; ++gidx) {
            MVCHK(c, hipEventRecord(c->gev_done[gidx], c->gstream[gidx]));
            MVCHK(c, hipStreamWaitEvent(c->stream, c->gev_done[gidx], 0));
        }
    }
    else MVCHK(c, launch_mv_chain_run(graph_of(c), a, c->stream));
    MVCHK(c, hipEventRecord(c->ev1, c->stream));
    c->ev_valid = true;
    c->step += steps; c->last_steps = steps;
    return MCMC_OK;
}

int mcmc_mv_chains_get_profile(mcmc_mv* c, uint64_t* out32)
{
    if (!c || !out32 || !c->d_prof) return mfail(c, MCMC_E_STATE, "mv_chains_get_profile: set option profile and run first");
    MVCHK(c, hipSetDevice(c->device));
    MVCHK(c, hipMemcpyAsync(out32, c->d_prof, 32 * 8, hipMemcpyDeviceToHost, c->stream));
    MVCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

double mcmc_mv_last_run_ms(mcmc_mv* c)
{
    if (!c || !c->ev_valid) return -1.0;
    if (hipSetDevice(c->device) != hipSuccess || hipEventSynchronize(c->ev1) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0;
    return (double)ms;
}

int mcmc_mv_chains_get_trace(mcmc_mv* c, uint8_t* flags, double* ratio)
{
    if (!c || !c->trace || c->last_steps == 0 || c->last_steps > c->trace_cap) return mfail(c, MCMC_E_STATE, "mv_chains_get_trace: no trace recorded");
    MVCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->n * c->last_steps;
    if (flags) MVCHK(c, hipMemcpyAsync(flags, c->trace, n, hipMemcpyDeviceToHost, c->stream));
    if (ratio) MVCHK(c, hipMemcpyAsync(ratio, c->trace_a, n * 8, hipMemcpyDeviceToHost, c->stream));
    MVCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_mv_chains_get_labels(mcmc_mv* c, int which, uint32_t* out)
{
    if (!c || !out || which < 0 || which > 1) return mfail(c, MCMC_E_BADARG, "mv_chains_get_labels: bad argument");
    if (c->n == 0) return mfail(c, MCMC_E_STATE, "mv_chains_get_labels: no chains");
    MVCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->n * c->V;
    std::vector<uint16_t> tmp(n);
    MVCHK(c, hipMemcpyAsync(tmp.data(), which ? c->own : c->cur, n * 2, hipMemcpyDeviceToHost, c->stream));
    MVCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; ++i) out[i] = tmp[i];
    return MCMC_OK;
}

int mcmc_mv_chains_get_scalars(mcmc_mv* c, uint64_t* n_accept, uint64_t* n_split, uint64_t* n_skip,
                               uint64_t* trace_hash, int32_t* nseg, int32_t* status)
{
    if (!c || c->n == 0) return mfail(c, MCMC_E_STATE, "mv_chains_get_scalars: no chains");
    MVCHK(c, hipSetDevice(c->device));
    const size_t N = (size_t)c->n;
    auto cp = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        return dst ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream) : hipSuccess;
    };
    MVCHK(c, cp(n_accept, c->n_accept, N * 8)); MVCHK(c, cp(n_split, c->n_split, N * 8));
    MVCHK(c, cp(n_skip, c->n_skip, N * 8)); MVCHK(c, cp(trace_hash, c->trace_hash, N * 8));
    MVCHK(c, cp(nseg, c->nseg, N * 4)); MVCHK(c, cp(status, c->status, N * 4));
    MVCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_mv_chains_get_planes(mcmc_mv* c, int chain, double* out)
{
    if (!c || !out || chain < 0 || chain >= c->n) return mfail(c, MCMC_E_BADARG, "mv_chains_get_planes: bad argument");
    MVCHK(c, hipSetDevice(c->device));
    int ns = 0;
    MVCHK(c, hipMemcpyAsync(&ns, c->nseg + chain, 4, hipMemcpyDeviceToHost, c->stream));
    MVCHK(c, hipStreamSynchronize(c->stream));
    MVCHK(c, hipMemcpyAsync(out, c->plane + (size_t)chain * c->seg_cap * 5, (size_t)ns * 40, hipMemcpyDeviceToHost, c->stream));
    MVCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

}  // extern "C"
