// kernels.hpp -- argument blocks and launchers shared by the C ABI and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mc {

// hipFuncSetAttribute is a property of a function on ONE device: a flag per (call site, device).  Setting it
// twice is harmless, so two host threads racing on the first launch need no lock.
struct DeviceOnce {
    bool done[64] = {};
    bool* slot()
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
        return &done[d];
    }
};


// per-chain segment cache row (f64): 10 exact sums, sum of ssr, probabilityL, member count, pad
constexpr int kSegCacheStride = 16, kSegCacheUsed = 13, kSegCacheP = 11, kSegCacheN = 12;

struct MomentArgs {
    const int16_t* disp;       // [F][H][W]
    const uint16_t* label;     // [F][H][W]
    const int* S;              // [F] superpixels per frame
    unsigned long long* mom;   // [F][s_max][10] int64 bit patterns
    size_t frame_stride;       // pixels per frame
    int W, H, s_max, bands, first_frame;
    int table_rows;            // rows of the LDS table (>= every S in the launch)
    int debug;                 // tuning experiments only (0 = normal)
    int lane_px;               // left/right kernel: pixels per lane and row, 8 or 16
};

struct FitArgs {
    const unsigned long long* mom;
    const int* S;
    double* fit;               // [F][s_max][12]
    uint8_t* valid;            // [F][s_max]
    double* local_err;         // [F][s_max]
    double var;
    int W, H, s_max, first_frame;
};

struct EdgeArgs {
    const double* fit;
    const uint8_t* valid;
    const uint32_t* rowp;      // [F][s_max+1]
    const uint32_t* col;       // [F][e2_max]
    const int* S;
    const uint32_t* E2;        // [F]
    double* q;                 // [F][e2_max]
    int* status;               // [F] bit0 NaN ratio, bit1 inf energy
    double var;
    int s_max, e2_max, first_frame;
};

struct SegArgs {
    const double* fit;         // one frame
    const uint8_t* mem;        // [S]
    double* out;               // probL err theta3 inf_flag
    double var;
    int S;
};

// Per-chain state in HBM between launches (chain-major, stride s_max).
struct ChainArgs {
    // frame tables
    const double* fit; const uint8_t* valid; const uint32_t* rowp; const uint32_t* col;
    const double* q; const int* S; const uint32_t* E2;
    int s_max, e2_max;
    // chains
    int n_chains, chains_per_frame;
    uint32_t seed, chain_offset;
    unsigned long long step0, burn_in, iters;
    double var, new_seg_prob;
    uint16_t* label;           // [C][s_max] min-id labels
    double* segC;              // [C][s_max][kSegCacheStride] cache row of the segment named by the row index
    double* energy; double* best_energy;
    uint16_t* best_label; uint16_t* top_label;
    unsigned long long* state_hash; unsigned long long* trace_hash;
    unsigned long long* n_accept; unsigned long long* n_case; // [C][3]
    int* status;
    unsigned int* fault;       // one word per context: set when a chain ends a run with status -2
    unsigned long long* visit_hash; unsigned long long* visit_count; // [C][hcap]
    unsigned long long* overflow; unsigned long long* top_count; unsigned long long* top_hash;
    double* last_targetL; double* last_prop;   // [C] ratios of the last step (Problem::propose's out-parameters)
    unsigned long long* marg;  // [F][s_max][lmax]
    uint8_t* trace; unsigned long long trace_stride; // optional
    unsigned long long* slots_g; // [C][queue slots] FIFO of the growth when the frame's tables do not fit LDS (else unused)
    unsigned long long* prof;  // optional: 12 per-phase cycle sums (diagnostic build of the 4-wave LDS variant)
    int hcap, lmax;
    int spec_variant;          // several waves per chain, speculating on the next steps: -1 never, 0 as many as fit (up to 4) when the chains are at most four per CU, 1 the same whatever the chain count, 2..4 exactly that many
    int dense_variant;         // eight chains per workgroup with the records in L2: -1 never, 0 when the chains exceed four per CU, 1 always
};

hipError_t launch_moments(const MomentArgs& a, int n_frames, int threads, size_t lds_bytes, int mode, hipStream_t st);
size_t moments_lr_lds_bytes(int table_rows, int threads);
hipError_t launch_local_fit(const FitArgs& a, int n_frames, hipStream_t st);
hipError_t launch_edge_prob(const EdgeArgs& a, int n_frames, hipStream_t st);
hipError_t launch_segment_prob(const SegArgs& a, hipStream_t st);
hipError_t launch_mark_members(const uint32_t* list, int n, uint8_t* mem, hipStream_t st);  // list ids < size of mem (checked by the caller)
hipError_t launch_selftest_math(const double* x, const double* y, int n, double* ol, double* oe, double* od, hipStream_t st);
hipError_t launch_selftest_philox(const uint32_t* ck, uint32_t* out, hipStream_t st);
hipError_t launch_selftest_wave_sums(const double* x, int n_waves, double* plain, double* folded, hipStream_t st);

// chain kernel: returns hipErrorInvalidValue if the per-chain state does not fit LDS
hipError_t launch_chain_init(const ChainArgs& a, hipStream_t st);
hipError_t launch_chain_run(const ChainArgs& a, int s_used_max, int e2_used_max, hipStream_t st, int* variant_out);
size_t chain_run_global_slots(const ChainArgs& a, int s_used_max, int e2_used_max);
hipError_t launch_chain_energy(const ChainArgs& a, double* out, hipStream_t st);
hipError_t launch_visit_rehash(const unsigned long long* oh, const unsigned long long* oc, int ocap, unsigned long long* nh,
                               unsigned long long* nc, int ncap, int n_chains, hipStream_t st);
// rec: 2 + (s_max + 3) / 4 words -- key, global chain id, labels (four u16 per word)
hipError_t launch_best_record(const ChainArgs& a, int which, int canonical, unsigned long long* rec, hipStream_t st);
hipError_t launch_visit_occupancy(const unsigned long long* vc, int hcap, int n_chains, unsigned int* out_max, hipStream_t st);  // *out_max zeroed by the caller
hipError_t launch_canonical_labels(const uint16_t* src, const uint8_t* valid, const int* S, int s_max,
                                   int n_chains, int chains_per_frame, uint16_t* dst, hipStream_t st);

// ---- Path B (multiview) ------------------------------------------------------
struct MvGraph {
    int V;
    const float4* pts;         // PCL PointXYZ stride: x y z pad
    const uint32_t* pt_ptr;    // [V+1]
    const double* vm;          // [V][10] per-vertex moments
    const uint32_t* rowp;      // [V+1]
    const uint32_t* col;       // [E2] ascending within a row
    const double* q;           // [E2] SPEdge::BernoulliProbability
    const uint2* et;           // [E2] .x rank of the undirected edge among (i<j) entries (the trial's draw index),
                               //      .y largest 32-bit draw x with x / 4294967295.0 <= q[e]: one load per trial
    // the adjacency once more as bitmaps, for the growth scan of the chain step (built by mcmc_mv_set_graph)
    int W64;                                // 64-bit words per row, (V + 63) / 64
    const unsigned long long* abm;          // [V][W64] bit j of row v: edge (v, j)
    const uint32_t* wbase;                  // [V][W64] entry index of the first edge of that word
    // the points once more for the L1 pass of the split iteration (built by mcmc_mv_set_graph): doubles, four per
    // point (x, y, z, 1): a row of the A operand of v_mfma_f64_16x16x4_f64; 16 points of padding at the end
    const double* ptsd;                     // [P + 16][4]
    const uint32_t* vorder;                 // [V] vertex ids by descending point count
};

struct MvChainArgs {
    int n_chains, seg_cap;
    uint32_t seed, chain_offset;
    double split_prob;
    unsigned long long step0, steps;
    uint16_t* cur;             // [C][V] initial partition ids (never updated, as in the reference)
    uint16_t* own;             // [C][V] vertex-set membership
    double* plane;             // [C][seg_cap][5]
    double* segP;              // [C][seg_cap] cached probability
    uint8_t* segP_valid;       // [C][seg_cap]
    int* nseg; int* status;
    unsigned long long *n_accept, *n_split, *n_skip, *trace_hash;
    uint8_t* trace; double* trace_a;   // optional [C][trace_stride], this launch writes from trace_first on
    unsigned long long trace_stride, trace_first;
    unsigned long long* prof;  // optional [16]: shader-clock cycles per phase of the first half, summed over chains
};

hipError_t launch_mv_vertex_moments(const MvGraph& g, double* out, hipStream_t st);
hipError_t launch_mv_fit_segment(const MvGraph& g, const uint8_t* mem, double* out, hipStream_t st);
hipError_t launch_mv_chain_init(const MvGraph& g, const MvChainArgs& a, hipStream_t st);
hipError_t launch_mv_chain_run(const MvGraph& g, const MvChainArgs& a, hipStream_t st);
// device buffers of the split iteration (capi_multiview.hip allocates them; sizes from the functions below)
// L1 pass: an item is one vertex under 16, 32, 48 or 64 (chain, plane) pairs, whichever keeps it near kMvErrItemCost matrix
// instructions (blocks of 16 points x groups of 16 pairs): a wave holds a share of its SIMD's matrix pipe, so the largest
// item sets how long the kernel lasts (64 pairs whatever the vertex: 248 instructions for the largest vertex of BASELINE
// config 4 against a mean of 108).  Host and device: the host orders the vertices by the cost of their items.
#ifndef MC_MV_ERR_ITEM_COST
#define MC_MV_ERR_ITEM_COST 128
#endif
constexpr uint32_t kMvErrItemCost = MC_MV_ERR_ITEM_COST;
__host__ __device__ inline uint32_t mv_err_groups_per_item(uint32_t cnt)
{
    const uint32_t nblk = (cnt + 15u) >> 4;
    const uint32_t gq = nblk ? kMvErrItemCost / nblk : 4u;
    return gq < 1u ? 1u : (gq > 4u ? 4u : gq);
}
constexpr int kMvExchangeCounters = 2;         // words of MvSplitBufs::counters per group of chains
struct MvSplitBufs {
    void* sh; uint8_t* inE; uint8_t* need; double* err;     // [C] shared blocks, [C][V], [V][C], [C][4][V]
    uint16_t* plist; uint4* queue; uint32_t* counters;      // mv_split_plist_entries, mv_split_queue_entries, [kMvExchangeCounters] zeroed once
};
// growth a step ahead: two hand-over slots (the set's flags [C][V], the shared blocks [C] x mv_exchange_shared_bytes) and the
// events that order the growth stream and the main stream
struct MvAheadBufs {
    uint8_t* inE[2]; void* sh[2];
    hipEvent_t ev_g[2], ev_c[2], ev_start;
};
// growth_waves: 8 (the growth kernel of the split-growth path) or 4 waves per chain
hipError_t launch_mv_chain_run_ahead(const MvGraph& g, const MvChainArgs& a, const MvSplitBufs& bufs, const MvAheadBufs& ab,
                                     hipStream_t st, hipStream_t gst, int growth_waves, bool have_first, bool grow_next);
size_t mv_split_plist_entries(int V, int n_chains);
size_t mv_split_queue_entries(int V, int n_chains);
// split_growth: the first half as two kernels (component growth with more waves per chain, then the plane fits)
// fuse_halves: the second half of a step and the first half of the next as one kernel; opens / closes: this call holds the
// first / last step of the run (a run may come as one call per step, as the groups of chains do)
hipError_t launch_mv_chain_run_split(const MvGraph& g, const MvChainArgs& a, const MvSplitBufs& bufs, hipStream_t st, bool split_growth = false,
                                     bool fuse_halves = false, bool opens = true, bool closes = true);
size_t mv_exchange_shared_bytes();
size_t mv_chain_lds_bytes(int V);
hipError_t launch_mv_world_bits(const float* th, const float* w, int dim, int V, int n0, const uint32_t* in_rowp,
                                const uint32_t* in_col, uint32_t in_e2, double in_thr, double thr, double T, int words,
                                uint32_t* bits, uint32_t* deg, hipStream_t st);
hipError_t launch_mv_world_fill(const float* th, const float* w, int dim, int V, int words, const uint32_t* bits,
                                const uint32_t* rowp, double T, uint32_t* col, double* prob, hipStream_t st);
hipError_t launch_mv_pmerge(const float* t1, const float* t2, const float* w, int dim, int n, double T, double* out, hipStream_t st);

}  // namespace mc
