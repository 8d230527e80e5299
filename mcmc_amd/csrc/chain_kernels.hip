// chain_kernels.hip -- K3 chain_step (+ K5 canonical relabel / statistics).
//
// One independent Markov chain per 64-lane wavefront; a workgroup of WPB
// wavefronts shares one frame.  When the frame's tables fit, they are staged
// in LDS once per launch (11 sums per superpixel, 88 B, CSR columns as u16,
// edge probabilities f64) next to each chain's private state (labels, BFS
// queue, per-segment log-probability cache, bitsets), so a whole run of
// burn_in+iters steps touches HBM only for statistics.  The step itself is
// latency-bound (a data-dependent BFS), not bandwidth-bound: parallelism comes
// from chains, and the lanes of a wave cooperate on one chain's neighbour
// scans, segment sums and reductions.
//
// Restates (behaviour, not code): Problem::propose (Problem.cpp:116-224),
// Segmentation::createComponent/selectSegment/pickSuperPixel
// (Segmentation.cpp:10-183), Segment fits (Graph.cpp:10-138, MyMath.h:189-275),
// jumpProbability/jump (Problem.cpp:32-74,475-487),
// FastMetropolisHastings::iterate (MCMC.h:207-238), MCMC::run (MCMC.h:43-89),
// Segmentation::computeHash (Segmentation.cpp:197-237), Statistics::update
// (Statistics.h:33-47).
//
// Canonical order (the reference iterates std::set/std::map in heap-pointer
// order, i.e. arbitrarily): superpixels by id; a segment is named by its
// smallest member id.
#include "device_math.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace mc {

constexpr int kLdsLimit = 160 * 1024;
constexpr int kSpecAuto = 4;    // waves per chain the speculative form takes by itself when they fit

MC_D size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }

// ---- RNG: the uniform at position p of a step is word (p>>6)&3 of Philox block (p>>8)*64 + (p&63),
// so the 64 lanes generate positions [256 b, 256 b + 256) with one Philox call each.
// p = 0 start index, 1 discarded, 2 selectSegment uniform, 3 selectSegment index, 4 MH uniform,
// p = 8 + n: trial of the n-th CSR entry examined by the component growth.
constexpr uint32_t kTrialPos0 = 8u;
struct Rng {
    uint32_t k0, k1, step_lo, step_hi;
    int batch;
    uint32_t* buf;                // LDS, [4][64]: the batch's raw 32-bit draws (1 KB; a draw becomes rand()/RAND_MAX at its use)
};
// philox4x32_10 (device_math.hpp).  The 10 round keys are wave-uniform and loop-invariant, so the compiler would keep
// all 20 in SGPRs across the growth loop, run out, and park them in lanes of a spill VGPR (a v_readlane and a hazard
// nop per round); pinned in VGPRs they cost 20 registers of a kernel that wants four waves per SIMD.  Here the key
// pair is hidden from the optimiser at every call, so the schedule is ten scalar adds per call that live nowhere.
MC_D u32x4 philox_pinned(u32x4 c, const Rng& g)
{
    uint32_t a = g.k0, b = g.k1;
    asm volatile("" : "+s"(a), "+s"(b));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)kPhiloxM0 * c.x;
        const uint64_t p1 = (uint64_t)kPhiloxM1 * c.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ a;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ b;
        n.w = (uint32_t)p0;
        c = n;
        a += kPhiloxW0; b += kPhiloxW1;
    }
    return c;
}

// rand()/RAND_MAX as one multiplication by the f64 nearest to 1/(2^32 - 1): the largest word gives 1.0
MC_D double to_unit(uint32_t x) { return (double)x * 2.3283064370807974e-10; }
MC_D void wave_fence();

// generates positions [256 batch, 256 batch + 256) into g.buf; returns the lane's raw words
MC_D u32x4 rng_gen(Rng& g, int batch, int lane)
{
    u32x4 c = { g.step_lo, g.step_hi, (uint32_t)(batch * 64 + lane), kPhiloxTag };
    u32x4 r = philox_pinned(c, g);
    wave_fence();
    g.buf[lane] = r.x; g.buf[64 + lane] = r.y; g.buf[128 + lane] = r.z; g.buf[192 + lane] = r.w;
    wave_fence();
    g.batch = batch;
    return r;
}

// Uniforms of positions p0 + lane, lane < count (p0, count wave-uniform): position p sits at
// buf[(p >> 6) & 3][p & 63] of its batch.
MC_D uint32_t rng_trial(Rng& g, uint32_t p0, uint32_t count, int lane)
{
    const uint32_t p = p0 + (uint32_t)lane;
    const uint32_t slot = p & 255u;
    const int b0 = (int)(p0 >> 8);
    if (b0 != g.batch) rng_gen(g, b0, lane);
    uint32_t res = g.buf[slot];
    if ((int)((p0 + count - 1u) >> 8) != b0) {                // the pass crosses into the next 256 positions
        rng_gen(g, b0 + 1, lane);
        const uint32_t v = g.buf[slot];
        if ((int)(p >> 8) != b0) res = v;
    }
    return res;
}

// Lanes of one wave that exchange data through LDS: hardware runs a wave's LDS operations in order, the
// fence only stops the compiler from reordering or forwarding across it (wavefront scope, no cache action).
MC_D void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

// ---- wave64 inclusive scans in registers (DPP: row_shr 1,2,4,8 then row_bcast 15 / 31) -------
template <int CTRL, int ROW_MASK>
MC_D uint32_t dpp_u32(uint32_t identity, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false);
}
MC_D uint32_t wave_scan_add_u32(uint32_t v)
{
    v += dpp_u32<0x111, 0xf>(0u, v);
    v += dpp_u32<0x112, 0xf>(0u, v);
    v += dpp_u32<0x114, 0xf>(0u, v);
    v += dpp_u32<0x118, 0xf>(0u, v);
    v += dpp_u32<0x142, 0xa>(0u, v);
    v += dpp_u32<0x143, 0xc>(0u, v);
    return v;
}
MC_D uint32_t wave_scan_max_u32(uint32_t v)
{
    v = max(v, dpp_u32<0x111, 0xf>(0u, v));
    v = max(v, dpp_u32<0x112, 0xf>(0u, v));
    v = max(v, dpp_u32<0x114, 0xf>(0u, v));
    v = max(v, dpp_u32<0x118, 0xf>(0u, v));
    v = max(v, dpp_u32<0x142, 0xa>(0u, v));
    v = max(v, dpp_u32<0x143, 0xc>(0u, v));
    return v;
}

// idx-th set bit (ascending) of a bit set of NW 32-bit words; wave-uniform result = the bit's index
MC_D uint32_t nth_flag(const uint32_t* words, int NW, uint32_t idx, int lane)
{
    uint32_t acc = 0, result = 0xFFFFFFFFu;
    for (int wb = 0; wb < NW; wb += kWave) {
        const int w = wb + lane;
        const uint32_t word = w < NW ? words[w] : 0u;
        const int pc = __popc(word);
        const int incl = (int)wave_scan_add_u32((uint32_t)pc);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane(incl, 63);
        if (idx < acc + total) {
            const uint32_t rel = idx - acc;
            const uint32_t excl = (uint32_t)(incl - pc);
            const bool here = rel >= excl && rel < (uint32_t)incl;
            uint32_t ww = word;
            const uint32_t k = here ? rel - excl : 0u;
            for (uint32_t t = 0; t < k; ++t) ww &= ww - 1u;
            const uint32_t val = (uint32_t)w * 32u + (uint32_t)(__ffs((int)ww) - 1);
            const uint64_t hm = __ballot(here);
            result = (uint32_t)__builtin_amdgcn_readlane((int)val, __ffsll((unsigned long long)hm) - 1);
            break;
        }
        acc += total;
    }
    return result;
}

// per-superpixel record of the chain kernel: the 10 exact sums and ssr (fit record entries 0..8, 12, 13)
constexpr int kRecStride = 11;

struct SegFit {
    double probL;
    int n, minid, inf_flag;
};

// ---------------------------------------------------------------------------
// LDS carve-up (all offsets multiples of 16 B).
// ---------------------------------------------------------------------------
struct LdsLayout {
    size_t fit, q, col, rowp, vlist, sv, chain0, chain_bytes, wave0, wave_bytes, total;
    size_t c_label, c_rbits, c_rpre, c_scal, c_ctl, c_ring;   // per chain: labels, root bit set + prefix counts (statistics, used by the committing wave), shared scalars, control words, changed-label ring (SPEC)
    size_t w_slots, w_cl, w_nflag, w_rng;                // per wave
};
// control words of a chain (SPEC): committed and version share an aligned 64-bit word, one LDS read per poll
constexpr int kCtlFailed = 1, kCtlCommitted = 2, kCtlVersion = 3;
constexpr int kRingEntries = 4;    // accepted moves whose changed labels are remembered (a wave is at most three steps ahead)
constexpr int kSpecScalars = 16;   // 64-bit words of a chain's scalars handed from wave to wave (speculative form)

// queue slots: one per node of the component, plus one per further 8 neighbours of a node
__host__ __device__ inline size_t queue_slots(int S, int E2) { return (size_t)S + (size_t)E2 / 8 + 8; }

// The FIFO lives in LDS next to the tables (one 32-bit word per slot: node | first CSR entry << 16; the entry count
// follows from the row pointers); frames whose tables do not fit keep it in HBM (a.slots_g, 64-bit slots), which
// leaves room for several chains per workgroup.
// rec_lds: the per-superpixel records (88 B each) sit in LDS too; without them eight chains fit a workgroup
// topo_lds: frames whose tables do not fit keep at least the CSR topology (row pointers, u16 columns) in LDS
// nspec: wavefronts per chain (1 = the plain form; 2..4 = the speculative form): labels and scalars are per chain,
//        the rest per wave
__host__ __device__ inline LdsLayout make_layout(int S, int E2, int wpb, bool tab_lds, bool rec_lds, bool topo_lds = false,
                                                 int nspec = 1)
{
    auto A = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const bool spec = nspec > 1;
    LdsLayout L;
    size_t o = 0;
    L.fit = o; if (tab_lds && rec_lds) o += A((size_t)S * kRecStride * 8);
    L.q = o; if (tab_lds) o += A((size_t)E2 * 8);
    L.col = o; if (tab_lds || topo_lds) o += A((size_t)E2 * 2);
    L.rowp = o; if (tab_lds || topo_lds) o += A((size_t)(S + 1) * 4);
    L.vlist = o; o += A((size_t)S * 2);
    L.sv = o; o += 16;
    const size_t NW = (size_t)(S + 31) / 32;
    size_t cb = 0;
    L.c_label = cb; cb += A((size_t)S * 2);
    L.c_rbits = cb; cb += A(NW * 4);
    L.c_rpre = cb; cb += A(NW * 4);
    L.c_scal = cb; if (spec) cb += (size_t)kSpecScalars * 8;
    L.c_ctl = cb; if (spec) cb += 16;
    L.c_ring = cb; if (spec) cb += kRingEntries * 4 * 2;
    L.chain0 = o; L.chain_bytes = cb;
    o += cb * (size_t)wpb;
    L.wave0 = o;
    size_t w = 0;
    L.w_slots = w; if (tab_lds) w += A(queue_slots(S, E2) * 4);
    L.w_cl = w; w += A((size_t)S);
    L.w_nflag = w; w += A(NW * 4);
    L.w_rng = w; w += 4 * 64 * 4;
    L.wave_bytes = w;
    L.total = o + w * (size_t)wpb * (size_t)(spec ? nspec : 1);
    return L;
}

// ---------------------------------------------------------------------------
// The chain kernel.
// ---------------------------------------------------------------------------
// PROF: diagnostic build that accumulates shader-clock cycles per phase of the step into
// a.prof[12] (never used for reported numbers: the stamps serialise the wave).
#define MC_PHASE(k)                                                    \
    do {                                                               \
        if constexpr (PROF) {                                          \
            const unsigned long long _t = __builtin_readcyclecounter(); \
            pacc[k] += _t - ptime;                                     \
            ptime = _t;                                                \
        }                                                              \
    } while (0)

constexpr uint32_t kFree = 0xFFu;          // cl[j] (one byte per superpixel): j is not in the component; else the lane that claimed it
constexpr uint32_t kPartShift = 20u;       // FIFO slot word: node (16) | entries (4) | part (12)

// SPEC: NSPEC (2..4) wavefronts per chain.  A step's proposal is a pure function of (chain state, step index) -- the
// Philox counter is (step, position) -- and at C2 about 94 % of the steps leave the state as it was, so a wave may
// evaluate step t+1, t+2, ... against the state BEFORE the steps ahead of it have been decided and is right whenever
// they change nothing.  The waves of a chain take step indices from a ticket counter; a wave evaluates its step at
// once (read-only on the labels and the segment cache, which belong to the chain; FIFO, claim words, flags and RNG
// batch are its own), then waits for its turn -- steps commit strictly in order.  A version word counts the changes
// of the state (odd while a commit rewrites it).  An accepted move only touches the labels of the segments it involves
// (old segment, target segment, the two new names), and it publishes those four labels before it rewrites anything; an
// evaluation that has seen none of them -- its own segment and the neighbouring segments it flagged are all it ever
// read -- is an evaluation of the new state as well and stays valid.  A wave whose evaluation did see one of them
// evaluates again as soon as the new state is complete, without waiting for its turn, so that after an accepted move
// the waves behind it recover side by side, not one after the other.  Committing (statistics, trace, relabel, cache rows,
// counters) is the same code as in the one-wave form and runs in step order, so the chain is the same chain bit for
// bit; the scalars it carries travel from wave to wave through LDS.  With one chain per SIMD the step is bound by
// the latency of its dependent LDS round trips, and the further waves fill the issue slots the first leaves idle.
// Waits are bounded by lack of progress (no step of the chain committed for ~0.1 s): such a wave marks the chain
// failed (MCMC_E_DEVICE) instead of hanging.
template <int WPB, bool TAB_LDS, bool REC_LDS, bool PROF, bool TOPO_LDS = false, int NSPEC = 1>
__global__ __launch_bounds__(WPB * 64 * NSPEC) void chain_run_kernel(ChainArgs a, int s_lds, int e2_lds)
{
    constexpr bool SPEC = NSPEC > 1;
    constexpr int NT = WPB * 64 * NSPEC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = (int)threadIdx.x;
    const int wave = uni(tid >> 6);            // scalar: addresses derived from it stay in SGPRs
    const int lane = tid & 63;
    const int chain0 = (int)blockIdx.x * WPB;
    const int frame = chain0 / a.chains_per_frame;
    const int S = a.S[frame];
    const int E2 = (int)a.E2[frame];
    const int NW = (S + 31) / 32;
    const LdsLayout L = make_layout(s_lds, e2_lds, WPB, TAB_LDS, REC_LDS, TOPO_LDS, NSPEC);

    const double* g_fit = a.fit + (size_t)frame * a.s_max * kFitStride;
    const double* g_q = a.q + (size_t)frame * a.e2_max;
    const uint32_t* g_col = a.col + (size_t)frame * a.e2_max;
    const uint32_t* g_rowp = a.rowp + (size_t)frame * (a.s_max + 1);
    const uint8_t* g_valid = a.valid + (size_t)frame * a.s_max;

    double* l_fit = reinterpret_cast<double*>(smem + L.fit);
    double* l_q = reinterpret_cast<double*>(smem + L.q);
    uint16_t* l_col = reinterpret_cast<uint16_t*>(smem + L.col);
    uint32_t* l_rowp = reinterpret_cast<uint32_t*>(smem + L.rowp);
    uint16_t* vlist = reinterpret_cast<uint16_t*>(smem + L.vlist);

    // ---- stage the frame tables (block-cooperative) ----
    if (TAB_LDS && REC_LDS) {
        for (int i = tid; i < S * kRecStride; i += NT) {
            const int sp = i / kRecStride, k = i - sp * kRecStride;
            l_fit[i] = g_fit[(size_t)sp * kFitStride + (k < 9 ? k : k + 3)];
        }
    }
    if (TAB_LDS) for (int i = tid; i < E2; i += NT) l_q[i] = g_q[i];
    if (TAB_LDS || TOPO_LDS) {
        for (int i = tid; i < E2; i += NT) l_col[i] = (uint16_t)g_col[i];
        for (int i = tid; i <= S; i += NT) l_rowp[i] = g_rowp[i];
    }
    // list of superpixels that passed the A1 filter, ascending (wave 0)
    int* s_Sv = reinterpret_cast<int*>(smem + L.sv);
    if (wave == 0) {
        int cnt = 0;
        for (int b = 0; b < S; b += kWave) {
            const int i = b + lane;
            const bool ok = i < S && g_valid[i];
            const uint64_t m = __ballot(ok);
            if (ok) vlist[cnt + mbcnt(m)] = (uint16_t)i;
            cnt += __popcll((unsigned long long)m);
        }
        // largest degree of the frame (bounds the per-node append loop of the growth)
        uint32_t md = 0;
        for (int i = lane; i < S; i += kWave) md = max(md, g_rowp[i + 1] - g_rowp[i]);
        md = wave_scan_max_u32(md);
        if (lane == 0) *s_Sv = cnt;
        if (lane == 63) s_Sv[1] = (int)md;
    }
    const int cw = SPEC ? uni(wave / NSPEC) : wave;          // chain within the workgroup
    const int role = SPEC ? wave - cw * NSPEC : 0;          // which of the chain's waves (SPEC)
    const int chain = chain0 + cw;
    unsigned char* cbase = smem + L.chain0 + (size_t)cw * L.chain_bytes;
    uint16_t* label = reinterpret_cast<uint16_t*>(cbase + L.c_label);
    unsigned long long* sscal = reinterpret_cast<unsigned long long*>(cbase + L.c_scal);   // SPEC only
    volatile uint32_t* ctl = reinterpret_cast<volatile uint32_t*>(cbase + L.c_ctl);       // SPEC: next ticket, failed, committed, version
    // SPEC: the labels an accepted move touched (old segment, target, the two new names), entry (version / 2) % kRingEntries
    volatile uint16_t* ring = reinterpret_cast<volatile uint16_t*>(cbase + L.c_ring);
    uint16_t* g_label = a.label + (size_t)(chain < a.n_chains ? chain : 0) * a.s_max;
    if (chain < a.n_chains && role == 0) {
        // the chain's labels (and, in the speculative form, its scalars and control words) before anyone steps
        for (int i = lane; i < S; i += kWave) label[i] = g_label[i];
        if constexpr (SPEC) {
            if (lane == 0) {
                sscal[0] = (unsigned long long)__double_as_longlong(a.energy[chain]);
                sscal[1] = (unsigned long long)__double_as_longlong(a.best_energy[chain]);
                sscal[2] = a.state_hash[chain]; sscal[3] = a.trace_hash[chain]; sscal[4] = a.n_accept[chain];
                sscal[5] = a.n_case[(size_t)chain * 3 + 0]; sscal[6] = a.n_case[(size_t)chain * 3 + 1];
                sscal[7] = a.n_case[(size_t)chain * 3 + 2];
                sscal[8] = (unsigned long long)(long long)a.status[chain];
                sscal[9] = a.top_count[chain]; sscal[10] = a.top_hash[chain]; sscal[11] = a.overflow[chain];
                sscal[12] = 0ull;   // dwell
                sscal[13] = (unsigned long long)__double_as_longlong(a.last_targetL[chain]);
                sscal[14] = (unsigned long long)__double_as_longlong(a.last_prop[chain]);
                ctl[0] = 0u; ctl[kCtlCommitted] = 0u; ctl[kCtlVersion] = 0u; ctl[kCtlFailed] = 0u;
            }
        }
    }
    __syncthreads();
    const int Sv = *s_Sv;
    const uint32_t maxdeg = (uint32_t)s_Sv[1];
    const uint32_t maxparts = (maxdeg + 7u) >> 3;
    if (chain >= a.n_chains || Sv == 0) return;

    // ---- table accessors: LDS or HBM/L2 ----
    const double* qv; const uint32_t* rowp;
    if constexpr (TAB_LDS) qv = l_q; else qv = g_q;
    if constexpr (TAB_LDS || TOPO_LDS) rowp = l_rowp; else rowp = g_rowp;
    // sums of superpixel i: 0..8 the moments, 9 Sbb, 10 ssr
    auto recAt = [&](uint32_t i, int k) -> double {
        if constexpr (TAB_LDS && REC_LDS) return l_fit[(size_t)i * kRecStride + k];
        else return g_fit[(size_t)i * kFitStride + (k < 9 ? k : k + 3)];
    };
    auto colAt = [&](uint32_t e) -> uint32_t {
        if constexpr (TAB_LDS || TOPO_LDS) return (uint32_t)l_col[e]; else return g_col[e];
    };

    // ---- wave-private state ----
    unsigned char* wbase = smem + L.wave0 + (size_t)wave * L.wave_bytes;
    // FIFO of the growth.  A slot stands for at most 8 neighbours of one node (a node with more takes consecutive
    // slots, "parts").  In LDS a slot is one word, node | first CSR entry << 16 (S, E2 < 65536 whenever the tables
    // fit LDS): the number of entries is min(8, rowp[node + 1] - first), and the slot is a node's first iff
    // first == rowp[node].  In HBM (tables behind L2) a slot is .x = node | entries << 16 (4 bits) | part << 20
    // (12 bits), .y = first CSR entry.
    uint32_t* slots32 = reinterpret_cast<uint32_t*>(wbase + L.w_slots);
    uint2* slots64 = reinterpret_cast<uint2*>(a.slots_g) + ((size_t)chain * 4 + role) * queue_slots(s_lds, e2_lds);
    auto put_slot = [&](uint32_t idx, uint32_t node, uint32_t first, uint32_t cnt, uint32_t part) __attribute__((always_inline)) {
        if constexpr (TAB_LDS) slots32[idx] = node | (first << 16);
        else slots64[idx] = make_uint2(node | (cnt << 16) | (part << kPartShift), first);
    };
    // slot idx (valid = idx is a live slot; an idle lane reads slot 0 and gets cnt = 0)
    auto get_slot = [&](uint32_t idx, bool valid, uint32_t& node, uint32_t& first, uint32_t& cnt) __attribute__((always_inline)) {
        if constexpr (TAB_LDS) {
            const uint32_t w = slots32[valid ? idx : 0u];
            node = w & 0xFFFFu; first = w >> 16;
            const uint32_t left = rowp[node + 1u] - first;
            cnt = valid ? min(8u, left) : 0u;
        } else {
            const uint2 ent = slots64[valid ? idx : 0u];
            node = ent.x & 0xFFFFu; first = ent.y;
            cnt = valid ? ((ent.x >> 16) & 0xFu) : 0u;
        }
    };
    // appends to the FIFO are read by other lanes of this wave in a later pass
    auto queue_fence = [&]() {
        if constexpr (TAB_LDS) wave_fence(); else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    };
    uint8_t* cl = wbase + L.w_cl;                                         // kFree, or in the component
    uint32_t* clw = reinterpret_cast<uint32_t*>(wbase + L.w_cl);
    uint32_t* nflagw = reinterpret_cast<uint32_t*>(wbase + L.w_nflag);    // bit set by label: 1 = neighbouring segment
    const int NWB = (int)(((size_t)S + 15) & ~(size_t)15) / 4;            // words of the claim bytes (padded)
    auto nflag_set = [&](uint32_t l) __attribute__((always_inline)) { atomicOr(&nflagw[l >> 5], 1u << (l & 31u)); };
    auto nflag_get = [&](uint32_t l) -> uint32_t { return (nflagw[l >> 5] >> (l & 31u)) & 1u; };
    uint32_t* rbits = reinterpret_cast<uint32_t*>(cbase + L.c_rbits);
    uint32_t* rpre = reinterpret_cast<uint32_t*>(cbase + L.c_rpre);

    // per-segment cache, row = the segment's name: 10 exact sums, sum of ssr, probabilityL, member count
    double* g_segC = a.segC + (size_t)chain * a.s_max * kSegCacheStride;
    for (int w = lane; w < NWB; w += kWave) clw[w] = 0xFFFFFFFFu;
    for (int w = lane; w < NW; w += kWave) nflagw[w] = 0u;

    double energy = a.energy[chain], best_energy = a.best_energy[chain];
    uint64_t state_hash = a.state_hash[chain], trace_hash = a.trace_hash[chain];
    uint64_t n_accept = a.n_accept[chain];
    uint64_t n_case0 = a.n_case[(size_t)chain * 3 + 0], n_case1 = a.n_case[(size_t)chain * 3 + 1],
             n_case2 = a.n_case[(size_t)chain * 3 + 2];
    int status = a.status[chain];
    uint64_t top_count = a.top_count[chain], top_hash = a.top_hash[chain], overflow = a.overflow[chain];
    uint64_t dwell = 0;
    // targetRatio (log) and proposalRatio of the last step taken, as Problem::propose hands them to its caller
    // (Problem.cpp:213-217; MCMC.h:214-219)
    double last_targetL = a.last_targetL[chain], last_prop = a.last_prop[chain];

    const double var = a.var, nsp = a.new_seg_prob;
    const double log_var = det_log(var);

    Rng g;
    g.k0 = a.seed; g.k1 = uni(a.chain_offset + (uint32_t)chain);
    g.buf = reinterpret_cast<uint32_t*>(wbase + L.w_rng);

    unsigned long long* marg = a.marg + (size_t)frame * a.s_max * a.lmax;
    unsigned long long* vh = a.visit_hash + (size_t)chain * a.hcap;
    unsigned long long* vc = a.visit_count + (size_t)chain * a.hcap;

    // Book `dwell` visits of the CURRENT state: Statistics::update (Statistics.h:33-47)
    // and the label-marginal histogram.  The state only changes on an accepted
    // split/merge, so visits are booked when the chain leaves a state.
    auto flush_stats = [&]() {
        if (dwell == 0) return;
        // visit table (Statistics::entries_, Statistics.h:30-47): open addressing with linear probing, count == 0
        // marks an empty slot.  64 slots per trip, one per lane; the first slot in probe order that holds this
        // state or is empty takes the visits.
        uint64_t cnt = 0;
        int snap = 0;
        {
            const uint32_t hcap = (uint32_t)a.hcap;
            const uint32_t h0 = (uint32_t)(state_hash % (uint64_t)hcap);
            bool done = false;
            for (uint32_t p0 = 0; p0 < hcap && !done; p0 += kWave) {
                const uint32_t p = p0 + (uint32_t)lane;
                uint32_t idx = h0 + p;
                if (idx >= hcap) idx -= hcap;
                const bool in = p < hcap;
                const uint64_t c = in ? vc[idx] : 0ull;
                const uint64_t hh = in ? vh[idx] : 0ull;
                const uint64_t m = __ballot(in && (c == 0ull || hh == state_hash));
                if (m) {
                    const int w = __ffsll((unsigned long long)m) - 1;
                    uint64_t nc = c + dwell;
                    if (lane == w) { if (c == 0ull) vh[idx] = state_hash; vc[idx] = nc; }
                    cnt = (uint64_t)__shfl((unsigned long long)nc, w, kWave);
                    done = true;
                }
            }
            // the slot is read back by other lanes of this wave at a later flush (same CU, same L1)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (!done) overflow += dwell;
            else if (cnt > top_count || (cnt == top_count && state_hash < top_hash)) {
                snap = (state_hash != top_hash || top_count == 0) ? 1 : 0;
                top_count = cnt; top_hash = state_hash;
            }
        }
        if (snap) {
            uint16_t* tl = a.top_label + (size_t)chain * a.s_max;
            for (int i = lane; i < S; i += kWave) tl[i] = label[i];
        }
        // first-appearance rank of every root (Segmentation.cpp:210-227)
        uint32_t running = 0;
        for (int b = 0; b < S; b += kWave) {
            const int i = b + lane;
            const bool root = i < S && label[i] == (uint16_t)i;
            const uint64_t m = __ballot(root);
            if (lane == 0) {
                rbits[(b >> 5)] = (uint32_t)m; rpre[(b >> 5)] = running;
                if ((b >> 5) + 1 < NW) { rbits[(b >> 5) + 1] = (uint32_t)(m >> 32); rpre[(b >> 5) + 1] = running + (uint32_t)__popc((uint32_t)m); }
            }
            running += (uint32_t)__popcll((unsigned long long)m);
        }
        for (int i = lane; i < S; i += kWave) {
            const uint32_t Lb = label[i];
            if (Lb == kNoLabel) continue;
            const uint32_t rank = rpre[Lb >> 5] + (uint32_t)__popc(rbits[Lb >> 5] & ((1u << (Lb & 31u)) - 1u));
            const uint32_t bin = rank < (uint32_t)a.lmax ? rank : (uint32_t)a.lmax - 1u;
            atomicAdd(marg + (size_t)i * a.lmax + bin, (unsigned long long)dwell);
        }
        dwell = 0;
    };

    unsigned long long pacc[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long ptime = PROF ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long total = a.burn_in + a.iters;

    // ---- a step = evaluate (reads the chain's state, writes only this wave's scratch) + commit (in step order) ----
    // what an evaluation hands to the commit
    uint32_t start = 0, orig = 0, next = 0, csize = 0, qt = 0;
    int kase = 0, st = 0;
    bool have1 = false, accept = false;
    double targetL = 0.0, prop = 0.0;
    SegFit f1 = { 0.0, 0, 0, 0 }, f2 = { 0.0, 0, 0, 0 };
    // the sums of the two proposed segments as cache rows, one entry per lane: lane l of every row of 16 lanes holds
    // entry l & 15 (0..9 the exact sums, 10 the sum of ssr); a register pair each instead of twenty
    double m2v = 0.0, m1v = 0.0;
    // products: lane = 8 * slot + neighbour (the oracle's wave order of the factors)
    const uint32_t g8 = (uint32_t)lane >> 3, s8 = (uint32_t)lane & 7u;
    // growth: FIFO place within a pass = 63 - lane, i.e. the first slot's first neighbour sits in lane 63.  When several
    // lanes store to one LDS address in one instruction the highest lane's store is the one that lands on this hardware
    // (tools/micro/lds_write_order.hip), so the claim of the first edge in FIFO order lands by itself; the arbitration
    // below is correct whichever store lands and only repeats stores when it was another one.
    const uint32_t rl = 63u - (uint32_t)lane;
    const uint32_t gg8 = rl >> 3, gs8 = rl & 7u;

    // the step's scratch (claim words of the component, neighbour flags) back to empty
    auto cleanup = [&]() __attribute__((always_inline)) {
        for (uint32_t h = (uint32_t)lane; h < qt; h += kWave) {
            uint32_t node;
            if constexpr (TAB_LDS) node = slots32[h] & 0xFFFFu; else node = slots64[h].x & 0xFFFFu;
            cl[node] = (uint8_t)kFree;
        }
        for (int w = lane; w < NW; w += kWave) nflagw[w] = 0u;
        wave_fence();
    };

    // commit of an evaluated step, in step order: MH bookkeeping, the jump if accepted, counters, trace
    // scalars of the chain travel through LDS in the speculative form: loaded at the start of a wave's turn, stored at its end
    auto load_scalars = [&]() __attribute__((always_inline)) {
        energy = __longlong_as_double((long long)sscal[0]); best_energy = __longlong_as_double((long long)sscal[1]);
        state_hash = sscal[2]; trace_hash = sscal[3]; n_accept = sscal[4];
        n_case0 = sscal[5]; n_case1 = sscal[6]; n_case2 = sscal[7];
        status = (int)(long long)sscal[8];
        top_count = sscal[9]; top_hash = sscal[10]; overflow = sscal[11]; dwell = sscal[12];
        last_targetL = __longlong_as_double((long long)sscal[13]); last_prop = __longlong_as_double((long long)sscal[14]);
    };
    auto store_scalars = [&]() __attribute__((always_inline)) {
        if (lane == 0) {
            sscal[0] = (unsigned long long)__double_as_longlong(energy); sscal[1] = (unsigned long long)__double_as_longlong(best_energy);
            sscal[2] = state_hash; sscal[3] = trace_hash; sscal[4] = n_accept;
            sscal[5] = n_case0; sscal[6] = n_case1; sscal[7] = n_case2;
            sscal[8] = (unsigned long long)(long long)status;
            sscal[9] = top_count; sscal[10] = top_hash; sscal[11] = overflow; sscal[12] = dwell;
            sscal[13] = (unsigned long long)__double_as_longlong(last_targetL); sscal[14] = (unsigned long long)__double_as_longlong(last_prop);
        }
    };
    // control words, read as wave-uniform scalars
    auto ctl_rd = [&](int k) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl[k]); };
    // committed count and version in one LDS read
    auto ctl_poll = [&](uint32_t& committed, uint32_t& version) __attribute__((always_inline)) {
        const unsigned long long w = *reinterpret_cast<volatile unsigned long long*>(const_cast<uint32_t*>(ctl) + kCtlCommitted);
        committed = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w);
        version = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w >> 32));
    };
    // ~0.1 s of s_sleep polling WITHOUT any step of the chain being committed: far beyond any step (a commit that books
    // visits in a large table included, since the count restarts whenever the chain advances); then the chain is marked failed
    constexpr uint32_t kSpinMax = 1u << 22;
#ifdef MC_SPEC_STATS
    unsigned long long st_wait = 0, st_eval = 0, st_nev = 0, st_kept = 0, st_redo = 0, st_commit = 0, st_t0 = __builtin_readcyclecounter();
#endif
    bool failed = false;
    unsigned long long ts_next = 0;
    for (;;) {
        // ---- the step this wave takes ----
        unsigned long long ts;
        if constexpr (SPEC) {
            uint32_t mine = 0;
            if (lane == 0) mine = atomicAdd(const_cast<uint32_t*>(ctl), 1u);
            ts = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)mine);
            if (ts >= total) break;
        } else {
            ts = ts_next++;
            if (ts >= total) break;
        }
        // nothing an evaluation hands to the commit outlives its step: said here, or the compiler carries every one of
        // these around the loop (registers and SGPR spills through the growth; 5 % of the eight-chain form's step)
        start = 0; orig = 0; next = 0; csize = 0; qt = 0; kase = 0; st = 0; have1 = false; accept = false;
        targetL = 0.0; prop = 0.0; m2v = 0.0; m1v = 0.0;
        f1 = SegFit{ 0.0, 0, 0, 0 }; f2 = SegFit{ 0.0, 0, 0, 0 };
        const unsigned long long t = a.step0 + ts;
        bool evaluated = false;
        uint32_t v0 = 0;
        // Plain form: one evaluation.  Speculative form: evaluate against the present version of the state, then wait
        // for this step's turn; whenever a newer version shows up first, evaluate again at once.
#pragma nounroll
        for (;;) {
            bool need;
            if constexpr (SPEC) {
                // a state that is not being rewritten (even version)
                uint32_t spin = 0, seen = ctl_rd(kCtlCommitted);
                uint32_t cur = ctl_rd(kCtlVersion);
                while ((cur & 1u) && spin < kSpinMax && ctl_rd(kCtlFailed) == 0u) {
                    __builtin_amdgcn_s_sleep(1);
                    cur = ctl_rd(kCtlVersion);
                    const uint32_t c1 = ctl_rd(kCtlCommitted);
                    if (c1 != seen) { seen = c1; spin = 0; } else ++spin;
                }
                if (cur & 1u) { failed = true; if (lane == 0) ctl[kCtlFailed] = 1u; break; }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                need = (!evaluated || cur != v0) && (int)(long long)sscal[8] == 0;
                if (need && evaluated) cleanup();
                v0 = cur;
            } else {
                need = status == 0;
            }
#ifdef MC_SPEC_STATS
            const unsigned long long st_e0 = __builtin_readcyclecounter();
            if (need) { st_nev++; if (evaluated) st_redo++; }
#endif
            if (need) {
                g.step_lo = (uint32_t)t; g.step_hi = (uint32_t)(t >> 32);
                const u32x4 raw0 = rng_gen(g, 0, lane);
                // pickSuperPixel: rand() % S, then one unused draw (Segmentation.cpp:155,161); the step's other
                // scalar draws are taken out of the batch now, before the growth moves on to later positions
                start = vlist[(uint32_t)__builtin_amdgcn_readlane((int)raw0.x, 0) % (uint32_t)Sv];
                const uint32_t u1_raw = (uint32_t)__builtin_amdgcn_readlane((int)raw0.x, 2);
                const uint32_t idx_draw = (uint32_t)__builtin_amdgcn_readlane((int)raw0.x, 3);
                const uint32_t mh_raw = (uint32_t)__builtin_amdgcn_readlane((int)raw0.x, 4);
                orig = label[start];
                // the old segment's cache row (sums, sum of ssr, probabilityL, size), entry l & 15 in lane l: in flight
                // during the growth, one register pair
                const double co_v = g_segC[(size_t)orig * kSegCacheStride + ((uint32_t)lane & 15u)];
                MC_PHASE(0);

                // ---- createComponent BFS (Segmentation.cpp:74-120), 8 queue slots x 8 neighbours per pass ----
                // The reference pops one node at a time and tries its neighbours in order.  Here the FIFO is made of
                // slots of at most 8 neighbours (a node with more takes consecutive slots), a pass takes the next 8
                // slots, lane = 8 * slot + neighbour, and the trial of neighbour s of slot k owns the uniform at
                // position 8 + 8 k + s (see Rng) whether or not it is needed.  A node reached through two open edges
                // of one pass is taken by the lower lane, which is the edge the sequential loop reaches first, so the
                // component and the FIFO order are exactly the sequential ones.
                uint32_t qn = 1, head = 0;
                {
                    const uint32_t rs0 = rowp[start], d0 = rowp[start + 1] - rs0;
                    const uint32_t parts0 = d0 <= 8u ? 1u : (d0 + 7u) >> 3;
                    for (uint32_t pp = (uint32_t)lane; pp < parts0; pp += kWave)
                        put_slot(pp, start, rs0 + 8u * pp, min(8u, d0 - 8u * pp), pp);
                    if (lane == 0) cl[start] = (uint8_t)0;
                    qt = parts0;
                    queue_fence();
                }
                while (head < qt) {
                    if (!TAB_LDS && qt - head > 8u) {
                        // Tables and FIFO behind L2 (large frames: long round trips, wide frontiers): two passes' worth
                        // of slots (head .. head+7 = A, head+8 .. head+15 = B) in one trip, so that their round trips
                        // overlap.  The sequential order is kept by the claim priority (every lane of A beats every lane
                        // of B: a node both reach goes to A, as it would if B ran after A) and by appending A's
                        // winners before B's.  B reads the claim words before A's claims land; the lanes that lose the
                        // claim this way are exactly those that would have found the node taken.  Trial positions are
                        // per slot, so the draws do not move.  (With everything in LDS this measured 5 % slower.)
                        const uint32_t K2 = min(qt - head, 16u);
                        uint32_t nodeA, firstA, cntA, nodeB, firstB, cntB;
                        get_slot(head + gg8, true, nodeA, firstA, cntA);
                        get_slot(head + 8u + gg8, 8u + gg8 < K2, nodeB, firstB, cntB);
                        const bool actA = gs8 < cntA, actB = gs8 < cntB;
                        const uint32_t eA = actA ? firstA + gs8 : 0u, eB = actB ? firstB + gs8 : 0u;
                        const double rA = to_unit(rng_trial(g, kTrialPos0 + 8u * head, 64u, (int)rl));
                        const double rB = to_unit(rng_trial(g, kTrialPos0 + 8u * head + 64u, 8u * (K2 - 8u), (int)rl));
                        const uint32_t jA = colAt(eA), jB = colAt(eB);
                        const double qxA = qv[eA], qxB = qv[eB];       // entry 0 for an idle lane: no branch around the load
                        const double qA = actA ? qxA : -1.0, qB = actB ? qxB : -1.0;
                        const uint32_t cA = cl[jA], cB = cl[jB];
                        const uint32_t lA = label[jA], lB = label[jB];
                        const uint32_t rjA = rowp[jA], djA = rowp[jA + 1] - rjA;
                        const uint32_t rjB = rowp[jB], djB = rowp[jB + 1] - rjB;
                        const bool freshA = !(qA < 0.0) && cA == kFree, freshB = !(qB < 0.0) && cB == kFree;
                        const bool joinA = freshA && lA == orig && (rA < qA);
                        const bool joinB = freshB && lB == orig && (rB < qB);
                        if (freshA && lA != orig) nflag_set(lA);
                        if (freshB && lB != orig) nflag_set(lB);
                        const uint64_t jmA = __ballot(joinA), jmB = __ballot(joinB);
                        if (jmA | jmB) {
                            // the first claimant of a node in FIFO order wins: tags 64 + lane for A, lane for B, the
                            // highest tag is the first.  Byte stores, B's before A's; then every claimant looks at what
                            // landed and a higher tag that lost stores again
                            const uint32_t tagA = 64u + (uint32_t)lane, tagB = (uint32_t)lane;
                            if (joinB) cl[jB] = (uint8_t)tagB;
                            wave_fence();
                            if (joinA) cl[jA] = (uint8_t)tagA;
                            wave_fence();
                            uint32_t vA = joinA ? (uint32_t)cl[jA] : 0xFFu, vB = joinB ? (uint32_t)cl[jB] : 0xFFu;
                            bool lostA = joinA && tagA > vA, lostB = joinB && tagB > vB;
                            while (__ballot(lostA || lostB)) {
                                if (lostB) cl[jB] = (uint8_t)tagB;
                                wave_fence();
                                if (joinA && tagA > (uint32_t)cl[jA]) cl[jA] = (uint8_t)tagA;
                                wave_fence();
                                vA = joinA ? (uint32_t)cl[jA] : 0xFFu; vB = joinB ? (uint32_t)cl[jB] : 0xFFu;
                                lostA = joinA && tagA > vA; lostB = joinB && tagB > vB;
                            }
                            const bool winA = joinA && vA == tagA;
                            const bool winB = joinB && vB == tagB;
                            const uint64_t wmA = __ballot(winA), wmB = __ballot(winB);
                            const uint32_t nA = (uint32_t)__popcll((unsigned long long)wmA), nB = (uint32_t)__popcll((unsigned long long)wmB);
                            // winners are appended in FIFO order, i.e. by descending lane
                            const uint32_t posA = nA - (uint32_t)mbcnt(wmA) - 1u, posB = nB - (uint32_t)mbcnt(wmB) - 1u;
                            if (maxdeg <= 8u) {
                                if (winA) put_slot(qt + posA, jA, rjA, djA, 0u);
                                if (winB) put_slot(qt + nA + posB, jB, rjB, djB, 0u);
                                qt += nA + nB;
                            } else if (maxdeg <= 16u) {
                                const bool bigA = winA && djA > 8u, bigB = winB && djB > 8u;
                                const uint64_t bmA = __ballot(bigA), bmB = __ballot(bigB);
                                const uint32_t nbA = (uint32_t)__popcll((unsigned long long)bmA), nbB = (uint32_t)__popcll((unsigned long long)bmB);
                                const uint32_t tA = nA + nbA;
                                const uint32_t baseA = qt + posA + (nbA - (uint32_t)mbcnt(bmA) - (bigA ? 1u : 0u));
                                const uint32_t baseB = qt + tA + posB + (nbB - (uint32_t)mbcnt(bmB) - (bigB ? 1u : 0u));
                                if (winA) put_slot(baseA, jA, rjA, min(8u, djA), 0u);
                                if (bigA) put_slot(baseA + 1u, jA, rjA + 8u, djA - 8u, 1u);
                                if (winB) put_slot(baseB, jB, rjB, min(8u, djB), 0u);
                                if (bigB) put_slot(baseB + 1u, jB, rjB + 8u, djB - 8u, 1u);
                                qt += tA + nB + nbB;
                            } else {
                                const uint32_t partsA = winA ? (djA <= 8u ? 1u : (djA + 7u) >> 3) : 0u;
                                const uint32_t partsB = winB ? (djB <= 8u ? 1u : (djB + 7u) >> 3) : 0u;
                                const uint32_t inclA = wave_scan_add_u32(partsA), inclB = wave_scan_add_u32(partsB);
                                const uint32_t tA = (uint32_t)__builtin_amdgcn_readlane((int)inclA, 63);
                                const uint32_t tB = (uint32_t)__builtin_amdgcn_readlane((int)inclB, 63);
                                const uint32_t baseA = qt + (tA - inclA), baseB = qt + tA + (tB - inclB);   // the parts of the lanes above
                                for (uint32_t pp = 0; pp < maxparts; ++pp) {
                                    if (pp < partsA) put_slot(baseA + pp, jA, rjA + 8u * pp, min(8u, djA - 8u * pp), pp);
                                    if (pp < partsB) put_slot(baseB + pp, jB, rjB + 8u * pp, min(8u, djB - 8u * pp), pp);
                                }
                                qt += tA + tB;
                            }
                            qn += nA + nB;
                            queue_fence();
                        }
                        head += K2;
                        if constexpr (PROF) pacc[15] += 1;
                        continue;
                    }
                    const uint32_t K = min(qt - head, 8u);
                    uint32_t node, first, cnt;
                    get_slot(head + gg8, gg8 < K, node, first, cnt);
                    const bool act = gs8 < cnt;
                    const uint32_t e = act ? first + gs8 : 0u;
                    const double r = to_unit(rng_trial(g, kTrialPos0 + 8u * head, 64u, (int)rl));
                    const uint32_t j = colAt(e);
                    const double qx = qv[e];                            // entry 0 for an idle lane: no branch around the load
                    const double qe = act ? qx : -1.0;
                    const uint32_t cj = cl[j];
                    const uint32_t lj = label[j];
                    const uint32_t rj = rowp[j], dj = rowp[j + 1] - rj;
                    const bool live = !(qe < 0.0);
                    const bool fresh = live && cj == kFree;
                    const bool join = fresh && lj == orig && (r < qe);   // strict (Segmentation.cpp:110)
                    if (fresh && lj != orig) nflag_set(lj);
                    const uint64_t jm0 = __ballot(join);
                    if (jm0) {
                        // several lanes may hold an open edge to the same node: the first in FIFO order -- the highest
                        // lane -- takes it.  Byte stores, then every claimant looks at what landed and a higher lane
                        // that lost stores again: the claim byte ends as the highest claiming lane whichever store the
                        // hardware lets land (on gfx950 the highest lane's, so nothing is stored twice)
                        if (join) cl[j] = (uint8_t)lane;
                        wave_fence();
                        uint32_t v = join ? (uint32_t)cl[j] : 0xFFu;
                        bool lost = join && (uint32_t)lane > v;
                        while (__ballot(lost)) {
                            if (lost) cl[j] = (uint8_t)lane;
                            wave_fence();
                            v = join ? (uint32_t)cl[j] : 0xFFu;
                            lost = join && (uint32_t)lane > v;
                        }
                        const bool win = join && v == (uint32_t)lane;
                        const uint64_t wm = __ballot(win);
                        const uint32_t nw = (uint32_t)__popcll((unsigned long long)wm);
                        const uint32_t pos = nw - (uint32_t)mbcnt(wm) - 1u;      // winners in FIFO order = by descending lane
                        if (maxdeg <= 8u) {
                            if (win) put_slot(qt + pos, j, rj, dj, 0u);
                            qt += nw;
                        } else if (maxdeg <= 16u) {
                            const bool big = win && dj > 8u;
                            const uint64_t bm = __ballot(big);
                            const uint32_t nb = (uint32_t)__popcll((unsigned long long)bm);
                            const uint32_t base = qt + pos + (nb - (uint32_t)mbcnt(bm) - (big ? 1u : 0u));
                            if (win) put_slot(base, j, rj, min(8u, dj), 0u);
                            if (big) put_slot(base + 1u, j, rj + 8u, dj - 8u, 1u);
                            qt += nw + nb;
                        } else {
                            const uint32_t parts = win ? (dj <= 8u ? 1u : (dj + 7u) >> 3) : 0u;
                            const uint32_t incl = wave_scan_add_u32(parts);
                            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                            const uint32_t base = qt + (tot - incl);             // the parts of the lanes above
                            for (uint32_t pp = 0; pp < maxparts; ++pp)
                                if (pp < parts) put_slot(base + pp, j, rj + 8u * pp, min(8u, dj - 8u * pp), pp);
                            qt += tot;
                        }
                        qn += nw;
                        queue_fence();
                    }
                    head += K;
                    if constexpr (PROF) pacc[15] += 1;
                }
                csize = qn;
                const uint32_t segn_orig = (uint32_t)readlane_f64(co_v, kSegCacheN);
                const bool same = csize == segn_orig;  // Segmentation.cpp:19
                MC_PHASE(1);

                // ---- selectSegment (Segmentation.cpp:10-57) ----
                next = orig; double ratio;
                if (!same && to_unit(u1_raw) < nsp) { kase = 1; ratio = nsp; }
                else {
                    int nnl = 0;
                    for (int w = lane; w < NW; w += kWave) nnl += __popc(nflagw[w]);
                    const uint32_t numSeg = (uint32_t)wave_sum_i32(nnl) + 1u;
                    const double existing = same ? 1.0 : (1.0 - nsp);
                    const uint32_t idx = idx_draw % numSeg;
                    if (idx == numSeg - 1u) { kase = 0; next = orig; }
                    else { kase = 2; next = nth_flag(nflagw, NW, idx, lane); }
                    ratio = existing / (double)numSeg;
                }

                MC_PHASE(2);
                targetL = 1.0; prop = ratio;  // Case A (Problem.cpp:161-171): targetRatio = 1.0 (sic)
                double logprop = 0.0;
                f1 = SegFit{ 0.0, 0, 0, 0 }; f2 = SegFit{ 0.0, 0, 0, 0 };
                m2v = 0.0; m1v = 0.0;
                st = 0;
                have1 = !same;  // sB1 = sA1 minus the component is not empty
                if (kase != 0) {
                    // cached row of the target segment (Segment::combine / diff on sums instead of a refit from
                    // the member list, Graph.cpp:35-67): issued now, needed after the pass over the component
                    // (entry l & 15 in lane l, as the old segment's)
                    double cn_v = 0.0;
                    if (kase == 2) cn_v = g_segC[(size_t)next * kSegCacheStride + ((uint32_t)lane & 15u)];
                    // separation / merge products over the component's cut edges (Segmentation.cpp:127-146;
                    // Graph.cpp:10-33), wave order over the examined-edge queue
                    double ps = 1.0, pm = 1.0;
                    // four blocks of 8 slots per trip: their LDS round trips overlap, the factors are applied in
                    // block order
                    for (uint32_t b8 = 0; b8 < qt; b8 += 32u) {
                        uint32_t first4[4], cnt4[4];
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t sl = b8 + 8u * (uint32_t)u + g8;
                            uint32_t node;
                            get_slot(sl, sl < qt, node, first4[u], cnt4[u]);
                        }
                        double qe[4]; uint32_t jj[4];
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool act = s8 < cnt4[u];
                            const uint32_t e = act ? first4[u] + s8 : 0u;
                            const double qx = qv[e];
                            qe[u] = act ? qx : -1.0;
                            jj[u] = colAt(e);
                        }
                        uint32_t lj[4], cj[4];
    #pragma unroll
                        for (int u = 0; u < 4; ++u) { lj[u] = label[jj[u]]; cj[u] = cl[jj[u]]; }
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool live = !(qe[u] < 0.0);
                            const double w = 1.0 - qe[u];
                            // a factor of 1.0 is exact, so the untaken edges need no branch
                            ps = ps * ((live && lj[u] == orig && cj[u] == kFree) ? w : 1.0);
                            pm = pm * ((live && kase == 2 && lj[u] == next && lj[u] != orig) ? w : 1.0);
                        }
                    }
                    // the component's sums, wave order over the FIFO slots (a node counts at its first slot)
                    double rp = 0.0;
                    double mc[kSumStride];
    #pragma unroll
                    for (int k = 0; k < kSumStride; ++k) mc[k] = 0.0;
                    int mn = 0x7FFFFFFF;
                    for (uint32_t h = (uint32_t)lane; h < qt; h += kWave) {
                        uint32_t i;
                        if constexpr (TAB_LDS) {
                            const uint32_t w = slots32[h];
                            i = w & 0xFFFFu;
                            if ((w >> 16) != rowp[i]) continue;
                        } else {
                            const uint32_t ex = slots64[h].x;
                            if ((ex >> kPartShift) != 0u) continue;
                            i = ex & 0xFFFFu;
                        }
    #pragma unroll
                        for (int k = 0; k < kSumStride; ++k) mc[k] += recAt(i, k);
                        rp += recAt(i, kSumStride);
                        mn = min(mn, (int)i);
                    }
                    MC_PHASE(3);
                    const double sep = wave_prod_f64(ps);
                    const double mrg = kase == 2 ? wave_prod_f64(pm) : 1.0;
                    // the eleven sums in one folded butterfly (device_math.hpp: the bits of eleven wave_sum_f64 for half
                    // the instructions), left one per lane: entry l & 15 in lane l, like the cache rows
                    double part[kSumStride + 1];
    #pragma unroll
                    for (int k = 0; k < kSumStride; ++k) part[k] = mc[k];
                    part[kSumStride] = rp;
                    const double sv = wave_sum11_lanes_f64(part, lane);
                    const int minC = wave_min_i32(mn);
                    MC_PHASE(4);
                    // Segment::combine / diff on sums (Graph.cpp:35-67): sB2 = sA2 + C (or C alone), sB1 = sA1 - C
                    m2v = kase == 2 ? cn_v + sv : sv;
                    m1v = co_v - sv;
                    if (kase == 1) prop = ratio * (1.0 / sep);              // Problem.cpp:188
                    else prop = ratio * (mrg / sep);                        // Problem.cpp:206
                    // the two evaluations run side by side: lanes 0..31 take sB2, lanes 32..63 sB1; the three
                    // pivot logarithms of each and log(proposal ratio) are taken by different lane groups
                    {
                        const bool hi = lane >= 32 && have1;   // without sB1 both halves evaluate sB2
                        // entry k of the segment a half evaluates, to every lane of the half (LDS crossbar, no vector op)
                        const double zv = hi ? m1v : m2v;
                        Sym3 A = { bcast_half_f64<0>(zv), bcast_half_f64<1>(zv), bcast_half_f64<2>(zv),
                                   bcast_half_f64<3>(zv), bcast_half_f64<4>(zv), bcast_half_f64<5>(zv) };
                        Vec3 rr = { bcast_half_f64<6>(zv), bcast_half_f64<7>(zv), bcast_half_f64<8>(zv) }, th, d;
                        const double quad = ldl3(A, rr, th, d);
                        const double err = ((bcast_half_f64<9>(zv) - quad) - bcast_half_f64<10>(zv)) / (2.0 * var);
                        const int gg = (lane >> 3) & 3;
                        const double lx = det_log(gg == 0 ? d.x : (gg == 1 ? d.y : (gg == 2 ? d.z : prop)));
                        logprop = readlane_f64(lx, 24);
                        const double logdet2 = ((readlane_f64(lx, 0) + readlane_f64(lx, 8)) + readlane_f64(lx, 16)) - 3.0 * log_var;
                        const double logdet1 = ((readlane_f64(lx, 32) + readlane_f64(lx, 40)) + readlane_f64(lx, 48)) - 3.0 * log_var;
                        double p2 = -readlane_f64(err, 0) + 0.5 * (3.0 * kLog2Pi - logdet2);
                        double p1 = -readlane_f64(err, 32) + 0.5 * (3.0 * kLog2Pi - logdet1);
                        if (p2 != p2) p2 = 0.75;                 // Graph.cpp:134
                        if (p1 != p1) p1 = 0.75;
                        f2.probL = p2; f2.inf_flag = (p2 - p2 != 0.0) ? 1 : 0;   // Graph.cpp:137
                        f1.probL = have1 ? p1 : 0.0; f1.inf_flag = (have1 && p1 - p1 != 0.0) ? 1 : 0;
                    }
                    f2.n = (int)csize + (kase == 2 ? (int)readlane_f64(cn_v, kSegCacheN) : 0);
                    f2.minid = (kase == 2 && (int)next < minC) ? (int)next : minC;
                    f1.n = have1 ? (int)segn_orig - (int)csize : 0;
                    MC_PHASE(5);
                    // jumpProbability (Problem.cpp:475-487)
                    const double sB2_sA1 = f2.probL - readlane_f64(co_v, kSegCacheP);
                    const double sB1_sA2 = (have1 ? f1.probL : 0.0) - (kase == 2 ? readlane_f64(cn_v, kSegCacheP) : 0.0);
                    targetL = sB2_sA1 + sB1_sA2;
                    if (f1.inf_flag || f2.inf_flag) st = -5;
                    if (targetL != targetL || prop != prop) st = -4;        // Problem.cpp:216-217
                    MC_PHASE(6);
                }

                MC_PHASE(7);
                // ---- FastMetropolisHastings::iterate (MCMC.h:218-227) ----
                if (kase == 0) logprop = det_log(prop);
                const double aL = targetL + logprop;
                const double acc = det_exp(aL);
                const double r = to_unit(mh_raw);  // always drawn
                accept = true;
                if (aL < 0.0) { if (r > acc) accept = false; }
                if (st != 0) accept = false;
                evaluated = true;
            }
#ifdef MC_SPEC_STATS
            const unsigned long long st_w0 = __builtin_readcyclecounter();
            if (need) st_eval += st_w0 - st_e0;
#endif
            if constexpr (!SPEC) break;
            else {
                // this step's turn, or a change of the state, whichever comes first
                bool redo = false;
                for (;;) {
                    uint32_t spin = 0, seen, cur;
                    ctl_poll(seen, cur);
                    while (seen != (uint32_t)ts && cur == v0 && spin < kSpinMax) {
                        __builtin_amdgcn_s_sleep(2);
                        uint32_t c1;
                        ctl_poll(c1, cur);
                        if (c1 != seen) { seen = c1; spin = 0; }
                        else if ((++spin & 1023u) == 0u && ctl_rd(kCtlFailed) != 0u) break;
                    }
                    if (cur != v0) {
                        // accepted moves since this evaluation (the last one perhaps still being written: its labels
                        // were published before the version went odd): did it read anything they touch?
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                        bool ok = evaluated;
                        if (evaluated) {
                            for (uint32_t k = v0 >> 1; k < ((cur + 1u) >> 1); ++k) {
                                const volatile uint16_t* d = ring + (k & (uint32_t)(kRingEntries - 1)) * 4u;
    #pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    const uint32_t dl = d[c];
                                    ok = ok && dl != orig && nflag_get(dl) == 0u;
                                }
                            }
                        }
#ifdef MC_SPEC_STATS
                        if (__builtin_amdgcn_readfirstlane((int)ok)) st_kept++;
#endif
                        if (__builtin_amdgcn_readfirstlane((int)ok)) { v0 = (cur + 1u) & ~1u; continue; }   // still an evaluation of the newest state
                        redo = true;
                        break;
                    }
                    if (seen == (uint32_t)ts) {
                        // every earlier step is committed: nobody else changes the state now
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                        if (ctl_rd(kCtlVersion) == v0) break;
                        continue;                                  // the last commit ahead changed it: look at what it touched
                    }
                    failed = true; if (lane == 0) ctl[kCtlFailed] = 1u;      // no progress (or another wave gave up)
                    break;
                }
#ifdef MC_SPEC_STATS
                st_wait += __builtin_readcyclecounter() - st_w0;
#endif
                if (!redo) break;
            }
        }
#ifdef MC_SPEC_STATS
        const unsigned long long st_c0 = __builtin_readcyclecounter();
#endif
        if constexpr (SPEC) {
            if (failed) break;
            load_scalars();
        }

        // ---- commit, in step order: MH bookkeeping, the jump if accepted, counters, trace ----
        uint8_t tb = 0xFF;
        if (status == 0) {
            MC_PHASE(8);
            last_targetL = targetL; last_prop = prop;
            if (st != 0) status = st;
            uint32_t newlab = next;
            if (accept && kase != 0) {
                flush_stats();  // leaving the current state
                MC_PHASE(9);
                // jump (Problem.cpp:32-74) with smallest-member naming
                const uint32_t L2 = (uint32_t)f2.minid;
                uint32_t L1 = orig;
                if (have1 && cl[orig] != kFree) {   // the old name moves with the component: smallest id left behind
                    int m1 = 0x7FFFFFFF;
                    for (int i = lane; i < S; i += kWave)
                        if (label[i] == orig && cl[i] == kFree) m1 = min(m1, i);
                    L1 = (uint32_t)wave_min_i32(m1);
                }
                if constexpr (SPEC) {
                    // what this move touches, then odd: the state is being rewritten
                    if (lane == 0) {
                        const uint32_t v = ctl[kCtlVersion];
                        volatile uint16_t* d = ring + ((v >> 1) & (uint32_t)(kRingEntries - 1)) * 4u;
                        d[0] = (uint16_t)orig; d[1] = (uint16_t)(kase == 2 ? next : orig); d[2] = (uint16_t)L2; d[3] = (uint16_t)L1;
                        ctl[kCtlVersion] = v + 1u;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                }
                uint64_t dh = 0;
                for (int i = lane; i < S; i += kWave) {
                    const uint32_t li = label[i];
                    if (li == kNoLabel) continue;
                    uint32_t nl = li;
                    if (cl[i] != kFree || (kase == 2 && li == next)) nl = L2;
                    else if (have1 && li == orig) nl = L1;
                    if (nl != li) {
                        dh += mix64(((uint64_t)i << 32) | nl) - mix64(((uint64_t)i << 32) | li);
                        label[i] = (uint16_t)nl;
                    }
                }
                state_hash += wave_sum_u64(dh);
                // the new segments' rows go to the cache; the stores are complete before any later step of
                // this wave reads them back (same CU, same L1: workgroup scope)
                if (lane < kSegCacheUsed) {
                    double v2 = m2v, v1 = m1v;                  // entries 0..10 as they are
                    if (lane == kSegCacheN) { v2 = (double)f2.n; v1 = (double)f1.n; }
                    if (lane == kSegCacheP) { v2 = f2.probL; v1 = f1.probL; }
                    g_segC[(size_t)L2 * kSegCacheStride + lane] = v2;
                    if (have1) g_segC[(size_t)L1 * kSegCacheStride + lane] = v1;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                if constexpr (SPEC) { if (lane == 0) ctl[kCtlVersion] = ctl[kCtlVersion] + 1u; }                                                    // even again: a new state
                energy += targetL;
                if (energy > best_energy) {
                    best_energy = energy;
                    uint16_t* bl = a.best_label + (size_t)chain * a.s_max;
                    for (int i = lane; i < S; i += kWave) bl[i] = label[i];
                }
                newlab = L2;
            }
            MC_PHASE(10);
            if constexpr (PROF) { pacc[12] += csize; pacc[13] += 8ull * qt; pacc[14] += (kase != 0 ? (unsigned long long)(f1.n + f2.n) : 0ull); }
            if (accept) n_accept++;
            if (kase == 0) n_case0++; else if (kase == 1) n_case1++; else n_case2++;
            tb = (uint8_t)((accept ? 1 : 0) | (kase << 1));
            const uint64_t wd = (uint64_t)tb | ((uint64_t)start << 8) | ((uint64_t)csize << 24) | ((uint64_t)newlab << 40);
            trace_hash = mix64(mix64(trace_hash ^ wd) ^ state_hash);
        }
        if (a.trace && lane == 0) a.trace[(size_t)chain * a.trace_stride + ts] = tb;
        if (ts >= a.burn_in && status == 0) dwell++;
        if constexpr (SPEC) {
            // the next step's turn begins before this wave tidies its own scratch
            store_scalars();
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            if (lane == 0) ctl[kCtlCommitted] = (uint32_t)ts + 1u;
        }
#ifdef MC_SPEC_STATS
        st_commit += __builtin_readcyclecounter() - st_c0;
#endif
        if (evaluated) cleanup();
        MC_PHASE(11);
    }
#ifdef MC_SPEC_STATS
    if (lane == 0 && a.prof) {
        atomicAdd(a.prof + 0, st_wait); atomicAdd(a.prof + 1, st_eval); atomicAdd(a.prof + 2, st_nev); atomicAdd(a.prof + 3, st_kept);
        atomicAdd(a.prof + 4, st_redo); atomicAdd(a.prof + 5, st_commit); atomicAdd(a.prof + 6, __builtin_readcyclecounter() - st_t0);
    }
#endif
    if constexpr (SPEC) {
        // the chain's first wave closes the run once every step is committed
        if (role != 0) return;
        uint32_t spin = 0, seen = ctl_rd(kCtlCommitted);
        while (seen != (uint32_t)total && spin < kSpinMax && ctl_rd(kCtlFailed) == 0u) {
            __builtin_amdgcn_s_sleep(1);
            const uint32_t c1 = ctl_rd(kCtlCommitted);
            if (c1 != seen) { seen = c1; spin = 0; } else ++spin;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        load_scalars();
        if (failed || seen != (uint32_t)total) {              // MCMC_E_DEVICE: a wait ran out (never seen; a hang otherwise)
            status = -2;
            if (lane == 0 && a.fault) atomicOr(a.fault, 1u);      // mcmc_ctx_synchronize reports it
        }
    }
    flush_stats();
    if constexpr (PROF) {
        if (lane == 0 && a.prof)
            for (int k = 0; k < 16; ++k) atomicAdd(a.prof + k, pacc[k]);
    }

    // ---- write the chain back ----
    for (int i = lane; i < S; i += kWave) g_label[i] = label[i];
    if (lane == 0) {
        a.energy[chain] = energy; a.best_energy[chain] = best_energy;
        a.state_hash[chain] = state_hash; a.trace_hash[chain] = trace_hash;
        a.n_accept[chain] = n_accept;
        a.n_case[(size_t)chain * 3 + 0] = n_case0; a.n_case[(size_t)chain * 3 + 1] = n_case1;
        a.n_case[(size_t)chain * 3 + 2] = n_case2;
        a.status[chain] = status;
        a.top_count[chain] = top_count; a.top_hash[chain] = top_hash; a.overflow[chain] = overflow;
        a.last_targetL[chain] = last_targetL; a.last_prop[chain] = last_prop;
    }
}

// ---------------------------------------------------------------------------
// Problem::initializeState (Problem.cpp:447-473): every superpixel that passed
// the filter in ONE segment.  One wavefront per chain.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void chain_init_kernel(ChainArgs a)
{
    const int chain = (int)blockIdx.x;
    const int lane = lane_id();
    const int frame = chain / a.chains_per_frame;
    const int S = a.S[frame];
    const double* fit = a.fit + (size_t)frame * a.s_max * kFitStride;
    const uint8_t* valid = a.valid + (size_t)frame * a.s_max;
    double s[kSumStride];
#pragma unroll
    for (int k = 0; k < kSumStride; ++k) s[k] = 0.0;
    double rp = 0.0;
    int n = 0, mn = 0x7FFFFFFF;
    for (int i = lane; i < S; i += kWave) {
        if (!valid[i]) continue;
        const double* f = fit + (size_t)i * kFitStride;
#pragma unroll
        for (int k = 0; k < 9; ++k) s[k] += f[k];
        s[9] += f[12];
        rp += f[13];
        n++; mn = min(mn, i);
    }
#pragma unroll
    for (int k = 0; k < kSumStride; ++k) s[k] = wave_sum_f64(s[k]);
    const double R = wave_sum_f64(rp);
    n = wave_sum_i32(n); mn = wave_min_i32(mn);
    double p = 0.0; int inf_flag = 0;
    if (n > 0) {
        Vec3 th; double err;
        p = segment_from_sums(s, R, a.var, det_log(a.var), th, err, &inf_flag);
    }
    const uint16_t root = n > 0 ? (uint16_t)mn : (uint16_t)0;
    uint16_t* lab = a.label + (size_t)chain * a.s_max;
    uint16_t* bl = a.best_label + (size_t)chain * a.s_max;
    uint16_t* tl = a.top_label + (size_t)chain * a.s_max;
    uint64_t h = 0;
    for (int i = lane; i < a.s_max; i += kWave) {
        const bool ok = i < S && valid[i];
        const uint16_t l = ok ? root : kNoLabel;
        lab[i] = l; bl[i] = l; tl[i] = l;
        if (ok) h += mix64(((uint64_t)i << 32) | l);
    }
    h = wave_sum_u64(h);
    if (n > 0 && lane < kSegCacheUsed) {
        double v = (double)n;
#pragma unroll
        for (int k = 0; k < kSumStride; ++k) if (lane == k) v = s[k];
        if (lane == kSumStride) v = R;
        if (lane == kSegCacheP) v = p;
        a.segC[((size_t)chain * a.s_max + root) * kSegCacheStride + lane] = v;
    }
    for (int i = lane; i < a.hcap; i += kWave) {
        a.visit_hash[(size_t)chain * a.hcap + i] = 0ull;
        a.visit_count[(size_t)chain * a.hcap + i] = 0ull;
    }
    if (lane == 0) {
        a.energy[chain] = p; a.best_energy[chain] = p;
        a.state_hash[chain] = h; a.trace_hash[chain] = 0ull;
        a.n_accept[chain] = 0ull;
        a.n_case[(size_t)chain * 3 + 0] = 0ull; a.n_case[(size_t)chain * 3 + 1] = 0ull;
        a.n_case[(size_t)chain * 3 + 2] = 0ull;
        a.status[chain] = inf_flag ? -5 : (n > 0 ? 0 : -7);
        a.top_count[chain] = 0ull; a.top_hash[chain] = 0ull; a.overflow[chain] = 0ull;
        a.last_targetL[chain] = 0.0; a.last_prop[chain] = 1.0;
    }
}

// Statistics grows: every (hash, count) entry of a chain's old visit table goes to its new, larger one (same
// probing as flush_stats).  One wavefront per chain; the new table was zeroed on the stream before.
__global__ __launch_bounds__(64) void visit_rehash_kernel(const unsigned long long* oh, const unsigned long long* oc, int ocap,
                                                          unsigned long long* nh, unsigned long long* nc, int ncap)
{
    const int chain = (int)blockIdx.x;
    const int lane = lane_id();
    oh += (size_t)chain * ocap; oc += (size_t)chain * ocap;
    nh += (size_t)chain * ncap; nc += (size_t)chain * ncap;
    for (int b = 0; b < ocap; b += kWave) {
        const int i = b + lane;
        const uint64_t c = i < ocap ? oc[i] : 0ull, h = i < ocap ? oh[i] : 0ull;
        uint64_t live = __ballot(c != 0ull);
        while (live) {
            const int src = __ffsll((unsigned long long)live) - 1;
            live &= live - 1;
            const uint64_t hs = (uint64_t)__shfl((unsigned long long)h, src, kWave);
            const uint64_t cs = (uint64_t)__shfl((unsigned long long)c, src, kWave);
            const uint32_t h0 = (uint32_t)(hs % (uint64_t)ncap);
            for (uint32_t p0 = 0; p0 < (uint32_t)ncap; p0 += kWave) {
                const uint32_t p = p0 + (uint32_t)lane;
                uint32_t idx = h0 + p;
                if (idx >= (uint32_t)ncap) idx -= (uint32_t)ncap;
                const bool in = p < (uint32_t)ncap;
                const uint64_t m = __ballot(in && nc[idx] == 0ull);
                if (m) {
                    if (lane == __ffsll((unsigned long long)m) - 1) { nh[idx] = hs; nc[idx] = cs; }
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
    }
}

// Largest number of occupied slots over the chains' visit tables (what decides whether a table must grow).
__global__ __launch_bounds__(64) void visit_occupancy_kernel(const unsigned long long* vc, int hcap, unsigned int* out_max)
{
    const int lane = lane_id();
    vc += (size_t)blockIdx.x * hcap;
    int n = 0;
    for (int i = lane; i < hcap; i += kWave) n += vc[i] != 0ull;
    n = wave_sum_i32(n);
    if (lane == 0) atomicMax(out_max, (unsigned int)n);
}

// Segmentation::computeProbability (Segmentation.cpp:185-195): sum of the cached
// segment log-probabilities, wave order over roots by id.
__global__ __launch_bounds__(64) void chain_energy_kernel(ChainArgs a, double* out)
{
    const int chain = (int)blockIdx.x;
    const int lane = lane_id();
    const int frame = chain / a.chains_per_frame;
    const int S = a.S[frame];
    const uint16_t* lab = a.label + (size_t)chain * a.s_max;
    const double* segC = a.segC + (size_t)chain * a.s_max * kSegCacheStride;
    double part = 0.0;
    for (int i = lane; i < S; i += kWave)
        if (lab[i] == (uint16_t)i) part += segC[(size_t)i * kSegCacheStride + kSegCacheP];
    const double e = wave_sum_f64(part);
    if (lane == 0) out[chain] = e;
}

// First-appearance relabel (Segmentation.cpp:197-233) of n_chains label rows.
__global__ __launch_bounds__(64) void canonical_labels_kernel(const uint16_t* src, const int* S_arr, int s_max,
                                                              int chains_per_frame, uint16_t* dst)
{
    extern __shared__ uint32_t sh[];  // rbits[NW] | rpre[NW]
    const int chain = (int)blockIdx.x;
    const int lane = lane_id();
    const int S = S_arr[chain / chains_per_frame];
    const int NW = (s_max + 63) / 64 * 2;
    uint32_t* rbits = sh;
    uint32_t* rpre = sh + NW;
    const uint16_t* lab = src + (size_t)chain * s_max;
    uint16_t* out = dst + (size_t)chain * s_max;
    uint32_t running = 0;
    for (int b = 0; b < S; b += kWave) {
        const int i = b + lane;
        const bool root = i < S && lab[i] == (uint16_t)i;
        const uint64_t m = __ballot(root);
        if (lane == 0) {
            rbits[b >> 5] = (uint32_t)m; rpre[b >> 5] = running;
            rbits[(b >> 5) + 1] = (uint32_t)(m >> 32); rpre[(b >> 5) + 1] = running + (uint32_t)__popc((uint32_t)m);
        }
        running += (uint32_t)__popcll((unsigned long long)m);
    }
    __syncthreads();
    for (int i = lane; i < s_max; i += kWave) {
        uint16_t o = kNoLabel;
        if (i < S) {
            const uint32_t Lb = lab[i];
            if (Lb != kNoLabel)
                o = (uint16_t)(rpre[Lb >> 5] + (uint32_t)__popc(rbits[Lb >> 5] & ((1u << (Lb & 31u)) - 1u)));
        }
        out[i] = o;
    }
}

// The one labeling a caller prints (printResults, can05_sampler.cpp:23-45; north_star's "best-energy labeling" that the
// ranks all-gather): arg-max over the chains on the device, ties to the lowest chain id, and that chain's labels, as
// one packed record rec[0] = the key (which = 1: best energy, f64 bits; which = 2: visits of the most-visited state),
// rec[1] = global chain id, then the labels, four u16 per word (kNoLabel beyond the frame's superpixels).
__global__ __launch_bounds__(64) void best_record_kernel(ChainArgs a, int which, int canonical, unsigned long long* rec)
{
    extern __shared__ uint32_t sh[];  // rbits[NW] | rpre[NW]
    const int lane = lane_id();
    int besti = 0x7FFFFFFF;
    double bestv = 0.0;
    unsigned long long bestc = 0ull;
    for (int k = lane; k < a.n_chains; k += kWave) {
        if (which == 1) {
            const double v = a.best_energy[k];
            if (besti == 0x7FFFFFFF ? v == v : v > bestv) { bestv = v; besti = k; }   // NaN never wins
        } else {
            const unsigned long long c = a.top_count[k];
            if (besti == 0x7FFFFFFF || c > bestc) { bestc = c; besti = k; }
        }
    }
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const int oi = __shfl_xor(besti, off, kWave);
        const double ov = __shfl_xor(bestv, off, kWave);
        const unsigned long long oc = __shfl_xor(bestc, off, kWave);
        bool take;
        if (oi == 0x7FFFFFFF) take = false;
        else if (besti == 0x7FFFFFFF) take = true;
        else if (which == 1) take = ov > bestv || (ov == bestv && oi < besti);
        else take = oc > bestc || (oc == bestc && oi < besti);
        if (take) { besti = oi; bestv = ov; bestc = oc; }
    }
    if (besti == 0x7FFFFFFF) { besti = 0; bestv = a.best_energy[0]; bestc = a.top_count[0]; }   // every energy NaN
    const int chain = besti;
    const int S = a.S[chain / a.chains_per_frame];
    const uint16_t* lab = (which == 1 ? a.best_label : a.top_label) + (size_t)chain * a.s_max;
    uint16_t* out = reinterpret_cast<uint16_t*>(rec + 2);
    const int NW = (a.s_max + 63) / 64 * 2;
    uint32_t* rbits = sh;
    uint32_t* rpre = sh + NW;
    if (canonical) {
        uint32_t running = 0;
        for (int b = 0; b < S; b += kWave) {
            const int i = b + lane;
            const bool root = i < S && lab[i] == (uint16_t)i;
            const uint64_t m = __ballot(root);
            if (lane == 0) {
                rbits[b >> 5] = (uint32_t)m; rpre[b >> 5] = running;
                rbits[(b >> 5) + 1] = (uint32_t)(m >> 32); rpre[(b >> 5) + 1] = running + (uint32_t)__popc((uint32_t)m);
            }
            running += (uint32_t)__popcll((unsigned long long)m);
        }
        __syncthreads();
    }
    const int padded = (a.s_max + 3) & ~3;
    for (int i = lane; i < padded; i += kWave) {
        uint16_t o = kNoLabel;
        if (i < S) {
            const uint32_t Lb = lab[i];
            if (Lb != kNoLabel)
                o = canonical ? (uint16_t)(rpre[Lb >> 5] + (uint32_t)__popc(rbits[Lb >> 5] & ((1u << (Lb & 31u)) - 1u))) : (uint16_t)Lb;
        }
        out[i] = o;
    }
    if (lane == 0) {
        rec[0] = which == 1 ? (unsigned long long)d2u(bestv) : bestc;
        rec[1] = (unsigned long long)a.chain_offset + (unsigned long long)chain;
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_chain_init(const ChainArgs& a, hipStream_t st)
{
    hipLaunchKernelGGL(chain_init_kernel, dim3(a.n_chains), dim3(64), 0, st, a);
    return hipGetLastError();
}

template <int WPB, bool TAB, bool REC = TAB, bool PROF = false, bool TOPO = false, int NSPEC = 1>
static hipError_t launch_variant(const ChainArgs& a, int s_lds, int e2_lds, size_t bytes, hipStream_t st)
{
    if constexpr (!PROF && WPB == 4 && TAB && REC) {
        if (a.prof) return launch_variant<4, true, true, true>(a, s_lds, e2_lds, bytes, st);
    }
    static DeviceOnce once;
    bool* attr_done = once.slot();
    if (!attr_done || !*attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_run_kernel<WPB, TAB, REC, PROF, TOPO, NSPEC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit - 64);
        if (e != hipSuccess) return e;
        if (attr_done) *attr_done = true;
    }
    const int blocks = (a.n_chains + WPB - 1) / WPB;
    hipLaunchKernelGGL((chain_run_kernel<WPB, TAB, REC, PROF, TOPO, NSPEC>), dim3(blocks), dim3(WPB * 64 * NSPEC), bytes, st, a, s_lds, e2_lds);
    return hipGetLastError();
}

// slots per chain of the HBM FIFO the launch will use, 0 if the tables (and the FIFO) fit LDS
size_t chain_run_global_slots(const ChainArgs& a, int s_used_max, int e2_used_max)
{
    const size_t cap = kLdsLimit - 64;
    const int wpbs[3] = { 4, 2, 1 };
    for (int wi = 0; wi < 3; ++wi) {
        if (a.chains_per_frame % wpbs[wi]) continue;
        if (make_layout(s_used_max, e2_used_max, wpbs[wi], true, true).total <= cap) return 0;
    }
    return 4 * queue_slots(s_used_max, e2_used_max);      // a FIFO per wave, up to four waves per chain
}

hipError_t launch_chain_run(const ChainArgs& a, int s_used_max, int e2_used_max, hipStream_t st, int* variant_out)
{
    const size_t cap = kLdsLimit - 64;
    const int wpbs[3] = { 4, 2, 1 };
    // More chains than four per CU can hold at once: eight per workgroup, with the per-superpixel records read
    // through L2 (only the component's rows, once per step) so that the CSR tables and eight private states fit LDS.
    if (!a.prof && a.dense_variant >= 0 && a.chains_per_frame % 8 == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) {
            const LdsLayout L8 = make_layout(s_used_max, e2_used_max, 8, true, false);
            if (L8.total <= cap && e2_used_max <= 0xFFFF && (a.dense_variant > 0 || a.n_chains > 4 * cus)) {
                // two waves per chain when their state fits as well (sixteen waves per CU instead of eight): variant 128
                const LdsLayout L82 = make_layout(s_used_max, e2_used_max, 8, true, false, false, 2);
                if (L82.total <= cap && a.spec_variant >= 0 && a.burn_in + a.iters < 0x7FFFFFFFull) {
                    if (variant_out) *variant_out = 128;
                    return launch_variant<8, true, false, false, false, 2>(a, s_used_max, e2_used_max, L82.total, st);
                }
                if (variant_out) *variant_out = 28;
                return launch_variant<8, true, false>(a, s_used_max, e2_used_max, L8.total, st);
            }
        }
    }
    // No more chains than four per CU, i.e. at most one wave per SIMD in the plain form: several waves per chain, the
    // further ones evaluating the next steps speculatively (see the kernel).  Records through L2, as in the dense form.
    // spec_variant: 0 = as many waves as fit, up to kSpecAuto; 1 = the same whatever the chain count; 2..4 = exactly that many
#ifdef MC_SPEC_STATS
    if (a.spec_variant >= 0 && a.chains_per_frame % 4 == 0
#else
    if (!a.prof && a.spec_variant >= 0 && a.chains_per_frame % 4 == 0
#endif
    && a.burn_in + a.iters < 0x7FFFFFFFull && e2_used_max <= 0xFFFF) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            (a.spec_variant > 0 || a.n_chains <= 4 * cus)) {
            const int hi = a.spec_variant >= 2 ? min(a.spec_variant, 4) : kSpecAuto, lo = a.spec_variant >= 2 ? hi : 2;
            for (int ns = hi; ns >= lo; --ns) {
                const LdsLayout Ls = make_layout(s_used_max, e2_used_max, 4, true, false, false, ns);
                if (Ls.total > cap) continue;
                if (variant_out) *variant_out = 44 + 100 * (ns - 2);
                if (ns == 4) return launch_variant<4, true, false, false, false, 4>(a, s_used_max, e2_used_max, Ls.total, st);
                if (ns == 3) return launch_variant<4, true, false, false, false, 3>(a, s_used_max, e2_used_max, Ls.total, st);
                return launch_variant<4, true, false, false, false, 2>(a, s_used_max, e2_used_max, Ls.total, st);
            }
        }
    }
    for (int tab = 1; tab >= 0; --tab) {
        if (!tab && a.dense_variant >= 0 && s_used_max <= 0xFFFF) {
            // tables too large for LDS: keep at least the CSR topology there (two of the four dependent round trips
            // of a growth pass), q, records and the FIFO behind L2
            // ... and, LDS leaving room for one workgroup per CU only, several waves per chain (the speculative form):
            // as many chains per workgroup as a CU gets (1, 2 or 4), then as many waves per chain as fit, up to four
            // per SIMD.  Variant 30 + chains per workgroup + 100 * (waves per chain - 2).
            if (!a.prof && a.spec_variant >= 0 && a.burn_in + a.iters < 0x7FFFFFFFull) {
                int dev = 0, cus = 0;
                if (!a.slots_g) return hipErrorInvalidValue;
                if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) {
                    const int per_cu = (a.n_chains + cus - 1) / cus;
                    int wpb = per_cu <= 1 ? 1 : (per_cu <= 2 ? 2 : 4);
                    while (wpb > 1 && a.chains_per_frame % wpb) wpb >>= 1;
                    const int want = a.spec_variant >= 2 ? min(a.spec_variant, 4) : 4;
                    for (int ns = min(want, 16 / wpb); ns >= 2; --ns) {
                        const LdsLayout L = make_layout(s_used_max, e2_used_max, wpb, false, false, true, ns);
                        if (L.total > cap) continue;
                        if (variant_out) *variant_out = 30 + wpb + 100 * (ns - 2);
                        if (wpb == 4) {
                            if (ns == 4) return launch_variant<4, false, false, false, true, 4>(a, s_used_max, e2_used_max, L.total, st);
                            if (ns == 3) return launch_variant<4, false, false, false, true, 3>(a, s_used_max, e2_used_max, L.total, st);
                            return launch_variant<4, false, false, false, true, 2>(a, s_used_max, e2_used_max, L.total, st);
                        }
                        if (wpb == 2) {
                            if (ns == 4) return launch_variant<2, false, false, false, true, 4>(a, s_used_max, e2_used_max, L.total, st);
                            if (ns == 3) return launch_variant<2, false, false, false, true, 3>(a, s_used_max, e2_used_max, L.total, st);
                            return launch_variant<2, false, false, false, true, 2>(a, s_used_max, e2_used_max, L.total, st);
                        }
                        if (ns == 4) return launch_variant<1, false, false, false, true, 4>(a, s_used_max, e2_used_max, L.total, st);
                        if (ns == 3) return launch_variant<1, false, false, false, true, 3>(a, s_used_max, e2_used_max, L.total, st);
                        return launch_variant<1, false, false, false, true, 2>(a, s_used_max, e2_used_max, L.total, st);
                    }
                }
            }
            for (int wi = 0; wi < 3; ++wi) {
                const int wpb = wpbs[wi];
                if (a.chains_per_frame % wpb) continue;
                const LdsLayout L = make_layout(s_used_max, e2_used_max, wpb, false, false, true);
                if (L.total > cap) continue;
                if (!a.slots_g) return hipErrorInvalidValue;
                if (variant_out) *variant_out = 30 + wpb;
                if (wpb == 4) return launch_variant<4, false, false, false, true>(a, s_used_max, e2_used_max, L.total, st);
                if (wpb == 2) return launch_variant<2, false, false, false, true>(a, s_used_max, e2_used_max, L.total, st);
                return launch_variant<1, false, false, false, true>(a, s_used_max, e2_used_max, L.total, st);
            }
        }
        for (int wi = 0; wi < 3; ++wi) {
            const int wpb = wpbs[wi];
            if (a.chains_per_frame % wpb) continue;
            const LdsLayout L = make_layout(s_used_max, e2_used_max, wpb, tab != 0, tab != 0);
            if (L.total > cap || (tab && e2_used_max > 0xFFFF)) continue;
            if (!tab && !a.slots_g) return hipErrorInvalidValue;
            if (variant_out) *variant_out = tab * 10 + wpb;
            if (tab) {
                if (wpb == 4) return launch_variant<4, true>(a, s_used_max, e2_used_max, L.total, st);
                if (wpb == 2) return launch_variant<2, true>(a, s_used_max, e2_used_max, L.total, st);
                return launch_variant<1, true>(a, s_used_max, e2_used_max, L.total, st);
            } else {
                if (wpb == 4) return launch_variant<4, false>(a, s_used_max, e2_used_max, L.total, st);
                if (wpb == 2) return launch_variant<2, false>(a, s_used_max, e2_used_max, L.total, st);
                return launch_variant<1, false>(a, s_used_max, e2_used_max, L.total, st);
            }
        }
    }
    return hipErrorInvalidValue;
}

hipError_t launch_best_record(const ChainArgs& a, int which, int canonical, unsigned long long* rec, hipStream_t st)
{
    const int NW = (a.s_max + 63) / 64 * 2;
    hipLaunchKernelGGL(best_record_kernel, dim3(1), dim3(64), (size_t)NW * 8, st, a, which, canonical, rec);
    return hipGetLastError();
}

hipError_t launch_visit_rehash(const unsigned long long* oh, const unsigned long long* oc, int ocap, unsigned long long* nh,
                               unsigned long long* nc, int ncap, int n_chains, hipStream_t st)
{
    hipLaunchKernelGGL(visit_rehash_kernel, dim3(n_chains), dim3(64), 0, st, oh, oc, ocap, nh, nc, ncap);
    return hipGetLastError();
}

hipError_t launch_visit_occupancy(const unsigned long long* vc, int hcap, int n_chains, unsigned int* out_max, hipStream_t st)
{
    hipLaunchKernelGGL(visit_occupancy_kernel, dim3(n_chains), dim3(64), 0, st, vc, hcap, out_max);
    return hipGetLastError();
}

hipError_t launch_chain_energy(const ChainArgs& a, double* out, hipStream_t st)
{
    hipLaunchKernelGGL(chain_energy_kernel, dim3(a.n_chains), dim3(64), 0, st, a, out);
    return hipGetLastError();
}

hipError_t launch_canonical_labels(const uint16_t* src, const uint8_t* /*valid*/, const int* S, int s_max,
                                   int n_chains, int chains_per_frame, uint16_t* dst, hipStream_t st)
{
    const int NW = (s_max + 63) / 64 * 2;
    hipLaunchKernelGGL(canonical_labels_kernel, dim3(n_chains), dim3(64), (size_t)NW * 8, st, src, S, s_max,
                       chains_per_frame, dst);
    return hipGetLastError();
}

}  // namespace mc
