// capi.hip -- the C ABI of include/mcmc_amd.h: context, HBM residency, launches.
//
// HBM layout (one allocation per array, frame-major / chain-major):
//   disp  int16 [F][H][W]      label uint16[F][H][W]
//   mom   int64 [F][s_max][10] fit   f64   [F][s_max][16]   valid u8[F][s_max]
//   rowp  u32   [F][s_max+1]   col   u32   [F][e2_max]      q     f64[F][e2_max]
//   chain state [C][s_max] (labels u16, segment cache f64/u16) + per-chain scalars
// Everything stays resident across calls; only getters copy back to the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mcmc_amd.h"
#include "kernels.hpp"

using namespace mc;

// ---- static scratch of the allocation-free entry point (mcmc_segment_probability_noalloc) -----------------
// Device memory that exists from load time on every GPU the library runs on, so the entry point never calls
// hipMalloc.  Layout: [0, 256) result doubles | [256, 256 + 65536) membership bytes | then 65536 member ids.
// In host code `scratch` is only a handle for hipGetSymbolAddress: the device address is taken once per device
// in mcmc_ctx_create (with that device current) and kept in g_scratch[device]; host code uses that pointer only.
constexpr int kScratchMaxS = 65536;            // superpixels (the ABI's s_max is < 65535 anyway)
constexpr int kScratchMaxMembers = 65536;      // ids in one member list
constexpr size_t kScratchOutOff = 0, kScratchMemOff = 256, kScratchListOff = 256 + (size_t)kScratchMaxS;
constexpr size_t kScratchBytes = kScratchListOff + (size_t)kScratchMaxMembers * 4;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
constexpr int kMaxDevices = 64;
static unsigned char* g_scratch[kMaxDevices];  // per device: each GPU holds its own copy at its own address

namespace {

enum { K_MOM = 0, K_FIT = 1, K_EDGE = 2, K_CHAIN = 3, K_COUNT = 4 };

struct Chains {
    int n = 0, cpf = 0;
    uint32_t seed = 0, offset = 0;
    unsigned long long step = 0;
    uint16_t *label = nullptr, *best_label = nullptr, *top_label = nullptr, *canon = nullptr;
    double *segC = nullptr, *energy = nullptr, *best_energy = nullptr, *tmp_energy = nullptr;
    unsigned long long *state_hash = nullptr, *trace_hash = nullptr, *n_accept = nullptr, *n_case = nullptr;
    unsigned long long *visit_hash = nullptr, *visit_count = nullptr, *overflow = nullptr, *top_count = nullptr,
                       *top_hash = nullptr;
    int* status = nullptr;
    uint8_t* trace = nullptr;
    unsigned long long trace_cap = 0, last_steps = 0;
    unsigned long long* slots_g = nullptr; size_t slots_cap = 0;
    double *last_targetL = nullptr, *last_prop = nullptr;
    unsigned long long* best_rec = nullptr; // packed record of mcmc_chains_get_best: key, global chain id, labels
    int hcap = 0;                          // slots per chain of the visit table (grows when visit_capacity == 0)
    unsigned long long measured = 0;       // Statistics::update calls made so far per chain (steps after burn-in)
    unsigned long long occ_known = 0, since_known = 0;   // largest table occupancy when last read; measured steps since
    unsigned int* d_occ = nullptr;
    unsigned int* d_fault = nullptr;       // set by the chain kernel when a chain gives up a wait (status -2)
    bool fault_unread = false;             // a run was launched since d_fault was last read
};

constexpr int kVisitCapAuto0 = 512;        // first size of an automatically sized visit table
constexpr int kVisitCapMax = 1 << 22;      // 4 Mi slots (64 MB) per chain: beyond this visits count as overflow

}  // namespace

struct mcmc_ctx {
    int device = 0;
    mcmc_params p{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // frames
    int n_frames = 0, s_max = 0, e2_max = 0;
    size_t px = 0;
    size_t fstride = 0;      // pixels between two frames of a plane (>= px; padding spreads the frames over HBM channels)
    int16_t* d_disp = nullptr; uint16_t* d_label = nullptr;   // d_label may sit `label_skew` bytes into its allocation
    uint16_t* d_label_alloc = nullptr;
    int* d_S = nullptr; uint32_t* d_E2 = nullptr; int* d_fstatus = nullptr;
    unsigned long long* d_mom = nullptr; double* d_fit = nullptr; uint8_t* d_valid = nullptr;
    double* d_lerr = nullptr; uint32_t* d_rowp = nullptr; uint32_t* d_col = nullptr; double* d_q = nullptr;
    unsigned long long* d_marg = nullptr;
    uint8_t* d_mem = nullptr; double* d_segout = nullptr; double* d_self = nullptr; size_t self_cap = 0;
    std::vector<int> h_S; std::vector<uint32_t> h_E2;
    std::vector<uint8_t> st_pixels, st_graph, st_mom, st_q;
    Chains ch;
    // timing
    bool timing = false;
    hipEvent_t ev0[K_COUNT]{}, ev1[K_COUNT]{};
    bool ev_valid[K_COUNT]{};
    int chain_variant = 0;
    // tuning overrides (0 / -1 = automatic)
    int opt_stride_pad = -1, opt_label_skew = -1;   // tuning: frame stride padding (pixels) / label plane offset (bytes)
    int opt_k1_px = 0;                               // left/right kernel: 0 automatic, 8 or 16 pixels per lane
    int opt_k1_bands = 0, opt_k1_threads = 0, opt_k1_lds = -1, opt_k1_debug = 0, opt_k3_prof = 0, opt_k3_dense = 0, opt_k3_spec = 0;
    unsigned long long* d_prof = nullptr;
};

namespace {

int fail(mcmc_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(c, expr)                                                                         \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail((c), MCMC_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

template <typename T>
hipError_t dalloc(T** p, size_t n)
{
    *p = nullptr;
    return hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T));
}

template <typename T>
void dfree(T*& p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

void free_chains(Chains& h)
{
    dfree(h.label); dfree(h.best_label); dfree(h.top_label); dfree(h.canon);
    dfree(h.segC); dfree(h.energy); dfree(h.best_energy); dfree(h.tmp_energy);
    dfree(h.state_hash); dfree(h.trace_hash); dfree(h.n_accept); dfree(h.n_case);
    dfree(h.visit_hash); dfree(h.visit_count); dfree(h.overflow); dfree(h.top_count); dfree(h.top_hash);
    dfree(h.status); dfree(h.trace); dfree(h.slots_g); dfree(h.last_targetL); dfree(h.last_prop); dfree(h.best_rec); dfree(h.d_occ); dfree(h.d_fault);
    h = Chains();
}

void free_frames(mcmc_ctx* c)
{
    dfree(c->d_disp); dfree(c->d_label_alloc); c->d_label = nullptr; dfree(c->d_S); dfree(c->d_E2); dfree(c->d_fstatus);
    dfree(c->d_mom); dfree(c->d_fit); dfree(c->d_valid); dfree(c->d_lerr); dfree(c->d_rowp);
    dfree(c->d_col); dfree(c->d_q); dfree(c->d_marg); dfree(c->d_mem); dfree(c->d_segout);
    c->n_frames = 0;
}

struct Timer {
    mcmc_ctx* c; int k;
    Timer(mcmc_ctx* c_, int k_) : c(c_), k(k_) { if (c->timing) (void)hipEventRecord(c->ev0[k], c->stream); }
    ~Timer() { if (c->timing) { (void)hipEventRecord(c->ev1[k], c->stream); c->ev_valid[k] = true; } }
};

bool frame_ok(const mcmc_ctx* c, int f) { return c && f >= 0 && f < c->n_frames; }

ChainArgs chain_args(mcmc_ctx* c)
{
    ChainArgs a{};
    a.fit = c->d_fit; a.valid = c->d_valid; a.rowp = c->d_rowp; a.col = c->d_col; a.q = c->d_q;
    a.S = c->d_S; a.E2 = c->d_E2; a.s_max = c->s_max; a.e2_max = c->e2_max;
    a.n_chains = c->ch.n; a.chains_per_frame = c->ch.cpf; a.seed = c->ch.seed; a.chain_offset = c->ch.offset;
    a.var = c->p.disparity_variance; a.new_seg_prob = c->p.new_segment_prob;
    a.label = c->ch.label; a.segC = c->ch.segC;
    a.energy = c->ch.energy; a.best_energy = c->ch.best_energy;
    a.best_label = c->ch.best_label; a.top_label = c->ch.top_label;
    a.state_hash = c->ch.state_hash; a.trace_hash = c->ch.trace_hash;
    a.n_accept = c->ch.n_accept; a.n_case = c->ch.n_case; a.status = c->ch.status; a.fault = c->ch.d_fault;
    a.visit_hash = c->ch.visit_hash; a.visit_count = c->ch.visit_count;
    a.overflow = c->ch.overflow; a.top_count = c->ch.top_count; a.top_hash = c->ch.top_hash;
    a.last_targetL = c->ch.last_targetL; a.last_prop = c->ch.last_prop;
    a.marg = c->d_marg; a.trace = nullptr; a.trace_stride = 0;
    a.hcap = c->ch.hcap; a.lmax = c->p.marginal_bins;
    return a;
}

// Statistics keeps every visited state (Statistics.h:30-47).  With visit_capacity == 0 the per-chain table grows
// with the run so that it never fills: a chain meets at most one new state per measured step, so a table with
// `occupied + steps about to be made` slots (and a quarter of slack for the probing) cannot overflow.  `occupied` is
// measured, not assumed: when the worst case since the last look would no longer fit, the real occupancy is read
// from the device (a chain visits ~1.5 new states per 100 steps at BASELINE config 2), and the table only grows --
// doubling, re-inserting the old entries -- when that does not fit either.  If the larger table cannot be had
// (memory, or the 4 Mi-slot limit), the run goes on with the old one and visits of further states are counted in
// `overflow`, as with a caller-fixed capacity.
int ensure_visit_capacity(mcmc_ctx* c, unsigned long long steps_next)
{
    Chains& h = c->ch;
    if (c->p.visit_capacity > 0) return MCMC_OK;          // fixed by the caller: overflow is counted, not avoided
    auto fits = [&](unsigned long long occupied) {
        const unsigned long long need = occupied + steps_next;
        return need + need / 4 <= (unsigned long long)h.hcap;
    };
    if (fits(h.occ_known + h.since_known)) return MCMC_OK;
    // look before growing
    if (!h.d_occ) HIPCHK(c, dalloc(&h.d_occ, 1));
    unsigned int occ = 0;
    HIPCHK(c, hipMemsetAsync(h.d_occ, 0, sizeof(unsigned int), c->stream));
    HIPCHK(c, launch_visit_occupancy(h.visit_count, h.hcap, h.n, h.d_occ, c->stream));
    HIPCHK(c, hipMemcpyAsync(&occ, h.d_occ, sizeof occ, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    h.occ_known = occ; h.since_known = 0;
    if (fits(occ)) return MCMC_OK;
    unsigned long long want = (unsigned long long)occ + steps_next;
    want += want / 4;
    if (want > (unsigned long long)kVisitCapMax) want = (unsigned long long)kVisitCapMax;
    if ((unsigned long long)h.hcap >= want) return MCMC_OK;   // at the limit: overflow is counted from here on
    int ncap = h.hcap;
    while ((unsigned long long)ncap < want) ncap *= 2;
    const size_t NH = (size_t)h.n * (size_t)ncap;
    unsigned long long *nh = nullptr, *nc = nullptr;
    if (dalloc(&nh, NH) != hipSuccess || dalloc(&nc, NH) != hipSuccess) {
        dfree(nh); dfree(nc);
        (void)hipGetLastError();                          // the failed allocation is not this call's error
        c->err = "visit table: no device memory for " + std::to_string(ncap) + " slots per chain; states beyond the present "
                 "table are counted in `overflow`";
        return MCMC_OK;
    }
    hipError_t e = hipMemsetAsync(nc, 0, NH * 8, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(nh, 0, NH * 8, c->stream);
    if (e == hipSuccess) e = launch_visit_rehash(h.visit_hash, h.visit_count, h.hcap, nh, nc, ncap, h.n, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { dfree(nh); dfree(nc); return fail(c, MCMC_E_DEVICE, std::string("visit table growth: ") + hipGetErrorString(e)); }
    dfree(h.visit_hash); dfree(h.visit_count);
    h.visit_hash = nh; h.visit_count = nc; h.hcap = ncap;
    return MCMC_OK;
}

}  // namespace

extern "C" {

const char* mcmc_version(void) { return "mcmc_amd 0.1 (gfx950)"; }

void mcmc_default_params(mcmc_params* p)
{
    if (!p) return;
    p->width = 640; p->height = 480;
    p->disparity_variance = 7.0;
    p->new_segment_prob = 0.8;
    p->marginal_bins = 64;
    p->visit_capacity = 0;   /* sized from the run, never fills */
}

int mcmc_ctx_create(int device, const mcmc_params* params, mcmc_ctx** out)
{
    if (!out) return MCMC_E_BADARG;
    *out = nullptr;
    mcmc_params p;
    if (params) p = *params; else mcmc_default_params(&p);
    // the variance may be +inf or NaN: both only reach the reference's own error paths (Graph.cpp:134-137), which
    // the chains then report as the reference does; zero or negative is refused
    if (p.width <= 0 || p.height <= 0 || p.width > 4096 || p.height > 4096 || p.disparity_variance <= 0.0 ||
        !(p.new_segment_prob >= 0.0 && p.new_segment_prob <= 1.0) || p.marginal_bins < 1 || p.visit_capacity < 0)
        return MCMC_E_BADARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return MCMC_E_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MCMC_E_DEVICE;
    if (device < kMaxDevices && !g_scratch[device]) {
        // the device current now is the one whose copy of the array this address belongs to
        unsigned char* sp = nullptr;
        if (hipGetSymbolAddress(reinterpret_cast<void**>(&sp), HIP_SYMBOL(scratch)) != hipSuccess || !sp) return MCMC_E_DEVICE;
        g_scratch[device] = sp;
    }
    mcmc_ctx* c = new mcmc_ctx();
    c->device = device; c->p = p;
    c->px = (size_t)p.width * (size_t)p.height;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return MCMC_E_DEVICE; }
    c->own_stream = true;
    for (int k = 0; k < K_COUNT; ++k) {
        if (hipEventCreate(&c->ev0[k]) != hipSuccess || hipEventCreate(&c->ev1[k]) != hipSuccess) {
            delete c; return MCMC_E_DEVICE;
        }
    }
    *out = c;
    return MCMC_OK;
}

int mcmc_ctx_destroy(mcmc_ctx* c)
{
    if (!c) return MCMC_E_BADARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_chains(c->ch);
    free_frames(c);
    dfree(c->d_self); dfree(c->d_prof);
    for (int k = 0; k < K_COUNT; ++k) { (void)hipEventDestroy(c->ev0[k]); (void)hipEventDestroy(c->ev1[k]); }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return MCMC_OK;
}

const char* mcmc_last_error(const mcmc_ctx* c) { return c ? c->err.c_str() : "null context"; }

int mcmc_ctx_set_stream(mcmc_ctx* c, void* s)
{
    if (!c) return MCMC_E_BADARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    c->stream = reinterpret_cast<hipStream_t>(s);
    return MCMC_OK;
}

int mcmc_ctx_synchronize(mcmc_ctx* c)
{
    if (!c) return MCMC_E_BADARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // a chain that gave up waiting for its turn (speculative forms; never seen) has status -2 in get_scalars: say so here,
    // where the caller learns that the run is over
    if (c->ch.n && c->ch.fault_unread && c->ch.d_fault) {
        unsigned int f = 0;
        HIPCHK(c, hipMemcpy(&f, c->ch.d_fault, sizeof f, hipMemcpyDeviceToHost));
        c->ch.fault_unread = false;
        if (f) {
            HIPCHK(c, hipMemset(c->ch.d_fault, 0, sizeof f));
            return fail(c, MCMC_E_DEVICE, "chains_run: a chain's wait for its turn ran out; its status in mcmc_chains_get_scalars is -2 and it no longer moves");
        }
    }
    return MCMC_OK;
}

int mcmc_ctx_set_option(mcmc_ctx* c, const char* name, int value)
{
    if (!c || !name) return MCMC_E_BADARG;
    const std::string n(name);
    if (n == "k1_bands") c->opt_k1_bands = value;
    else if (n == "k1_threads") c->opt_k1_threads = value;
    else if (n == "k1_use_lds") c->opt_k1_lds = value;
    else if (n == "k1_debug") c->opt_k1_debug = value;
    else if (n == "k3_profile") c->opt_k3_prof = value;
    else if (n == "k3_dense") c->opt_k3_dense = value;
    else if (n == "k3_spec") c->opt_k3_spec = value;
    else if (n == "k1_px") c->opt_k1_px = value;
    else if (n == "stride_pad") c->opt_stride_pad = value;      // takes effect at the next mcmc_frames_alloc
    else if (n == "label_skew") c->opt_label_skew = value;
    else return fail(c, MCMC_E_BADARG, "set_option: unknown option " + n);
    return MCMC_OK;
}

int mcmc_set_timing(mcmc_ctx* c, int enable)
{
    if (!c) return MCMC_E_BADARG;
    c->timing = enable != 0;
    return MCMC_OK;
}

double mcmc_last_kernel_ms(mcmc_ctx* c, int k)
{
    if (!c || k < 0 || k >= K_COUNT || !c->ev_valid[k]) return -1.0;
    if (hipSetDevice(c->device) != hipSuccess) return -1.0;
    if (hipEventSynchronize(c->ev1[k]) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev0[k], c->ev1[k]) != hipSuccess) return -1.0;
    return (double)ms;
}

// ---------------------------------------------------------------- frames
int mcmc_frames_alloc(mcmc_ctx* c, int n_frames, int s_max, int e2_max)
{
    if (!c || n_frames <= 0 || s_max <= 0 || s_max >= 0xFFFF || e2_max < 0) return fail(c, MCMC_E_BADARG, "frames_alloc: bad sizes");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_chains(c->ch);
    free_frames(c);
    const size_t F = (size_t)n_frames;
    {
        const size_t pad = c->opt_stride_pad >= 0 ? (size_t)c->opt_stride_pad : 0;
        const size_t skew = c->opt_label_skew >= 0 ? (size_t)c->opt_label_skew : 0;
        c->fstride = ((c->px + pad + 7) / 8) * 8;             // frames stay 16-byte aligned
        HIPCHK(c, dalloc(&c->d_disp, F * c->fstride));
        HIPCHK(c, dalloc(&c->d_label_alloc, F * c->fstride + (skew + 1) / 2 + 8));
        c->d_label = c->d_label_alloc + ((skew / 16) * 16) / 2;
    }
    HIPCHK(c, dalloc(&c->d_S, F));
    HIPCHK(c, dalloc(&c->d_E2, F));
    HIPCHK(c, dalloc(&c->d_fstatus, F));
    HIPCHK(c, dalloc(&c->d_mom, F * s_max * MCMC_MOMENT_STRIDE));
    HIPCHK(c, dalloc(&c->d_fit, F * s_max * MCMC_FIT_STRIDE));
    HIPCHK(c, dalloc(&c->d_valid, F * s_max));
    HIPCHK(c, dalloc(&c->d_lerr, F * s_max));
    HIPCHK(c, dalloc(&c->d_rowp, F * ((size_t)s_max + 1)));
    HIPCHK(c, dalloc(&c->d_col, F * (size_t)e2_max));
    HIPCHK(c, dalloc(&c->d_q, F * (size_t)e2_max));
    HIPCHK(c, dalloc(&c->d_marg, F * s_max * (size_t)c->p.marginal_bins));
    HIPCHK(c, dalloc(&c->d_mem, (size_t)s_max));
    HIPCHK(c, dalloc(&c->d_segout, 8));
    HIPCHK(c, hipMemsetAsync(c->d_S, 0, F * sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_E2, 0, F * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_fstatus, 0, F * sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_rowp, 0, F * ((size_t)s_max + 1) * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_marg, 0, F * s_max * (size_t)c->p.marginal_bins * 8, c->stream));
    c->n_frames = n_frames; c->s_max = s_max; c->e2_max = e2_max;
    c->h_S.assign(F, 0); c->h_E2.assign(F, 0);
    c->st_pixels.assign(F, 0); c->st_graph.assign(F, 0); c->st_mom.assign(F, 0); c->st_q.assign(F, 0);
    return MCMC_OK;
}

static int upload_common(mcmc_ctx* c, int frame, const void* disp, const void* label, int S, hipMemcpyKind kind)
{
    if (!frame_ok(c, frame) || !disp || !label || S <= 0 || S > c->s_max) return fail(c, MCMC_E_BADARG, "frame_upload: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->d_disp + (size_t)frame * c->fstride, disp, c->px * 2, kind, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_label + (size_t)frame * c->fstride, label, c->px * 2, kind, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_S + frame, &S, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // &S is a stack temporary
    c->h_S[frame] = S;
    c->st_pixels[frame] = 1; c->st_mom[frame] = 0; c->st_q[frame] = 0;
    return MCMC_OK;
}

int mcmc_frame_upload(mcmc_ctx* c, int frame, const int16_t* disp, const uint16_t* label, int S)
{
    return upload_common(c, frame, disp, label, S, hipMemcpyHostToDevice);
}

int mcmc_frame_upload_device(mcmc_ctx* c, int frame, const void* d_disp, const void* d_label, int S)
{
    return upload_common(c, frame, d_disp, d_label, S, hipMemcpyDeviceToDevice);
}

int mcmc_frame_set_graph(mcmc_ctx* c, int frame, const uint32_t* row_ptr, const uint32_t* col, uint32_t e2)
{
    if (!frame_ok(c, frame) || !row_ptr || (!col && e2) || e2 > (uint32_t)c->e2_max) return fail(c, MCMC_E_BADARG, "frame_set_graph: bad argument");
    const int S = c->h_S[frame];
    if (S <= 0) return fail(c, MCMC_E_STATE, "frame_set_graph: upload the frame first");
    if (row_ptr[0] != 0 || row_ptr[S] != e2) return fail(c, MCMC_E_BADARG, "frame_set_graph: row_ptr does not span e2");
    for (int i = 0; i < S; ++i) {
        if (row_ptr[i + 1] < row_ptr[i] || row_ptr[i + 1] > e2) return fail(c, MCMC_E_BADARG, "frame_set_graph: row_ptr not monotone");
        // the growth FIFO addresses a superpixel's neighbours in parts of 8 with a 12-bit part index
        if (row_ptr[i + 1] - row_ptr[i] > (uint32_t)MCMC_MAX_DEGREE) return fail(c, MCMC_E_TOO_LARGE, "frame_set_graph: a superpixel has more than MCMC_MAX_DEGREE neighbours");
    }
    // neighbours strictly ascending within a row: SuperPixel::neighbors_ is a map keyed by the neighbour, so no
    // neighbour appears twice (MyMath.h:181); a duplicate entry would be tried and counted twice here
    for (int i = 0; i < S; ++i)
        for (uint32_t e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
            if (col[e] >= (uint32_t)S) return fail(c, MCMC_E_BADARG, "frame_set_graph: neighbour id out of range");
            if (e > row_ptr[i] && col[e] <= col[e - 1]) return fail(c, MCMC_E_BADARG, "frame_set_graph: neighbours of a superpixel must be strictly ascending (no duplicates)");
        }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->d_rowp + (size_t)frame * (c->s_max + 1), row_ptr, ((size_t)S + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (e2) HIPCHK(c, hipMemcpyAsync(c->d_col + (size_t)frame * c->e2_max, col, (size_t)e2 * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_E2 + frame, &e2, 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->h_E2[frame] = e2;
    c->st_graph[frame] = 1; c->st_q[frame] = 0;
    return MCMC_OK;
}

int mcmc_frames_moments(mcmc_ctx* c, int first, int count)
{
    if (!c || count <= 0 || !frame_ok(c, first) || !frame_ok(c, first + count - 1)) return fail(c, MCMC_E_BADARG, "frames_moments: bad range");
    int s_used = 0;
    for (int f = first; f < first + count; ++f) {
        if (!c->st_pixels[f]) return fail(c, MCMC_E_STATE, "frames_moments: frame not uploaded");
        s_used = std::max(s_used, c->h_S[f]);
    }
    HIPCHK(c, hipSetDevice(c->device));
    const int W = c->p.width, H = c->p.height;
    MomentArgs a{};
    a.disp = c->d_disp; a.label = c->d_label; a.S = c->d_S; a.mom = c->d_mom;
    a.frame_stride = c->fstride; a.W = W; a.H = H; a.s_max = c->s_max; a.first_frame = first;
    const bool generic = (W % 8) != 0 || W / 8 > 1024;
    int threads = 256, bands = 1, mode = 2;
    size_t lds = 0;
    if (generic) {
        bands = std::min(H, std::max(std::max(1, 256 / count), 64));
    } else {
        // left/right kernel: a workgroup is nsub sub-bands x G column strips of 8 or 16 pixels; a strip has at
        // most 128 rows x 8 pixels or 64 rows x 16 pixels (32-bit strip-local sums); about one workgroup per CU or more
        // 8 pixels per lane unless option "k1_px" = 16 asks for the 16-pixel form (measured slower at 640x480: half
        // the waves, and the per-row work it shares is not what bounds the kernel)
        const int npx = (c->opt_k1_px == 16 && W % 16 == 0) ? 16 : 8;
        const int G = W / npx, max_rows = npx == 16 ? 64 : 128;
        const int tmax = npx == 16 ? 512 : 1024;     // the 16-pixel form needs more registers per lane
        const int nsub_max = std::max(1, (c->opt_k1_threads > 0 ? std::min(tmax, c->opt_k1_threads) : tmax) / G);
        bands = std::max(1, (256 + count - 1) / count);
        bands = std::min(bands, std::max(1, H / 8));
        if (c->opt_k1_bands > 0) bands = std::min(c->opt_k1_bands, H);
        int nsub = nsub_max;
        for (;;) {
            const int rpb = (H + bands - 1) / bands;
            nsub = std::min(nsub_max, std::max(1, rpb / 4));
            if ((rpb + nsub - 1) / nsub <= max_rows) break;
            bands++;
        }
        threads = G * nsub;
        mode = 3;
        a.lane_px = npx;
        lds = moments_lr_lds_bytes(s_used, threads);
        if (lds > 160 * 1024 - 64 || c->opt_k1_lds == 0) { mode = 4; lds = moments_lr_lds_bytes(0, threads); }
    }
    a.bands = bands;
    a.table_rows = s_used;
    a.debug = c->opt_k1_debug;
    if (bands > 1 || mode != 3)
        HIPCHK(c, hipMemsetAsync(c->d_mom + (size_t)first * c->s_max * MCMC_MOMENT_STRIDE, 0,
                                 (size_t)count * c->s_max * MCMC_MOMENT_STRIDE * 8, c->stream));
    {
        Timer t(c, K_MOM);
        HIPCHK(c, launch_moments(a, count, threads, lds, mode, c->stream));
    }
    FitArgs fa{};
    fa.mom = c->d_mom; fa.S = c->d_S; fa.fit = c->d_fit; fa.valid = c->d_valid; fa.local_err = c->d_lerr;
    fa.var = c->p.disparity_variance; fa.W = W; fa.H = H; fa.s_max = c->s_max; fa.first_frame = first;
    {
        Timer t(c, K_FIT);
        HIPCHK(c, launch_local_fit(fa, count, c->stream));
    }
    for (int f = first; f < first + count; ++f) { c->st_mom[f] = 1; c->st_q[f] = 0; }
    return MCMC_OK;
}

int mcmc_frames_edge_probs(mcmc_ctx* c, int first, int count)
{
    if (!c || count <= 0 || !frame_ok(c, first) || !frame_ok(c, first + count - 1)) return fail(c, MCMC_E_BADARG, "frames_edge_probs: bad range");
    for (int f = first; f < first + count; ++f)
        if (!c->st_mom[f] || !c->st_graph[f]) return fail(c, MCMC_E_STATE, "frames_edge_probs: needs moments and graph");
    HIPCHK(c, hipSetDevice(c->device));
    EdgeArgs a{};
    a.fit = c->d_fit; a.valid = c->d_valid; a.rowp = c->d_rowp; a.col = c->d_col; a.S = c->d_S; a.E2 = c->d_E2;
    a.q = c->d_q; a.status = c->d_fstatus; a.var = c->p.disparity_variance;
    a.s_max = c->s_max; a.e2_max = c->e2_max; a.first_frame = first;
    // the kernel only ever sets bits: a frame re-uploaded with good data must not keep an old NaN / inf flag
    HIPCHK(c, hipMemsetAsync(c->d_fstatus + first, 0, (size_t)count * sizeof(int), c->stream));
    {
        Timer t(c, K_EDGE);
        HIPCHK(c, launch_edge_prob(a, count, c->stream));
    }
    for (int f = first; f < first + count; ++f) c->st_q[f] = 1;
    return MCMC_OK;
}

int mcmc_frame_get_moments(mcmc_ctx* c, int frame, int64_t* out)
{
    if (!frame_ok(c, frame) || !out) return fail(c, MCMC_E_BADARG, "get_moments: bad argument");
    if (!c->st_mom[frame]) return fail(c, MCMC_E_STATE, "get_moments: moments not computed");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->d_mom + (size_t)frame * c->s_max * MCMC_MOMENT_STRIDE,
                             (size_t)c->h_S[frame] * MCMC_MOMENT_STRIDE * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_frame_get_local_fit(mcmc_ctx* c, int frame, double* fit, uint8_t* valid, double* local_err)
{
    if (!frame_ok(c, frame)) return fail(c, MCMC_E_BADARG, "get_local_fit: bad frame");
    if (!c->st_mom[frame]) return fail(c, MCMC_E_STATE, "get_local_fit: moments not computed");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t S = (size_t)c->h_S[frame];
    if (fit) HIPCHK(c, hipMemcpyAsync(fit, c->d_fit + (size_t)frame * c->s_max * MCMC_FIT_STRIDE, S * MCMC_FIT_STRIDE * 8, hipMemcpyDeviceToHost, c->stream));
    if (valid) HIPCHK(c, hipMemcpyAsync(valid, c->d_valid + (size_t)frame * c->s_max, S, hipMemcpyDeviceToHost, c->stream));
    if (local_err) HIPCHK(c, hipMemcpyAsync(local_err, c->d_lerr + (size_t)frame * c->s_max, S * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_frame_get_edge_probs(mcmc_ctx* c, int frame, double* q)
{
    if (!frame_ok(c, frame) || !q) return fail(c, MCMC_E_BADARG, "get_edge_probs: bad argument");
    if (!c->st_q[frame]) return fail(c, MCMC_E_STATE, "get_edge_probs: edge probabilities not computed");
    HIPCHK(c, hipSetDevice(c->device));
    int fs = 0;
    if (c->h_E2[frame])
        HIPCHK(c, hipMemcpyAsync(q, c->d_q + (size_t)frame * c->e2_max, (size_t)c->h_E2[frame] * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&fs, c->d_fstatus + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (fs & 1) return fail(c, MCMC_E_NAN_RATIO, "The edge probability is nan!");
    if (fs & 2) return fail(c, MCMC_E_INF_ENERGY, "The subgraph log probability is inf");
    return MCMC_OK;
}

int mcmc_segment_probability(mcmc_ctx* c, int frame, const uint32_t* members, int n, double* probL, double* theta3, double* err)
{
    if (!frame_ok(c, frame) || !members || n <= 0) return fail(c, MCMC_E_BADARG, "segment_probability: bad argument");
    if (!c->st_mom[frame]) return fail(c, MCMC_E_STATE, "segment_probability: moments not computed");
    const int S = c->h_S[frame];
    std::vector<uint8_t> mem((size_t)S, 0);
    for (int k = 0; k < n; ++k) {
        if (members[k] >= (uint32_t)S) return fail(c, MCMC_E_BADARG, "segment_probability: member id out of range");
        mem[members[k]] = 1;
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->d_mem, mem.data(), (size_t)S, hipMemcpyHostToDevice, c->stream));
    SegArgs a{};
    a.fit = c->d_fit + (size_t)frame * c->s_max * MCMC_FIT_STRIDE; a.mem = c->d_mem; a.out = c->d_segout;
    a.var = c->p.disparity_variance; a.S = S;
    HIPCHK(c, launch_segment_prob(a, c->stream));
    double out[6];
    HIPCHK(c, hipMemcpyAsync(out, c->d_segout, sizeof out, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (probL) *probL = out[0];
    if (err) *err = out[1];
    if (theta3) { theta3[0] = out[2]; theta3[1] = out[3]; theta3[2] = out[4]; }
    if (out[5] != 0.0) return fail(c, MCMC_E_INF_ENERGY, "The subgraph log probability is inf");
    return MCMC_OK;
}

// Same result as mcmc_segment_probability without any device allocation: the membership bytes, the member list
// and the result are staged in the library's static scratch (one per GPU, shared by every context on it).
int mcmc_segment_probability_noalloc(mcmc_ctx* c, int frame, const uint32_t* members, int n, double* probL, double* theta3, double* err)
{
    if (!frame_ok(c, frame) || !members || n <= 0) return fail(c, MCMC_E_BADARG, "segment_probability_noalloc: bad argument");
    if (!c->st_mom[frame]) return fail(c, MCMC_E_STATE, "segment_probability_noalloc: moments not computed");
    const int S = c->h_S[frame];
    if (n > kScratchMaxMembers || S > kScratchMaxS)
        return fail(c, MCMC_E_TOO_LARGE, "segment_probability_noalloc: at most 65536 member ids and 65536 superpixels fit the static scratch");
    for (int k = 0; k < n; ++k)
        if (members[k] >= (uint32_t)S) return fail(c, MCMC_E_BADARG, "segment_probability_noalloc: member id out of range");
    if (c->device >= kMaxDevices || !g_scratch[c->device]) return fail(c, MCMC_E_DEVICE, "segment_probability_noalloc: no scratch address for this device");
    HIPCHK(c, hipSetDevice(c->device));
    unsigned char* sp = g_scratch[c->device];
    double* d_out = reinterpret_cast<double*>(sp + kScratchOutOff);
    uint8_t* d_mem = sp + kScratchMemOff;
    uint32_t* d_list = reinterpret_cast<uint32_t*>(sp + kScratchListOff);
    // the scratch keeps the last call's contents: the membership bytes must start at zero
    HIPCHK(c, hipMemsetAsync(d_mem, 0, (size_t)S, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_list, members, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_mark_members(d_list, n, d_mem, c->stream));
    SegArgs a{};
    a.fit = c->d_fit + (size_t)frame * c->s_max * MCMC_FIT_STRIDE; a.mem = d_mem; a.out = d_out;
    a.var = c->p.disparity_variance; a.S = S;
    HIPCHK(c, launch_segment_prob(a, c->stream));
    double out[6];
    HIPCHK(c, hipMemcpyAsync(out, d_out, sizeof out, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (probL) *probL = out[0];
    if (err) *err = out[1];
    if (theta3) { theta3[0] = out[2]; theta3[1] = out[3]; theta3[2] = out[4]; }
    if (out[5] != 0.0) return fail(c, MCMC_E_INF_ENERGY, "The subgraph log probability is inf");
    return MCMC_OK;
}

// ---------------------------------------------------------------- chains
int mcmc_chains_init(mcmc_ctx* c, int n_chains, int cpf, uint32_t seed, uint32_t chain_offset)
{
    if (!c || n_chains <= 0 || cpf <= 0) return fail(c, MCMC_E_BADARG, "chains_init: bad argument");
    if (c->n_frames == 0) return fail(c, MCMC_E_STATE, "chains_init: no frames");
    const int frames_needed = (n_chains + cpf - 1) / cpf;
    if (frames_needed > c->n_frames) return fail(c, MCMC_E_BADARG, "chains_init: more chains than frames * chains_per_frame");
    for (int f = 0; f < frames_needed; ++f)
        if (!c->st_q[f]) return fail(c, MCMC_E_STATE, "chains_init: edge probabilities not computed for every frame");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_chains(c->ch);
    Chains& h = c->ch;
    h.n = n_chains; h.cpf = cpf; h.seed = seed; h.offset = chain_offset; h.step = 0;
    h.hcap = c->p.visit_capacity > 0 ? c->p.visit_capacity : kVisitCapAuto0;
    h.measured = 0;
    const size_t N = (size_t)n_chains, NS = N * (size_t)c->s_max, NH = N * (size_t)h.hcap;
    HIPCHK(c, dalloc(&h.label, NS)); HIPCHK(c, dalloc(&h.best_label, NS));
    HIPCHK(c, dalloc(&h.top_label, NS)); HIPCHK(c, dalloc(&h.canon, NS)); HIPCHK(c, dalloc(&h.segC, NS * mc::kSegCacheStride));
    HIPCHK(c, dalloc(&h.energy, N)); HIPCHK(c, dalloc(&h.best_energy, N)); HIPCHK(c, dalloc(&h.tmp_energy, N));
    HIPCHK(c, dalloc(&h.state_hash, N)); HIPCHK(c, dalloc(&h.trace_hash, N)); HIPCHK(c, dalloc(&h.n_accept, N));
    HIPCHK(c, dalloc(&h.n_case, N * 3)); HIPCHK(c, dalloc(&h.visit_hash, NH)); HIPCHK(c, dalloc(&h.visit_count, NH));
    HIPCHK(c, dalloc(&h.overflow, N)); HIPCHK(c, dalloc(&h.top_count, N)); HIPCHK(c, dalloc(&h.top_hash, N));
    HIPCHK(c, dalloc(&h.status, N));
    HIPCHK(c, dalloc(&h.d_fault, 1)); HIPCHK(c, hipMemsetAsync(h.d_fault, 0, sizeof(unsigned int), c->stream));
    HIPCHK(c, dalloc(&h.last_targetL, N)); HIPCHK(c, dalloc(&h.last_prop, N));
    HIPCHK(c, dalloc(&h.best_rec, 2 + ((size_t)c->s_max + 3) / 4));
    HIPCHK(c, hipMemsetAsync(c->d_marg, 0, (size_t)c->n_frames * c->s_max * (size_t)c->p.marginal_bins * 8, c->stream));
    ChainArgs a = chain_args(c);
    HIPCHK(c, launch_chain_init(a, c->stream));
    std::vector<int> st(N);
    HIPCHK(c, hipMemcpyAsync(st.data(), h.status, N * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < N; ++k) {
        if (st[k] == MCMC_E_NOSUPERPIXELS) return fail(c, MCMC_E_NOSUPERPIXELS, "chains_init: a frame has no valid superpixel");
        if (st[k] == MCMC_E_INF_ENERGY) return fail(c, MCMC_E_INF_ENERGY, "The subgraph log probability is inf");
    }
    return MCMC_OK;
}

int mcmc_chains_set_trace(mcmc_ctx* c, uint64_t cap)
{
    if (!c || c->ch.n == 0) return fail(c, MCMC_E_STATE, "chains_set_trace: no chains");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dfree(c->ch.trace);
    c->ch.trace_cap = 0;
    if (cap) { HIPCHK(c, dalloc(&c->ch.trace, (size_t)c->ch.n * cap)); c->ch.trace_cap = cap; }
    return MCMC_OK;
}

int mcmc_chains_run(mcmc_ctx* c, uint64_t burn_in, uint64_t iters)
{
    if (!c || c->ch.n == 0) return fail(c, MCMC_E_STATE, "chains_run: chains_init first");
    HIPCHK(c, hipSetDevice(c->device));
    Chains& h = c->ch;
    { const int rc = ensure_visit_capacity(c, iters); if (rc != MCMC_OK) return rc; }
    ChainArgs a = chain_args(c);
    a.step0 = h.step; a.burn_in = burn_in; a.iters = iters;
    const unsigned long long total = burn_in + iters;
    if (h.trace && total <= h.trace_cap) { a.trace = h.trace; a.trace_stride = total; }
    if (c->opt_k3_prof) {
        if (!c->d_prof) { HIPCHK(c, dalloc(&c->d_prof, 16)); }
        HIPCHK(c, hipMemsetAsync(c->d_prof, 0, 16 * 8, c->stream));
        a.prof = c->d_prof;
    }
    a.dense_variant = c->opt_k3_dense;
    a.spec_variant = c->opt_k3_spec;
    int s_used = 0, e_used = 0;
    const int frames_needed = (h.n + h.cpf - 1) / h.cpf;
    for (int f = 0; f < frames_needed; ++f) { s_used = std::max(s_used, c->h_S[f]); e_used = std::max(e_used, (int)c->h_E2[f]); }
    {
        const size_t per_chain = chain_run_global_slots(a, s_used, e_used);
        if (per_chain * (size_t)h.n > h.slots_cap) {
            dfree(h.slots_g);
            HIPCHK(c, dalloc(&h.slots_g, per_chain * (size_t)h.n));
            h.slots_cap = per_chain * (size_t)h.n;
        }
        a.slots_g = h.slots_g;
    }
    {
        Timer t(c, K_CHAIN);
        hipError_t e = launch_chain_run(a, s_used, e_used, c->stream, &c->chain_variant);
        if (e == hipErrorInvalidValue) return fail(c, MCMC_E_TOO_LARGE, "chains_run: per-chain state does not fit 160 KB of LDS");
        HIPCHK(c, e);
    }
    h.step += total; h.last_steps = total; h.measured += iters; h.since_known += iters;
    h.fault_unread = true;
    return MCMC_OK;
}

int mcmc_chains_reserve_visits(mcmc_ctx* c, uint64_t measured_steps)
{
    if (!c || c->ch.n == 0) return fail(c, MCMC_E_STATE, "chains_reserve_visits: chains_init first");
    HIPCHK(c, hipSetDevice(c->device));
    return ensure_visit_capacity(c, measured_steps);
}

int mcmc_chains_visit_capacity(const mcmc_ctx* c)
{
    return c ? c->ch.hcap : MCMC_E_BADARG;
}

int mcmc_chains_get_last_ratios(mcmc_ctx* c, double* targetL, double* proposal)
{
    if (!c || c->ch.n == 0) return fail(c, MCMC_E_STATE, "chains_get_last_ratios: no chains");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = (size_t)c->ch.n;
    if (targetL) HIPCHK(c, hipMemcpyAsync(targetL, c->ch.last_targetL, N * 8, hipMemcpyDeviceToHost, c->stream));
    if (proposal) HIPCHK(c, hipMemcpyAsync(proposal, c->ch.last_prop, N * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_chains_last_variant(const mcmc_ctx* c)
{
    return c ? c->chain_variant : MCMC_E_BADARG;
}

int mcmc_chains_get_profile(mcmc_ctx* c, uint64_t* out12)
{
    if (!c || !out12 || !c->d_prof) return fail(c, MCMC_E_STATE, "chains_get_profile: set option k3_profile and run first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out12, c->d_prof, 16 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_chains_get_trace(mcmc_ctx* c, uint8_t* out)
{
    if (!c || !out || !c->ch.trace || c->ch.last_steps == 0 || c->ch.last_steps > c->ch.trace_cap) return fail(c, MCMC_E_STATE, "chains_get_trace: no trace recorded");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->ch.trace, (size_t)c->ch.n * c->ch.last_steps, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_chains_get_labels(mcmc_ctx* c, int which, int canonical, uint16_t* out)
{
    if (!c || !out || which < 0 || which > 2) return fail(c, MCMC_E_BADARG, "chains_get_labels: bad argument");
    if (c->ch.n == 0) return fail(c, MCMC_E_STATE, "chains_get_labels: no chains");
    HIPCHK(c, hipSetDevice(c->device));
    Chains& h = c->ch;
    const uint16_t* src = which == 0 ? h.label : (which == 1 ? h.best_label : h.top_label);
    if (canonical) {
        HIPCHK(c, launch_canonical_labels(src, c->d_valid, c->d_S, c->s_max, h.n, h.cpf, h.canon, c->stream));
        src = h.canon;
    }
    HIPCHK(c, hipMemcpyAsync(out, src, (size_t)h.n * c->s_max * 2, hipMemcpyDeviceToHost, c->stream));
    std::vector<unsigned long long> ovf;
    if (which == 2) {
        ovf.resize((size_t)h.n);
        HIPCHK(c, hipMemcpyAsync(ovf.data(), h.overflow, (size_t)h.n * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // a full visit table stops counting new states: "most visited" may then be wrong.  The labels are still
    // written; the status says so (only possible with a caller-fixed visit_capacity).
    for (unsigned long long o : ovf)
        if (o) return fail(c, MCMC_E_OVERFLOW, "chains_get_labels: a chain's visit table overflowed (visit_capacity too small); the most-visited state may be wrong");
    return MCMC_OK;
}

// One labeling for the whole job (printResults, can05_sampler.cpp:23-45): arg-max on the device, one record back.
int mcmc_chains_best_record_device(mcmc_ctx* c, int which, int canonical, const void** d_record, size_t* bytes)
{
    if (!c || (which != 1 && which != 2) || !d_record) return fail(c, MCMC_E_BADARG, "chains_best_record_device: bad argument");
    if (c->ch.n == 0) return fail(c, MCMC_E_STATE, "chains_best_record_device: no chains");
    HIPCHK(c, hipSetDevice(c->device));
    ChainArgs a = chain_args(c);
    HIPCHK(c, launch_best_record(a, which, canonical, c->ch.best_rec, c->stream));
    *d_record = c->ch.best_rec;
    if (bytes) *bytes = (2 + ((size_t)c->s_max + 3) / 4) * 8;
    return MCMC_OK;
}

int mcmc_chains_get_best(mcmc_ctx* c, int which, int canonical, double* energy, uint64_t* visits, uint64_t* chain, uint16_t* labels)
{
    const void* d = nullptr;
    size_t bytes = 0;
    const int rc = mcmc_chains_best_record_device(c, which, canonical, &d, &bytes);
    if (rc != MCMC_OK) return rc;
    std::vector<unsigned long long> rec(bytes / 8);
    HIPCHK(c, hipMemcpyAsync(rec.data(), d, bytes, hipMemcpyDeviceToHost, c->stream));
    unsigned long long ovf = 0ull, cnt = 0ull; double e = 0.0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t local = (size_t)(rec[1] - (unsigned long long)c->ch.offset);
    // the other figure of the chosen chain, and whether its visit table overflowed (most-visited may then be wrong)
    HIPCHK(c, hipMemcpyAsync(&e, c->ch.best_energy + local, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&cnt, c->ch.top_count + local, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&ovf, c->ch.overflow + local, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (energy) *energy = e;
    if (visits) *visits = cnt;
    if (chain) *chain = rec[1];
    if (labels) std::memcpy(labels, rec.data() + 2, (size_t)c->s_max * 2);
    if (which == 2 && ovf) return fail(c, MCMC_E_OVERFLOW, "chains_get_best: the chosen chain's visit table overflowed (visit_capacity too small); the most-visited state may be wrong");
    return MCMC_OK;
}

int mcmc_frame_marginals_device(mcmc_ctx* c, int frame, const void** d_counts, size_t* count)
{
    if (!frame_ok(c, frame) || !d_counts) return fail(c, MCMC_E_BADARG, "frame_marginals_device: bad argument");
    const size_t n = (size_t)c->s_max * (size_t)c->p.marginal_bins;
    *d_counts = c->d_marg + (size_t)frame * n;
    if (count) *count = n;
    return MCMC_OK;
}

int mcmc_chains_get_scalars(mcmc_ctx* c, double* energy, double* best_energy, uint64_t* state_hash, uint64_t* trace_hash,
                            uint64_t* n_accept, uint64_t* n_case, int32_t* status, uint64_t* top_count,
                            uint64_t* top_hash, uint64_t* overflow)
{
    if (!c || c->ch.n == 0) return fail(c, MCMC_E_STATE, "chains_get_scalars: no chains");
    HIPCHK(c, hipSetDevice(c->device));
    Chains& h = c->ch;
    const size_t N = (size_t)h.n;
    auto cp = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        return dst ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream) : hipSuccess;
    };
    HIPCHK(c, cp(energy, h.energy, N * 8)); HIPCHK(c, cp(best_energy, h.best_energy, N * 8));
    HIPCHK(c, cp(state_hash, h.state_hash, N * 8)); HIPCHK(c, cp(trace_hash, h.trace_hash, N * 8));
    HIPCHK(c, cp(n_accept, h.n_accept, N * 8)); HIPCHK(c, cp(n_case, h.n_case, N * 24));
    HIPCHK(c, cp(status, h.status, N * sizeof(int))); HIPCHK(c, cp(top_count, h.top_count, N * 8));
    HIPCHK(c, cp(top_hash, h.top_hash, N * 8)); HIPCHK(c, cp(overflow, h.overflow, N * 8));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_chains_get_visits(mcmc_ctx* c, uint64_t* hash, uint64_t* count)
{
    if (!c || !hash || !count) return fail(c, MCMC_E_BADARG, "chains_get_visits: bad argument");
    if (c->ch.n == 0) return fail(c, MCMC_E_STATE, "chains_get_visits: no chains");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t NH = (size_t)c->ch.n * (size_t)c->ch.hcap;
    HIPCHK(c, hipMemcpyAsync(hash, c->ch.visit_hash, NH * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(count, c->ch.visit_count, NH * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_frame_get_marginals(mcmc_ctx* c, int frame, uint64_t* out)
{
    if (!frame_ok(c, frame) || !out) return fail(c, MCMC_E_BADARG, "get_marginals: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->s_max * (size_t)c->p.marginal_bins;
    HIPCHK(c, hipMemcpyAsync(out, c->d_marg + (size_t)frame * n, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_chains_recompute_energy(mcmc_ctx* c, double* out)
{
    if (!c || !out) return fail(c, MCMC_E_BADARG, "recompute_energy: bad argument");
    if (c->ch.n == 0) return fail(c, MCMC_E_STATE, "recompute_energy: no chains");
    HIPCHK(c, hipSetDevice(c->device));
    ChainArgs a = chain_args(c);
    HIPCHK(c, launch_chain_energy(a, c->ch.tmp_energy, c->stream));
    HIPCHK(c, hipMemcpyAsync(out, c->ch.tmp_energy, (size_t)c->ch.n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

// ---------------------------------------------------------------- self tests
int mcmc_selftest_math(mcmc_ctx* c, const double* x, const double* y, int n, double* out_log, double* out_exp, double* out_div)
{
    if (!c || !x || !y || n <= 0 || !out_log || !out_exp || !out_div) return fail(c, MCMC_E_BADARG, "selftest_math: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = (size_t)n;
    if (c->self_cap < 5 * N) { dfree(c->d_self); HIPCHK(c, dalloc(&c->d_self, 5 * N)); c->self_cap = 5 * N; }
    double* d = c->d_self;
    HIPCHK(c, hipMemcpyAsync(d, x, N * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + N, y, N * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_selftest_math(d, d + N, n, d + 2 * N, d + 3 * N, d + 4 * N, c->stream));
    HIPCHK(c, hipMemcpyAsync(out_log, d + 2 * N, N * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(out_exp, d + 3 * N, N * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(out_div, d + 4 * N, N * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_selftest_wave_sums(mcmc_ctx* c, const double* x, int n_waves, double* out_plain, double* out_folded)
{
    if (!c || !x || n_waves <= 0 || n_waves > 65535 || !out_plain || !out_folded) return fail(c, MCMC_E_BADARG, "selftest_wave_sums: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t NI = (size_t)n_waves * 11 * 64, NO = (size_t)n_waves * 11;
    if (c->self_cap < NI + 2 * NO) { dfree(c->d_self); HIPCHK(c, dalloc(&c->d_self, NI + 2 * NO)); c->self_cap = NI + 2 * NO; }
    double* d = c->d_self;
    HIPCHK(c, hipMemcpyAsync(d, x, NI * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_selftest_wave_sums(d, n_waves, d + NI, d + NI + NO, c->stream));
    HIPCHK(c, hipMemcpyAsync(out_plain, d + NI, NO * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(out_folded, d + NI + NO, NO * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

int mcmc_selftest_philox(mcmc_ctx* c, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    if (!c || !ctr || !key || !out) return fail(c, MCMC_E_BADARG, "selftest_philox: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->self_cap < 8) { dfree(c->d_self); HIPCHK(c, dalloc(&c->d_self, 8)); c->self_cap = 8; }
    uint32_t h[6] = { ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1] };
    uint32_t* d = reinterpret_cast<uint32_t*>(c->d_self);
    HIPCHK(c, hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_selftest_philox(d, d + 8, c->stream));
    HIPCHK(c, hipMemcpyAsync(out, d + 8, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCMC_OK;
}

}  // extern "C"
