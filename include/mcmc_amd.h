/*
 * mcmc_amd.h -- C ABI of the MI355X (gfx950) batched proposal-evaluation path
 * of a-price/MCMC.  Plain pointers and sizes only; no C++ or torch types.
 *
 * This is the drop-in boundary for the PROBLEM concept of MCMC<PROBLEM>
 * (reference include/MCMC.h:19-95; include/Problem.h:20-29): everything the
 * reference computes between OverSegmentation handing over a Graph and
 * Statistics being read back.  Each entry point cites the reference interface
 * it replaces.  Status: 0 = ok, negative = error; mcmc_last_error() gives the
 * text.  The reference reports the same conditions by throwing Exception_
 * (include/MyMath.h:36-44).
 *
 * Threading: one context per (host thread, GPU); a context is not shared
 * between threads (the reference is single-threaded and not re-entrant).
 * All host buffers are owned by the caller.  Calls are asynchronous on the
 * context's stream unless they return data to the host.
 */
#ifndef MCMC_AMD_H
#define MCMC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCMC_OK 0
#define MCMC_E_BADARG (-1)       /* null pointer, size out of range, bad frame index */
#define MCMC_E_DEVICE (-2)       /* HIP runtime error (text in mcmc_last_error) */
#define MCMC_E_STATE (-3)        /* call out of order (e.g. chains before moments) */
#define MCMC_E_NAN_RATIO (-4)    /* "Target/Proposal ratio is nan!" Problem.cpp:216-217, :443 */
#define MCMC_E_INF_ENERGY (-5)   /* "log probability is inf" Graph.cpp:137 */
#define MCMC_E_TOO_LARGE (-6)    /* superpixel count beyond what the chain kernel's LDS state holds */
#define MCMC_E_NOSUPERPIXELS (-7)/* frame has no superpixel that passes the A1 filter */
#define MCMC_E_RANGE (-8)        /* "Weird channel value." OverSegmentation.cpp:382 */
#define MCMC_E_OVERFLOW (-9)     /* a caller-fixed visit table filled up: Statistics results may be incomplete */

#define MCMC_NOLABEL 0xFFFFu     /* pixel / superpixel that belongs to no superpixel / segment */
#define MCMC_MOMENT_STRIDE 10    /* N Su Sv Suu Suv Svv Sb Sub Svb Sbb (int64) */
#define MCMC_MAX_DEGREE 32760    /* neighbours of one superpixel (12-bit part index x 8 in the growth FIFO) */
#define MCMC_FIT_STRIDE 16       /* Suu Suv Svv Su Sv N | Sub Svb Sb | theta_a theta_b theta_c | Sbb ssr 0 0 (f64, centred) */

typedef struct mcmc_ctx mcmc_ctx;

/* Compile-time constants of the reference made run-time parameters
 * (include/MyMath.h:28-32; src/libraries/Segmentation.cpp:16). */
typedef struct mcmc_params {
    int32_t width;               /* kWidth  = 640 */
    int32_t height;              /* kHeight = 480 */
    double disparity_variance;   /* kDisparityVariance = 7 */
    double new_segment_prob;     /* newSegmentProb = 0.8 */
    int32_t marginal_bins;       /* Lmax of the label-marginal histogram (build's choice, 64) */
    int32_t visit_capacity;      /* Statistics::entries_ (Statistics.h:30): 0 = grows with the run so that every
                                  * visited state is kept, as in the reference (default: starts at 512 slots per
                                  * chain and grows from the MEASURED occupancy -- slots in use + steps about to be
                                  * made, not twice the steps ever made; if the larger table cannot be allocated the
                                  * run goes on and further states count as `overflow`); > 0 = fixed number of
                                  * slots per chain, visits of further states are counted in `overflow` */
} mcmc_params;

void mcmc_default_params(mcmc_params *p);

/* ---- context ----------------------------------------------------------- */
int mcmc_ctx_create(int device, const mcmc_params *params, mcmc_ctx **out);
int mcmc_ctx_destroy(mcmc_ctx *ctx);
const char *mcmc_last_error(const mcmc_ctx *ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of the context's own. */
int mcmc_ctx_set_stream(mcmc_ctx *ctx, void *hip_stream);
/* Waits for the context's stream.  After a mcmc_chains_run it also reports MCMC_E_DEVICE if a chain of the
 * several-waves-per-chain kernel gave up waiting for its turn (about 0.1 s without any step of that chain being
 * committed; never observed -- the alternative would be a hang): that chain's status in mcmc_chains_get_scalars
 * is -2 and it no longer moves; the other chains are unaffected. */
int mcmc_ctx_synchronize(mcmc_ctx *ctx);
const char *mcmc_version(void);

/* ---- frames: replaces what OverSegmentation::createSuperPixels leaves in
 * Graph::superPixels_ (A_, b_, neighbors_, plane_) -- reference
 * src/libraries/OverSegmentation.cpp:89-233, include/MyMath.h:177-183. ------ */

/* Reserve device storage for n_frames frames of width x height pixels with at
 * most s_max superpixels and e2_max directed adjacency entries each. */
int mcmc_frames_alloc(mcmc_ctx *ctx, int n_frames, int s_max, int e2_max);

/* Disparity plane (int16, the reference's .at<short>, OverSegmentation.cpp:148)
 * and superpixel id per pixel (uint16; MCMC_NOLABEL = dropped pixel), host memory. */
int mcmc_frame_upload(mcmc_ctx *ctx, int frame, const int16_t *disp, const uint16_t *label,
                      int n_superpixels);
/* Same, from device memory already resident in HBM (device-to-device copy). */
int mcmc_frame_upload_device(mcmc_ctx *ctx, int frame, const void *d_disp, const void *d_label,
                             int n_superpixels);
/* Adjacency in CSR form, neighbours strictly ascending by id (no duplicates): SuperPixel::neighbors_
 * keys (include/MyMath.h:181; filled at OverSegmentation.cpp:186-190).  MCMC_E_BADARG for a malformed CSR,
 * MCMC_E_TOO_LARGE for a superpixel with more than MCMC_MAX_DEGREE neighbours. */
int mcmc_frame_set_graph(mcmc_ctx *ctx, int frame, const uint32_t *row_ptr, const uint32_t *col,
                         uint32_t e2);

/* K1 sp_moments + K1b local fit for frames [first, first+count):
 * `new Plane(A, b)` -> solve() (OverSegmentation.cpp:219; MyMath.h:120-166)
 * and the drop filter (OverSegmentation.cpp:198-222, 422-455). */
int mcmc_frames_moments(mcmc_ctx *ctx, int first, int count);
/* K2: Problem::computeEdgeProbs (src/libraries/Problem.cpp:76-114, 420-445). */
int mcmc_frames_edge_probs(mcmc_ctx *ctx, int first, int count);

int mcmc_frame_get_moments(mcmc_ctx *ctx, int frame, int64_t *out /* S*10 */);
int mcmc_frame_get_local_fit(mcmc_ctx *ctx, int frame, double *fit /* S*MCMC_FIT_STRIDE */, uint8_t *valid /* S */,
                             double *local_err /* S, Plane::error_, may be NULL */);
int mcmc_frame_get_edge_probs(mcmc_ctx *ctx, int frame, double *q /* e2; -1 = edge to a dropped superpixel */);

/* Segment(superPixels) constructor: computeHash + computeProbability
 * (include/Graph.h:33-40; Graph.cpp:69-138).  theta3 is (a, b, c) of
 * disparity = a*(u - W/2) + b*(v - H/2) + c. */
int mcmc_segment_probability(mcmc_ctx *ctx, int frame, const uint32_t *members, int n_members,
                             double *probL, double *theta3, double *err);

/* No counterpart in the reference.  mcmc_segment_probability for callers that cannot allocate device memory (tight
 * serving loops, harnesses that forbid hipMalloc): same arguments, same result bit for bit, but the membership
 * bytes, the member list and the result are staged in a static device array of the library (328 KB per GPU, held
 * until the process ends) instead of buffers the context allocates, and the call makes no hipMalloc / hipFree.
 * It is NOT capturable into a HIP graph: it copies the member list from the caller's (pageable) host memory, copies
 * the result back and synchronises the stream before it returns.  Limits: n_members <= 65536 and at most 65536
 * superpixels, else MCMC_E_TOO_LARGE.  Every context on one GPU shares that GPU's array: calls that overlap on one
 * GPU (two host threads, two contexts) race; calls one after the other are safe, each clears what it needs. */
int mcmc_segment_probability_noalloc(mcmc_ctx *ctx, int frame, const uint32_t *members, int n_members,
                                     double *probL, double *theta3, double *err);

/* ---- chains: Problem::initializeState + MCMC<Problem>::run with
 * FastMetropolisHastings::iterate (Problem.cpp:447-473, 116-224; MCMC.h:43-89,
 * 207-238; Segmentation.cpp:10-237; Statistics.h:33-70). -------------------- */

/* n_chains independent chains; chain k samples frame k / chains_per_frame and
 * draws from Philox4x32-10 keyed (seed, chain_offset + k), so a run sharded
 * over several GPUs by chain_offset reproduces the single-GPU chains exactly. */
int mcmc_chains_init(mcmc_ctx *ctx, int n_chains, int chains_per_frame, uint32_t seed,
                     uint32_t chain_offset);
/* burn_in steps without statistics then iters steps with Statistics::update
 * after each (MCMC.h:55-82).  May be called repeatedly; steps continue. */
int mcmc_chains_run(mcmc_ctx *ctx, uint64_t burn_in, uint64_t iters);
/* Optional accept trace for parity tests: one byte per chain per step of the
 * NEXT mcmc_chains_run, bit0 accept, bits1-2 case (0 own segment, 1 split,
 * 2 merge/exchange), 0xFF = chain frozen by an error.  capacity_steps bytes
 * per chain are reserved on the device; 0 disables. */
int mcmc_chains_set_trace(mcmc_ctx *ctx, uint64_t capacity_steps);
int mcmc_chains_get_trace(mcmc_ctx *ctx, uint8_t *out /* n_chains*steps of last run */);

/* which: 0 current state, 1 best-energy state, 2 most-visited state
 * (printResults uses the most visited, src/exe/can05_sampler.cpp:36).
 * canonical != 0: first-appearance relabel of Segmentation::computeHash
 * (Segmentation.cpp:197-233); else the smallest member id of each segment.
 * which = 2 returns MCMC_E_OVERFLOW (labels still written) if a caller-fixed visit table filled up. */
int mcmc_chains_get_labels(mcmc_ctx *ctx, int which, int canonical, uint16_t *out /* n_chains*s_max */);
/* The ONE labeling a caller prints or exchanges (printResults, src/exe/can05_sampler.cpp:23-45 takes
 * Statistics::getOrderedStates().front(), Statistics.h:50-70; north_star's all-gather of best-energy labelings):
 * the arg-max over this context's chains is taken on the device -- which = 1: highest best-energy, which = 2: most
 * visits of a chain's most-visited state; ties go to the lowest chain id -- and only that chain's labeling comes
 * back (s_max labels, MCMC_NOLABEL beyond the frame's superpixels; canonical as in mcmc_chains_get_labels).
 * energy / visits: the chosen chain's best energy and top count; chain: its GLOBAL id (chain_offset + k).  Any
 * output pointer may be NULL.  which = 2 returns MCMC_E_OVERFLOW (outputs still written) if that chain's
 * caller-fixed visit table filled up. */
int mcmc_chains_get_best(mcmc_ctx *ctx, int which, int canonical, double *energy, uint64_t *visits,
                         uint64_t *chain, uint16_t *labels /* s_max */);
/* The same record left on the device for a collective (RCCL all-gather without a host copy): 8-byte words
 * [0] the key (which = 1: energy as f64 bits; 2: visit count), [1] global chain id, [2..] the labels, four per
 * word; *bytes = (2 + (s_max + 3) / 4) * 8.  Valid on the context's stream until the next call of either. */
int mcmc_chains_best_record_device(mcmc_ctx *ctx, int which, int canonical, const void **d_record, size_t *bytes);
/* Any pointer may be NULL.  n_case: 3 per chain.  status: per-chain MCMC_E_*. */
int mcmc_chains_get_scalars(mcmc_ctx *ctx, double *energy, double *best_energy, uint64_t *state_hash,
                            uint64_t *trace_hash, uint64_t *n_accept, uint64_t *n_case,
                            int32_t *status, uint64_t *top_count, uint64_t *top_hash,
                            uint64_t *overflow);
/* Statistics::entries_ per chain (Statistics.h:30-47): mcmc_chains_visit_capacity() (hash, count) slots, count 0 =
 * empty.  Statistics::getOrderedStates (Statistics.h:50-70) = these entries sorted by count, descending;
 * Statistics::getCount(state) (:73-83) = the count stored with the state's hash. */
int mcmc_chains_get_visits(mcmc_ctx *ctx, uint64_t *hash, uint64_t *count);
/* Slots per chain of the visit table right now (it grows with the run when params.visit_capacity == 0). */
int mcmc_chains_visit_capacity(const mcmc_ctx *ctx);
/* Optional: make room now for `measured_steps` further Statistics::update calls per chain, so that no later
 * mcmc_chains_run has to look at the table or grow it (looking synchronises the stream; growing allocates and
 * re-inserts). */
int mcmc_chains_reserve_visits(mcmc_ctx *ctx, uint64_t measured_steps);
/* The out-parameters of Problem::propose for the last step each chain took (Problem.cpp:116,213-217;
 * MCMC.h:214-219): targetRatio (log domain, the literal 1.0 of case A included) and proposalRatio (linear).
 * Either pointer may be NULL; n_chains values each. */
int mcmc_chains_get_last_ratios(mcmc_ctx *ctx, double *targetL, double *proposal);
/* Label marginals of one frame summed over its chains: s_max*marginal_bins counts. */
int mcmc_frame_get_marginals(mcmc_ctx *ctx, int frame, uint64_t *out);
/* The same counts where they live (device memory, uint64[s_max * marginal_bins]) for a collective on the device;
 * read-only for the caller, changed by every mcmc_chains_run. */
int mcmc_frame_marginals_device(mcmc_ctx *ctx, int frame, const void **d_counts, size_t *count);
/* Segmentation::computeProbability (Segmentation.cpp:185-195) recomputed from the segment cache. */
int mcmc_chains_recompute_energy(mcmc_ctx *ctx, double *out /* n_chains */);

/* ---- measurement -------------------------------------------------------- */
/* Device time of the most recent launch of each kernel, measured with HIP
 * events on the context's stream: 0 sp_moments, 1 local_fit, 2 edge_prob,
 * 3 chain_step.  Returns milliseconds, < 0 if never launched. */
double mcmc_last_kernel_ms(mcmc_ctx *ctx, int kernel);
int mcmc_set_timing(mcmc_ctx *ctx, int enable);
/* Launch-shape overrides for tuning runs (0 / -1 = automatic): "k1_bands",
 * "k1_threads", "k1_use_lds", "k1_debug", "k3_profile", "k3_dense" (eight chains per workgroup with the records
 * read through L2: -1 never, 0 when the chains exceed four per CU, 1 always), "k3_spec" (several waves per chain in
 * step order, the further ones evaluating the next steps against the undecided state: -1 never, 0 = as many as fit
 * LDS, up to four, when there are at most four chains per CU, 1 = the same whatever the chain count, 2..4 = exactly
 * that many if they fit).  Results never depend on them. */
int mcmc_ctx_set_option(mcmc_ctx *ctx, const char *name, int value);
/* Diagnostic (no counterpart in the reference): which form of the chain kernel the last mcmc_chains_run
 * launched, as 10 * placement + chains per workgroup; placement 1 = all frame tables in LDS, 2 = tables in LDS
 * except the per-superpixel records (eight chains per workgroup), 3 = only the CSR topology in LDS (large frames),
 * 4 = the speculative form (44 with two waves per chain, 144 with three, 244 with four), 0 = everything behind
 * L2.  Results never depend on it. */
int mcmc_chains_last_variant(const mcmc_ctx *ctx);

/* After option "k3_profile" = 1 and a run: shader-clock cycles summed over chains for 12
 * phases of the step, then sums of component size, draws and refit members (diagnostic build of the chain kernel; never used for reported numbers). */
int mcmc_chains_get_profile(mcmc_ctx *ctx, uint64_t *out16);

/* Device self-test of the deterministic math shared by every kernel: fills
 * out_log[i] = det_log(x[i]), out_exp[i] = det_exp(x[i]),
 * out_div[i] = x[i] / y[i] (parity of IEEE division), n values. */
int mcmc_selftest_math(mcmc_ctx *ctx, const double *x, const double *y, int n, double *out_log,
                       double *out_exp, double *out_div);
int mcmc_selftest_philox(mcmc_ctx *ctx, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* Device self-test of the wave reductions the chain kernel sums with (no counterpart in the reference: its sums are
 * sequential loops, Segmentation.cpp:127-146; the order here is the fixed xor butterfly of DESIGN.md section 3 that the
 * oracle mirrors).  x holds n_waves * 11 * 64 values, [wave][value][lane]; out_plain[wave][value] = the butterfly sum
 * taken one value at a time, out_folded[wave][value] = the same sums from the folded form that takes eleven at once.
 * Both must equal the butterfly computed on the host, bit for bit. */
int mcmc_selftest_wave_sums(mcmc_ctx *ctx, const double *x, int n_waves, double *out_plain, double *out_folded);

/* ======================================================================== *
 * Over-segmentation front end: the producer of the
 * label map and superpixel graph above.  Replaces OverSegmentation::overSegment
 * steps 1-4 (src/libraries/OverSegmentation.cpp:19-48); step 5 is
 * mcmc_frames_edge_probs.
 * ======================================================================== */
typedef struct mcmc_overseg mcmc_overseg;

/* OverSegmentationParameters (include/OverSegmentation.h:18-44), file order of
 * output/overSegmentationParams.txt (IO.cpp:105-134). */
typedef struct mcmc_overseg_params {
    int window_size;          /* windowSize_: < 2 skips the filter */
    double spatial_stdev;     /* spatialStdev_ */
    double disparity_stdev;   /* disparityStdev_ */
    int weight_threshold;     /* weightThreshold_ (size_t in the reference, the c of segment_graph) */
    double color_factor, disparity_factor, distance_factor;
} mcmc_overseg_params;

void mcmc_overseg_default_params(mcmc_overseg_params *p);   /* 9 / 5.0 / 3.0 / 300 / 1 / 1 / 1 */
int mcmc_overseg_create(int device, int width, int height, mcmc_overseg **out);
int mcmc_overseg_destroy(mcmc_overseg *os);
const char *mcmc_overseg_last_error(const mcmc_overseg *os);

/* overSegment steps 1-4 (OverSegmentation.cpp:25-48): cross-bilateral filter of
 * the colours by the disparities (:304-391), 4 edges per pixel weighted by
 * colour, disparity and distance (:236-301), Felzenszwalb-Huttenlocher
 * union-find over the sorted edges (segment-graph.h:47-111: std::sort and the sequential scan
 * on the host, as in the reference; option "tie_order" = 0 sorts on the device instead, see
 * mcmc_overseg_set_option), createSuperPixels (:89-194: dense labels in ascending root
 * pixel order = the std::map order, neighbours through LEFT / TOP / TOP-RIGHT
 * contacts).  colors: H*W*3 bytes in cv::Vec3b order (B, G, R); disp: H*W shorts.
 * Errors: MCMC_E_BADARG for non-positive deviations (the reference throws,
 * :309-310), MCMC_E_RANGE for "Weird channel value." (:382), MCMC_E_TOO_LARGE
 * when more than 65534 superpixels come out. */
int mcmc_overseg_run(mcmc_overseg *os, const uint8_t *colors, const int16_t *disp,
                     const mcmc_overseg_params *params, int *n_superpixels, uint32_t *n_directed_edges);
/* "tie_order": the order of edges of EQUAL weight in segment_graph's scan.  The reference sorts with std::sort on
 * the weight alone (segment-graph.h:35-37,53), which leaves equal weights wherever that standard library's
 * introsort happens to put them; equal weights are common (colour and disparity distances come from integers).
 * 1 (default): that very call -- std::sort of this library's C++ runtime on the host over the edges in creation
 * order, i.e. what the reference's binary does when built with the same runtime; 0: stable radix sort on the device
 * (rocPRIM, a library sort), equal weights in creation order -- faster (0.4 ms against ~35 ms at 640x480), but a
 * superpixel layout of this library's own wherever weights tie; mcmc_overseg_last_tie_count says how many did. */
int mcmc_overseg_set_option(mcmc_overseg *os, const char *name, int value);
/* Results of the last run.  The device buffers (int16 disparities, uint16 labels)
 * can be handed to mcmc_frame_upload_device, the graph to mcmc_frame_set_graph. */
int mcmc_overseg_get_smoothed(mcmc_overseg *os, uint8_t *out /* H*W*3 */);
long long mcmc_overseg_get_edges(mcmc_overseg *os, int32_t *a, int32_t *b, float *w /* creation order, any may be NULL; returns the count */);
int mcmc_overseg_get_roots(mcmc_overseg *os, int32_t *out /* H*W: universe::find(k) */);
int mcmc_overseg_get_labels(mcmc_overseg *os, uint16_t *out /* H*W */);
int mcmc_overseg_get_graph(mcmc_overseg *os, uint32_t *row_ptr /* S+1 */, uint32_t *col /* e2 */);
int mcmc_overseg_device_buffers(mcmc_overseg *os, const void **d_disp, const void **d_label);
int mcmc_overseg_last_ms(mcmc_overseg *os, float *filter_ms, float *edges_ms, float *union_find_ms, float *graph_ms);
/* Diagnostic (no counterpart in the reference): edges of the last run whose weight equals that of the edge before
 * them in scan order, i.e. how much of the scan order the tie rule decided (< 0: no run yet). */
long long mcmc_overseg_last_tie_count(mcmc_overseg *os);
/* Diagnostic (no counterpart in the reference): the two parts of union_find_ms that are not transfers --
 * the device sort and the host scan. */
int mcmc_overseg_last_sort_ms(mcmc_overseg *os, float *device_sort_ms, float *host_scan_ms);

/* A batch of frames through overSegment (OverSegmentation.cpp:19-54, once per frame; the reference has no batch
 * form: can05_sampler.cpp:84 handles one frame per process).  The sort and the union-find scan of a frame run on a
 * host thread, so the batch owns n_workers handles and threads and hands frames out to them; frame f's colours are
 * colors + f*H*W*3, its disparities disp + f*H*W.  Results of every frame are kept until the next run: the
 * disparity and label planes on the device (mcmc_overseg_batch_device_buffers -> mcmc_frame_upload_device), the
 * neighbour CSR on the host.  n_superpixels / n_directed_edges: one entry per frame.  Options as for a single handle
 * (tie_order).  On an error the first failing frame is named in mcmc_overseg_batch_last_error and no result is kept. */
typedef struct mcmc_overseg_batch mcmc_overseg_batch;
int mcmc_overseg_batch_create(int device, int width, int height, int n_workers, mcmc_overseg_batch **out);
int mcmc_overseg_batch_destroy(mcmc_overseg_batch *b);
const char *mcmc_overseg_batch_last_error(const mcmc_overseg_batch *b);
int mcmc_overseg_batch_set_option(mcmc_overseg_batch *b, const char *name, int value);
int mcmc_overseg_batch_run(mcmc_overseg_batch *b, int n_frames, const uint8_t *colors, const int16_t *disp,
                           const mcmc_overseg_params *params, int *n_superpixels, uint32_t *n_directed_edges);
int mcmc_overseg_batch_get_graph(mcmc_overseg_batch *b, int frame, uint32_t *row_ptr /* S+1 */, uint32_t *col /* e2 */);
int mcmc_overseg_batch_device_buffers(mcmc_overseg_batch *b, int frame, const void **d_disp, const void **d_label);
int mcmc_overseg_batch_get_labels(mcmc_overseg_batch *b, int frame, uint16_t *out /* H*W */);
long long mcmc_overseg_batch_tie_count(mcmc_overseg_batch *b, int frame);
float mcmc_overseg_batch_last_wall_ms(const mcmc_overseg_batch *b);

/* ======================================================================== *
 * Path B: multiview segmentation over an SPGraph (world graph of several
 * views).  Replaces SPGraph / MultiviewSegment / MultiviewSegmentation /
 * MathUtils for the proposal loop of src/exe/PatternedMatcher.cpp:58-130.
 * ======================================================================== */
typedef struct mcmc_mv mcmc_mv;

#define MCMC_MV_PLANE_STRIDE 5   /* nx ny nz d det(cov): PlaneModel (include/MathUtils.h) */
#define MCMC_MV_MOMENT_STRIDE 10 /* N Sx Sy Sz Sxx Sxy Sxz Syy Syz Szz (f64) */

int mcmc_mv_create(int device, mcmc_mv **out);
int mcmc_mv_destroy(mcmc_mv *mv);
const char *mcmc_mv_last_error(const mcmc_mv *mv);
int mcmc_mv_synchronize(mcmc_mv *mv);

/* The SPGraph (include/SPGraph.h:43-86): vertex v owns the XYZ points
 * points_xyzw[4*pt_ptr[v] .. 4*pt_ptr[v+1]) (SPNode::subCloudCartesian, PCL
 * PointXYZ = 4 floats, MathUtils.cpp:26); out-edges in CSR form ascending by
 * target with SPEdge::BernoulliProbability per entry (symmetric). */
int mcmc_mv_set_graph(mcmc_mv *mv, int n_vertices, const uint32_t *pt_ptr, const float *points_xyzw,
                      const uint32_t *row_ptr, const uint32_t *col, const double *bernoulli, uint32_t e2);
/* Per-vertex sufficient statistics of the plane fit (what fitPlaneSVD's
 * concatenate-and-SVD needs from a vertex); runs once per graph. */
int mcmc_mv_vertex_moments(mcmc_mv *mv);
int mcmc_mv_get_vertex_moments(mcmc_mv *mv, double *out /* V*10 */);
/* MultiviewSegment::computeFitPlane + computeProbability for an explicit vertex
 * set (MultiviewSegment.cpp:33-96; MathUtils::fitPlaneSVD MathUtils.cpp:22-54;
 * SPNode::getErrorForPlaneModel SPGraph.cpp:89-104). */
int mcmc_mv_fit_segment(mcmc_mv *mv, const uint32_t *members, int n_members, double *plane5,
                        double *probability, double *local_err);
/* pMerge(theta1, theta2, weights, temperature) for n_pairs pairs of dim-vectors
 * (GraphUtils.cpp:11-20), host arrays in, host array out. */
int mcmc_mv_pmerge(mcmc_mv *mv, const float *theta1, const float *theta2, const float *weights, int dim,
                   int n_pairs, double temperature, double *out);

/* The world graph of several scans, built as the reference builds it: getPlanarAdjacencyGraph
 * (GraphUtils.cpp:52-138) keeps an in-view neighbour pair when pMerge > in_view_threshold, and
 * mergeNewScanGraph (GraphUtils.cpp:140-202) joins every vertex of a later scan with EVERY vertex
 * already in the graph -- earlier vertices of its own scan included -- when pMerge > merge_threshold
 * (SimpleSeg.cpp:409), then re-links the scan's in-view edges.  Vertices [0, n_first_scan) are the
 * first scan, which only gets its in-view edges.  theta: V x dim full models (SPNode::getFullModel),
 * in_row_ptr / in_col: in-view neighbour CSR over the same vertex ids (may be NULL).
 * Output: symmetric CSR, neighbours ascending, prob = SPEdge::BernoulliProbability; e2_out is always
 * set, MCMC_E_TOO_LARGE if it exceeds capacity (call again with room). */
int mcmc_mv_build_world_graph(mcmc_mv *mv, int V, int dim, const float *theta, const float *weights,
                              int n_first_scan, const uint32_t *in_row_ptr, const uint32_t *in_col,
                              double in_view_threshold, double merge_threshold, double temperature,
                              uint32_t *row_ptr_out, uint32_t *col_out, double *prob_out, uint32_t capacity,
                              uint32_t *e2_out);

/* MultiviewSegmentation(graph) per chain (MultiviewSegmentation.cpp:11-69):
 * Bernoulli every edge, connected components, fit every segment.  Philox key
 * (seed, chain_offset + k).  max_new_segments bounds the segments split
 * proposals append (a chain that runs out freezes with MCMC_E_TOO_LARGE);
 * split_prob is PatternedMatcher.cpp:77's 0.10. */
int mcmc_mv_chains_init(mcmc_mv *mv, int n_chains, uint32_t seed, uint32_t chain_offset,
                        int max_new_segments, double split_prob);
/* `steps` iterations of the loop at PatternedMatcher.cpp:58-130. */
/* Tuning: "split_step" 0 one fused kernel per run; 1 or -1 (default) four kernels per iteration with the L1
 * residuals of all chains evaluated on the f64 matrix cores (same results bit for bit; no reference counterpart;
 * faster at every chain count: 111 us against 605 us per iteration for one chain at BASELINE config 4).
 * "growth_ahead" -1 (default: on up to 3072 chains unless "groups", "split_growth" or "fuse_halves" are set by hand)
 * / 0 / 1: the component growth of iteration t + 1 -- which needs nothing of the iterations before it: its trials belong
 * to the edges and the iteration, the segments it walks are the initial ones -- runs as a kernel of its own on a second
 * stream while the main stream finishes iteration t - 1, fits the planes of iteration t and evaluates its residuals
 * (config 4, per iteration: one chain 68 -> 42 us, 256 chains 118 -> 71, 1024 chains 152 -> 141, 2048 chains 242 -> 238);
 * same chains.  A run also grows the step after its last one, and the next mcmc_mv_chains_run on the same chains starts
 * from it (mcmc_mv_chains_init, mcmc_mv_set_graph, mcmc_mv_set_option or a run of another kind in between discard it).  "growth_ahead_waves" 0 (automatic) / 4 / 8: waves per chain of that kernel.  The options below
 * shape the iteration when the growth does not run ahead (more than 3072 chains, or by hand).
 * "groups" 0 (default: two from 1024 chains on) .. 8: the chains run in that many groups, each on a stream of its
 * own, so that a group waits only for its own slowest chain; same chains whatever the number.  "split_growth" -1
 * (default: up to 256 chains) / 0 / 1: the first half of an iteration as a growth kernel with eight waves per chain
 * followed by a plane-fit kernel (a chain's latency, which is what few chains wait for: 92 -> 83 us per iteration for
 * one chain at config 4); same chains.  "fuse_halves" -1 (default: on unless the growth kernel runs apart) / 0 / 1:
 * the second half of an iteration and the first half of the next as one kernel (a run of n iterations is first half,
 * n - 1 fused kernels, second half: 186 -> 174 us per iteration for 1024 chains in one group at config 4); same chains.
 * "profile" 1: cycle counters for mcmc_mv_chains_get_profile (one group, halves apart). */
int mcmc_mv_set_option(mcmc_mv *mv, const char *name, int value);
int mcmc_mv_chains_run(mcmc_mv *mv, uint64_t steps);
/* Per step of the NEXT run and chain: flags (bit0 accepted, bit1 split proposal,
 * bit2 skipped: no neighbour segment; 0xFF frozen) and the ratio a. */
int mcmc_mv_chains_set_trace(mcmc_mv *mv, uint64_t capacity_steps);
int mcmc_mv_chains_get_trace(mcmc_mv *mv, uint8_t *flags, double *ratio);
/* which: 0 = SPNodeState::currentSegmentID (the initial partition; the reference
 * never updates it), 1 = the segments' vertex sets (moved by split proposals). */
int mcmc_mv_chains_get_labels(mcmc_mv *mv, int which, uint32_t *out /* n_chains*V */);
int mcmc_mv_chains_get_scalars(mcmc_mv *mv, uint64_t *n_accept, uint64_t *n_split, uint64_t *n_skip,
                               uint64_t *trace_hash, int32_t *n_segments, int32_t *status);
int mcmc_mv_chains_get_planes(mcmc_mv *mv, int chain, double *out /* n_segments*5 */);
/* Device time of the last mcmc_mv_chains_run launch, milliseconds (< 0: none). */
/* Diagnostic (no counterpart in the reference): with option "profile" = 1 the first half of a split
 * iteration sums shader-clock cycles per phase over all chains of the last run: [0] set-up, [1] start vertex,
 * [2] growth, [3] move type / neighbour segments, [4] planes, [5] hand-over, [6] write-back, [7] the counting for [10..12] itself (not a phase of the product), [8] growth levels,
 * [9] grown vertices, [10] points in SB1 u SB2, [11] (vertex, plane) residual evaluations requested, [12] point
 * residuals requested; and the L1 kernel, summed over one item in 16: [16] cycles before the first block (item record,
 * pair list, planes), [17] cycles in the block loop, [18] cycles after it, [19] items, [20] blocks of 16 points,
 * [21] matrix instructions, [22] waves that had an item, [23] their cycles from first to last item. */
int mcmc_mv_chains_get_profile(mcmc_mv *mv, uint64_t *out32);
double mcmc_mv_last_run_ms(mcmc_mv *mv);

#ifdef __cplusplus
}
#endif
#endif /* MCMC_AMD_H */
