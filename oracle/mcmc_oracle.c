/*
 * mcmc_oracle.c -- ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, double precision, no FMA contraction) of the
 * Path-A hot path of a-price/MCMC: superpixel moments, local plane fit, fused
 * segment fit + log-energy, edge probabilities, Swendsen-Wang component growth,
 * target selection, Metropolis-Hastings accept, state hash and visit statistics.
 *
 * Nothing under mcmc_amd/ may include, link or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker ("port" baseline), never as the thing shipped.
 *
 * PARITY PINNING: the reference (/root/reference) holds NO golden vectors, no
 * known-answer tests and no seeded runs for this path,
 * and it cannot be compiled here (gtsam/Eigen/OpenCV absent).  Its arithmetic
 * lives in gtsam 2.x (un-vendored, un-pinned).  This restatement is therefore
 * pinned by (1) tests/literal_ref.py, a numpy emulation of the gtsam call
 * sequence (whitened QR per superpixel -> stacked QR -> factor errors ->
 * covariance determinant) checked to 1e-9 relative, (2) Random123's published
 * Philox4x32-10 known-answer vectors, (3) libm log/exp for det_log/det_exp.
 * Where none of those reach (sampler trajectories) it is "parity unpinned":
 * the reference seeds libc rand() from time(NULL) and iterates containers in
 * heap-pointer order, so it has no reproducible trajectory to match.
 *
 * Every function cites the reference file:line it follows.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_NLANE 64
#define ORC_NOLABEL 0xFFFFu
#define ORC_FIT_STRIDE 16
#define ORC_MOM_STRIDE 10

/* status codes shared with include/mcmc_amd.h by value */
#define ORC_OK 0
#define ORC_E_BADARG (-1)
#define ORC_E_NAN_RATIO (-4)
#define ORC_E_INF_ENERGY (-5)

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants).       */
/* The reference draws from libc rand() (can05_sampler.cpp:52); Philox is the */
/* build's replacement so CPU and GPU can share one counter-based stream.     */
/* key = (seed, global chain id); counter = (step_lo, step_hi, block, TAG).   */
/* ------------------------------------------------------------------------- */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u
#define PHILOX_TAG 0x4D434D43u /* "MCMC" */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/*
 * Draws of one step are counter-addressed by a position p: the uniform at position
 * p is word (p>>6)&3 of Philox block (p>>8)*64 + (p&63), i.e. 64 lanes of a
 * wavefront generate positions [256 b, 256 b + 256) with one Philox call each.
 *   p = 0  start index      (rand()%S,            Segmentation.cpp:155)
 *   p = 1  discarded draw   (                     Segmentation.cpp:161)
 *   p = 2  selectSegment's uniform                (Segmentation.cpp:23)
 *   p = 3  selectSegment's index                  (Segmentation.cpp:40)
 *   p = 4  the Metropolis-Hastings uniform        (MCMC.h:223)
 *   p = 8 + 8 k + s  the Swendsen-Wang trial of neighbour s of FIFO slot k of
 *          createComponent: a popped node takes one slot per 8 neighbours (at
 *          least one), slots are numbered in FIFO order, and a neighbour owns its
 *          position whether or not it needs a trial.
 * The reference hands a trial whatever rand() returns next, which makes the draw a
 * trial gets depend on how many trials happened before it.  Numbering by FIFO
 * slot keeps the trials independent uniforms in sequential order (an entry is
 * examined at most once per step) and makes the position of a trial a function of
 * the node's place in the FIFO alone, so the GPU can try the neighbours of 8 FIFO
 * slots in one pass and serve four passes from one Philox call.  The component, the FIFO order and every
 * later decision are unchanged functions of those uniforms.
 */
#define ORC_TRIAL_POS0 8u
typedef struct {
    uint32_t key[2];
    uint32_t step_lo, step_hi;
    uint32_t cur_batch;   /* batch cached in buf */
    uint8_t have[64];
    uint32_t buf[64][4];
} orc_rng;

static void rng_begin(orc_rng *g, uint32_t seed, uint32_t chain, uint64_t step)
{
    g->key[0] = seed; g->key[1] = chain;
    g->step_lo = (uint32_t)step; g->step_hi = (uint32_t)(step >> 32);
    g->cur_batch = 0xFFFFFFFFu;
}

static uint32_t rng_u32_at(orc_rng *g, uint32_t p)
{
    uint32_t batch = p >> 8, lane = p & 63u, w = (p >> 6) & 3u;
    if (batch != g->cur_batch) { memset(g->have, 0, sizeof g->have); g->cur_batch = batch; }
    if (!g->have[lane]) {
        uint32_t ctr[4] = { g->step_lo, g->step_hi, batch * 64u + lane, PHILOX_TAG };
        orc_philox4x32_10(ctr, g->key, g->buf[lane]);
        g->have[lane] = 1;
    }
    return g->buf[lane][w];
}

/* rand()/RAND_MAX is inclusive of 1.0 (MCMC.h:223, Segmentation.cpp:23,109): one
 * multiplication by the f64 nearest to 1/(2^32 - 1); the largest word gives 1.0. */
static double rng_unif_at(orc_rng *g, uint32_t p)
{
    return (double)rng_u32_at(g, p) * 2.3283064370807974e-10;
}

/* ------------------------------------------------------------------------- */
/* Deterministic log / exp from IEEE + - * / and bit moves only, so that the  */
/* CPU and the GPU produce the same bits (libm and OCML differ in the last    */
/* ulp).  Accuracy ~2 ulp; checked against libm in tests.                     */
/* ------------------------------------------------------------------------- */
static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

#define DET_LN2_HI 6.93147180369123816490e-01 /* fdlibm split of ln 2 */
#define DET_LN2_LO 1.90821492927058770002e-10
#define DET_INV_LN2 1.44269504088896338700e+00
#define DET_SQRT2 1.41421356237309514547e+00

double orc_det_log(double x)
{
    if (x != x) return x;
    if (x < 0.0) return u2d(0x7FF8000000000000ull);
    if (x == 0.0) return u2d(0xFFF0000000000000ull);
    uint64_t u = d2u(x);
    if (u == 0x7FF0000000000000ull) return x;
    int e = 0;
    if ((u >> 52) == 0) { /* subnormal: scale by 2^54 (exact) */
        x = x * 18014398509481984.0;
        u = d2u(x);
        e = -54;
    }
    e += (int)(u >> 52) - 1023;
    double m = u2d((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > DET_SQRT2) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    /* atanh series: log(m) = 2s (1 + z/3 + z^2/5 + ... + z^11/23) */
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
    return de * DET_LN2_HI + (lm + de * DET_LN2_LO);
}

double orc_det_exp(double x)
{
    if (x != x) return x;
    if (x > 709.782712893384) return u2d(0x7FF0000000000000ull);
    if (x < -745.2) return 0.0;
    double t = x * DET_INV_LN2;
    long long k = (long long)(t + (t < 0.0 ? -0.5 : 0.5)); /* round half away */
    double dk = (double)k;
    double r = (x - dk * DET_LN2_HI) - dk * DET_LN2_LO;
    /* Taylor to r^13, Horner; |r| <= 0.347 */
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
    /* scale by 2^k in two exact steps so subnormal results round once */
    long long k1 = k / 2, k2 = k - k1;
    double s1 = u2d((uint64_t)(k1 + 1023) << 52);
    double s2 = u2d((uint64_t)(k2 + 1023) << 52);
    return (p * s1) * s2;
}

/* ------------------------------------------------------------------------- */
/* Wave-order reductions.  The only order-sensitive f64 reductions on the     */
/* path are mirrored between CPU and GPU as: 64 lane partials, then an xor    */
/* butterfly with offsets 1,2,4,8,16,32.  (The reference's order is the heap  */
/* address order of std::set/std::map, i.e. arbitrary.)                        */
/* ------------------------------------------------------------------------- */
static double wave_sum(const double *p)
{
    double a[ORC_NLANE], b[ORC_NLANE];
    memcpy(a, p, sizeof a);
    for (int off = 1; off < ORC_NLANE; off <<= 1) {
        for (int l = 0; l < ORC_NLANE; ++l) b[l] = a[l] + a[l ^ off];
        memcpy(a, b, sizeof a);
    }
    return a[0];
}

static double wave_prod(const double *p)
{
    double a[ORC_NLANE], b[ORC_NLANE];
    memcpy(a, p, sizeof a);
    for (int off = 1; off < ORC_NLANE; off <<= 1) {
        for (int l = 0; l < ORC_NLANE; ++l) b[l] = a[l] * a[l ^ off];
        memcpy(a, b, sizeof a);
    }
    return a[0];
}

/* ------------------------------------------------------------------------- */
/* A1/A2: moments.  createSuperPixels fills A_ rows [u v 1] and b_ with the   */
/* short disparity (OverSegmentation.cpp:145-162); solve() forms the weighted */
/* normal equations (MyMath.h:120-154).  Sufficient statistics, exact int64:  */
/*   N, Su, Sv, Suu, Suv, Svv, Sb, Sub, Svb, Sbb.                              */
/* ------------------------------------------------------------------------- */
int orc_moments(const int16_t *disp, const uint16_t *label, int W, int H, int S, int64_t *mom)
{
    memset(mom, 0, sizeof(int64_t) * (size_t)S * ORC_MOM_STRIDE);
    for (int v = 0; v < H; ++v) {
        for (int u = 0; u < W; ++u) {
            size_t k = (size_t)v * W + u;
            unsigned s = label[k];
            if (s >= (unsigned)S) continue;
            int64_t b = disp[k];
            int64_t *m = mom + (size_t)s * ORC_MOM_STRIDE;
            m[0] += 1; m[1] += u; m[2] += v;
            m[3] += (int64_t)u * u; m[4] += (int64_t)u * v; m[5] += (int64_t)v * v;
            m[6] += b; m[7] += u * b; m[8] += v * b; m[9] += b * b;
        }
    }
    return ORC_OK;
}

/* 3x3 SPD solve by LDL^T without pivoting, fixed evaluation order.  Stands   */
/* in for gtsam's QR elimination (MyMath.h:141-143, 238-240); same solution   */
/* of the same normal equations.  Returns pivots in d[3] and r' A^-1 r from    */
/* the factorisation (sum y_k^2 / d_k), which is how the residual of the fit   */
/* is obtained without a pass over the data: |A x - b|^2 = b'b - r' A^-1 r.    */
static double ldl3(const double a[6] /* a11 a12 a22 a13 a23 a33 */, const double r[3],
                   double x[3], double d[3])
{
    double a11 = a[0], a12 = a[1], a22 = a[2], a13 = a[3], a23 = a[4], a33 = a[5];
    double d1 = a11;
    double l21 = a12 / d1;
    double l31 = a13 / d1;
    double d2 = a22 - l21 * a12;
    double t32 = a23 - l31 * a12;
    double l32 = t32 / d2;
    double d3 = (a33 - l31 * a13) - l32 * t32;
    double y1 = r[0];
    double y2 = r[1] - l21 * y1;
    double y3 = (r[2] - l31 * y1) - l32 * y2;
    double z1 = y1 / d1, z2 = y2 / d2, z3 = y3 / d3;
    double x3 = z3;
    double x2 = z2 - l32 * x3;
    double x1 = (z1 - l21 * x2) - l31 * x3;
    x[0] = x1; x[1] = x2; x[2] = x3;
    d[0] = d1; d[1] = d2; d[2] = d3;
    return (y1 * z1 + y2 * z2) + y3 * z3;
}

/*
 * A2 local fit (MyMath.h:120-170) + A1 filter (OverSegmentation.cpp:198-222,
 * 422-455).  Coordinates are shifted to the frame centre (cx=W/2, cy=H/2) in
 * exact integer arithmetic before going to double: energies are invariant
 * under the shift (unimodular change of basis), conditioning improves ~1e5x.
 * fit record (16 doubles): Suu Suv Svv Su Sv N | Sub Svb Sb | theta_a theta_b theta_c |
 * Sbb ssr 0 0.  theta is in centred coordinates: disparity = a*(u-cx) + b*(v-cy) + c.
 * ssr = |A theta - b|^2 = Sbb - r' A^-1 r (from the factorisation);
 * local_err = ssr / (2 var) (MyMath.h:146-150).
 */
int orc_local_fit(const int64_t *mom, int W, int H, int S, double var,
                  double *fit, uint8_t *valid, double *local_err)
{
    int64_t cx = W / 2, cy = H / 2;
    for (int s = 0; s < S; ++s) {
        const int64_t *m = mom + (size_t)s * ORC_MOM_STRIDE;
        double *f = fit + (size_t)s * ORC_FIT_STRIDE;
        int64_t n = m[0];
        int64_t su = m[1] - cx * n, sv = m[2] - cy * n;
        int64_t suu = m[3] - 2 * cx * m[1] + cx * cx * n;
        int64_t suv = m[4] - cx * m[2] - cy * m[1] + cx * cy * n;
        int64_t svv = m[5] - 2 * cy * m[2] + cy * cy * n;
        int64_t sb = m[6];
        int64_t sub = m[7] - cx * m[6];
        int64_t svb = m[8] - cy * m[6];
        /* isLinear: all u equal or all v equal <=> N*Suu == Su^2 (integers) */
        __int128 du = (__int128)n * suu - (__int128)su * su;
        __int128 dv = (__int128)n * svv - (__int128)sv * sv;
        /* isZeroDisparity (OverSegmentation.cpp:422-432): dropped only if NO entry is >= 0.5.  Entries are shorts
         * widened through size_t (OverSegmentation.cpp:148), so any non-zero value -- a negative one becomes
         * ~1.8e19 -- keeps the superpixel: "some b != 0" <=> Sbb != 0 (sum of squares; signs cannot cancel). */
        int ok = (n >= 10) && (du != 0) && (dv != 0) && (m[9] != 0);
        valid[s] = (uint8_t)ok;
        f[0] = (double)suu; f[1] = (double)suv; f[2] = (double)svv;
        f[3] = (double)su;  f[4] = (double)sv;  f[5] = (double)n;
        f[6] = (double)sub; f[7] = (double)svb; f[8] = (double)sb;
        if (ok) {
            double a[6] = { f[0], f[1], f[2], f[3], f[4], f[5] };
            double r[3] = { f[6], f[7], f[8] };
            double x[3], d[3];
            double quad = ldl3(a, r, x, d);
            f[9] = x[0]; f[10] = x[1]; f[11] = x[2];
            f[12] = (double)m[9]; f[13] = f[12] - quad;
        } else {
            f[9] = f[10] = f[11] = 0.0;
            f[12] = (double)m[9]; f[13] = 0.0;
        }
        f[14] = f[15] = 0.0;
        if (local_err) local_err[s] = f[13] / (2.0 * var);
    }
    return ORC_OK;
}

#define LOG_2PI 1.8378770664093453

/*
 * A3/A4: fit() (MyMath.h:189-275) + Segment::fitGlobalSurface/computeProbability
 * (Graph.cpp:69-138).  Information-form fusion: M = sum M_i, m = sum m_i (exact:
 * integer-valued doubles), theta = M^-1 m.  totalError is the sum of the member
 * factors' errors at the fused mean, sum_i 1/2 (theta-theta_i)' M_i (theta-theta_i) / var
 * (MyMath.h:247-252; the members' local error_ is NOT added, Graph.cpp:101).  Each
 * term is the member's residual at theta minus its own least-squares residual, so
 *     2 var * totalError = (Sbb - m' M^-1 m) - sum_i ssr_i
 * with Sbb = sum_i Sbb_i exact.  This form needs only sums over the members, which
 * is what lets a proposal be evaluated from the cached sums of the two segments and
 * the sums over the component: sB2 = sA2 + C, sB1 = sA1 - C.
 * probabilityL = -error + log sqrt det(2 pi Lambda^-1).
 * Quirk kept: NaN -> 0.75 (Graph.cpp:134); +-inf -> status (Graph.cpp:137).
 */
#define ORC_SUM_STRIDE 10   /* Suu Suv Svv Su Sv N Sub Svb Sb Sbb */
typedef struct {
    double probL, err, theta[3], logdet;
    double m[ORC_SUM_STRIDE], R;  /* the sums the result was computed from */
    int n, minid, status;
} orc_segfit;

/* probabilityL from the segment's sums: m[10] exact, R = sum of member ssr */
static void seg_finish(const double *m, double R, double var, double log_var, orc_segfit *out)
{
    double x[3], d[3];
    double quad = ldl3(m, m + 6, x, d);
    double err = ((m[9] - quad) - R) / (2.0 * var);
    double logdet = ((orc_det_log(d[0]) + orc_det_log(d[1])) + orc_det_log(d[2])) - 3.0 * log_var;
    double p = -err + 0.5 * (3.0 * LOG_2PI - logdet);
    out->status = ORC_OK;
    if (p != p) p = 0.75;
    if (p - p != 0.0) out->status = ORC_E_INF_ENERGY;
    out->probL = p; out->err = err; out->logdet = logdet;
    out->theta[0] = x[0]; out->theta[1] = x[1]; out->theta[2] = x[2];
    for (int k = 0; k < ORC_SUM_STRIDE; ++k) out->m[k] = m[k];
    out->R = R;
}

static inline void rec_sums(const double *f, double *m)
{
    for (int k = 0; k < 9; ++k) m[k] += f[k];
    m[9] += f[12];
}

/* from scratch: mem[i]==which selects the members; R in wave order by id */
static void seg_eval(const double *fit, int S, const uint8_t *mem, uint8_t which,
                     double var, double log_var, orc_segfit *out)
{
    double m[ORC_SUM_STRIDE] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    double part[ORC_NLANE];
    for (int l = 0; l < ORC_NLANE; ++l) part[l] = 0.0;
    int n = 0, minid = -1;
    for (int i = 0; i < S; ++i) {
        if (mem[i] != which) continue;
        const double *f = fit + (size_t)i * ORC_FIT_STRIDE;
        rec_sums(f, m);
        part[i & (ORC_NLANE - 1)] += f[13];
        if (minid < 0) minid = i;
        n++;
    }
    memset(out, 0, sizeof *out);
    out->n = n; out->minid = minid; out->status = ORC_OK;
    if (n == 0) return;
    seg_finish(m, wave_sum(part), var, log_var, out);
    out->n = n; out->minid = minid;
}

/* public wrapper: members given as an id list */
int orc_segment_probability(const double *fit, int S, const uint32_t *members, int n, double var,
                            double *probL, double *theta3, double *err)
{
    uint8_t *mem = (uint8_t *)calloc((size_t)S, 1);
    for (int k = 0; k < n; ++k) {
        if (members[k] >= (uint32_t)S) { free(mem); return ORC_E_BADARG; }
        mem[members[k]] = 1;
    }
    orc_segfit r;
    seg_eval(fit, S, mem, 1, var, orc_det_log(var), &r);
    free(mem);
    if (probL) *probL = r.probL;
    if (err) *err = r.err;
    if (theta3) { theta3[0] = r.theta[0]; theta3[1] = r.theta[1]; theta3[2] = r.theta[2]; }
    return r.status;
}

/*
 * A11: Problem::computeEdgeProbability (Problem.cpp:420-445) with
 * createSegmentations (:372-418) and addLogProbs (MyMath.h:284-296).
 * q = exp(T - logaddexp(T, Sep)); T = probL({i,j}), Sep = probL({i})+probL({j}).
 * An edge with an endpoint that failed the A1 filter does not exist in the
 * reference's neighbor maps (OverSegmentation.cpp:205-214): marked q = -1.
 */
int orc_edge_probs(const double *fit, const uint8_t *valid, int S, const uint32_t *row_ptr,
                   const uint32_t *col, double var, double *q)
{
    double log_var = orc_det_log(var);
    uint8_t *mem = (uint8_t *)calloc((size_t)S, 1);
    int status = ORC_OK;
    for (int i = 0; i < S; ++i) {
        for (uint32_t e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
            uint32_t j = col[e];
            if (j >= (uint32_t)S || (int)j == i || !valid[i] || !valid[j]) { q[e] = -1.0; continue; }
            orc_segfit t, si, sj;
            mem[i] = 1; mem[j] = 1;
            seg_eval(fit, S, mem, 1, var, log_var, &t);
            mem[j] = 0;
            seg_eval(fit, S, mem, 1, var, log_var, &si);
            mem[i] = 0; mem[j] = 1;
            seg_eval(fit, S, mem, 1, var, log_var, &sj);
            mem[j] = 0;
            double T = t.probL;
            double Sp = si.probL + sj.probL; /* Segmentation::computeProbability, Segmentation.cpp:185-195 */
            double lB, lS;
            if (T > Sp) { lB = T; lS = Sp; } else { lS = T; lB = Sp; }
            double sum = lB + orc_det_log(1.0 + orc_det_exp(lS - lB));
            double ratio = orc_det_exp(T - sum);
            if (ratio != ratio) status = ORC_E_NAN_RATIO; /* Problem.cpp:443 */
            q[e] = ratio;
        }
    }
    free(mem);
    return status;
}

/* ------------------------------------------------------------------------- */
/* Sampler state.  Canonical order (the reference's is heap-pointer order):   */
/* superpixels by id; a segment is named by its smallest member id, so the    */
/* label array itself is canonical and "set order" of segments is ascending   */
/* label.                                                                      */
/* ------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

typedef struct {
    uint64_t hash;
    uint64_t count;
} orc_visit;

typedef struct orc_chains {
    /* frame (borrowed) */
    int S, Sv;
    const double *fit; const uint8_t *valid;
    const uint32_t *row_ptr, *col; const double *q;
    uint32_t *vlist;
    double var, log_var, new_seg_prob;
    /* chains */
    int n_chains; uint32_t seed, chain_offset;
    uint64_t step;       /* steps already taken by every chain */
    uint16_t *label;     /* n_chains*S, min-id labels */
    double *segP;        /* n_chains*S, probabilityL_ by root label */
    uint32_t *segN;      /* n_chains*S, sizes by root label */
    double *segM;        /* n_chains*S*10, the segment's exact sums by root label */
    double *segR;        /* n_chains*S, sum of the members' ssr by root label */
    double *energy;      /* running sum of probabilityL_ (Segmentation::computeProbability) */
    double *best_energy; uint16_t *best_label;
    uint64_t *state_hash, *trace_hash;
    uint64_t *n_accept, *n_case; /* n_case: 3 per chain */
    int *status;
    /* Statistics (Statistics.h:33-47): visit counts keyed by state hash */
    int hcap; orc_visit *visits; int *n_visits; uint64_t *overflow;
    uint64_t *top_count; uint64_t *top_hash; uint16_t *top_label;
    double *last_targetL, *last_prop; /* Problem::propose's out-parameters of the last step (Problem.cpp:213-217) */
    /* marginals of first-appearance labels, summed over chains */
    int lmax; uint64_t *marg; /* S*lmax */
    /* scratch */
    uint8_t *inC, *mem, *nbr; uint32_t *queue;
} orc_chains;

static uint64_t state_hash_of(const uint16_t *label, const uint8_t *valid, int S)
{
    uint64_t h = 0;
    for (int i = 0; i < S; ++i)
        if (valid[i]) h += mix64(((uint64_t)i << 32) | label[i]);
    return h;
}

/* Segmentation::computeHash's first-appearance relabel (Segmentation.cpp:197-233):
 * rank of each root among roots, in id order.  Invalid superpixels -> 0xFFFF. */
static void canonical_labels(const uint16_t *label, const uint8_t *valid, int S, uint16_t *out)
{
    uint16_t *rank = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)S);
    uint16_t r = 0;
    for (int i = 0; i < S; ++i) rank[i] = (valid[i] && label[i] == i) ? r++ : 0;
    for (int i = 0; i < S; ++i) out[i] = valid[i] ? rank[label[i]] : (uint16_t)ORC_NOLABEL;
    free(rank);
}

void orc_chains_free(orc_chains *c)
{
    if (!c) return;
    free(c->vlist); free(c->label); free(c->segP); free(c->segN); free(c->segM); free(c->segR); free(c->energy);
    free(c->best_energy); free(c->best_label); free(c->state_hash); free(c->trace_hash);
    free(c->n_accept); free(c->n_case); free(c->status); free(c->visits); free(c->n_visits);
    free(c->overflow); free(c->top_count); free(c->top_hash); free(c->top_label); free(c->marg);
    free(c->inC); free(c->mem); free(c->nbr); free(c->queue);
    free(c->last_targetL); free(c->last_prop);
    free(c);
}

/* Problem::initializeState (Problem.cpp:447-473): every superpixel in ONE segment. */
orc_chains *orc_chains_create(int S, const double *fit, const uint8_t *valid,
                              const uint32_t *row_ptr, const uint32_t *col, const double *q,
                              double var, double new_seg_prob, int n_chains, uint32_t seed,
                              uint32_t chain_offset, int hcap, int lmax)
{
    orc_chains *c = (orc_chains *)calloc(1, sizeof *c);
    c->S = S; c->fit = fit; c->valid = valid; c->row_ptr = row_ptr; c->col = col; c->q = q;
    c->var = var; c->log_var = orc_det_log(var); c->new_seg_prob = new_seg_prob;
    c->n_chains = n_chains; c->seed = seed; c->chain_offset = chain_offset;
    c->hcap = hcap; c->lmax = lmax;
    c->vlist = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(S > 0 ? S : 1));
    for (int i = 0; i < S; ++i) if (valid[i]) c->vlist[c->Sv++] = (uint32_t)i;
    size_t NS = (size_t)n_chains * (size_t)S;
    c->label = (uint16_t *)malloc(sizeof(uint16_t) * NS);
    c->segP = (double *)calloc(NS, sizeof(double));
    c->segN = (uint32_t *)calloc(NS, sizeof(uint32_t));
    c->segM = (double *)calloc(NS * ORC_SUM_STRIDE, sizeof(double));
    c->segR = (double *)calloc(NS, sizeof(double));
    c->energy = (double *)calloc((size_t)n_chains, sizeof(double));
    c->best_energy = (double *)calloc((size_t)n_chains, sizeof(double));
    c->best_label = (uint16_t *)malloc(sizeof(uint16_t) * NS);
    c->state_hash = (uint64_t *)calloc((size_t)n_chains, 8);
    c->trace_hash = (uint64_t *)calloc((size_t)n_chains, 8);
    c->n_accept = (uint64_t *)calloc((size_t)n_chains, 8);
    c->n_case = (uint64_t *)calloc((size_t)n_chains * 3, 8);
    c->status = (int *)calloc((size_t)n_chains, sizeof(int));
    c->visits = (orc_visit *)calloc((size_t)n_chains * (size_t)hcap, sizeof(orc_visit));
    c->n_visits = (int *)calloc((size_t)n_chains, sizeof(int));
    c->overflow = (uint64_t *)calloc((size_t)n_chains, 8);
    c->top_count = (uint64_t *)calloc((size_t)n_chains, 8);
    c->top_hash = (uint64_t *)calloc((size_t)n_chains, 8);
    c->top_label = (uint16_t *)malloc(sizeof(uint16_t) * NS);
    c->marg = (uint64_t *)calloc((size_t)S * (size_t)lmax, 8);
    c->last_targetL = (double *)calloc((size_t)n_chains, 8);
    c->last_prop = (double *)malloc((size_t)n_chains * 8);
    for (int k = 0; k < n_chains; ++k) c->last_prop[k] = 1.0;
    c->inC = (uint8_t *)calloc((size_t)S, 1);
    c->mem = (uint8_t *)calloc((size_t)S, 1);
    c->nbr = (uint8_t *)calloc((size_t)S, 1);
    c->queue = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(S > 0 ? S : 1));

    /* one segment holding every valid superpixel, named by the smallest id */
    uint16_t root = c->Sv ? (uint16_t)c->vlist[0] : 0;
    for (int i = 0; i < S; ++i) { c->mem[i] = valid[i] ? 1 : 0; }
    orc_segfit f0;
    seg_eval(fit, S, c->mem, 1, var, c->log_var, &f0);
    memset(c->mem, 0, (size_t)S);
    for (int k = 0; k < n_chains; ++k) {
        uint16_t *lab = c->label + (size_t)k * S;
        for (int i = 0; i < S; ++i) lab[i] = valid[i] ? root : (uint16_t)ORC_NOLABEL;
        if (c->Sv) {
            c->segP[(size_t)k * S + root] = f0.probL;
            c->segN[(size_t)k * S + root] = (uint32_t)c->Sv;
            memcpy(c->segM + ((size_t)k * S + root) * ORC_SUM_STRIDE, f0.m, sizeof f0.m);
            c->segR[(size_t)k * S + root] = f0.R;
        }
        c->energy[k] = f0.probL;
        c->best_energy[k] = f0.probL;
        memcpy(c->best_label + (size_t)k * S, lab, sizeof(uint16_t) * (size_t)S);
        memcpy(c->top_label + (size_t)k * S, lab, sizeof(uint16_t) * (size_t)S);
        c->state_hash[k] = state_hash_of(lab, valid, S);
        c->status[k] = f0.status;
    }
    return c;
}

/* Statistics::update (Statistics.h:33-47), bounded to hcap distinct states. */
static void visit_update(orc_chains *c, int k)
{
    orc_visit *tab = c->visits + (size_t)k * c->hcap;
    uint64_t h = c->state_hash[k];
    int n = c->n_visits[k];
    uint64_t cnt = 0;
    int found = 0;
    for (int i = 0; i < n; ++i)
        if (tab[i].hash == h) { cnt = ++tab[i].count; found = 1; break; }
    if (!found) {
        if (n < c->hcap) { tab[n].hash = h; tab[n].count = 1; c->n_visits[k] = n + 1; cnt = 1; }
        else { c->overflow[k]++; return; }
    }
    /* "best" = most visited (can05_sampler.cpp:36); tie -> smaller hash */
    if (cnt > c->top_count[k] || (cnt == c->top_count[k] && h < c->top_hash[k])) {
        if (h != c->top_hash[k] || c->top_count[k] == 0)
            memcpy(c->top_label + (size_t)k * c->S, c->label + (size_t)k * c->S,
                   sizeof(uint16_t) * (size_t)c->S);
        c->top_count[k] = cnt; c->top_hash[k] = h;
    }
}

static void marg_update(orc_chains *c, int k)
{
    int S = c->S;
    uint16_t *canon = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)S);
    canonical_labels(c->label + (size_t)k * S, c->valid, S, canon);
    for (int i = 0; i < S; ++i) {
        if (!c->valid[i]) continue;
        int b = canon[i] < c->lmax ? canon[i] : c->lmax - 1;
        c->marg[(size_t)i * c->lmax + b]++;
    }
    free(canon);
}

/*
 * One sampler step for chain k at global step t:
 * FastMetropolisHastings::iterate (MCMC.h:207-238) -> Problem::propose
 * (Problem.cpp:116-224) -> Segmentation::createComponent / selectSegment
 * (Segmentation.cpp:59-149, 10-57) -> Segment ctor fits -> jumpProbability
 * (Problem.cpp:475-487) -> jump (Problem.cpp:32-74).
 * Returns trace byte: bit0 accept, bits1-2 case (0=A back to own segment,
 * 1=B split, 2=C merge/exchange).
 */
static uint8_t chain_step(orc_chains *c, int k, uint64_t t)
{
    const int S = c->S;
    uint16_t *lab = c->label + (size_t)k * S;
    double *segP = c->segP + (size_t)k * S;
    uint32_t *segN = c->segN + (size_t)k * S;
    orc_rng g;
    rng_begin(&g, c->seed, c->chain_offset + (uint32_t)k, t);

    /* pickSuperPixel (Segmentation.cpp:151-183): rand()%S then one unused draw */
    uint32_t start = c->vlist[rng_u32_at(&g, 0) % (uint32_t)c->Sv];
    const double u1 = rng_unif_at(&g, 2);        /* Segmentation.cpp:23 */
    const uint32_t idx_draw = rng_u32_at(&g, 3); /* Segmentation.cpp:40 (used only on that branch) */
    const double r_mh = rng_unif_at(&g, 4);      /* MCMC.h:223, always drawn */
    const uint16_t orig = lab[start];

    /* createComponent BFS (Segmentation.cpp:74-120) */
    /* The FIFO is counted in slots of at most 8 neighbours (a node with more takes consecutive
     * slots); the trial of neighbour s of slot k owns the uniform at position 8 + 8 k + s. */
    uint32_t qn = 0, slot = 0;
    c->queue[qn++] = start; c->inC[start] = 1;
    for (uint32_t head = 0; head < qn; ++head) {
        uint32_t i = c->queue[head];
        const uint32_t rs = c->row_ptr[i], deg = c->row_ptr[i + 1] - rs;
        for (uint32_t e = rs; e < rs + deg; ++e) {
            const uint32_t kk = e - rs;
            const uint32_t p = ORC_TRIAL_POS0 + 8u * (slot + (kk >> 3)) + (kk & 7u);
            double qe = c->q[e];
            if (qe < 0.0) continue;               /* not an edge in the reference graph */
            uint32_t j = c->col[e];
            if (c->inC[j]) continue;              /* :94 already in component */
            if (lab[j] != orig) { c->nbr[lab[j]] = 1; continue; } /* :100-105 */
            double r = rng_unif_at(&g, p);        /* :109 */
            if (r < qe) { c->inC[j] = 1; c->queue[qn++] = j; } /* :110-117, strict */
        }
        slot += deg <= 8u ? 1u : (deg + 7u) >> 3;
    }
    const uint32_t csize = qn;
    const int same = (csize == segN[orig]);      /* Segmentation.cpp:19 */

    /* selectSegment (Segmentation.cpp:10-57) */
    int kase; uint16_t next = orig; double ratio;
    uint32_t nn = 0;
    for (int i = 0; i < S; ++i) nn += c->nbr[i];
    if (!same && u1 < c->new_seg_prob) {         /* :25 */
        kase = 1; ratio = c->new_seg_prob;       /* :28-29 */
    } else {
        double existing = same ? 1.0 : (1.0 - c->new_seg_prob); /* :36 */
        uint32_t numSeg = nn + 1;                /* :39 */
        uint32_t idx = idx_draw % numSeg;        /* :40 */
        if (idx == numSeg - 1) { next = orig; kase = 0; } /* :43-45 */
        else {
            uint32_t seen = 0; kase = 2;
            for (int i = 0; i < S; ++i)
                if (c->nbr[i]) { if (seen == idx) { next = (uint16_t)i; break; } seen++; }
        }
        ratio = existing / (double)numSeg;       /* :55 */
    }

    double targetL, prop;
    orc_segfit fB1, fB2;
    memset(&fB1, 0, sizeof fB1); memset(&fB2, 0, sizeof fB2);
    fB1.minid = -1; fB2.minid = -1;
    int st = ORC_OK;
    const int have1 = !same;                     /* sB1 = sA1 \ C is not empty */
    if (kase == 0) {
        /* Case A (Problem.cpp:161-171): targetRatio = 1.0 (sic, it is used as a log) */
        prop = ratio; targetL = 1.0;
    } else {
        /* separation / merge products (Segmentation.cpp:127-146, Graph.cpp:10-33) and the
         * wave order over the stream of examined entries, and the component's sums, wave order over
         * the BFS queue */
        double psep[ORC_NLANE], pmrg[ORC_NLANE], rpart[ORC_NLANE];
        double Mc[ORC_SUM_STRIDE] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        int minC = S;
        for (int l = 0; l < ORC_NLANE; ++l) { psep[l] = 1.0; pmrg[l] = 1.0; rpart[l] = 0.0; }
        uint32_t slot = 0;                       /* FIFO slot of the node's first 8 neighbours */
        for (uint32_t h = 0; h < qn; ++h) {
            uint32_t i = c->queue[h];
            const double *f = c->fit + (size_t)i * ORC_FIT_STRIDE;
            const uint32_t rs = c->row_ptr[i], deg = c->row_ptr[i + 1] - rs;
            rec_sums(f, Mc);
            rpart[slot & (ORC_NLANE - 1)] += f[13];
            if ((int)i < minC) minC = (int)i;
            for (uint32_t e = rs; e < rs + deg; ++e) {
                const uint32_t kk = e - rs;
                int l = (int)((8u * (slot + (kk >> 3)) + (kk & 7u)) & (ORC_NLANE - 1));
                double qe = c->q[e];
                if (qe < 0.0) continue;
                uint32_t j = c->col[e];
                if (lab[j] == orig) { if (!c->inC[j]) psep[l] = psep[l] * (1.0 - qe); }
                else if (kase == 2 && lab[j] == next) pmrg[l] = pmrg[l] * (1.0 - qe);
            }
            slot += deg <= 8u ? 1u : (deg + 7u) >> 3;
        }
        double sep = wave_prod(psep);
        const double Rc = wave_sum(rpart);
        /* Segment::combine / diff (Graph.cpp:35-67) on the cached sums instead of a refit from the
         * member list: sB2 = sA2 + C (or C alone for a new segment), sB1 = sA1 - C */
        double M2[ORC_SUM_STRIDE], M1[ORC_SUM_STRIDE], R2, R1 = 0.0;
        const double *Mo = c->segM + ((size_t)k * S + orig) * ORC_SUM_STRIDE;
        const double *Mn = c->segM + ((size_t)k * S + next) * ORC_SUM_STRIDE;
        for (int v = 0; v < ORC_SUM_STRIDE; ++v) {
            M2[v] = kase == 2 ? Mn[v] + Mc[v] : Mc[v];
            M1[v] = Mo[v] - Mc[v];
        }
        R2 = kase == 2 ? c->segR[(size_t)k * S + next] + Rc : Rc;
        seg_finish(M2, R2, c->var, c->log_var, &fB2);
        fB2.n = (int)csize + (kase == 2 ? (int)segN[next] : 0);
        fB2.minid = (kase == 2 && (int)next < minC) ? (int)next : minC;
        if (have1) {
            R1 = c->segR[(size_t)k * S + orig] - Rc;
            seg_finish(M1, R1, c->var, c->log_var, &fB1);
            fB1.n = (int)segN[orig] - (int)csize;
        }
        if (kase == 1) {
            prop = ratio * (1.0 / sep);          /* Problem.cpp:188 */
        } else {
            double mrg = wave_prod(pmrg);
            prop = ratio * (mrg / sep);          /* Problem.cpp:206 */
        }
        /* jumpProbability (Problem.cpp:475-487) */
        double sB2_sA1 = fB2.probL - segP[orig];
        double sB1_sA2 = (have1 ? fB1.probL : 0.0) - (kase == 2 ? segP[next] : 0.0);
        targetL = sB2_sA1 + sB1_sA2;
        if (fB1.status) st = fB1.status;
        if (fB2.status) st = fB2.status;
        if (targetL != targetL || prop != prop) st = ORC_E_NAN_RATIO; /* Problem.cpp:216-217 */
    }

    c->last_targetL[k] = targetL; c->last_prop[k] = prop;

    /* FastMetropolisHastings::iterate (MCMC.h:218-227) */
    double aL = targetL + orc_det_log(prop);
    double a = orc_det_exp(aL);
    const double r = r_mh;
    int accept = 1;
    if (aL < 0.0) { if (r > a) accept = 0; }
    if (st != ORC_OK) { accept = 0; c->status[k] = st; }

    uint16_t newlab = next;
    if (accept && kase != 0) {
        /* jump (Problem.cpp:32-74) with min-id naming */
        uint16_t L2 = (uint16_t)fB2.minid, L1 = 0;
        if (have1) {
            /* smallest id left in the old segment */
            int m1 = orig;
            if (c->inC[orig]) { m1 = -1; for (int i = 0; i < S && m1 < 0; ++i) if (lab[i] == orig && !c->inC[i]) m1 = i; }
            fB1.minid = m1; L1 = (uint16_t)m1;
        }
        for (int i = 0; i < S; ++i) {
            if (!c->valid[i]) continue;
            if (c->inC[i] || (kase == 2 && lab[i] == next)) lab[i] = L2;
            else if (have1 && lab[i] == orig) lab[i] = L1;
        }
        segP[L2] = fB2.probL; segN[L2] = (uint32_t)fB2.n;
        memcpy(c->segM + ((size_t)k * S + L2) * ORC_SUM_STRIDE, fB2.m, sizeof fB2.m);
        c->segR[(size_t)k * S + L2] = fB2.R;
        if (have1) {
            segP[L1] = fB1.probL; segN[L1] = (uint32_t)fB1.n;
            memcpy(c->segM + ((size_t)k * S + L1) * ORC_SUM_STRIDE, fB1.m, sizeof fB1.m);
            c->segR[(size_t)k * S + L1] = fB1.R;
        }
        c->energy[k] += targetL;
        c->state_hash[k] = state_hash_of(lab, c->valid, S);
        if (c->energy[k] > c->best_energy[k]) {
            c->best_energy[k] = c->energy[k];
            memcpy(c->best_label + (size_t)k * S, lab, sizeof(uint16_t) * (size_t)S);
        }
        newlab = L2;
    }
    if (accept) c->n_accept[k]++;
    c->n_case[(size_t)k * 3 + kase]++;

    /* clear scratch */
    for (uint32_t h = 0; h < qn; ++h) c->inC[c->queue[h]] = 0;
    memset(c->nbr, 0, (size_t)S);

    uint8_t tb = (uint8_t)(accept | (kase << 1));
    uint64_t w = (uint64_t)tb | ((uint64_t)start << 8) | ((uint64_t)csize << 24) | ((uint64_t)newlab << 40);
    c->trace_hash[k] = mix64(mix64(c->trace_hash[k] ^ w) ^ c->state_hash[k]);
    return tb;
}

/*
 * MCMC<PROBLEM>::run (MCMC.h:43-89): burn_in steps without statistics, then
 * iters steps each followed by Statistics::update.  trace (optional) receives
 * one byte per chain per step, chain-major: trace[k*(burn_in+iters) + s].
 * A chain whose status is non-zero is frozen (the reference would have thrown).
 */
int orc_chains_run(orc_chains *c, uint64_t burn_in, uint64_t iters, uint8_t *trace)
{
    if (c->Sv == 0) return ORC_E_BADARG;
    uint64_t total = burn_in + iters;
    int worst = ORC_OK;
    for (int k = 0; k < c->n_chains; ++k) {
        for (uint64_t s = 0; s < total; ++s) {
            uint8_t tb = 0xFF;
            if (c->status[k] == ORC_OK) tb = chain_step(c, k, c->step + s);
            if (trace) trace[(size_t)k * total + s] = tb;
            if (s >= burn_in && c->status[k] == ORC_OK) { visit_update(c, k); marg_update(c, k); }
        }
        if (c->status[k] != ORC_OK) worst = c->status[k];
    }
    c->step += total;
    return worst;
}

/* getters ------------------------------------------------------------------ */
int orc_chains_get_labels(const orc_chains *c, int which /*0 cur,1 best-energy,2 most-visited*/,
                          int canonical, uint16_t *out)
{
    const uint16_t *src = which == 0 ? c->label : (which == 1 ? c->best_label : c->top_label);
    for (int k = 0; k < c->n_chains; ++k) {
        if (canonical) canonical_labels(src + (size_t)k * c->S, c->valid, c->S, out + (size_t)k * c->S);
        else memcpy(out + (size_t)k * c->S, src + (size_t)k * c->S, sizeof(uint16_t) * (size_t)c->S);
    }
    return ORC_OK;
}
void orc_chains_get_scalars(const orc_chains *c, double *energy, double *best_energy,
                            uint64_t *state_hash, uint64_t *trace_hash, uint64_t *n_accept,
                            uint64_t *n_case, int *status, uint64_t *top_count, uint64_t *top_hash,
                            uint64_t *overflow)
{
    size_t n = (size_t)c->n_chains;
    if (energy) memcpy(energy, c->energy, n * 8);
    if (best_energy) memcpy(best_energy, c->best_energy, n * 8);
    if (state_hash) memcpy(state_hash, c->state_hash, n * 8);
    if (trace_hash) memcpy(trace_hash, c->trace_hash, n * 8);
    if (n_accept) memcpy(n_accept, c->n_accept, n * 8);
    if (n_case) memcpy(n_case, c->n_case, n * 24);
    if (status) memcpy(status, c->status, n * sizeof(int));
    if (top_count) memcpy(top_count, c->top_count, n * 8);
    if (top_hash) memcpy(top_hash, c->top_hash, n * 8);
    if (overflow) memcpy(overflow, c->overflow, n * 8);
}
/* visits: out_hash/out_count sized n_chains*hcap; n_out per chain */
void orc_chains_get_visits(const orc_chains *c, uint64_t *out_hash, uint64_t *out_count, int *n_out)
{
    for (int k = 0; k < c->n_chains; ++k) {
        n_out[k] = c->n_visits[k];
        for (int i = 0; i < c->hcap; ++i) {
            out_hash[(size_t)k * c->hcap + i] = c->visits[(size_t)k * c->hcap + i].hash;
            out_count[(size_t)k * c->hcap + i] = c->visits[(size_t)k * c->hcap + i].count;
        }
    }
}
void orc_chains_get_last_ratios(const orc_chains *c, double *targetL, double *prop)
{
    memcpy(targetL, c->last_targetL, (size_t)c->n_chains * 8);
    memcpy(prop, c->last_prop, (size_t)c->n_chains * 8);
}
void orc_chains_get_marginals(const orc_chains *c, uint64_t *out)
{
    memcpy(out, c->marg, sizeof(uint64_t) * (size_t)c->S * (size_t)c->lmax);
}
/* Segmentation::computeProbability (Segmentation.cpp:185-195) recomputed from
 * scratch for the current labels (wave order over roots by id). */
void orc_chains_recompute_energy(const orc_chains *c, double *out)
{
    for (int k = 0; k < c->n_chains; ++k) {
        const uint16_t *lab = c->label + (size_t)k * c->S;
        double part[ORC_NLANE];
        for (int l = 0; l < ORC_NLANE; ++l) part[l] = 0.0;
        for (int i = 0; i < c->S; ++i)
            if (c->valid[i] && lab[i] == i) part[i & (ORC_NLANE - 1)] += c->segP[(size_t)k * c->S + i];
        out[k] = wave_sum(part);
    }
}
uint64_t orc_mix64(uint64_t x) { return mix64(x); }
double orc_wave_sum(const double *p) { return wave_sum(p); }
double orc_wave_prod(const double *p) { return wave_prod(p); }
