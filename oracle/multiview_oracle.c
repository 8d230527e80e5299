/*
 * multiview_oracle.c -- ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, f64, no FMA contraction; the one explicit fma chain is vertex_error's) of Path B of a-price/MCMC,
 * the "Price multiview" proposal evaluation:
 *
 *   B1  MathUtils::fitPlaneSVD            src/libraries/MathUtils.cpp:22-54
 *   B2  SPNode::getErrorForPlaneModel     src/libraries/SPGraph.cpp:89-104
 *   B3  MultiviewSegment::computeFitPlane / computeProbability
 *                                         src/libraries/MultiviewSegment.cpp:33-96
 *   B4  MultiviewSegmentation ctor, getNewConnectedSet, getNeighborSegments,
 *       addNewSegment, moveSuperpixels, computeProposalRatio
 *                                         src/libraries/MultiviewSegmentation.cpp:11-274
 *       and the step loop                 src/exe/PatternedMatcher.cpp:58-130
 *       pMerge                            src/libraries/GraphUtils.cpp:11-26
 *
 * Nothing under mcmc_amd/ may include, link or call this file.
 *
 * PARITY PINNING.  The reference cannot be built here (Eigen, PCL, Boost absent)
 * and holds no golden vectors for this path; its only tests are property checks
 * (TestMathUtils.cpp:57-58: unit normal, direction; TestGraphUtils.cpp:64-70:
 * pMerge identity and symmetry), which tests/test_multiview.py repeats.  The
 * plane fit and probability are pinned against numpy's f64 SVD of the centred
 * cloud (the reference's formula, in f64) to 1e-6 relative.  The sampler loop is
 * "parity unpinned": the reference seeds rand()/mt19937 from time().
 *
 * Restated behaviour, including what the unfinished reference actually does:
 *  - plane normal = left singular vector of the smallest singular value of the
 *    centred 3xN cloud == eigenvector of the smallest eigenvalue of the scatter
 *    matrix (computed here from pooled moments N, sum p, sum p p^T in f64; the
 *    reference does a float32 JacobiSVD).  coefficients[3] = + n . centroid (the
 *    sign is inconsistent with the +d distance formula; kept).
 *    covariance = scatter / 2, so det(cov) = l0 l1 l2 / 8.
 *  - segment error = sum over member vertices of the MEAN |n.p + d| of the
 *    vertex's points; P = exp(-err/2) * sqrt(2 pi det(cov)); NaN -> 0.
 *  - the node's currentSegmentID is never updated by the loop (only the proposed
 *    state is written), so component growth and neighbour lookup always see the
 *    INITIAL partition, while a split proposal (moveType < 0.10) really moves
 *    the vertices between the segments' vertex sets and appends a segment,
 *    accepted or not.  A transfer proposal changes nothing.  The accept decision
 *    u < a is computed and only reported (PatternedMatcher.cpp:111-121).
 *  - a segment that lost vertices in a split keeps its old plane (not refitted).
 *
 * Canonical choices where the reference is arbitrary or uses another RNG:
 * Philox4x32-10 keyed (seed, global chain id), the trial of an undirected edge tied to the edge (draw
 * 16 + rank) instead of to the order of the walk; out-edges ascending by target id
 * (boost setS out-edge lists are ordered by target); the initial Bernoulli pass
 * visits undirected edges in (i < j) lexicographic order; neighbour segments in
 * ascending id (std::set<MultiviewSegment*> of vector elements = index order);
 * f64 instead of float / x87 long double; order-sensitive sums in wave order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NLANE 64
#define MV_MOM 10           /* N Sx Sy Sz Sxx Sxy Sxz Syy Syz Szz */
#define MV_PLANE 5          /* nx ny nz d det(cov) */
#define PHILOX_TAG_MV 0x4D564D43u /* "MVMC" */
#define MV_INIT_STEP 0xFFFFFFFFFFFFFFFFull

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_det_exp(double x);
double orc_wave_sum(const double *p);
uint64_t orc_mix64(uint64_t x);

static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* sqrt from IEEE + * / only (Newton), identical on CPU and GPU. */
double orc_det_sqrt(double x)
{
    if (x != x) return x;
    if (x < 0.0) return u2d(0x7FF8000000000000ull);
    if (x == 0.0) return x;
    uint64_t u = d2u(x);
    if (u == 0x7FF0000000000000ull) return x;
    int e = 0;
    if ((u >> 52) == 0) { x = x * 18014398509481984.0 * 18014398509481984.0; u = d2u(x); e = -108; }
    e += (int)(u >> 52) - 1023;
    int odd = e & 1;                 /* works for negative e in two's complement */
    e -= odd;
    double m = u2d((u & 0x000FFFFFFFFFFFFFull) | ((uint64_t)(1023 + odd) << 52)); /* [1, 4) */
    double y = (m + 2.0) / 3.0;
    for (int i = 0; i < 6; ++i) y = 0.5 * (y + m / y);
    return y * u2d((uint64_t)(e / 2 + 1023) << 52);
}

/* draw d of a step: word (d>>6)&3 of Philox block (d>>8)*64 + (d&63) */
typedef struct { uint32_t key[2], lo, hi, nd; } mv_rng;
static uint32_t mv_u32_at(const mv_rng *g, uint32_t d)
{
    uint32_t ctr[4] = { g->lo, g->hi, (d >> 8) * 64u + (d & 63u), PHILOX_TAG_MV }, out[4];
    orc_philox4x32_10(ctr, g->key, out);
    return out[(d >> 6) & 3u];
}
/* MathUtils::randbetween(0,1) is rand()/RAND_MAX, inclusive of 1 (MathUtils.cpp:17-20): u32 / (2^32 - 1) */

/* ---- B1 inputs: per-vertex moments, wave order over the vertex's points ---- */
void orc_mv_vertex_moments(const float *pts /* 4 per point */, const uint32_t *pt_ptr, int V, double *out)
{
    for (int v = 0; v < V; ++v) {
        double part[MV_MOM][NLANE];
        memset(part, 0, sizeof part);
        for (uint32_t i = pt_ptr[v]; i < pt_ptr[v + 1]; ++i) {
            int l = (int)((i - pt_ptr[v]) & 63u);
            double x = pts[4 * i], y = pts[4 * i + 1], z = pts[4 * i + 2];
            part[0][l] += 1.0; part[1][l] += x; part[2][l] += y; part[3][l] += z;
            part[4][l] += x * x; part[5][l] += x * y; part[6][l] += x * z;
            part[7][l] += y * y; part[8][l] += y * z; part[9][l] += z * z;
        }
        for (int k = 0; k < MV_MOM; ++k) out[(size_t)v * MV_MOM + k] = orc_wave_sum(part[k]);
    }
}

/* 1 / sqrt(x) for a normal positive x from + - * alone (the rotations below need two square roots each, and a
 * division costs the device as much as this whole function): x = m 2^e, m in [1, 2); a quadratic seed for
 * 1 / sqrt(m), three Newton steps r <- r (1.5 - (m / 2) r^2), 1 / sqrt(2) for an odd e, the power of two.
 * A few units in the last place; every operation is one IEEE rounding, so the device returns the same bits. */
double orc_det_rsqrt(double x)
{
    uint64_t u;
    memcpy(&u, &x, 8);
    int e = (int)(u >> 52) - 1023;
    const int odd = e & 1;
    e -= odd;
    uint64_t mu = (u & 0x000FFFFFFFFFFFFFull) | (1023ull << 52);
    double m;
    memcpy(&m, &mu, 8);
    const double hm = 0.5 * m;
    double r = (0.14496475 * m - 0.72223657) * m + 1.57368075;
    r = r * (1.5 - hm * (r * r));
    r = r * (1.5 - hm * (r * r));
    r = r * (1.5 - hm * (r * r));
    if (odd) r = r * 0.70710678118654752440;
    uint64_t su = (uint64_t)(1023 - e / 2) << 52;
    double sc;
    memcpy(&sc, &su, 8);
    return r * sc;
}

/* Symmetric 3x3 eigen-decomposition: cyclic Jacobi, at most 8 sweeps.
 * A rotation is left out when its off-diagonal element is below rounding against the diagonal,
 * apq^2 <= 2^-106 |app aqq| (the relative criterion, which keeps the small eigenvalue of a flat cloud accurate);
 * the sweeps end with the first one that rotates nothing.  A rotation by an angle too small for t^2 to show in
 * 1 + t^2 (|theta| >= 2^27) takes t = 1 / (2 theta) and c = 1, which is what the long form rounds to.
 * (The reference's Eigen::JacobiSVD iterates to convergence as well: MathUtils.cpp:35.) */
static void jacobi3(double A[3][3], double Vm[3][3])
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Vm[i][j] = i == j ? 1.0 : 0.0;
    static const int P[3] = { 0, 0, 1 }, Q[3] = { 1, 2, 2 };
    for (int sweep = 0; sweep < 8; ++sweep) {
        int rotated = 0;
        for (int r = 0; r < 3; ++r) {
            const int p = P[r], q = Q[r];
            const double apq = A[p][q];
            const double dd = A[p][p] * A[q][q];
            if (apq * apq <= 0x1p-106 * (dd < 0.0 ? -dd : dd)) continue;
            rotated = 1;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double at = theta < 0.0 ? -theta : theta;
            double t, c;
            if (at >= 0x1p27) { t = 1.0 / (2.0 * at); c = 1.0; }
            else {
                const double h2 = at * at + 1.0;
                t = 1.0 / (at + h2 * orc_det_rsqrt(h2));
                c = orc_det_rsqrt(t * t + 1.0);
            }
            if (theta < 0.0) t = -t;
            const double s = t * c;
            const int k = 3 - p - q;
            const double akp = A[k][p], akq = A[k][q];
            A[p][p] = A[p][p] - t * apq;
            A[q][q] = A[q][q] + t * apq;
            A[p][q] = 0.0; A[q][p] = 0.0;
            A[k][p] = c * akp - s * akq; A[p][k] = A[k][p];
            A[k][q] = s * akp + c * akq; A[q][k] = A[k][q];
            for (int i = 0; i < 3; ++i) {
                const double vip = Vm[i][p], viq = Vm[i][q];
                Vm[i][p] = c * vip - s * viq;
                Vm[i][q] = s * vip + c * viq;
            }
        }
        if (!rotated) break;
    }
}

/* fitPlaneSVD on pooled moments m[10]; plane = nx ny nz d det(cov) (MathUtils.cpp:22-54) */
static void plane_from_moments(const double *m, double *plane)
{
    const double N = m[0];
    const double cx = m[1] / N, cy = m[2] / N, cz = m[3] / N;
    double A[3][3], Vm[3][3];
    A[0][0] = m[4] - cx * m[1]; A[0][1] = m[5] - cx * m[2]; A[0][2] = m[6] - cx * m[3];
    A[1][1] = m[7] - cy * m[2]; A[1][2] = m[8] - cy * m[3]; A[2][2] = m[9] - cz * m[3];
    A[1][0] = A[0][1]; A[2][0] = A[0][2]; A[2][1] = A[1][2];
    jacobi3(A, Vm);
    int k = 0;
    if (A[1][1] < A[k][k]) k = 1;
    if (A[2][2] < A[k][k]) k = 2;
    plane[0] = Vm[0][k]; plane[1] = Vm[1][k]; plane[2] = Vm[2][k];
    plane[3] = (plane[0] * cx + plane[1] * cy) + plane[2] * cz;   /* + n.centroid (sic) */
    plane[4] = ((A[0][0] * A[1][1]) * A[2][2]) / 8.0;              /* det(U S^2 U^T / 2) */
}

/* B2: mean |n.p + d| over the vertex's points (SPGraph.cpp:89-104; the reference evaluates n.p + d in float and
 * adds the |.| in a double, point after point).
 * The residual of a point is the chain  fma(d, 1, fma(nz, z, fma(ny, y, fma(nx, x, 0)))) : products enter the sum
 * unrounded, one rounding per term, in this order -- which is bit for bit what the f64 matrix instruction of the
 * device (v_mfma_f64_16x16x4_f64, a row of 4 products added to C = 0) returns; measured with
 * tools/micro/mfma_f64_probe.hip against exact rational arithmetic, 6144 of 6144 results.
 * The |.| are added into 16 partial sums by point index mod 16 (ascending index), and the partials m = 4 r + g are
 * combined as a fixed tree: over r, then over g (the way the device holds them: a 16 x 16 block of the matrix
 * instruction puts rows 4 r + g of a column into register r of lane group g). */
static double vertex_tree_sum(const double *p /* 16 */)
{
    double s[4];
    for (int g = 0; g < 4; ++g) s[g] = (p[g] + p[4 + g]) + (p[8 + g] + p[12 + g]);
    return (s[0] + s[1]) + (s[2] + s[3]);
}

static double vertex_error(const float *pts, const uint32_t *pt_ptr, int v, const double *plane)
{
    double part[16];
    memset(part, 0, sizeof part);
    const uint32_t b = pt_ptr[v], e = pt_ptr[v + 1];
    for (uint32_t i = b; i < e; ++i) {
        double x = pts[4 * i], y = pts[4 * i + 1], z = pts[4 * i + 2];
        double d = fma(x, plane[0], 0.0);
        d = fma(y, plane[1], d);
        d = fma(z, plane[2], d);
        d = fma(1.0, plane[3], d);
        part[(i - b) & 15u] += fabs(d);
    }
    return vertex_tree_sum(part) / (double)(e - b);
}

typedef struct {
    int V; const float *pts; const uint32_t *pt_ptr; const double *vm;
    const uint32_t *rowp, *col; const double *q;
} mv_graph;

/* pooled moments of the vertices with mem[v] != 0, wave order by vertex id */
static int pool_moments(const mv_graph *g, const uint8_t *mem, double *m)
{
    double part[MV_MOM][NLANE];
    memset(part, 0, sizeof part);
    int n = 0;
    for (int v = 0; v < g->V; ++v) {
        if (!mem[v]) continue;
        n++;
        for (int k = 0; k < MV_MOM; ++k) part[k][v & 63] += g->vm[(size_t)v * MV_MOM + k];
    }
    for (int k = 0; k < MV_MOM; ++k) m[k] = orc_wave_sum(part[k]);
    return n;
}

/* B3 computeProbability (MultiviewSegment.cpp:71-96) */
static double set_probability(const mv_graph *g, const uint8_t *mem, const double *plane, double *err_out)
{
    double part[NLANE];
    memset(part, 0, sizeof part);
    for (int v = 0; v < g->V; ++v)
        if (mem[v]) part[v & 63] += vertex_error(g->pts, g->pt_ptr, v, plane);
    const double err = orc_wave_sum(part);
    if (err_out) *err_out = err;
    double p = orc_det_exp(-err / 2.0) * orc_det_sqrt((2.0 * 3.14159265358979323846) * plane[4]);
    if (p != p) p = 0.0;
    return p;
}

/* Segment(members): fit + probability for an explicit member list */
int orc_mv_fit_segment(int V, const float *pts, const uint32_t *pt_ptr, const double *vm,
                       const uint32_t *members, int n, double *plane5, double *prob, double *err)
{
    if (n <= 0) return -1;
    uint8_t *mem = calloc((size_t)V, 1);
    for (int k = 0; k < n; ++k) { if (members[k] >= (uint32_t)V) { free(mem); return -1; } mem[members[k]] = 1; }
    mv_graph g = { V, pts, pt_ptr, vm, NULL, NULL, NULL };
    double m[MV_MOM];
    pool_moments(&g, mem, m);
    plane_from_moments(m, plane5);
    *prob = set_probability(&g, mem, plane5, err);
    free(mem);
    return 0;
}

/* pMerge (GraphUtils.cpp:11-26): exp(-|w (.) |t2 - t1||^2 * T / 2), float inputs */
double orc_mv_pmerge(const float *t1, const float *t2, const float *w, int n, double temperature)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        double d = (double)t2[i] - (double)t1[i];
        if (d < 0.0) d = -d;
        double e = (double)w[i] * d;
        s += e * e;
    }
    return orc_det_exp(-s * temperature / 2.0);
}

/* World graph: getPlanarAdjacencyGraph (GraphUtils.cpp:52-138) + mergeNewScanGraph (:140-202).
 * Edge {i, j}: in-view neighbours with pMerge > in_thr, or max(i, j) >= n0 (a vertex of a later
 * scan against everything already in the graph, its own scan included) with pMerge > thr.
 * theta1 of a pair is the lower id.  Returns the number of directed entries (CSR ascending);
 * col / prob may be NULL to count. */
long long orc_mv_world_graph(int V, int dim, const float *theta, const float *w, int n0, const uint32_t *in_rowp,
                             const uint32_t *in_col, double in_thr, double thr, double T, uint32_t *row_ptr,
                             uint32_t *col, double *prob)
{
    uint8_t *adj = (uint8_t *)calloc((size_t)V * V, 1);
    for (int i = 0; i < V; ++i)
        for (int j = i + 1; j < V; ++j)
            if (j >= n0 && orc_mv_pmerge(theta + (size_t)i * dim, theta + (size_t)j * dim, w, dim, T) > thr)
                adj[(size_t)i * V + j] = adj[(size_t)j * V + i] = 1;
    if (in_rowp)
        for (int i = 0; i < V; ++i)
            for (uint32_t e = in_rowp[i]; e < in_rowp[i + 1]; ++e) {
                const int j = (int)in_col[e];
                if (j >= V || j == i) continue;
                const int a = i < j ? i : j, b = i < j ? j : i;
                if (orc_mv_pmerge(theta + (size_t)a * dim, theta + (size_t)b * dim, w, dim, T) > in_thr)
                    adj[(size_t)i * V + j] = adj[(size_t)j * V + i] = 1;
            }
    long long e2 = 0;
    for (int i = 0; i < V; ++i) {
        row_ptr[i] = (uint32_t)e2;
        for (int j = 0; j < V; ++j)
            if (adj[(size_t)i * V + j]) {
                if (col) col[e2] = (uint32_t)j;
                if (prob) {
                    const int a = i < j ? i : j, b = i < j ? j : i;
                    prob[e2] = orc_mv_pmerge(theta + (size_t)a * dim, theta + (size_t)b * dim, w, dim, T);
                }
                ++e2;
            }
    }
    row_ptr[V] = (uint32_t)e2;
    free(adj);
    return e2;
}

/* ------------------------------------------------------------------------- */
/* chains                                                                     */
/* ------------------------------------------------------------------------- */
typedef struct {
    mv_graph g;
    uint32_t *erank;        /* [E2] rank of the undirected edge among (i<j) entries */
    int n, seg_cap; uint32_t seed, offset; double split_prob; uint64_t step;
    uint32_t *cur, *own;    /* [n][V] */
    double *plane;          /* [n][seg_cap][5] */
    int *nseg, *status;
    uint64_t *n_accept, *n_split, *n_skip, *trace_hash;
} mv_chains;

void orc_mv_chains_free(mv_chains *c)
{
    if (!c) return;
    free(c->erank); free(c->cur); free(c->own); free(c->plane); free(c->nseg); free(c->status);
    free(c->n_accept); free(c->n_split); free(c->n_skip); free(c->trace_hash); free(c);
}

mv_chains *orc_mv_chains_create(int V, const float *pts, const uint32_t *pt_ptr, const double *vm,
                                const uint32_t *rowp, const uint32_t *col, const double *q,
                                int n_chains, uint32_t seed, uint32_t offset, int max_new, double split_prob)
{
    mv_chains *c = calloc(1, sizeof *c);
    c->g = (mv_graph){ V, pts, pt_ptr, vm, rowp, col, q };
    c->n = n_chains; c->seed = seed; c->offset = offset; c->split_prob = split_prob;
    c->seg_cap = V + max_new;
    const uint32_t E2 = rowp[V];
    c->erank = malloc(sizeof(uint32_t) * (E2 ? E2 : 1));
    uint32_t r = 0;
    for (int i = 0; i < V; ++i)
        for (uint32_t e = rowp[i]; e < rowp[i + 1]; ++e)
            if ((uint32_t)i < col[e]) c->erank[e] = r++;
    for (int i = 0; i < V; ++i)
        for (uint32_t e = rowp[i]; e < rowp[i + 1]; ++e)
            if ((uint32_t)i > col[e]) {
                const uint32_t j = col[e];
                for (uint32_t f = rowp[j]; f < rowp[j + 1]; ++f) if (col[f] == (uint32_t)i) c->erank[e] = c->erank[f];
            }
    c->cur = malloc(sizeof(uint32_t) * (size_t)n_chains * V);
    c->own = malloc(sizeof(uint32_t) * (size_t)n_chains * V);
    c->plane = calloc((size_t)n_chains * c->seg_cap * MV_PLANE, sizeof(double));
    c->nseg = calloc(n_chains, sizeof(int)); c->status = calloc(n_chains, sizeof(int));
    c->n_accept = calloc(n_chains, 8); c->n_split = calloc(n_chains, 8);
    c->n_skip = calloc(n_chains, 8); c->trace_hash = calloc(n_chains, 8);
    uint8_t *mem = malloc((size_t)V);
    int *stack = malloc(sizeof(int) * (size_t)V);
    for (int k = 0; k < n_chains; ++k) {
        /* MultiviewSegmentation ctor (MultiviewSegmentation.cpp:11-69) */
        mv_rng g = { { seed, offset + (uint32_t)k }, (uint32_t)MV_INIT_STEP, (uint32_t)(MV_INIT_STEP >> 32), 0 };
        uint32_t *cur = c->cur + (size_t)k * V, *own = c->own + (size_t)k * V;
        for (int v = 0; v < V; ++v) cur[v] = 0xFFFFFFFFu;
        int ncomp = 0;
        for (int s = 0; s < V; ++s) {
            if (cur[s] != 0xFFFFFFFFu) continue;
            int sp = 0; stack[sp++] = s; cur[s] = (uint32_t)ncomp;
            while (sp) {
                const int v = stack[--sp];
                for (uint32_t e = rowp[v]; e < rowp[v + 1]; ++e) {
                    const uint32_t j = col[e];
                    if (cur[j] != 0xFFFFFFFFu) continue;
                    const double u = (double)mv_u32_at(&g, c->erank[e]) / 4294967295.0;
                    if (u <= q[e]) { cur[j] = (uint32_t)ncomp; stack[sp++] = (int)j; }   /* inclusive, :23 */
                }
            }
            ncomp++;
        }
        memcpy(own, cur, sizeof(uint32_t) * (size_t)V);
        c->nseg[k] = ncomp;
        for (int s = 0; s < ncomp; ++s) {
            for (int v = 0; v < V; ++v) mem[v] = cur[v] == (uint32_t)s;
            double m[MV_MOM];
            pool_moments(&c->g, mem, m);
            plane_from_moments(m, c->plane + ((size_t)k * c->seg_cap + s) * MV_PLANE);
        }
    }
    free(mem); free(stack);
    return c;
}

/* one step of PatternedMatcher.cpp:58-130; returns flags: bit0 accept, bit1 split, bit2 skipped */
static uint8_t mv_step(mv_chains *c, int k, uint64_t t, double *a_out)
{
    const mv_graph *g = &c->g;
    const int V = g->V;
    uint32_t *cur = c->cur + (size_t)k * V, *own = c->own + (size_t)k * V;
    double *planes = c->plane + (size_t)k * c->seg_cap * MV_PLANE;
    mv_rng rg = { { c->seed, c->offset + (uint32_t)k }, (uint32_t)t, (uint32_t)(t >> 32), 0 };
    uint8_t *inE = calloc((size_t)V, 1), *mem = malloc((size_t)V);
    uint32_t *sv = malloc(sizeof(uint32_t) * (size_t)V), *se = malloc(sizeof(uint32_t) * (size_t)V);
    uint8_t flags = 0;
    double a = 0.0;

    /* draws of a step: 0 start vertex, 1 move type, 2 target index, 3 accept; 16 + r the trial of the
     * undirected edge of rank r (each is tried at most once per step, whichever end reaches it first) */
    const uint32_t start = mv_u32_at(&rg, 0) % (uint32_t)V;            /* random_vertex, :61 */
    const uint32_t orig = own[start];                                   /* getParentSegment, :64 */
    /* getNewConnectedSet: recursive DFS, one draw per same-segment unvisited neighbour (:207-257).  The set it
     * returns is the component of `start` among the same-segment edges whose trial succeeds, so with the
     * trial's uniform tied to the edge the set does not depend on the order of the walk. */
    int sp = 0, nE = 1;
    inE[start] = 1; sv[0] = start; se[0] = g->rowp[start]; sp = 1;
    while (sp) {
        const uint32_t v = sv[sp - 1];
        if (se[sp - 1] >= g->rowp[v + 1]) { sp--; continue; }
        const uint32_t e = se[sp - 1]++;
        const uint32_t j = g->col[e];
        if (cur[v] != cur[j]) continue;
        if (inE[j]) continue;
        if ((double)mv_u32_at(&rg, 16u + c->erank[e]) / 4294967295.0 <= g->q[e]) { inE[j] = 1; nE++; sv[sp] = j; se[sp] = g->rowp[j]; sp++; }
    }
    const double moveType = (double)mv_u32_at(&rg, 1) / 4294967295.0;   /* :76 */
    uint32_t target;
    if (moveType < c->split_prob) {
        /* addNewSegment (:85-110): really moves the vertices and appends a segment */
        if (c->n_split[k] >= (uint64_t)(c->seg_cap - V)) { c->status[k] = -6; flags = 0xFF; goto done; }   /* max_new_segments */
        target = (uint32_t)c->nseg[k]++;
        for (int v = 0; v < V; ++v) if (inE[v]) own[v] = target;
        double m[MV_MOM];
        pool_moments(g, inE, m);
        plane_from_moments(m, planes + (size_t)target * MV_PLANE);
        flags |= 2; c->n_split[k]++;
    } else {
        /* getNeighborSegments (:112-142) on the never-updated current ids */
        uint8_t *nb = calloc((size_t)c->seg_cap, 1);
        int nn = 0;
        for (int v = 0; v < V; ++v) {
            if (!inE[v]) continue;
            for (uint32_t e = g->rowp[v]; e < g->rowp[v + 1]; ++e) {
                const uint32_t sj = cur[g->col[e]];
                if (sj != cur[v] && !nb[sj]) { nb[sj] = 1; nn++; }
            }
        }
        if (nn < 1) { free(nb); flags |= 4; c->n_skip[k]++; goto hash; }   /* continue, :88-92 */
        /* round(randbetween(0, n-1)), half away from zero (:93) */
        const double x = (double)(nn - 1) * ((double)mv_u32_at(&rg, 2) / 4294967295.0);
        int idx = (int)(x + 0.5);
        target = 0;
        for (int s = 0, seen = 0; s < c->seg_cap; ++s) if (nb[s]) { if (seen == idx) { target = (uint32_t)s; break; } seen++; }
        free(nb);
    }
    {
        /* moveSuperpixels on copies (:188-205) + computeProposalRatio (:270-274) */
        double pl[MV_PLANE], m[MV_MOM], pSA1, pSA2, pSB1, pSB2;
        for (int v = 0; v < V; ++v) mem[v] = own[v] == orig;
        pSA1 = set_probability(g, mem, planes + (size_t)orig * MV_PLANE, NULL);
        for (int v = 0; v < V; ++v) mem[v] = own[v] == target;
        pSA2 = set_probability(g, mem, planes + (size_t)target * MV_PLANE, NULL);
        for (int v = 0; v < V; ++v) mem[v] = own[v] == orig && !inE[v];
        memcpy(pl, planes + (size_t)orig * MV_PLANE, sizeof pl);
        if (pool_moments(g, mem, m) > 0) plane_from_moments(m, pl);       /* empty: fit skipped (:37) */
        pSB1 = set_probability(g, mem, pl, NULL);
        for (int v = 0; v < V; ++v) mem[v] = own[v] == target || inE[v];
        pool_moments(g, mem, m);
        plane_from_moments(m, pl);
        pSB2 = set_probability(g, mem, pl, NULL);
        a = (pSB1 / pSA1) * (pSB2 / pSA2);
        const double u = (double)mv_u32_at(&rg, 3) / 4294967295.0;
        if (u < a) { flags |= 1; c->n_accept[k]++; }                       /* :112 */
    }
hash:
    c->trace_hash[k] = orc_mix64(orc_mix64(c->trace_hash[k] ^ ((uint64_t)flags | ((uint64_t)start << 8) | ((uint64_t)nE << 32))) ^ d2u(a));
done:
    if (a_out) *a_out = a;
    free(inE); free(mem); free(sv); free(se);
    return flags;
}

int orc_mv_chains_run(mv_chains *c, uint64_t steps, uint8_t *trace, double *trace_a)
{
    for (int k = 0; k < c->n; ++k)
        for (uint64_t s = 0; s < steps; ++s) {
            double a = 0.0;
            uint8_t f = 0xFF;
            if (c->status[k] == 0) f = mv_step(c, k, c->step + s, &a);
            if (trace) trace[(size_t)k * steps + s] = f;
            if (trace_a) trace_a[(size_t)k * steps + s] = a;
        }
    c->step += steps;
    return 0;
}

void orc_mv_chains_get_labels(const mv_chains *c, int which, uint32_t *out)
{
    memcpy(out, which ? c->own : c->cur, sizeof(uint32_t) * (size_t)c->n * c->g.V);
}

void orc_mv_chains_get_scalars(const mv_chains *c, uint64_t *n_accept, uint64_t *n_split, uint64_t *n_skip,
                               uint64_t *trace_hash, int32_t *nseg, int32_t *status)
{
    for (int k = 0; k < c->n; ++k) {
        if (n_accept) n_accept[k] = c->n_accept[k];
        if (n_split) n_split[k] = c->n_split[k];
        if (n_skip) n_skip[k] = c->n_skip[k];
        if (trace_hash) trace_hash[k] = c->trace_hash[k];
        if (nseg) nseg[k] = c->nseg[k];
        if (status) status[k] = c->status[k];
    }
}

void orc_mv_chains_get_planes(const mv_chains *c, int chain, double *out /* nseg*5 */)
{
    memcpy(out, c->plane + (size_t)chain * c->seg_cap * MV_PLANE, sizeof(double) * (size_t)c->nseg[chain] * MV_PLANE);
}
