/*
 * frontend_oracle.c -- ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, no FMA contraction) of the over-segmentation front end of
 * a-price/MCMC, the producer of every hot-path input:
 *
 *   F1  OverSegmentation::bilateralFilter      src/libraries/OverSegmentation.cpp:304-391
 *       createGaussianFilter / computeGaussian  :394-419, include/OverSegmentation.h:75-79
 *   F2  createEdges / computeWeight             :236-301
 *   F3  segment_graph (Felzenszwalb-Huttenlocher union-find)
 *                                               include/Segmentation/segment-graph.h:47-111,
 *                                               disjoint-set.h:42-79   -> frontend_fh_oracle.cpp
 *   F4  createSuperPixels (labels + neighbours)  src/libraries/OverSegmentation.cpp:89-194
 *
 * Nothing under mcmc_amd/ may include, link or call this file.
 *
 * PARITY UNPINNED.  The reference holds no test or golden vector for any of these stages; F1,
 * F2, F4 need OpenCV / gtsam to build, and building the two std-only headers of F3 was refused
 * by the environment.  They are pinned by literal numpy / Python
 * transcriptions of the cited lines in tests/test_frontend.py (f64 / f32 arithmetic) instead.
 *
 * Behaviour kept, including what looks accidental:
 *  - the filter window is anchored at its top-left corner (x, y) and its "middle" is at
 *    x + ceil(w / 2.0), one past the true centre for odd w (:335); only pixels with
 *    x < W - w and y < H - w are filtered, the rest of the smoothed image stays 0 (:322-327);
 *    the channel value is truncated, not rounded (:381);
 *  - edges go RIGHT, BOTTOM, BOTTOM-RIGHT, TOP-RIGHT from every pixel in row-major order
 *    (:272-296); colour and disparity terms are float, the weighted sum is taken in double
 *    (the factors are double) and stored as float (:261-265);
 *  - createSuperPixels compares every pixel with LEFT, TOP, TOP-LEFT, TOP-RIGHT through a ring
 *    buffer of W + 1 ids that is written before it is read, so the TOP-LEFT slot already
 *    holds the pixel's own id and that neighbour never registers (:174-186).
 * Canonical choices: exp / sqrt of the Gaussians are det_exp / det_sqrt (the same bits on CPU
 * and GPU; libm differs from them in the last ulp, which can move a truncated channel value
 * by one where the weighted mean is within 1e-13 of an integer); the sums over the window run
 * j (rows) outer, i (columns) inner; superpixels are numbered by ascending root pixel, which
 * is the order of the reference's std::map<SuperPixelID, SuperPixel*>.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

double orc_det_exp(double x);
double orc_det_sqrt(double x);

#define ORC_OK 0
#define ORC_E_BADARG (-1)
#define ORC_E_RANGE (-8)

static double gauss2(double x, double y, double stdev) /* OverSegmentation.h:75-79 */
{
    const double norm = 1.0 / orc_det_sqrt(2.0 * 3.14159265358979323846 * stdev * stdev);
    const double exponent = (x * x + y * y) / (2 * stdev * stdev);
    return norm * orc_det_exp(-exponent);
}

/* F1.  colors / smoothed: H x W x 3 bytes (cv::Vec3b), disp: H x W shorts. */
int orc_bilateral(const int16_t *disp, const uint8_t *colors, int W, int H, int w,
                  double spatial_stdev, double disparity_stdev, uint8_t *smoothed)
{
    if (!(spatial_stdev > 0.0) || !(disparity_stdev > 0.0) || w < 0 || w > 64) return ORC_E_BADARG;
    if (w < 2) { memcpy(smoothed, colors, (size_t)W * H * 3); return ORC_OK; }   /* :313-316 */
    memset(smoothed, 0, (size_t)W * H * 3);
    double *sk = (double *)malloc(sizeof(double) * (size_t)w * w);   /* spatial kernel, [i][j] */
    double *jk = (double *)malloc(sizeof(double) * (size_t)w * w);
    {
        double sum = 0.0;
        const double mid = (1.0 + w) / 2.0;
        for (int i = 1; i <= w; ++i)
            for (int j = 1; j <= w; ++j) {
                const double v = gauss2(mid - i, mid - j, spatial_stdev);
                sk[(i - 1) * w + (j - 1)] = v;
                sum += v;
            }
        for (int k = 0; k < w * w; ++k) sk[k] = sk[k] / sum;
    }
    const int c0 = (w + 1) / 2;   /* ceil(w / 2.0) */
    int rc = ORC_OK;
    for (int y = 0; y + w < H; ++y) {
        for (int x = 0; x + w < W; ++x) {
            const int x0 = x + c0, y0 = y + c0;
            double md = disp[(size_t)y0 * W + x0];
            if (w % 2 == 0)
                md = (disp[(size_t)y0 * W + x0] + disp[(size_t)y0 * W + x0 + 1] +
                      disp[(size_t)(y0 + 1) * W + x0] + disp[(size_t)(y0 + 1) * W + x0 + 1]) / 4.0;
            double sum = 0.0;
            for (int j = 0; j < w; ++j)
                for (int i = 0; i < w; ++i) {
                    const double dd = md - (double)disp[(size_t)(j + y) * W + (i + x)];
                    const double v = sk[i * w + j] * gauss2(dd, 0.0, disparity_stdev);
                    jk[i * w + j] = v;
                    sum += v;
                }
            for (int ch = 0; ch < 3; ++ch) {
                double t = 0.0;
                for (int j = 0; j < w; ++j)
                    for (int i = 0; i < w; ++i)
                        t += (double)colors[((size_t)(y + j) * W + (x + i)) * 3 + ch] * (jk[i * w + j] / sum);
                const long long value = (long long)t;
                if (!(value >= 0 && value < 256)) rc = ORC_E_RANGE;   /* "Weird channel value." */
                smoothed[((size_t)y * W + x) * 3 + ch] = (uint8_t)value;
            }
        }
    }
    free(sk); free(jk);
    return rc;
}

/* F2.  Returns the number of edges; ea/eb/ew sized 4 W H. */
long long orc_create_edges(const int16_t *disp, const uint8_t *smoothed, int W, int H, double color_factor,
                           double disparity_factor, double distance_factor, int32_t *ea, int32_t *eb, float *ew)
{
    static const int dir[8] = { 1, 0, 0, 1, 1, 1, 1, -1 };   /* RIGHT, BOTTOM, BOTTOM-RIGHT, TOP-RIGHT */
    long long n = 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int i = 0; i < 4; ++i) {
                const int x2 = x + dir[2 * i], y2 = y + dir[2 * i + 1];
                if (x2 >= W || y2 >= H || y2 < 0) continue;   /* :282 (a negative int compares as huge) */
                const uint8_t *c1 = smoothed + ((size_t)y * W + x) * 3, *c2 = smoothed + ((size_t)y2 * W + x2) * 3;
                const float dr = (float)c1[2] - (float)c2[2];
                const float dg = (float)c1[1] - (float)c2[1];
                const float db = (float)c1[0] - (float)c2[0];
                const float d1 = disp[(size_t)y * W + x], d2 = disp[(size_t)y2 * W + x2];
                const float colorv = sqrtf(dr * dr + dg * dg + db * db);
                const float dispv = fabsf(d1 - d2);
                const int ddx = x - x2, ddy = y - y2;
                const float distv = (float)sqrt((double)(ddx * ddx + ddy * ddy));
                const float wgt = (float)((color_factor * colorv + disparity_factor * dispv) + distance_factor * distv);
                ea[n] = y * W + x; eb[n] = y2 * W + x2; ew[n] = wgt;
                ++n;
            }
    return n;
}

/* F4.  root[k] = universe::find(k).  Outputs: label (dense ids by ascending root pixel),
 * *S_out, and the neighbour relation as a CSR (ascending ids) in row_ptr[S+1] / col (capacity
 * col_cap entries; returns ORC_E_RANGE if it does not fit).  adj_scratch: S x S bytes or NULL
 * (allocated here). */
int orc_superpixels(const int32_t *root, int W, int H, uint16_t *label, int *S_out, uint32_t *row_ptr,
                    uint32_t *col, long long col_cap, uint32_t *e2_out)
{
    const size_t N = (size_t)W * H;
    int32_t *rank = (int32_t *)malloc(sizeof(int32_t) * N);
    int S = 0;
    for (size_t k = 0; k < N; ++k) rank[k] = (root[k] == (int32_t)k) ? S++ : -1;
    if (S >= 0xFFFF) { free(rank); return ORC_E_RANGE; }
    for (size_t k = 0; k < N; ++k) {
        if (root[k] < 0 || (size_t)root[k] >= N || rank[root[k]] < 0) { free(rank); return ORC_E_BADARG; }
        label[k] = (uint16_t)rank[root[k]];
    }
    free(rank);
    *S_out = S;
    uint8_t *adj = (uint8_t *)calloc((size_t)S * S, 1);
    /* ring buffer of the last W + 1 ids, written before it is read (:174) */
    uint16_t *prev = (uint16_t *)calloc((size_t)W + 1, sizeof(uint16_t));
    static const int dir[8] = { -1, 0, 0, -1, -1, -1, 1, -1 };   /* left, top, top-left, top-right */
    size_t k = 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x, ++k) {
            const uint16_t id = label[k];
            prev[k % ((size_t)W + 1)] = id;
            for (int i = 0; i < 4; ++i) {
                const int x2 = x + dir[2 * i], y2 = y + dir[2 * i + 1];
                if (x2 < 0 || y2 < 0 || x2 >= W || y2 >= H) continue;   /* :181, unsigned compare */
                const size_t k2 = (size_t)x2 + (size_t)y2 * W;
                const uint16_t id2 = prev[k2 % ((size_t)W + 1)];
                if (id != id2) { adj[(size_t)id * S + id2] = 1; adj[(size_t)id2 * S + id] = 1; }
            }
        }
    free(prev);
    uint32_t e = 0;
    int rc = ORC_OK;
    for (int i = 0; i < S; ++i) {
        row_ptr[i] = e;
        for (int j = 0; j < S; ++j)
            if (adj[(size_t)i * S + j]) {
                if ((long long)e < col_cap) col[e] = (uint32_t)j; else rc = ORC_E_RANGE;
                ++e;
            }
    }
    row_ptr[S] = e;
    *e2_out = e;
    free(adj);
    return rc;
}
