// frontend_fh_oracle.cpp -- ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the graph-based segmentation the reference runs on the pixel grid
// (F3 of frontend_oracle.c): segment_graph, include/Segmentation/segment-graph.h:47-111, on the
// disjoint-set forest of include/Segmentation/disjoint-set.h:42-79.
// PARITY UNPINNED: the reference holds no test or golden vector for this stage, and building
// its headers here was refused by the environment.  Pinned instead by a
// literal Python transcription of the cited lines on inputs without weight ties
// (tests/test_frontend.py) and by partition properties.
//
// Kept: edges sorted by weight alone (segment-graph.h:35-37,53).  The reference uses std::sort, so the
// order of equal weights is whatever that standard library's introsort leaves -- the reference's result
// depends on it.  tie_order 1 does the same call with this standard library; tie_order 0 (the product's
// default) is the canonical order "equal weights in creation order" (std::stable_sort).  Merge iff w <= both thresholds (:92); threshold of a merged set =
// w + c / size in float (:102, THRESHOLD at :27); find() compresses only the queried node's
// link (disjoint-set.h:58-64); union by rank, ties go to the second root (:68-79).
#include <algorithm>
#include <cstdint>
#include <vector>

namespace {
struct WEdge { float w; int32_t a, b; };
struct Forest {
    std::vector<int32_t> parent, rank, size;
    explicit Forest(int n) : parent(n), rank(n, 0), size(n, 1) { for (int i = 0; i < n; ++i) parent[i] = i; }
    int find(int x) { int y = x; while (y != parent[y]) y = parent[y]; parent[x] = y; return y; }
    void join(int x, int y)
    {
        if (rank[x] > rank[y]) { parent[y] = x; size[x] += size[y]; }
        else { parent[x] = y; size[y] += size[x]; if (rank[x] == rank[y]) rank[y]++; }
    }
};
}  // namespace

// root_out[k] = find(k) after the pass, for every vertex in ascending k (the order in which
// createSuperPixels queries the universe, OverSegmentation.cpp:110).  Returns the number of sets.
extern "C" int orc_segment_graph(int nv, long long ne, const int32_t* ea, const int32_t* eb, const float* ew, float c,
                                 int tie_order, int32_t* root_out)
{
    std::vector<WEdge> edges((size_t)ne);
    for (long long i = 0; i < ne; ++i) { edges[i].w = ew[i]; edges[i].a = ea[i]; edges[i].b = eb[i]; }
    if (tie_order == 1) std::sort(edges.begin(), edges.end(), [](const WEdge& x, const WEdge& y) { return x.w < y.w; });
    else std::stable_sort(edges.begin(), edges.end(), [](const WEdge& x, const WEdge& y) { return x.w < y.w; });
    Forest f(nv);
    std::vector<float> thr((size_t)nv, c / 1);
    int sets = nv;
    for (const WEdge& e : edges) {
        const int ra = f.find(e.a), rb = f.find(e.b);
        if (ra == rb) continue;
        if (e.w <= thr[ra] && e.w <= thr[rb]) {
            f.join(ra, rb);
            --sets;
            const int r = f.find(ra);
            thr[r] = e.w + c / (float)(size_t)f.size[r];
        }
    }
    for (int k = 0; k < nv; ++k) root_out[k] = f.find(k);
    return sets;
}
