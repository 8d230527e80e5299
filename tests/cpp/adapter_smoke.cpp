// Drives include/mcmc_amd_adapter.hpp the way the reference's executable drives
// Problem through MCMC.h (can05_sampler.cpp:84-90): a FastMetropolisHastings-shaped
// loop over Problem::propose, then the batched Sampler.  Input: a scene file
// written by tests/test_adapter.py.  Output: hashes on stdout for the test to compare.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#include "mcmc_amd_adapter.hpp"

template <class PROBLEM>
struct FastMHShape {  // the control flow of FastMetropolisHastings::iterate (MCMC.h:207-238)
    mutable long double targetRatio = 0, proposalRatio = 0;   // kept for the test: what propose handed over last
    typename PROBLEM::State* iterate(typename PROBLEM::State* state) const
    {
        typename PROBLEM::State* newState = PROBLEM::propose(*state, targetRatio, proposalRatio);
        long double aL = targetRatio + std::log(proposalRatio);
        long double a = std::exp(aL);
        long double r = ((long double)rand()) / RAND_MAX;
        bool accept = true;
        if (aL < 0.0) { if (r > a) accept = false; }
        return accept ? newState : state;
    }
};

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: adapter_smoke scene.bin steps seed\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[4];
    if (fread(hdr, 4, 4, f) != 4) return 2;
    const int W = hdr[0], H = hdr[1], S = hdr[2], E2 = hdr[3];
    std::vector<int16_t> disp((size_t)W * H); std::vector<uint16_t> label((size_t)W * H);
    std::vector<uint32_t> rp(S + 1), col(E2);
    if (fread(disp.data(), 2, disp.size(), f) != disp.size()) return 2;
    if (fread(label.data(), 2, label.size(), f) != label.size()) return 2;
    if (fread(rp.data(), 4, rp.size(), f) != rp.size()) return 2;
    if (E2 && fread(col.data(), 4, col.size(), f) != col.size()) return 2;
    fclose(f);
    const int steps = atoi(argv[2]);
    const uint32_t seed = (uint32_t)atoi(argv[3]);

    mcmc_amd::FrameInput in;
    in.width = W; in.height = H; in.disp = disp.data(); in.label = label.data();
    in.n_superpixels = S; in.row_ptr = rp.data(); in.col = col.data(); in.e2 = (uint32_t)E2;

    try {
        // 1. PROBLEM-concept path: one device chain stepped through propose()
        mcmc_amd::Problem::Data data(in, 0, seed);
        mcmc_amd::Problem::State* st = mcmc_amd::Problem::initializeState(&data);
        FastMHShape<mcmc_amd::Problem> mh;
        int accepted = 0;
        for (int i = 0; i < steps; ++i) {
            mcmc_amd::Problem::State* nx = mh.iterate(st);
            // the chain is on the device: count what IT decided; whichever State the host-side draw keeps
            // carries the chain's current hash
            if (nx != st ? nx->accepted : false) accepted++;
            st = nx;
        }
        printf("ratios %a %a\n", (double)mh.targetRatio, (double)mh.proposalRatio);
        printf("problem hash %lu accepted %d trace %lu\n", (unsigned long)st->hash_, accepted,
               (unsigned long)st->sampler->scalars().trace_hash[0]);
        // 2. batched path: the same chain (chain 0 of 4) in one launch
        mcmc_amd::Sampler s(in, 0);
        s.run(4, seed, 0, (uint64_t)steps);
        mcmc_amd::Sampler::Scalars sc = s.scalars();
        printf("batched hash %lu accepted %lu trace %lu\n", (unsigned long)(long)sc.state_hash[0],
               (unsigned long)sc.n_accept[0], (unsigned long)sc.trace_hash[0]);
        std::vector<uint16_t> lab = s.most_visited_labels();
        unsigned long sum = 0;
        for (size_t i = 0; i < lab.size(); ++i) sum = sum * 31 + lab[i];
        printf("labels %lu visits %zu\n", sum, s.visits(0).size());
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
