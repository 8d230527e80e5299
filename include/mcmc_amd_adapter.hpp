// mcmc_amd_adapter.hpp -- host-side C++ mirror of the reference's operator
// interface for this path, above the C ABI of mcmc_amd.h.
//
//   mcmc_amd::Problem   the PROBLEM concept of MCMC<PROBLEM>
//                       (reference include/MCMC.h:19-95, include/Problem.h:20-29):
//                       typedef State; static State* initializeState(void*);
//                       static State* propose(const State&, long double&, long double&).
//   mcmc_amd::Sampler   the batched form of MCMC<Problem>::run (MCMC.h:43-89) for
//                       many chains at once; Statistics-shaped read-back.
//
// Errors are reported the way the reference does: by throwing (its Exception_,
// include/MyMath.h:36-44; here std::runtime_error with mcmc_last_error()).
// Header-only; link libmcmc_amd.so.
#ifndef MCMC_AMD_ADAPTER_HPP
#define MCMC_AMD_ADAPTER_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "mcmc_amd.h"

namespace mcmc_amd {

// What OverSegmentation::createSuperPixels leaves behind (OverSegmentation.cpp:89-233),
// as flat arrays: disparity plane, superpixel id per pixel, CSR adjacency.
struct FrameInput {
    int width = 640, height = 480;
    const int16_t* disp = nullptr;
    const uint16_t* label = nullptr;
    int n_superpixels = 0;
    const uint32_t* row_ptr = nullptr;
    const uint32_t* col = nullptr;
    uint32_t e2 = 0;
};

class Sampler {
public:
    Sampler(const FrameInput& in, int device = 0, double disparity_variance = 7.0, double new_segment_prob = 0.8)
        : s_(in.n_superpixels)
    {
        mcmc_params p;
        mcmc_default_params(&p);
        p.width = in.width; p.height = in.height;
        p.disparity_variance = disparity_variance; p.new_segment_prob = new_segment_prob;
        if (mcmc_ctx_create(device, &p, &ctx_) != MCMC_OK) throw std::runtime_error("mcmc_ctx_create failed (no MI355X?)");
        ck(mcmc_frames_alloc(ctx_, 1, in.n_superpixels, (int)in.e2));
        ck(mcmc_frame_upload(ctx_, 0, in.disp, in.label, in.n_superpixels));
        ck(mcmc_frame_set_graph(ctx_, 0, in.row_ptr, in.col, in.e2));
        ck(mcmc_frames_moments(ctx_, 0, 1));       // new Plane(A, b) per superpixel, OverSegmentation.cpp:219
        ck(mcmc_frames_edge_probs(ctx_, 0, 1));    // Problem::computeEdgeProbs, OverSegmentation.cpp:50
    }
    ~Sampler() { if (ctx_) mcmc_ctx_destroy(ctx_); }
    Sampler(const Sampler&) = delete;
    Sampler& operator=(const Sampler&) = delete;

    // Problem::initializeState for n chains (Problem.cpp:447-473)
    void init(int chains, uint32_t seed, uint32_t chain_offset = 0)
    {
        ck(mcmc_chains_init(ctx_, chains, chains, seed, chain_offset));
        n_ = chains;
    }
    // MCMC<Problem>::run (MCMC.h:43-89); may be called again to continue
    void run(uint64_t burn_in, uint64_t iterations)
    {
        ck(mcmc_chains_run(ctx_, burn_in, iterations));
        ck(mcmc_ctx_synchronize(ctx_));
    }
    void run(int chains, uint32_t seed, uint64_t burn_in, uint64_t iterations)
    {
        init(chains, seed);
        run(burn_in, iterations);
    }

    int chains() const { return n_; }
    int superpixels() const { return s_; }

    // which: 0 current, 1 best energy, 2 most visited (printResults, can05_sampler.cpp:36)
    std::vector<uint16_t> labels(int which, bool canonical = true)
    {
        std::vector<uint16_t> out((size_t)n_ * (size_t)s_);
        ck(mcmc_chains_get_labels(ctx_, which, canonical ? 1 : 0, out.data()));
        return out;
    }
    std::vector<uint16_t> most_visited_labels() { return labels(2); }

    struct Scalars {
        std::vector<double> energy, best_energy;
        std::vector<uint64_t> state_hash, trace_hash, n_accept, top_count, top_hash, overflow;
        std::vector<int32_t> status;
    };
    Scalars scalars()
    {
        Scalars r;
        const size_t n = (size_t)n_;
        r.energy.resize(n); r.best_energy.resize(n); r.state_hash.resize(n); r.trace_hash.resize(n);
        r.n_accept.resize(n); r.top_count.resize(n); r.top_hash.resize(n); r.overflow.resize(n); r.status.resize(n);
        ck(mcmc_chains_get_scalars(ctx_, r.energy.data(), r.best_energy.data(), r.state_hash.data(), r.trace_hash.data(),
                                   r.n_accept.data(), nullptr, r.status.data(), r.top_count.data(), r.top_hash.data(),
                                   r.overflow.data()));
        return r;
    }
    // Statistics::entries_ of one chain (Statistics.h:30): state hash -> visit count
    std::map<uint64_t, uint64_t> visits(int chain)
    {
        const int capacity = mcmc_chains_visit_capacity(ctx_);
        if (capacity <= 0 || chain < 0 || chain >= n_) throw std::runtime_error("mcmc_amd: visits: no chains / bad chain");
        std::vector<uint64_t> h((size_t)n_ * capacity), c((size_t)n_ * capacity);
        ck(mcmc_chains_get_visits(ctx_, h.data(), c.data()));
        std::map<uint64_t, uint64_t> m;
        for (int i = 0; i < capacity; ++i)
            if (c[(size_t)chain * capacity + i]) m[h[(size_t)chain * capacity + i]] = c[(size_t)chain * capacity + i];
        return m;
    }
    // Statistics::getOrderedStates (Statistics.h:50-70): (count, state hash) of one chain, most visited first
    std::vector<std::pair<uint64_t, uint64_t>> ordered_states(int chain)
    {
        std::vector<std::pair<uint64_t, uint64_t>> v;
        for (const auto& kv : visits(chain)) v.emplace_back(kv.second, kv.first);
        std::sort(v.begin(), v.end(), [](const std::pair<uint64_t, uint64_t>& a, const std::pair<uint64_t, uint64_t>& b) {
            return a.first != b.first ? a.first > b.first : a.second < b.second;
        });
        return v;
    }
    // Problem::propose's out-parameters for the last step of every chain (Problem.cpp:213-217)
    void last_ratios(std::vector<double>& targetL, std::vector<double>& proposal)
    {
        targetL.resize((size_t)n_); proposal.resize((size_t)n_);
        ck(mcmc_chains_get_last_ratios(ctx_, targetL.data(), proposal.data()));
    }
    // Segment(superPixels).probabilityL_ (Graph.h:33-40)
    double segment_probability(const std::vector<uint32_t>& members)
    {
        double p = 0.0;
        ck(mcmc_segment_probability(ctx_, 0, members.data(), (int)members.size(), &p, nullptr, nullptr));
        return p;
    }
    mcmc_ctx* ctx() { return ctx_; }

    void ck(int rc)
    {
        if (rc != MCMC_OK) throw std::runtime_error(std::string("mcmc_amd: ") + mcmc_last_error(ctx_));
    }

private:
    mcmc_ctx* ctx_ = nullptr;
    int n_ = 0, s_ = 0;
};

// ---- the PROBLEM concept -----------------------------------------------------
// State is a view of the one device-resident chain; hash_ is what Statistics keys on
// (Statistics.h:37).
struct State {
    mutable long hash_ = 0;   // always the device chain's current hash, whichever State object the caller keeps
    Sampler* sampler = nullptr;
    bool accepted = true;
};

struct Problem {
    typedef mcmc_amd::State State;

    struct Data {
        FrameInput in;
        int device;
        uint32_t seed;
        Data(const FrameInput& i, int dev = 0, uint32_t s = 1) : in(i), device(dev), seed(s) {}
    };

    // Problem::initializeState(void* data) (Problem.cpp:447-473); data is a Data*
    static State* initializeState(void* data = nullptr)
    {
        if (!data) throw std::runtime_error("initializeState: Initialize state needs a graph pointer");
        Data* d = static_cast<Data*>(data);
        Sampler* s = new Sampler(d->in, d->device);   // leaked like every State in the reference (MCMC.h:237)
        s->init(1, d->seed);
        State* st = new State;
        st->sampler = s;
        st->hash_ = (long)s->scalars().state_hash[0];
        return st;
    }

    // Problem::propose(state, targetRatio (log), proposalRatio) (Problem.cpp:116-224).  The ratios are the real
    // ones of the step (Problem.cpp:213-217).  The chain lives on the device, which has already taken the accept
    // decision with its own Philox draw (the reference's iterate draws rand(), MCMC.h:223); every State is a view
    // of that one chain, so both the State passed in and the one returned carry the chain's hash after the step,
    // and whichever of the two FastMetropolisHastings::iterate keeps (MCMC.h:224-237), Statistics::update
    // (Statistics.h:37) books the state the chain is really in.  `accepted` tells what the device decided.
    static State* propose(const State& state, long double& targetRatio, long double& proposalRatio)
    {
        Sampler* s = state.sampler;
        const uint64_t before = s->scalars().n_accept[0];
        s->run(0, 1);
        const Sampler::Scalars sc = s->scalars();
        if (sc.status[0] == MCMC_E_NAN_RATIO) throw std::runtime_error("propose: Target ratio is nan!");
        if (sc.status[0] == MCMC_E_INF_ENERGY) throw std::runtime_error("computeProbability: The subgraph log probability is inf");
        State* next = new State;
        next->sampler = s;
        next->hash_ = (long)sc.state_hash[0];
        next->accepted = sc.n_accept[0] != before;
        std::vector<double> t, p;
        s->last_ratios(t, p);
        targetRatio = (long double)t[0];
        proposalRatio = (long double)p[0];
        state.hash_ = next->hash_;
        return next;
    }
};

}  // namespace mcmc_amd
#endif
