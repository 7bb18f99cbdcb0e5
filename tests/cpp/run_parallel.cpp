// run_parallel.cpp -- the reference's parallel benchmark shape (b.RunParallel over shallow copies of one evaluator,
// schemes/ckks/ckks_benchmarks_test.go:116-207; schemes/bgv's MulRelin at the BASELINE config-3 shape) from a COMPILED host
// through the public interface only (include/hering.hpp): K OS threads, one ciphertext per call, one shared evaluator whose
// submission queue turns the concurrent calls into batched launches.  Prints one JSON line; bench.py embeds it.
//
//   run_parallel [K=64] [calls per thread=96] [sync_each=0|1] [coalesce=1|0]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "hering.hpp"
extern "C" {
#include "lattigo_oracle.h"
}
using u64v = std::vector<uint64_t>;

static u64v uniform(std::mt19937_64 &rng, const u64v &moduli, int N, int entries) {
    u64v out((size_t)entries * moduli.size() * N);
    size_t k = 0;
    for (int e = 0; e < entries; e++)
        for (uint64_t q : moduli) {
            std::uniform_int_distribution<uint64_t> d(0, q - 1);
            for (int j = 0; j < N; j++) out[k++] = d(rng);
        }
    return out;
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? std::atoi(argv[1]) : 64, iters = argc > 2 ? std::atoi(argv[2]) : 96;
    const bool sync_each = argc > 3 && std::atoi(argv[3]) != 0, coalesce = !(argc > 4 && std::atoi(argv[4]) == 0);
    const int logN = 15, N = 1 << logN;
    const uint64_t T = 65537;
    std::vector<int> logq(12, 45), logp(3, 55);
    logq[0] = 55;
    u64v q(logq.size()), p(logp.size());
    if (lo_gen_moduli(logN + 1, logq.data(), (int)logq.size(), logp.data(), (int)logp.size(), q.data(), p.data()) != 0) return 2;
    const int nq = (int)q.size(), np = (int)p.size(), level = nq - 1;
    try {
        hering::Context ctx(0);
        hering::Ring ringQ(ctx, logN, q), ringP(ctx, logN, p);
        hering::Evaluator eval(ringQ, ringP);
        std::mt19937_64 rng(0x1A77160 + 2);
        const int beta = lo_base_rns_decomposition_vector_size(level, np - 1);
        const u64v kq = uniform(rng, q, N, beta * 2), kp = uniform(rng, p, N, beta * 2);
        hering::EvaluationKey rlk = eval.NewEvaluationKey(beta, nq, np, kq, kp);
        if (coalesce) eval.SetCoalescing(64, 30); else eval.SetCoalescing(0, 0);
        const size_t words = (size_t)nq * N;
        struct Caller { hering::Ciphertext a, b, out; };
        std::vector<Caller> callers(K);
        u64v in0, in1;  // caller 0's operands, for the check
        for (int k = 0; k < K; k++) {
            const u64v a = uniform(rng, q, N, 2), b = uniform(rng, q, N, 2);
            if (k == 0) { in0 = a; in1 = b; }
            for (int c = 0; c < 2; c++) {
                hering::Poly pa = ringQ.NewScratch(), pb = ringQ.NewScratch();
                pa.Upload(a.data() + c * words, words);
                pb.Upload(b.data() + c * words, words);
                callers[k].a.Value.push_back(pa);
                callers[k].b.Value.push_back(pb);
                callers[k].out.Value.push_back(ringQ.NewPoly());
            }
        }
        auto run = [&](int n) {
            std::atomic<int> ready{0};
            std::atomic<bool> go{false};
            std::vector<std::thread> th;
            for (int k = 0; k < K; k++)
                th.emplace_back([&, k] {
                    ready++;
                    while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
                    for (int i = 0; i < n; i++) {
                        eval.MulRelinBGV(T, callers[k].a, callers[k].b, &rlk, callers[k].out);  // one ciphertext per call
                        if (sync_each) ctx.Sync();
                    }
                });
            while (ready.load() < K) std::this_thread::yield();
            const auto t0 = std::chrono::steady_clock::now();
            go.store(true, std::memory_order_release);
            for (auto &t : th) t.join();
            ctx.Sync();
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        };
        run(8);  // warm-up: plans, scratch arena, the queue's batch-size history
        const double dt = run(iters);
        // parity of what was timed: caller 0's last result against the oracle
        lo_ring *oQ = lo_ring_new(N, q.data(), nq), *oP = lo_ring_new(N, p.data(), np);
        lo_evaluator *oev = lo_evaluator_new(oQ, oP);
        lo_evk oevk{};
        oevk.beta = beta; oevk.nQk = nq; oevk.nPk = np; oevk.q = kq.data(); oevk.p = kp.data();
        for (int &v : oevk.nj) v = 1;
        u64v want(2 * words);
        lo_bgv_mul_relin(oev, level, T, in0.data(), in1.data(), &oevk, 1, want.data());
        const u64v g0 = callers[0].out.Value[0].Download(), g1 = callers[0].out.Value[1].Download();
        const bool ok = std::equal(g0.begin(), g0.end(), want.begin()) && std::equal(g1.begin(), g1.end(), want.begin() + words);
        std::printf("{\"host\": \"C++ (include/hering.hpp), std::thread per caller, public interface only\", \"K\": %d, \"calls_per_caller\": %d, "
                    "\"sync_each\": %s, \"coalescing\": %s, \"ops_per_s\": %.1f, \"verified\": %s}\n",
                    K, iters, sync_each ? "true" : "false", coalesce ? "true" : "false", (double)K * iters / dt, ok ? "true" : "false");
        return ok ? 0 : 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "run_parallel: %s\n", e.what());
        return 3;
    }
}
