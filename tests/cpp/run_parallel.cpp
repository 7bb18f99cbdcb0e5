// run_parallel.cpp -- the reference's parallel benchmark shape (b.RunParallel over shallow copies of one evaluator,
// schemes/ckks/ckks_benchmarks_test.go:116-207; schemes/bgv's MulRelin at the BASELINE config-3 shape) from a COMPILED host
// through the public interface only (include/hering.hpp): K OS threads, one ciphertext per call, one shared evaluator whose
// context's submission queue turns the concurrent calls into batched launches.  Prints one JSON line; bench.py embeds it.
//
//   run_parallel [K=64] [calls per thread=96] [sync_each=0|1] [coalesce=1|0] [workload=c3|c2] [max_batch=64] [window_us=30] [deferred depth=0]
//
// workload c3: BGV MulRelin, logN = 15, 12 + 3 limbs (BASELINE config 3).
// workload c2: CKKS Mul (degree 2, no relinearisation) + Rescale of the three polynomials, logN = 14, 8 + 1 limbs (BASELINE
//              config 2; schemes/ckks/evaluator.go:764-872, :477-515) -- two calls of the interface per operation (Mul, Rescale).
// EVERY caller's last result is compared with the oracle (on the host's threads), not a sample.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>
#include <sys/resource.h>
#include <time.h>

#include "hering.hpp"
#include "hering_debug.h"
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
    const std::string workload = argc > 5 ? argv[5] : "c3";
    const int max_batch = argc > 6 ? std::atoi(argv[6]) : 64, window_us = argc > 7 ? std::atoi(argv[7]) : 30;
    const int deferred = argc > 8 ? std::atoi(argv[8]) : 0;  // he_ctx_set_deferred: calls return once filed
    const bool c2 = workload == "c2";
    if (!c2 && workload != "c3") { std::fprintf(stderr, "run_parallel: workload c3 or c2\n"); return 2; }
    const int logN = c2 ? 14 : 15, N = 1 << logN;
    const uint64_t T = 65537;
    std::vector<int> logq(c2 ? 8 : 12, c2 ? 40 : 45), logp(c2 ? 1 : 3, c2 ? 60 : 55);
    logq[0] = c2 ? 50 : 55;
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
        if (coalesce) { ctx.SetCoalescing(max_batch, window_us); ctx.SetDeferred(deferred); } else ctx.SetCoalescing(0, 0);
        const size_t words = (size_t)nq * N;
        struct Caller { hering::Ciphertext a, b, out, res; };
        std::vector<Caller> callers(K);
        std::vector<u64v> in0(K), in1(K);  // every caller's operands, for the check
        for (int k = 0; k < K; k++) {
            in0[k] = uniform(rng, q, N, 2); in1[k] = uniform(rng, q, N, 2);
            for (int c = 0; c < 2; c++) {
                hering::Poly pa = ringQ.NewScratch(), pb = ringQ.NewScratch();
                pa.Upload(in0[k].data() + c * words, words);
                pb.Upload(in1[k].data() + c * words, words);
                callers[k].a.Value.push_back(pa);
                callers[k].b.Value.push_back(pb);
            }
            for (int c = 0; c < (c2 ? 3 : 2); c++) {
                callers[k].out.Value.push_back(ringQ.NewPoly());
                if (c2) callers[k].res.Value.push_back(ringQ.AtLevel(level - 1).NewPoly());
            }
        }
        std::atomic<long long> caller_cpu_ns{0};  // CPU time the callers spent inside their loops (asleep does not count)
        auto thread_cpu_ns = [] { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec; };
        auto run = [&](int n) {
            std::atomic<int> ready{0};
            std::atomic<bool> go{false};
            std::vector<std::thread> th;
            for (int k = 0; k < K; k++)
                th.emplace_back([&, k] {
                    ready++;
                    while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
                    const long long c0 = thread_cpu_ns();
                    struct Acc { std::atomic<long long> &a; long long c0; decltype(thread_cpu_ns) &f; ~Acc() { a += f() - c0; } } acc{caller_cpu_ns, c0, thread_cpu_ns};
                    for (int i = 0; i < n; i++) {  // one ciphertext per call
                        if (c2) {
                            eval.MulRelinCKKS(callers[k].a, callers[k].b, nullptr, callers[k].out);
                            eval.Rescale(1, callers[k].out, callers[k].res);
                        } else {
                            eval.MulRelinBGV(T, callers[k].a, callers[k].b, &rlk, callers[k].out);
                        }
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
        uint64_t st0[4] = {0, 0, 0, 0}, st1[4] = {0, 0, 0, 0};
        he_ctx_coalescing_stats(ctx.h(), st0);
        caller_cpu_ns = 0;
        rusage ru0{}, ru1{};
        getrusage(RUSAGE_SELF, &ru0);
        const double dt = run(iters);
        getrusage(RUSAGE_SELF, &ru1);
        const double proc_cpu_s = (ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) + (ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec) * 1e-6 +
                                  (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec) * 1e-6;
        const double sys_s = (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec) * 1e-6;
        he_ctx_coalescing_stats(ctx.h(), st1);
        const double mean_batch = st1[1] > st0[1] ? (double)(st1[0] - st0[0]) / (double)(st1[1] - st0[1]) : 0.0;
        if (std::getenv("HERING_QUEUE_DEBUG")) {  // where the queue's launching thread spent the run (hering_debug.h)
            uint64_t dbg[16] = {0};
            he_debug_queue_counters(ctx.h(), dbg);
            std::fprintf(stderr, "queue counters: everyone %llu timeout %llu full %llu | gather %llu us launch %llu us | ahead-cap %llu us, "
                                 "waiting (device busy) %llu us, waiting (device dry) %llu us, batches %llu | tables filled %llu reused %llu | wall %.0f us\n",
                         (unsigned long long)dbg[0], (unsigned long long)dbg[1], (unsigned long long)dbg[2], (unsigned long long)dbg[3],
                         (unsigned long long)dbg[4], (unsigned long long)dbg[8], (unsigned long long)dbg[9], (unsigned long long)dbg[10],
                         (unsigned long long)dbg[11], (unsigned long long)dbg[13], (unsigned long long)dbg[14], dt * 1e6);
        }
        // parity of what was timed: EVERY caller's last result against the oracle, on the host's threads
        lo_ring *oQ = lo_ring_new(N, q.data(), nq), *oP = lo_ring_new(N, p.data(), np);
        lo_evaluator *oev = lo_evaluator_new(oQ, oP);
        lo_evk oevk{};
        oevk.beta = beta; oevk.nQk = nq; oevk.nPk = np; oevk.q = kq.data(); oevk.p = kp.data();
        for (int &v : oevk.nj) v = 1;
        u64v op0((size_t)K * 2 * words), op1((size_t)K * 2 * words);
        for (int k = 0; k < K; k++) {
            std::memcpy(op0.data() + (size_t)k * 2 * words, in0[k].data(), 2 * words * 8);
            std::memcpy(op1.data() + (size_t)k * 2 * words, in1[k].data(), 2 * words * 8);
        }
        const size_t per = c2 ? (size_t)3 * (nq - 1) * N : 2 * words;  // lo_batch_op: [nb][3][level][N] (kind 2) / [nb][2][level+1][N]
        u64v want((size_t)K * per);
        const int host_threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        lo_batch_op(oev, c2 ? 2 : 0, level, c2 ? 0 : T, 0, op0.data(), op1.data(), c2 ? nullptr : &oevk, K, host_threads, want.data());
        int bad = 0;
        for (int k = 0; k < K; k++) {
            const hering::Ciphertext &r = c2 ? callers[k].res : callers[k].out;
            const size_t w = c2 ? (size_t)(nq - 1) * N : words;
            bool ok = true;
            for (size_t c = 0; c < r.Value.size(); c++) {
                const u64v g = r.Value[c].Download();
                ok = ok && g.size() >= w && std::equal(g.begin(), g.begin() + w, want.begin() + (size_t)k * per + c * w);
            }
            bad += ok ? 0 : 1;
        }
        std::printf("{\"host\": \"C++ (include/hering.hpp), std::thread per caller, public interface only\", \"workload\": \"%s\", \"K\": %d, "
                    "\"calls_per_caller\": %d, \"interface_calls_per_op\": %d, \"sync_each\": %s, \"coalescing\": %s, \"max_batch\": %d, "
                    "\"window_us\": %d, \"deferred_depth\": %d, \"mean_batch\": %.1f, \"ops_per_s\": %.1f, \"caller_cpu_us_per_interface_call\": %.2f, "
                    "\"process_cpu_cores_used\": %.1f, \"process_sys_share\": %.2f, \"verified\": %s, \"verified_callers\": \"%d/%d\"}\n",
                    workload.c_str(), K, iters, c2 ? 2 : 1, sync_each ? "true" : "false", coalesce ? "true" : "false", max_batch, window_us, coalesce ? deferred : 0,
                    mean_batch, (double)K * iters / dt, caller_cpu_ns.load() * 1e-3 / ((double)K * iters * (c2 ? 2 : 1)), proc_cpu_s / dt, sys_s / std::max(proc_cpu_s, 1e-9), bad == 0 ? "true" : "false", K - bad, K);
        return bad == 0 ? 0 : 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "run_parallel: %s\n", e.what());
        return 3;
    }
}
