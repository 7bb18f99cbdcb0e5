// replay.cpp -- he_debug_replay (include/hering_debug.h): K OS threads replaying a recorded sequence of ABI calls, each on its own
// polynomials.  Diagnostics, outside the drop-in boundary: the measurement harness of bench.py's `concurrent_b1` for whole circuits.
//
// The reference benchmarks a circuit under concurrency by running it from many goroutines at once, one ciphertext each
// (BenchmarkConcurrentBootstrap: b.RunParallel over bootstrappers, circuits/ckks/bootstrapping/evaluator_benchmarks_test.go:14-42).
// The drivers above the operator API exist here only as Python restatements (tests/drivers), and K interpreter threads serialise on
// the interpreter's lock.  So the call sequence of ONE run of the driver is recorded (lattigo_amd/_lib.py: trace_begin / trace_end:
// every he_* call with its arguments) and replayed by K pthreads: the same calls through the same public entry points, every handle
// the recorded run allocated replaced by one of the replaying thread's own, the run's input handles replaced per thread (`subst`),
// everything else (rings, evaluator, keys, plaintext diagonals) shared -- exactly what K callers of the one-ciphertext interface do.
//
// Program encoding (uint64 words): per call [fn, nargs, args...]; an argument is [kind, payload...]:
//   0 immediate value | 1 handle (mapped through the thread's table if present) | 2 OUT handle: the call creates an object, the
//   recorded value becomes the key of the thread's new handle | 3 u64 array [len, words...] | 4 handle array [len, handles...] | 5 null
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/hering.h"
#include "../../include/hering_debug.h"

namespace {
enum Fn : uint64_t {
    F_POLY_ALLOC = 0, F_POLY_ALLOC_SCRATCH, F_POLY_FREE, F_POLY_COPY, F_POLY_COPY_BATCH, F_POLY_ZERO,
    F_NTT, F_NTT_LAZY, F_INTT, F_INTT_LAZY, F_BINOP, F_UNOP, F_SCALAROP, F_MUL_RNS_SCALAR, F_ADD_BIGINT, F_SUB_BIGINT, F_MUL_BIGINT,
    F_MUL_BIGINT_THEN_ADD, F_DOUBLE_RNS, F_SHIFT, F_MONOMIAL, F_MUL_BY_VECTOR, F_DIV, F_RESCALE_POLYS, F_INDEX_CREATE, F_INDEX_DESTROY,
    F_AUTO_INDEX, F_AUTO_INDEX_ADD, F_AUTO_COEFF, F_MODUP_QP, F_MODUP_PQ, F_MODDOWN_BE, F_EVAL_MODDOWN, F_DECOMP_CREATE, F_DECOMP_DESTROY,
    F_DECOMPOSE_NTT, F_GP_LAZY, F_GP_HOISTED_LAZY, F_MODDOWN, F_GP, F_GP_HOISTED, F_RELIN, F_AUTO_CT, F_AUTO_HOISTED, F_AUTO_HOISTED_LAZY,
    F_CENTERED_LIFT, F_DECOMP_FILL, F_LINTRANS, F_CKKS_MUL, F_BGV_MUL, F_GIANT_STEP, F_COUNT
};
struct Arg {
    uint64_t kind = 0, val = 0;
    std::vector<uint64_t> arr;
};
struct Call {
    uint64_t fn;
    std::vector<Arg> a;
};
struct Worker {
    int idx, rounds, rc = 0;
    const std::vector<Call> *prog;
    std::unordered_map<uint64_t, uint64_t> base;  // recorded handle -> this thread's (the substituted inputs)
    const uint64_t *watch;
    int n_watch;
    uint64_t *watch_out;
    he_handle ctx;
    std::atomic<int> *go;  // 0: wait, 1: run, -1: give up (a thread could not be started)
    double t0 = 0, t1 = 0;
    std::string err;
};
std::atomic<uint64_t> g_fn_us[64], g_fn_n[64], g_fn_max[64];  // per function: microseconds inside the calls, calls, longest call (he_debug_replay_profile)
double mono_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int run_call(const Call &c, std::unordered_map<uint64_t, uint64_t> &map, std::vector<std::pair<uint64_t, uint64_t>> &made) {
    const std::vector<Arg> &a = c.a;
    auto H = [&](size_t i) -> he_handle {
        if (i >= a.size() || a[i].kind == 5) return 0;
        auto it = map.find(a[i].val);
        return it == map.end() ? a[i].val : it->second;
    };
    auto I = [&](size_t i) -> uint64_t { return i < a.size() ? a[i].val : 0; };
    auto A = [&](size_t i) -> const uint64_t * { return (i < a.size() && a[i].kind == 3) ? a[i].arr.data() : nullptr; };
    std::vector<std::vector<he_handle>> harr(a.size());
    auto HA = [&](size_t i) -> const he_handle * {
        if (i >= a.size() || a[i].kind != 4) return nullptr;
        harr[i].resize(a[i].arr.size());
        for (size_t k = 0; k < a[i].arr.size(); k++) {
            auto it = map.find(a[i].arr[k]);
            harr[i][k] = it == map.end() ? a[i].arr[k] : it->second;
        }
        return harr[i].data();
    };
    he_handle out = 0;
    auto OUT = [&](size_t i, int rc, int kind) -> int {  // the object a call created belongs to this thread
        if (rc == 0) { map[a[i].val] = out; made.emplace_back(out, (uint64_t)kind); }
        return rc;
    };
    // an object the recorded run had before the recording started is shared by the threads: they do not destroy it
    auto OWN = [&](size_t i) -> bool { return map.find(a[i].val) != map.end(); };
    auto DROP = [&](size_t i) {
        const he_handle h = H(i);
        map.erase(a[i].val);
        for (auto &m : made) if (m.first == h) m.first = 0;
    };
    switch (c.fn) {
        case F_POLY_ALLOC: return OUT(3, he_poly_alloc(H(0), (int)I(1), (int)I(2), &out), 0);
        case F_POLY_ALLOC_SCRATCH: return OUT(3, he_poly_alloc_scratch(H(0), (int)I(1), (int)I(2), &out), 0);
        case F_POLY_FREE: { if (!OWN(0)) return 0; const he_handle h = H(0); DROP(0); return he_poly_free(h); }
        case F_POLY_COPY: return he_poly_copy(H(0), H(1), (int)I(2));
        case F_POLY_COPY_BATCH: return he_poly_copy_batch(H(0), (int)I(1), H(2), (int)I(3), (int)I(4), (int)I(5));
        case F_POLY_ZERO: return he_poly_zero(H(0));
        case F_NTT: return he_ntt(H(0), (int)I(1), H(2), H(3));
        case F_NTT_LAZY: return he_ntt_lazy(H(0), (int)I(1), H(2), H(3));
        case F_INTT: return he_intt(H(0), (int)I(1), H(2), H(3));
        case F_INTT_LAZY: return he_intt_lazy(H(0), (int)I(1), H(2), H(3));
        case F_BINOP: return he_binop(H(0), (int)I(1), (int)I(2), H(3), H(4), H(5));
        case F_UNOP: return he_unop(H(0), (int)I(1), (int)I(2), H(3), H(4));
        case F_SCALAROP: return he_scalarop(H(0), (int)I(1), (int)I(2), H(3), I(4), H(5));
        case F_MUL_RNS_SCALAR: return he_mul_rns_scalar_montgomery(H(0), (int)I(1), H(2), A(3), H(4));
        case F_ADD_BIGINT: return he_add_scalar_bigint(H(0), (int)I(1), H(2), A(3), (int)I(4), H(5));
        case F_SUB_BIGINT: return he_sub_scalar_bigint(H(0), (int)I(1), H(2), A(3), (int)I(4), H(5));
        case F_MUL_BIGINT: return he_mul_scalar_bigint(H(0), (int)I(1), H(2), A(3), (int)I(4), H(5));
        case F_MUL_BIGINT_THEN_ADD: return he_mul_scalar_bigint_then_add(H(0), (int)I(1), H(2), A(3), (int)I(4), H(5));
        case F_DOUBLE_RNS: return he_double_rns_scalarop(H(0), (int)I(1), (int)I(2), H(3), A(4), A(5), H(6));
        case F_SHIFT: return he_shift(H(0), (int)I(1), H(2), (int)(int64_t)I(3), H(4));
        case F_MONOMIAL: return he_mult_by_monomial(H(0), (int)I(1), H(2), (int)(int64_t)I(3), H(4));
        case F_MUL_BY_VECTOR: return he_mul_by_vector_montgomery(H(0), (int)I(1), H(2), H(3), (int)I(4), H(5));
        case F_DIV: {  // [ring, level, nb, p0, p1, variant]: variant = round | ntt << 1 | many << 2
            const int v = (int)I(5);
            if ((v & 3) == 3) return he_div_round_by_last_modulus_many_ntt(H(0), (int)I(1), (int)I(2), H(3), H(4));
            if ((v & 3) == 1) return he_div_round_by_last_modulus_many(H(0), (int)I(1), (int)I(2), H(3), H(4));
            if ((v & 3) == 2) return he_div_floor_by_last_modulus_many_ntt(H(0), (int)I(1), (int)I(2), H(3), H(4));
            return he_div_floor_by_last_modulus_many(H(0), (int)I(1), (int)I(2), H(3), H(4));
        }
        case F_RESCALE_POLYS: return he_rescale_polys(H(0), (int)I(1), (int)I(2), (int)I(3), HA(4), HA(5));
        case F_INDEX_CREATE: return OUT(2, he_automorphism_index_create(H(0), I(1), &out), 1);
        case F_INDEX_DESTROY: { if (!OWN(0)) return 0; const he_handle h = H(0); DROP(0); return he_automorphism_index_destroy(h); }
        case F_AUTO_INDEX: return he_automorphism_ntt_with_index(H(0), (int)I(1), H(2), H(3), H(4));
        case F_AUTO_INDEX_ADD: return he_automorphism_ntt_with_index_then_add_lazy(H(0), (int)I(1), H(2), H(3), H(4));
        case F_AUTO_COEFF: return he_automorphism(H(0), (int)I(1), H(2), I(3), H(4));
        case F_MODUP_QP: return he_modup_q_to_p(H(0), (int)I(1), (int)I(2), H(3), H(4));
        case F_MODUP_PQ: return he_modup_p_to_q(H(0), (int)I(1), (int)I(2), H(3), H(4));
        case F_MODDOWN_BE: {  // [be, lq, lp, a, b, c, kind]
            const int k = (int)I(6);
            if (k == 0) return he_moddown_qp_to_q(H(0), (int)I(1), (int)I(2), H(3), H(4), H(5));
            if (k == 1) return he_moddown_qp_to_q_ntt(H(0), (int)I(1), (int)I(2), H(3), H(4), H(5));
            return he_moddown_qp_to_p(H(0), (int)I(1), (int)I(2), H(3), H(4), H(5));
        }
        case F_EVAL_MODDOWN: return he_eval_moddown_qp_to_q_ntt(H(0), (int)I(1), (int)I(2), H(3), H(4), H(5));
        case F_DECOMP_CREATE: return OUT(2, he_decomp_create(H(0), (int)I(1), &out), 2);
        case F_DECOMP_DESTROY: { if (!OWN(0)) return 0; const he_handle h = H(0); DROP(0); return he_decomp_destroy(h); }
        case F_DECOMPOSE_NTT: return he_decompose_ntt(H(0), (int)I(1), (int)I(2), (int)I(3), H(4), (int)I(5), H(6));
        case F_GP_LAZY: return he_gadget_product_lazy(H(0), (int)I(1), H(2), H(3), H(4), H(5), H(6), H(7));
        case F_GP_HOISTED_LAZY: return he_gadget_product_hoisted_lazy(H(0), (int)I(1), H(2), H(3), H(4), H(5), H(6), H(7));
        case F_MODDOWN: return he_moddown(H(0), (int)I(1), (int)(int64_t)I(2), H(3), H(4), H(5), H(6), H(7), H(8));
        case F_GP: return he_gadget_product(H(0), (int)I(1), H(2), H(3), H(4), H(5));
        case F_GP_HOISTED: return he_gadget_product_hoisted(H(0), (int)I(1), H(2), H(3), H(4), H(5));
        case F_RELIN: return he_relinearize(H(0), (int)I(1), H(2), H(3), H(4), H(5), H(6), H(7));
        case F_AUTO_CT: return he_automorphism_ct(H(0), (int)I(1), H(2), H(3), I(4), H(5), H(6), H(7));
        case F_AUTO_HOISTED: return he_automorphism_hoisted(H(0), (int)I(1), H(2), H(3), I(4), H(5), H(6), H(7));
        case F_AUTO_HOISTED_LAZY: return he_automorphism_hoisted_lazy(H(0), (int)I(1), H(2), H(3), I(4), H(5), H(6), H(7), H(8), H(9));
        case F_CENTERED_LIFT: return he_centered_lift(H(0), (int)I(1), H(2), (int)I(3), (int)I(4), H(5), (int)(int64_t)I(6), H(7));
        case F_DECOMP_FILL: return he_decomp_fill(H(0), (int)I(1), (int)I(2), H(3), H(4));
        case F_LINTRANS: return he_lintrans_mul_sum(H(0), (int)I(1), (int)I(2), (int)I(3), HA(4), HA(5), HA(6), HA(7), HA(8), HA(9), HA(10),
                                                    (int)I(11), H(12), H(13), H(14), H(15));
        case F_CKKS_MUL: return he_ckks_mul_relin(H(0), (int)I(1), H(2), H(3), H(4), H(5), H(6), H(7), H(8), H(9));
        case F_BGV_MUL: return he_bgv_mul_relin(H(0), (int)I(1), I(2), H(3), H(4), H(5), H(6), H(7), H(8), H(9), H(10));
        case F_GIANT_STEP: return he_lintrans_giant_step(H(0), (int)I(1), H(2), H(3), I(4), H(5), H(6), H(7), H(8), H(9), H(10), (int)I(11));
        default: return HE_EINVAL;
    }
}
void destroy(uint64_t h, uint64_t kind) {
    if (!h) return;
    if (kind == 0) he_poly_free(h);
    else if (kind == 1) he_automorphism_index_destroy(h);
    else he_decomp_destroy(h);
}
void *worker(void *vp) {
    Worker &w = *(Worker *)vp;
    int g;
    while ((g = w.go->load(std::memory_order_acquire)) == 0) sched_yield();
    if (g < 0) return nullptr;
    w.t0 = mono_s();
    for (int round = 0; round < w.rounds && w.rc == 0; round++) {
        std::unordered_map<uint64_t, uint64_t> map = w.base;
        std::vector<std::pair<uint64_t, uint64_t>> made;  // (handle, kind) of the objects this round created and has not destroyed
        for (const Call &c : *w.prog) {
            const double c0 = mono_s();
            w.rc = run_call(c, map, made);
            const uint64_t us = (uint64_t)((mono_s() - c0) * 1e6);
            g_fn_us[c.fn].fetch_add(us, std::memory_order_relaxed);
            g_fn_n[c.fn].fetch_add(1, std::memory_order_relaxed);
            uint64_t mx = g_fn_max[c.fn].load(std::memory_order_relaxed);
            while (us > mx && !g_fn_max[c.fn].compare_exchange_weak(mx, us, std::memory_order_relaxed)) {}
            if (w.rc != 0) { w.err = std::string("call of function ") + std::to_string(c.fn) + ": " + he_last_error(); break; }
        }
        const bool last = round + 1 == w.rounds;
        if (w.rc == 0 && last) {  // the watched results survive: the caller downloads and frees them
            for (int i = 0; i < w.n_watch; i++) {
                auto it = map.find(w.watch[i]);
                w.watch_out[i] = it == map.end() ? 0 : it->second;
                for (auto &m : made) if (m.first == w.watch_out[i]) m.first = 0;
            }
        }
        // what the recorded run left alive at the end of the recording (its result, the driver's survivors) is this round's garbage
        for (auto &m : made) destroy(m.first, m.second);
    }
    if (w.rc == 0) { w.rc = he_ctx_sync(w.ctx); if (w.rc) w.err = he_last_error(); }
    w.t1 = mono_s();
    return nullptr;
}
}  // namespace

// per function number of the program encoding: [3 f + 0] microseconds spent inside its calls (all threads), [3 f + 1] calls,
// [3 f + 2] the longest single call, since the last reset
extern "C" int he_debug_replay_profile(uint64_t *out, int n_fn, int reset) {
    for (int f = 0; f < n_fn && f < 64; f++) {
        if (out) { out[3 * f] = g_fn_us[f].load(); out[3 * f + 1] = g_fn_n[f].load(); out[3 * f + 2] = g_fn_max[f].load(); }
        if (reset) { g_fn_us[f] = 0; g_fn_n[f] = 0; g_fn_max[f] = 0; }
    }
    return HE_OK;
}
extern "C" int he_debug_replay(he_handle ctx, const uint64_t *program, size_t n_words, int n_threads, int rounds, const uint64_t *subst_from,
                               int n_subst, const uint64_t *subst_to, const uint64_t *watch, int n_watch, uint64_t *watch_out, double *wall_s,
                               char *err, size_t err_len) {
    auto say = [&](const std::string &m) { if (err && err_len) { std::strncpy(err, m.c_str(), err_len - 1); err[err_len - 1] = 0; } };
    if (!program || n_threads <= 0 || n_threads > 1024 || rounds <= 0 || !wall_s || (n_subst > 0 && (!subst_from || !subst_to)) ||
        (n_watch > 0 && (!watch || !watch_out))) { say("bad arguments"); return HE_EINVAL; }
    std::vector<Call> prog;
    size_t i = 0;
    while (i < n_words) {
        if (i + 2 > n_words) { say("truncated program"); return HE_EINVAL; }
        Call c;
        c.fn = program[i++];
        const uint64_t na = program[i++];
        if (c.fn >= F_COUNT || na > 32) { say("unknown function or argument count"); return HE_EINVAL; }
        for (uint64_t k = 0; k < na; k++) {
            if (i >= n_words) { say("truncated program"); return HE_EINVAL; }
            Arg a;
            a.kind = program[i++];
            if (a.kind == 3 || a.kind == 4) {
                if (i >= n_words || i + 1 + program[i] > n_words) { say("truncated array"); return HE_EINVAL; }
                const uint64_t len = program[i++];
                a.arr.assign(program + i, program + i + len);
                i += len;
            } else if (a.kind != 5) {
                if (i >= n_words) { say("truncated program"); return HE_EINVAL; }
                a.val = program[i++];
            }
            c.a.push_back(std::move(a));
        }
        prog.push_back(std::move(c));
    }
    std::vector<Worker> ws(n_threads);
    std::vector<pthread_t> th(n_threads);
    std::atomic<int> go{0};
    int started = 0;
    for (int t = 0; t < n_threads; t++) {
        Worker &w = ws[t];
        w.idx = t; w.rounds = rounds; w.prog = &prog; w.ctx = ctx; w.go = &go;
        for (int s = 0; s < n_subst; s++) w.base[subst_from[s]] = subst_to[(size_t)t * n_subst + s];
        w.watch = watch; w.n_watch = n_watch; w.watch_out = watch_out ? watch_out + (size_t)t * n_watch : nullptr;
    }
    // all threads or none: they wait for `go`, which says run only when every one of them exists
    for (int t = 0; t < n_threads; t++) {
        if (pthread_create(&th[t], nullptr, worker, &ws[t]) != 0) break;
        started++;
    }
    go.store(started == n_threads ? 1 : -1, std::memory_order_release);
    for (int t = 0; t < started; t++) pthread_join(th[t], nullptr);
    if (started != n_threads) { say("could not start the threads"); return HE_ENOMEM; }
    double lo = ws[0].t0, hi = ws[0].t1;
    for (const Worker &w : ws) { lo = std::min(lo, w.t0); hi = std::max(hi, w.t1); }
    *wall_s = hi - lo;
    for (const Worker &w : ws)
        if (w.rc != 0) { say("thread " + std::to_string(w.idx) + ": " + w.err); return w.rc; }
    return HE_OK;
}
