// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of libhering.
//
// Everything here is 64-bit integer modular arithmetic over RNS limbs: no MFMA
// (there is no dense low-precision contraction on this path).  The design rules
// are the HBM/LDS ones: limb-major coalesced loads, twiddles and butterfly stages
// staged through LDS / registers, one launch covering every (limb x batch entry).
//
// NTT decomposition (N = 2^n, n <= 20; the fused pipelines below stop at n = 17, beyond that the generic passes run):  a = max(0, n-12) "column" stages are done
// in registers on elements strided by N/2^a (ntt_cols), the remaining b = n-a <= 12
// stages on contiguous rows of 2^b coefficients held in LDS, in rounds of up to four
// stages with 16 coefficients per thread in registers (ntt_rows).  The transform is
// the reference's (ring/ntt.go:223-257, :570-606): Cooley-Tukey natural->bit-reversed
// forward with twiddle RootsForward[m+i], Gentleman-Sande inverse, same tables; the
// butterflies use the Harvey lazy form on [0,4q) / [0,2q) and only canonical
// outputs are promised (SURVEY.md section 8a note on lazy ranges).
#include "kernels.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace he {

// ------------------------------------------------------------------------------------
// optional per-launch HIP-event timing (off by default; bench.py's roofline leg turns it on)
// ------------------------------------------------------------------------------------
namespace {
// Profiling state is per stream (= per context, see api.cpp) behind one mutex: two contexts may launch from different
// threads while one of them is being profiled; a launch looks its own stream up and records only there.
struct ProfRec { int id; hipEvent_t e0, e1; double bytes; };
struct ProfState { std::vector<ProfRec> recs; };
std::mutex g_prof_mu;
std::atomic<int> g_prof_active{0};                       // number of streams being profiled (fast path: none)
std::unordered_map<hipStream_t, ProfState> g_prof;       // guarded by g_prof_mu
std::vector<hipEvent_t> g_prof_pool;                     // guarded by g_prof_mu
hipEvent_t prof_event_locked() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
// `bytes` = the ALGORITHMIC HBM bytes of the launch: every polynomial stream the kernel must read or write, once (a key row
// shared by the batch counts once; twiddles, constants and index tables are resident and excluded) -- the numerator of the
// per-kernel roofline figures bench.py reports, kept next to the launch so that it cannot drift from the pipeline
struct ProfScope {
    bool on; int id; hipStream_t s; hipEvent_t e0; double bytes;
    ProfScope(int id_, hipStream_t s_, double bytes_ = 0.0) : on(false), id(id_), s(s_), e0(nullptr), bytes(bytes_) {
        if (g_prof_active.load(std::memory_order_relaxed) == 0) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof.find(s) == g_prof.end()) return;
        on = true;
        e0 = prof_event_locked();
        (void)hipEventRecord(e0, s);
    }
    ~ProfScope() {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        auto it = g_prof.find(s);
        hipEvent_t e1 = prof_event_locked();
        (void)hipEventRecord(e1, s);
        if (it != g_prof.end()) it->second.recs.push_back(ProfRec{id, e0, e1, bytes});
        else { g_prof_pool.push_back(e0); g_prof_pool.push_back(e1); }
    }
};
}  // namespace
const char *kernel_name(int id) {
    static const char *names[K_COUNT] = {"ntt_cols_fwd", "ntt_rows_fwd", "ntt_rows_inv", "ntt_cols_inv", "ew", "gather",
                                         "automorphism_coeff", "build_index", "modup", "center_copy", "ks_inner",
                                         "tensor", "modmul_probe", "ci_fold", "mask_spread", "ntt_rows_fwd_f64",
                                         "ntt_rows_inv_f64", "ntt_mac_f64", "diag_mac"};
    return (id >= 0 && id < K_COUNT) ? names[id] : "?";
}
bool prof_active(hipStream_t s) {
    if (g_prof_active.load(std::memory_order_relaxed) == 0) return false;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    return g_prof.find(s) != g_prof.end();
}
void prof_begin(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto ins = g_prof.emplace(s, ProfState{});
    if (ins.second) g_prof_active.fetch_add(1);
    for (auto &r : ins.first->second.recs) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }  // a begin without an end
    ins.first->second.recs.clear();
}
int prof_end(hipStream_t s, int *counts, float *total_ms, double *total_bytes) {
    for (int i = 0; i < K_COUNT; i++) { counts[i] = 0; total_ms[i] = 0.f; if (total_bytes) total_bytes[i] = 0.0; }
    std::vector<ProfRec> recs;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        auto it = g_prof.find(s);
        if (it == g_prof.end()) return 0;
        recs.swap(it->second.recs);
        g_prof.erase(it);
        g_prof_active.fetch_sub(1);
    }
    for (auto &r : recs) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.e1);
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        counts[r.id]++; total_ms[r.id] += ms;
        if (total_bytes) total_bytes[r.id] += r.bytes;
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : recs) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
    return (int)recs.size();
}

// Streaming accesses: polynomial data is read once per pass, while twiddle and key rows are shared by the workgroups of a
// launch -- the non-temporal hint on the loads keeps L2 for the latter.  Stores carry the hint only where the consumer is
// several large kernels away (decomposition, key-switch accumulators, fused epilogues: +3 % on Rotate); the plain NTT passes,
// the tensor and the element-wise kernels feed the next launch from the infinity cache and lose 2-5 % with it.
template <class T> __device__ __forceinline__ T ldnt(const T *p) { return __builtin_nontemporal_load(p); }
template <class T> __device__ __forceinline__ void stnt(T *p, T v) { __builtin_nontemporal_store(v, p); }
// the 16-byte forms of the two-coefficients-per-thread kernels
typedef unsigned long long he_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ulonglong2 ldnt2(const uint64_t *p) {
    const he_u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const he_u64x2 *>(p));
    return make_ulonglong2(v.x, v.y);
}

// block-uniform 64-bit constants through the scalar cache: the constant address space tells the compiler that the table is
// never written by a kernel, so a load with a uniform address becomes an s_load instead of a vector load + wait
typedef const uint64_t __attribute__((address_space(4))) *he_cptr64;
__device__ __forceinline__ uint64_t ldc(const uint64_t *p, size_t i) { return ((he_cptr64)(uintptr_t)p)[i]; }
typedef const double __attribute__((address_space(4))) *he_cptrd;
__device__ __forceinline__ double ldcd(const double *p, size_t i) { return ((he_cptrd)(uintptr_t)p)[i]; }
__device__ __forceinline__ uint64_t mad32(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }  // v_mad_u64_u32
// word offset of batch entry z of a view (View::tab): through the entry table when the launch has one (block-uniform z: a scalar
// load), z * bstride otherwise
__device__ __forceinline__ size_t voff(const size_t *tab, size_t bs, size_t z) {
    return tab ? (size_t)ldc(reinterpret_cast<const uint64_t *>(tab), z) : z * bs;
}
// Automorphism fused into an epilogue's stores (ring/automorphism.go:50-77 applied where the result is written): the NTT-domain
// automorphism of Galois element g is the gather out[j] = in[index_g[j]], i.e. source coefficient e lands at index_{g^-1}[e] --
// computed here (two bit reversals and a multiply, AutomorphismNTTIndex :12-34 for NthRoot = 2N) instead of loaded.  The map sends
// every aligned block of 64 consecutive coefficients onto an aligned block of 64 (the low six index bits are the high exponent
// bits, which g permutes among themselves), so a wave's store still fills whole cache lines.  ginv = g^-1 mod 2N, 0 = no scatter.
__device__ __forceinline__ unsigned auto_dest(unsigned e, unsigned ginv, int logN) {
    const unsigned t1 = 2u * (__brev(e) >> (32 - logN)) + 1u;
    const unsigned t2 = (((ginv * t1) & ((2u << logN) - 1u)) - 1u) >> 1;
    return __brev(t2) >> (32 - logN);
}
static inline bool no_tab(std::initializer_list<View> vs) {
    for (const View &v : vs) if (v.tab) return false;
    return true;
}


// XCD-aware work order (MI355X: 8 XCDs with private L2s, workgroup b runs on XCD b % 8): workgroup `lin` of a launch takes
// work item (lin % 8) * (n / 8) + lin / 8, so that each XCD walks one contiguous eighth of the work list and the workgroups
// that share a key / twiddle row (consecutive items, batch fastest) hit the same L2.  Identity when n is not a multiple of 8.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned lin, unsigned n) {
    return (n & 7u) ? lin : (lin & 7u) * (n >> 3) + (lin >> 3);
}

// ------------------------------------------------------------------------------------
// butterflies
// ------------------------------------------------------------------------------------
// The butterflies take their Montgomery products through mred_lazy_w32 (modarith.h: two 32-bit reduction rounds, 8 v_mad_u64_u32 +
// 2 v_mul_lo_u32 + a few adds instead of the 11 multiplies and ~15 carry instructions of the full-width form): only the residue
// class of a butterfly output matters, and the operands satisfy its domain (V + q < 2^64, w < q).  HE_BFLY_W32 = 0 restores the
// full-width products for A/B builds.
#ifndef HE_BFLY_W32
#define HE_BFLY_W32 1
#endif
// Row kernels (HE_BFLY_ASM): the same product as ONE hand-written sequence of 16 instructions.  hipcc turns every form of it
// into 21-26: the 64-bit addend of v_mad_u64_u32 must be an (even-aligned) register pair, so each "high word, zero-extended" is a
// v_mov_b32 into a fresh pair, and a carry costs a 64-bit compare + select.  By column instead: L = x0 w0, M = x0 w1 + x1 w0,
// H = x1 w1 are three aligned 64-bit accumulators (for q < 2^61 and x < 4q the middle column cannot overflow: x1 w0 < 2^63,
// x0 w1 < 2^61), the two reduction rounds add m q0 / m q1 to (L, M) then (M, H), and what crosses from one accumulator to the
// next is a 32-bit word or a carry -- v_add_co / v_addc on the HIGH or LOW half of a pair.  Halves of a pair cannot be named
// through asm operands, so the seven scratch registers are fixed (v116..v122, declared clobbered: inside the 128-register
// budget of the four-wave row kernels), which is why only the row kernels take this form.
//   8 v_mad_u64_u32 + 2 v_mul_lo_u32 + 6 v_add(c)_co_u32, result (T + m q) / 2^64 in [0, 2q) as mred_lazy_w32.
#ifndef HE_BFLY_ASM
#define HE_BFLY_ASM 1
#endif
// gfx940 / gfx950: a VALU instruction that reads VCC (a carry-in) wants two wait states after the VALU instruction that wrote it
// -- hipcc pads its own carry chains with `s_nop 1` (llvm's VALUWriteSGPRVALURead rule), and nothing pads an asm string.  The
// sequence ran bit-exact without the pads through every suite of this round, which is no evidence for a hazard that shows on
// "some waves of some launches": it carries them.  An s_nop costs the wave two issue cycles, not the SIMD (HE_BFLY_VCC_PAD = 0
// for A/B builds).
#ifndef HE_BFLY_VCC_PAD
#define HE_BFLY_VCC_PAD 1
#endif
#if HE_BFLY_VCC_PAD
#define HE_VCC_PAD "s_nop 1\n\t"
#else
#define HE_VCC_PAD
#endif
__device__ __forceinline__ uint64_t mred_lazy_col_asm(uint64_t x, uint64_t w, uint64_t q, uint64_t qinv) {
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t q0 = (uint32_t)q, q1 = (uint32_t)(q >> 32), nq = (uint32_t)(0 - qinv);
    uint64_t r;
    asm("v_mad_u64_u32 v[116:117], vcc, %[x0], %[w0], 0\n\t"               // L = x0 w0
        "v_mad_u64_u32 v[118:119], vcc, %[x0], %[w1], 0\n\t"               // M = x0 w1
        "v_mad_u64_u32 v[120:121], vcc, %[x1], %[w1], 0\n\t"               // H = x1 w1
        "v_mad_u64_u32 v[118:119], vcc, %[x1], %[w0], v[118:119]\n\t"      // M += x1 w0            (no overflow, see above)
        "v_mul_lo_u32 v122, v116, %[nq]\n\t"                               // m = L.lo (-q^-1) mod 2^32
        "v_mad_u64_u32 v[116:117], vcc, v122, %[q0], v[116:117]\n\t"       // L += m q0: L.lo = 0, carry
        HE_VCC_PAD
        "v_addc_co_u32_e64 v119, vcc, 0, v119, vcc\n\t"                    //   -> bit 32 of M
        "v_mad_u64_u32 v[118:119], vcc, v122, %[q1], v[118:119]\n\t"       // M += m q1
        "v_add_co_u32_e32 v118, vcc, v118, v117\n\t"                       // M += L.hi             (T + m q) / 2^32 = M + H 2^32
        HE_VCC_PAD
        "v_addc_co_u32_e64 v119, vcc, 0, v119, vcc\n\t"
        "v_mul_lo_u32 v122, v118, %[nq]\n\t"                               // m = M.lo (-q^-1) mod 2^32
        "v_mad_u64_u32 v[118:119], vcc, v122, %[q0], v[118:119]\n\t"       // M += m q0: M.lo = 0, carry
        HE_VCC_PAD
        "v_addc_co_u32_e64 v121, vcc, 0, v121, vcc\n\t"                    //   -> bit 32 of H
        "v_add_co_u32_e32 v120, vcc, v120, v119\n\t"                       // H += M.hi
        HE_VCC_PAD
        "v_addc_co_u32_e64 v121, vcc, 0, v121, vcc\n\t"
        "v_mad_u64_u32 %[r], vcc, v122, %[q1], v[120:121]"                   // r = H + m q1 in [0, 2q)
        : [r] "=v"(r)
        : [x0] "v"(x0), [x1] "v"(x1), [w0] "v"(w0), [w1] "v"(w1), [q0] "s"(q0), [q1] "s"(q1), [nq] "s"(nq)
        : "v116", "v117", "v118", "v119", "v120", "v121", "v122", "vcc");
    return r;
}
template <bool ROWS = false>
__device__ __forceinline__ uint64_t bfly_mul(uint64_t v, uint64_t w, uint64_t q, uint64_t qinv) {
    if constexpr (HE_BFLY_ASM && ROWS) return mred_lazy_col_asm(v, w, q, qinv);
    else if constexpr (HE_BFLY_W32) return mred_lazy_w32(v, w, q, qinv);
    else return mred_lazy(v, w, q, qinv);
}
// forward: U,V in [0,4q) -> X,Y in [0,4q)
template <bool ROWS = false>
__device__ __forceinline__ void bfly_fwd(uint64_t &a, uint64_t &b, uint64_t w, uint64_t q, uint64_t twoq, uint64_t qinv) {
    uint64_t U = a >= twoq ? a - twoq : a;
    uint64_t V = bfly_mul<ROWS>(b, w, q, qinv);
    a = U + V;
    b = U + twoq - V;
}
// forward without range correction, for q < 2^58: every output is below (input bound + 2q), so the
// 15 stages of a logN=16 transform stay below 34q < 2^64; one Barrett reduction at the very end.
//   r = V*w*2^-64 in [0, 2q);  X = U + r,  Y = U + 2q - r
template <bool ROWS = false>
__device__ __forceinline__ void bfly_fwd_nc(uint64_t &a, uint64_t &b, uint64_t w, uint64_t q, uint64_t qinv) {
    const uint64_t r = bfly_mul<ROWS>(b, w, q, qinv);
    const uint64_t u = a;
    a = u + r;
    b = u + (q << 1) - r;
}
constexpr int kNoCorrBits = 58;
// inverse: U,V in [0,2q) -> X,Y in [0,2q)
__device__ __forceinline__ void bfly_inv(uint64_t &a, uint64_t &b, uint64_t w, uint64_t q, uint64_t twoq, uint64_t qinv) {
    uint64_t U = a, V = b;
    uint64_t X = U + V;
    a = X >= twoq ? X - twoq : X;
    // (either word-serial form -- C++ or hand-written -- costs the generic inverse row kernel its fourth wave: 27-45 spills; the
    // production row sizes take the lean variant of the kernel, which has the hand-written form)
    b = mred_lazy(U + twoq - V, w, q, qinv);
}
// the same two butterflies with the hand-written product (kernels with register room for its fixed scratch: the fused basis
// extension, HE_MODUP_ASM)
__device__ __forceinline__ void bfly_inv_asm(uint64_t &a, uint64_t &b, uint64_t w, uint64_t q, uint64_t twoq, uint64_t qinv) {
    const uint64_t U = a, V = b, X = U + V;
    a = X >= twoq ? X - twoq : X;
    b = mred_lazy_col_asm(U + twoq - V, w, q, qinv);
}
__device__ __forceinline__ void bfly_inv_scaled_asm(uint64_t &a, uint64_t &b, uint64_t wn, uint64_t ninv, uint64_t q,
                                                    uint64_t twoq, uint64_t qinv) {
    const uint64_t U = a, V = b;
    uint64_t X = U + V;
    X = X >= twoq ? X - twoq : X;  // the sequence wants its operand below 4q (U, V in [0, 2q): U + V is)
    a = cred(mred_lazy_col_asm(X, ninv, q, qinv), q);
    b = cred(mred_lazy_col_asm(U + twoq - V, wn, q, qinv), q);
}
// last inverse stage with N^-1 folded in: outputs canonical
__device__ __forceinline__ void bfly_inv_scaled(uint64_t &a, uint64_t &b, uint64_t wn, uint64_t ninv, uint64_t q,
                                                uint64_t twoq, uint64_t qinv) {
    uint64_t U = a, V = b;
    a = mred(U + V, ninv, q, qinv);
    b = mred(U + twoq - V, wn, q, qinv);
}

struct NttArgs {
    const uint64_t *in;
    uint64_t *out;
    size_t in_bs, out_bs;
    const ModConst *mc;
    const uint64_t *tw;
    const double *twd;  // same table as plain (non-Montgomery) integers in double precision, moduli < 2^47 only
    int N;
    int a;       // column stages already done (forward) / still to do (inverse)
    int flags;
    int scale;   // inverse: fold N^-1 into the last stage of THIS kernel
    LimbTab tab;
    // forward epilogue (see NttEpilogue)
    int epi;     // 0 none, 1 = MRed(v + 2q - y, s), 2 = w + that
    const uint64_t *epi_y, *epi_w;
    size_t epi_y_bs, epi_w_bs;
    uint64_t epi_s[kMaxLimbs];
    int zsplit;  // batch entries >= zsplit write the second output set
    int epi2;
    uint64_t *out2;
    size_t out2_bs;
    const uint64_t *epi_y2, *epi_w2;
    size_t epi_y2_bs, epi_w2_bs;
    int epi_tensor;            // tensor-mode epilogue (NttEpilogue::tensor)
    const uint64_t *ta0, *ta1, *tb0, *tb1;
    size_t ta0_bs, ta1_bs, tb0_bs, tb1_bs;
    uint64_t epi_ts[kMaxLimbs];
    uint64_t io_s[kMaxLimbs];  // NTT_ADD_SCALAR: per launch limb, added to the input words (forward) / to the canonical output (inverse)
    int epi_y_f64;  // f64 kernel only: y holds doubles
    int epi_y_reduce;  // f64 kernel only: y holds arbitrary 64-bit words (reduced before the conversion to double)
    int nbatch, iters;  // f64 kernel only: a workgroup transforms batch entries blockIdx.x * iters ... (+ iters - 1) of its row
    int nbatch_prof = 0;  // host only: batch entries of the launch when grid.x is not their number (rows_bytes)
    int tprod = 0;  // f64 inverse kernel only (NttProdIn): the input is formed here as T(ta1, tb1) with epi_ts, and also written to out2
    // entry tables (View::tab) of the caller-facing operands: the epilogue's outputs and addends, the product's inputs
    unsigned sc_ginv = 0;            // epilogue stores scattered by the automorphism of inverse Galois element sc_ginv (auto_dest); 0: none
    int sc_logN = 0;
    const size_t *in_tab = nullptr;  // the transform's input (launch_ntt_rows: the operand of a coalesced key switch)
    int cb = 0;  // column kernel only: stages done by an outer column pass (logN >= 19: the columns take two passes, see ntt_cols_kernel)
    const size_t *out_tab = nullptr, *out2_tab = nullptr, *epi_w_tab = nullptr, *epi_w2_tab = nullptr;
    const size_t *ta0_tab = nullptr, *ta1_tab = nullptr, *tb0_tab = nullptr, *tb1_tab = nullptr;
    const size_t *epi_y_tab = nullptr, *epi_y2_tab = nullptr;  // the epilogue's subtrahend (the rescale's own input)
};

__device__ __forceinline__ int lds_phys(int e) { return e + (e >> 4); }

// Synchronisation of the LDS exchanges of the row kernels.  With T = 2^LOGB / 16 threads, thread tau = (hi, lo) of round rho
// holds e = hi 2^(sh+4) + k 2^sh + lo (sh = LOGB - 4 rho - 4), and the exchange that follows the round moves data only inside
// groups of 2^sh consecutive threads (the consumer of element e is thread (hi 16 + k) 2^(sh-4) + (lo mod 2^(sh-4)), in the same
// group).  For sh <= 6 the group lies inside one wavefront: LDS operations of a wave are processed in order, so no workgroup
// barrier is needed, only a compiler fence.  A 4096-row (LOGB = 12) then has ONE barrier (after the first round, sh = 8) instead
// of three: the second exchange is local to 16 lanes and the last one -- back to a coalesced order for the global accesses -- is
// made wave-local by letting every wave keep its own contiguous 1024 coefficients (nat_e) instead of the k T + tau interleave.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void rows_sync(int sh) {
    if (sh <= 6) wave_lds_sync();
    else __syncthreads();
}
// coalesced ("natural") element k of thread tau: wave w owns coefficients [1024 w, 1024 (w+1)), lane l element 64 k + l of them
template <int T>
__device__ __forceinline__ int nat_e(int k, int tau) {
    constexpr int CH = T >= 64 ? 64 : T;
    return (tau / CH) * (16 * CH) + k * CH + (tau % CH);
}

// N = 2^n is split into a strided "column" stages and b contiguous "row" stages: rows of 4096 coefficients
// (b = 12) up to logN = 15, rows of 8192 (b = 13, two 512-thread workgroups per CU) from logN = 16 so that the
// fused basis extension never holds more than 8 strided coefficients per thread.
// HERING_ROWBITS16=12: logN = 16 on 4096-rows with four column stages inside the basis extension (A/B, NOTES.md round 6)
int ntt_row_bits(int n) {
    static const int rb16 = getenv("HERING_ROWBITS16") && atoi(getenv("HERING_ROWBITS16")) == 12 ? 12 : 13;
    return n <= 12 ? n : (n <= 15 ? 12 : (n == 16 ? rb16 : 13));
}

// ------------------------------------------------------------------------------------
// ntt_rows: b = LOGB stages on one contiguous row of 2^LOGB coefficients per workgroup.
// grid = (rows per limb = 2^a, limbs, batch), block = 2^LOGB / 16 threads.
// Rounds of g <= 4 stages; in a round a thread owns W = 16/2^g groups of 2^g coefficients
//   e = hi*2^(LOGB-s0) + k*2^sh + lo,  sh = LOGB-s0-g,  group id = tau*W + w = hi*2^sh + lo
// and stage s0+u uses twiddle index 2^(s0+u)*(2^a+row) + hi*2^u + (k >> (g-u)).
// The round loop is NOT unrolled (the three radix-16 rounds of a 4096-row share one copy of
// the 32-butterfly network; the fully unrolled kernel overflowed the instruction cache).
// NC: butterflies without range correction (all moduli of the launch below 2^58).
// ------------------------------------------------------------------------------------
template <int G4, bool INV, bool NC>
__device__ __forceinline__ void rows_round(uint64_t (&x)[16], const uint64_t *__restrict__ tw, int rowtw, int s0, int hi0,
                                           int tau, int sh, uint64_t q, uint64_t twoq, uint64_t qinv, const ModConst &mc,
                                           bool scale_last) {
    // G4 = log2 of the group size (g), W = 16 >> g groups per thread
    constexpr int g = G4, G = 1 << g, W = 16 / G;
    if constexpr (!INV) {
#pragma unroll
        for (int u = 0; u < g; u++) {
            const int d = 1 << (g - 1 - u);
#pragma unroll
            for (int w = 0; w < W; w++) {
                const int hi = (W == 1) ? hi0 : ((tau * W + w) >> sh);
                const int base = (rowtw << (s0 + u)) + (hi << u);
#pragma unroll
                for (int k = 0; k < G; k++) {
                    if (k & d) continue;
                    const uint64_t wv = tw[base + (k >> (g - u))];
                    if constexpr (NC) bfly_fwd_nc<true>(x[w * G + k], x[w * G + k + d], wv, q, qinv);
                    else bfly_fwd<true>(x[w * G + k], x[w * G + k + d], wv, q, twoq, qinv);
                }
            }
        }
    } else {
#pragma unroll
        for (int u = g - 1; u >= 0; u--) {
            const int d = 1 << (g - 1 - u);
            const bool last = scale_last && u == 0;
#pragma unroll
            for (int w = 0; w < W; w++) {
                const int hi = (W == 1) ? hi0 : ((tau * W + w) >> sh);
                const int base = (rowtw << (s0 + u)) + (hi << u);
#pragma unroll
                for (int k = 0; k < G; k++) {
                    if (k & d) continue;
                    const uint64_t wv = tw[base + (k >> (g - u))];
                    if (last) bfly_inv_scaled(x[w * G + k], x[w * G + k + d], mred(wv, mc.ninv, q, qinv), mc.ninv, q, twoq, qinv);
                    else bfly_inv(x[w * G + k], x[w * G + k + d], wv, q, twoq, qinv);
                }
            }
        }
    }
}

// radix-16 round with its fifteen twiddles preloaded (see rows_tw16_f64): t[(1 << u) - 1 + j] = tw[(rowtw << (s0 + u)) + (hi0 << u) + j]
__device__ __forceinline__ void rows_tw16(uint64_t (&t)[15], const uint64_t *__restrict__ tw, int rowtw, int s0, int hi0) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int base = (rowtw << (s0 + u)) + (hi0 << u);
#pragma unroll
        for (int j = 0; j < (1 << u); j++) t[(1 << u) - 1 + j] = tw[base + j];
    }
}
template <bool INV, bool NC>
__device__ __forceinline__ void rows_round16(uint64_t (&x)[16], const uint64_t (&t)[15], uint64_t q, uint64_t twoq, uint64_t qinv,
                                             const ModConst &mc, bool scale_last) {
    if constexpr (!INV) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int d = 1 << (3 - u);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k & d) continue;
                const uint64_t wv = t[(1 << u) - 1 + (k >> (4 - u))];
                if constexpr (NC) bfly_fwd_nc<true>(x[k], x[k + d], wv, q, qinv);
                else bfly_fwd<true>(x[k], x[k + d], wv, q, twoq, qinv);
            }
        }
    } else {
#pragma unroll
        for (int u = 3; u >= 0; u--) {
            const int d = 1 << (3 - u);
            const bool last = scale_last && u == 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k & d) continue;
                const uint64_t wv = t[(1 << u) - 1 + (k >> (4 - u))];
                if (last) bfly_inv_scaled(x[k], x[k + d], mred(wv, mc.ninv, q, qinv), mc.ninv, q, twoq, qinv);
                else bfly_inv(x[k], x[k + d], wv, q, twoq, qinv);
            }
        }
    }
}

// inverse radix-16 round without the N^-1 fold, products through the hand-written column Montgomery sequence (the production
// inverse row kernel: x[k + d] = (U + 2q - V) w with U + 2q - V < 4q, inside the sequence's domain)
__device__ __forceinline__ void rows_round16_inv_asm(uint64_t (&x)[16], const uint64_t (&t)[15], uint64_t q, uint64_t twoq, uint64_t qinv) {
#pragma unroll
    for (int u = 3; u >= 0; u--) {
        const int d = 1 << (3 - u);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k & d) continue;
            const uint64_t U = x[k], V = x[k + d];
            const uint64_t X = U + V;
            x[k] = X >= twoq ? X - twoq : X;
            x[k + d] = mred_lazy_col_asm(U + twoq - V, t[(1 << u) - 1 + (k >> (4 - u))], q, qinv);
        }
    }
}

// A radix-16 round (one group of sixteen per thread, W = 1): element k of the thread is e0 | (k << sh) with
// e0 = (hi << (LOGB - s0)) + lo, and because bits [sh, sh + 4) of e0 are clear, lds_phys(e) = lds_phys(e0) + c_k with
// c_k = (k << sh) + ((k << sh) >> 4) -- a compile-time constant where the call site knows sh (LDS offset immediates), a scalar
// otherwise: one base address per exchange instead of a shift / add / shift / add chain per element.
#ifndef HE_XFER16
#define HE_XFER16 1  // 0: the per-element address arithmetic of rounds 1-2 (A/B builds)
#endif
template <int LOGB, class Tw>
__device__ __forceinline__ void rows_lds_xfer16(Tw (&x)[16], Tw *lds, int tau, int s0, int sh, bool store) {
    const unsigned ut = (unsigned)tau, hi = ut >> sh, lo = ut & ((1u << sh) - 1u);
    const unsigned e0 = (hi << (LOGB - s0)) + lo;
    Tw *p = lds + (e0 + (e0 >> 4));
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const unsigned c = ((unsigned)k << sh) + (((unsigned)k << sh) >> 4);
        if (store) p[c] = x[k];
        else x[k] = p[c];
    }
}
template <int LOGB, int G4>
__device__ __forceinline__ void rows_lds_xfer(uint64_t (&x)[16], uint64_t *lds, int tau, int s0, int sh, bool store) {
    if constexpr (G4 == 4 && HE_XFER16) { rows_lds_xfer16<LOGB>(x, lds, tau, s0, sh, store); return; }
    constexpr int g = G4, G = 1 << g, W = 16 / G;
#pragma unroll
    for (int w = 0; w < W; w++) {
        const int gamma = tau * W + w, hi = gamma >> sh, lo = gamma & ((1 << sh) - 1);
#pragma unroll
        for (int k = 0; k < G; k++) {
            const int e = (hi << (LOGB - s0)) + (k << sh) + lo;
            if (store) lds[lds_phys(e)] = x[w * G + k];
            else x[w * G + k] = lds[lds_phys(e)];
        }
    }
}

// LEAN (inverse, production row sizes): no N^-1 fold (A.scale == 0 by contract) and the products through the hand-written
// Montgomery sequence -- without the fold's code the kernel keeps its four waves per SIMD with the sequence's fixed scratch
// registers (the generic inverse spills 27-45 with them).  Round 2 ran this variant on Shoup twiddle pairs (19 instructions per
// product and twice the twiddle bytes); the 16-instruction sequence on the ordinary table is 4 % faster and needs no second table.
// SCAT (forward, production row sizes): the epilogue's stores go through auto_dest (NttEpilogue::scatter_ginv); its own
// instantiation, so that the plain kernels keep their registers
template <int LOGB, bool INV, bool NC, bool LEAN = false, bool SCAT = false>
__global__ void __launch_bounds__((1 << LOGB) / 16 > 0 ? (1 << LOGB) / 16 : 1, 4) ntt_rows_kernel(NttArgs A) {
    constexpr int N2 = 1 << LOGB;
    constexpr int T = N2 / 16;
    constexpr int NR4 = LOGB / 4;       // full radix-16 rounds
    constexpr int GREM = LOGB % 4;      // stages of the trailing partial round (test sizes only)
    __shared__ uint64_t lds[N2 + N2 / 16];

    // grid = (batch, limbs, rows): the batch index varies fastest so that the workgroups sharing a twiddle row
    // (same limb and row) are dispatched together and hit it in L2
    const int tau = threadIdx.x;
    const int row = blockIdx.z;
    unsigned bzi = blockIdx.x;
    if (!INV && A.epi_tensor) {
        // launch order: [8 entries, component 0][the same 8 entries, component 1]... -> input entry = comp * zsplit + entry
        const unsigned ent = (bzi >> 4) * 8 + (bzi & 7);
        if ((int)ent >= A.zsplit) return;
        bzi = ((bzi >> 3) & 1) * (unsigned)A.zsplit + ent;
    }
    const int y = blockIdx.y;
    const int il = A.tab.in_limb[y], ol = A.tab.out_limb[y], mi = A.tab.mod[y];
    const ModConst mc = A.mc[mi];
    const uint64_t q = mc.q, qinv = mc.qinv, twoq = mc.q << 1;
    const uint64_t *__restrict__ tw = A.tw + (size_t)mi * A.N;
    const uint64_t *__restrict__ src = A.in + voff(A.in_tab, A.in_bs, bzi) + (size_t)il * A.N + (size_t)row * N2;
    // (with an epilogue the stores go through `op` below: entries of the second output set have no row in the entry table)
    uint64_t *__restrict__ dst = A.out + ((!INV && A.epi) ? (size_t)0 : voff(A.out_tab, A.out_bs, bzi)) + (size_t)ol * A.N + (size_t)row * N2;
    const int rowtw = (1 << A.a) + row;  // 2^a + r

    uint64_t x[16];
    static_assert(!LEAN || INV, "the lean variant is an inverse kernel");

    if constexpr (!INV) {
        constexpr int sh0 = LOGB - 4;
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = ldnt(&src[(k << sh0) + tau]);
        if (A.flags & NTT_ADD_SCALAR) {
            const uint64_t sadd = A.io_s[y];
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] += sadd;
        }
        if (A.flags & NTT_REDUCE_INPUT) {
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = bred_add_lazy(x[k], q, mc.brc0);
        }
        uint64_t t16[15];
        if constexpr (NR4 > 0) rows_tw16(t16, tw, rowtw, 0, tau >> (LOGB - 4));
#pragma unroll 1
        for (int rho = 0; rho < NR4; rho++) {
            const int s0 = 4 * rho, sh = LOGB - s0 - 4;
            if (rho > 0) rows_lds_xfer<LOGB, 4>(x, lds, tau, s0, sh, false);
            rows_round16<false, NC>(x, t16, q, twoq, qinv, mc, false);
            if (rho + 1 < NR4) rows_tw16(t16, tw, rowtw, s0 + 4, tau >> (sh - 4));  // in flight across the exchange
            rows_lds_xfer<LOGB, 4>(x, lds, tau, s0, sh, true);
            rows_sync(sh);
        }
        if constexpr (GREM > 0) {
            constexpr int s0 = 4 * NR4;
            rows_lds_xfer<LOGB, GREM>(x, lds, tau, s0, 0, false);
            rows_round<GREM, false, NC>(x, tw, rowtw, s0, 0, tau, 0, q, twoq, qinv, mc, false);
            rows_lds_xfer<LOGB, GREM>(x, lds, tau, s0, 0, true);
            rows_sync(0);
        }
        const bool lazy = (A.flags & NTT_LAZY_OUT) != 0;
        auto settle = [&](uint64_t v) -> uint64_t {
            if constexpr (NC) return bred_add_lazy(v, q, mc.brc0);  // [0, 36q) -> [0, 2q)
            else return v >= twoq ? v - twoq : v;
        };
        if (A.epi) {
            // out = [w +] MRed(x + 2q - y, s); the sixteen y (then w) loads are issued together, not one wait per element
            const bool second = A.zsplit && (int)bzi >= A.zsplit;
            const size_t zz = second ? bzi - A.zsplit : bzi;
            const size_t off = (size_t)ol * A.N + (size_t)row * N2;
            const uint64_t *yp = (second ? A.epi_y2 + voff(A.epi_y2_tab, A.epi_y2_bs, zz) : A.epi_y + voff(A.epi_y_tab, A.epi_y_bs, zz)) + off;
            const uint64_t *wp = (second ? A.epi_w2 + voff(A.epi_w2_tab, A.epi_w2_bs, zz) : A.epi_w + voff(A.epi_w_tab, A.epi_w_bs, zz)) + off;
            uint64_t *op = (second ? A.out2 + voff(A.out2_tab, A.out2_bs, zz) : A.out + voff(A.out_tab, A.out_bs, zz)) + off;
            const bool addw = (second ? A.epi2 : A.epi) == 2;
            const uint64_t sy = A.epi_s[y];
            uint64_t yv[16];
#pragma unroll
            for (int k = 0; k < 16; k++) yv[k] = ldnt(&yp[nat_e<T>(k, tau)]);
            if (A.epi_tensor) {
                // the addend from the product's inputs, eight coefficients at a time (plain loads: the other component's
                // workgroup reads the same rows from L2)
                const size_t toff = (size_t)ol * A.N + (size_t)row * N2;
                const uint64_t *pa0 = A.ta0 + voff(A.ta0_tab, A.ta0_bs, zz) + toff, *pa1 = A.ta1 + voff(A.ta1_tab, A.ta1_bs, zz) + toff;
                const uint64_t *pb0 = A.tb0 + voff(A.tb0_tab, A.tb0_bs, zz) + toff, *pb1 = A.tb1 + voff(A.tb1_tab, A.tb1_bs, zz) + toff;
                const uint64_t ts = A.epi_ts[y];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    uint64_t u[8], v[8], wv[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) { const int e = nat_e<T>(8 * h + k, tau); u[k] = pa0[e]; v[k] = second ? pb1[e] : pb0[e]; }
#pragma unroll
                    for (int k = 0; k < 8; k++) wv[k] = mred(mred(u[k], ts, q, qinv), v[k], q, qinv);
                    if (second) {
#pragma unroll
                        for (int k = 0; k < 8; k++) { const int e = nat_e<T>(8 * h + k, tau); u[k] = pa1[e]; v[k] = pb0[e]; }
#pragma unroll
                        for (int k = 0; k < 8; k++) wv[k] = cred(wv[k] + mred(mred(u[k], ts, q, qinv), v[k], q, qinv), q);
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int e = nat_e<T>(8 * h + k, tau);
                        stnt(&op[e], cred(wv[k] + mred(settle(lds[lds_phys(e)]) + twoq - yv[8 * h + k], sy, q, qinv), q));
                    }
                }
            } else if (addw) {
                uint64_t wv[16];
#pragma unroll
                for (int k = 0; k < 16; k++) wv[k] = ldnt(&wp[nat_e<T>(k, tau)]);
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int e = nat_e<T>(k, tau);
                    uint64_t *dp = &op[e];
                    if constexpr (SCAT) dp = op - (size_t)row * N2 + auto_dest((unsigned)(row * N2 + e), A.sc_ginv, A.sc_logN);
                    stnt(dp, cred(wv[k] + mred(settle(lds[lds_phys(e)]) + twoq - yv[k], sy, q, qinv), q));
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int e = nat_e<T>(k, tau);
                    uint64_t *dp = &op[e];
                    if constexpr (SCAT) dp = op - (size_t)row * N2 + auto_dest((unsigned)(row * N2 + e), A.sc_ginv, A.sc_logN);
                    stnt(dp, mred(settle(lds[lds_phys(e)]) + twoq - yv[k], sy, q, qinv));
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int e = nat_e<T>(k, tau);
                uint64_t v = settle(lds[lds_phys(e)]);
                if (!lazy) v = v >= q ? v - q : v;
                dst[e] = v;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = ldnt(&src[nat_e<T>(k, tau)]);  // the wave's own 1024 coefficients: the first exchange stays inside the wave
        if (A.flags & NTT_REDUCE_INPUT) {  // (only for a wave that met a word of 2q or more: see ntt_rows_f64_kernel)
            bool big = false;
#pragma unroll
            for (int k = 0; k < 16; k++) big = big || x[k] >= twoq;
            if (__any(big)) {
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = bred_add_lazy(x[k], q, mc.brc0);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; k++) lds[lds_phys(nat_e<T>(k, tau))] = x[k];
        rows_sync(0);
        if constexpr (GREM > 0) {
            constexpr int s0 = 4 * NR4;
            rows_lds_xfer<LOGB, GREM>(x, lds, tau, s0, 0, false);
            rows_round<GREM, true, false>(x, tw, rowtw, s0, 0, tau, 0, q, twoq, qinv, mc, false);
            rows_lds_xfer<LOGB, GREM>(x, lds, tau, s0, 0, true);
            rows_sync(GREM);
        }
        uint64_t t16[15];
        if constexpr (NR4 > 0) rows_tw16(t16, tw, rowtw, 4 * (NR4 - 1), tau >> (LOGB - 4 * NR4));
#pragma unroll 1
        for (int rho = NR4 - 1; rho >= 0; rho--) {
            const int s0 = 4 * rho, sh = LOGB - s0 - 4;
            rows_lds_xfer<LOGB, 4>(x, lds, tau, s0, sh, false);
            if constexpr (LEAN) rows_round16_inv_asm(x, t16, q, twoq, qinv);
            else rows_round16<true, false>(x, t16, q, twoq, qinv, mc, A.scale && rho == 0);
            if (rho > 0) {
                // the next round's, in flight across the exchange
                rows_tw16(t16, tw, rowtw, s0 - 4, tau >> (sh + 4));
                rows_lds_xfer<LOGB, 4>(x, lds, tau, s0, sh, true);
                rows_sync(sh + 4);  // the consumers are the next round's groups of 2^(sh + 4) threads
            }
        }
        constexpr int sh0 = LOGB - 4;
        if ((A.flags & NTT_ADD_SCALAR) && A.scale) {  // single-pass inverse: canonical outputs
            const uint64_t sadd = A.io_s[y];
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = cred(x[k] + sadd, q);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) dst[(k << sh0) + tau] = x[k];
    }
}

// ------------------------------------------------------------------------------------
// ntt_rows_f64: the same row transform for moduli below 2^47, with the residues carried as exact
// integers in double-precision registers.  gfx950 has no 64-bit integer multiplier (a Montgomery
// product costs ~30 VALU ops) but runs v_fma_f64 at the rate of one v_mad_u64_u32, so
//     h = a*w; l = fma(a,w,-h)            (error-free product, a*w = h + l exactly)
//     c = rint(h * (1/q)); r = fma(-c,q,h) + l   (integer, r == a*w mod q, |r| < 2q)
// is an exact modular product in 6 ops (measured 3.0x the integer MRedLazy rate, tools/f64_modmul_probe.hip).
// All values stay integers of magnitude < 2^53:  forward butterflies X = U + r, Y = U - r grow by < 2q per
// stage (34q + input < 2^53 for q < 2^47); the inverse reduces X once per radix-16 round.
// Inputs / outputs are the same uint64 words as the integer kernel (outputs canonical).
// ------------------------------------------------------------------------------------
constexpr int kF64Bits = 47;
__device__ __forceinline__ double modmul_f64(double a, double w, double q, double qi) {
    const double h = a * w;
    const double l = __fma_rn(a, w, -h);
    const double c = rint(h * qi);
    return __fma_rn(-c, q, h) + l;
}
// exact conversions of integers in [0, 2^52) through the 2^52 exponent trick: two VALU ops each, against four (u64 -> f64) and
// six (f64 -> u64) for the generic sequences.  Every residue on the double-precision paths is below 2^48.
__device__ __forceinline__ double u52_to_f64(uint64_t x) {
    return __longlong_as_double((long long)(x | 0x4330000000000000ull)) - 0x1p52;
}
__device__ __forceinline__ uint64_t f64_to_u52(double x) {  // x a non-negative integer
    return (uint64_t)__double_as_longlong(x + 0x1p52) & 0x000FFFFFFFFFFFFFull;
}
__device__ __forceinline__ double reduce_f64(double x, double q, double qi) {  // -> |x| < q
    return __fma_rn(-rint(x * qi), q, x);
}
__device__ __forceinline__ double canon_f64d(double x, double q, double qi) {  // any |x| < 2^53 -> [0, q), still a double
    double t = __fma_rn(-floor(x * qi), q, x);
    t = t < 0.0 ? t + q : t;
    t = t >= q ? t - q : t;
    return t;
}
__device__ __forceinline__ uint64_t canon_f64(double x, double q, double qi) { return f64_to_u52(canon_f64d(x, q, qi)); }

template <int G4, bool INV>
__device__ __forceinline__ void rows_round_f64(double (&x)[16], const double *__restrict__ tw, int rowtw, int s0, int hi0, int tau,
                                               int sh, double q, double qi) {
    constexpr int g = G4, G = 1 << g, W = 16 / G;
    if constexpr (!INV) {
#pragma unroll
        for (int u = 0; u < g; u++) {
            const int d = 1 << (g - 1 - u);
#pragma unroll
            for (int w = 0; w < W; w++) {
                const int hi = (W == 1) ? hi0 : ((tau * W + w) >> sh);
                const int base = (rowtw << (s0 + u)) + (hi << u);
#pragma unroll
                for (int k = 0; k < G; k++) {
                    if (k & d) continue;
                    const double r = modmul_f64(x[w * G + k + d], tw[base + (k >> (g - u))], q, qi);
                    const double U = x[w * G + k];
                    x[w * G + k] = U + r;
                    x[w * G + k + d] = U - r;
                }
            }
        }
    } else {
#pragma unroll
        for (int u = g - 1; u >= 0; u--) {
            const int d = 1 << (g - 1 - u);
#pragma unroll
            for (int w = 0; w < W; w++) {
                const int hi = (W == 1) ? hi0 : ((tau * W + w) >> sh);
                const int base = (rowtw << (s0 + u)) + (hi << u);
#pragma unroll
                for (int k = 0; k < G; k++) {
                    if (k & d) continue;
                    const double U = x[w * G + k], V = x[w * G + k + d];
                    x[w * G + k] = U + V;
                    x[w * G + k + d] = modmul_f64(U - V, tw[base + (k >> (g - u))], q, qi);
                }
            }
        }
    }
}
// The fifteen twiddles of one radix-16 round (one group of 16 per thread): stage u of the round uses
// t[(1 << u) - 1 + j] = tw[(rowtw << (s0 + u)) + (hi0 << u) + j], j < 2^u.  Loading them apart from the butterflies lets a
// kernel issue the NEXT round's twiddles before the LDS exchange and its barrier, so that their latency (an L2 round trip
// per round otherwise, exposed at two to four waves per SIMD) overlaps the exchange; the registers are the ones the
// finished round's twiddles occupied.
__device__ __forceinline__ void rows_tw16_f64(double (&t)[15], const double *__restrict__ tw, int rowtw, int s0, int hi0) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int base = (rowtw << (s0 + u)) + (hi0 << u);
#pragma unroll
        for (int j = 0; j < (1 << u); j++) t[(1 << u) - 1 + j] = tw[base + j];
    }
}
template <bool INV>
__device__ __forceinline__ void rows_round16_f64(double (&x)[16], const double (&t)[15], double q, double qi) {
    if constexpr (!INV) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int d = 1 << (3 - u);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k & d) continue;
                const double r = modmul_f64(x[k + d], t[(1 << u) - 1 + (k >> (4 - u))], q, qi);
                const double U = x[k];
                x[k] = U + r;
                x[k + d] = U - r;
            }
        }
    } else {
#pragma unroll
        for (int u = 3; u >= 0; u--) {
            const int d = 1 << (3 - u);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k & d) continue;
                const double U = x[k], V = x[k + d];
                x[k] = U + V;
                x[k + d] = modmul_f64(U - V, t[(1 << u) - 1 + (k >> (4 - u))], q, qi);
            }
        }
    }
}

template <int LOGB, int G4>
__device__ __forceinline__ void rows_lds_xfer_f64(double (&x)[16], double *lds, int tau, int s0, int sh, bool store) {
    if constexpr (G4 == 4 && HE_XFER16) { rows_lds_xfer16<LOGB>(x, lds, tau, s0, sh, store); return; }
    constexpr int g = G4, G = 1 << g, W = 16 / G;
#pragma unroll
    for (int w = 0; w < W; w++) {
        const int gamma = tau * W + w, hi = gamma >> sh, lo = gamma & ((1 << sh) - 1);
#pragma unroll
        for (int k = 0; k < G; k++) {
            const int e = (hi << (LOGB - s0)) + (k << sh) + lo;
            if (store) lds[lds_phys(e)] = x[w * G + k];
            else x[w * G + k] = lds[lds_phys(e)];
        }
    }
}

template <int LOGB, bool INV, bool TP = false, bool SCAT = false>
__global__ void __launch_bounds__((1 << LOGB) / 16 > 0 ? (1 << LOGB) / 16 : 1) ntt_rows_f64_kernel(NttArgs A) {
    static_assert(!TP || INV, "the product prologue belongs to the inverse transform");
    constexpr int N2 = 1 << LOGB;
    constexpr int T = N2 / 16;
    constexpr int NR4 = LOGB / 4;
    constexpr int GREM = LOGB % 4;
    __shared__ double lds[N2 + N2 / 16];

    const int tau = threadIdx.x;
    const int row = blockIdx.z;
    const int y = blockIdx.y;
    const int il = A.tab.in_limb[y], ol = A.tab.out_limb[y], mi = A.tab.mod[y];
    const ModConst mc = A.mc[mi];
    const double q = (double)mc.q, qi = mc.rq;
    const double *__restrict__ tw = A.twd + (size_t)mi * A.N;
    const int rowtw = (1 << A.a) + row;
    const size_t in_off = (size_t)il * A.N + (size_t)row * N2;

    // inverse: software pipeline over the workgroup's batch entries, the words of entry b + 1 are in flight while entry b is
    // transformed (-12 %).  The forward kernel keeps one entry per workgroup: with the 32 extra registers it drops from four
    // to two waves per SIMD and runs 1.5x slower.
    unsigned b0 = blockIdx.x * (unsigned)A.iters;
    if (!INV && A.epi_tensor) {  // see ntt_rows_kernel
        const unsigned ent = (b0 >> 4) * 8 + (b0 & 7);
        if ((int)ent >= A.zsplit) return;
        b0 = ((b0 >> 3) & 1) * (unsigned)A.zsplit + ent;
    }
    constexpr bool PIPE = INV && LOGB <= 12 && !TP;  // 512-thread rows (LOGB = 13) would fall to one workgroup per CU
    const unsigned b1 = PIPE ? min(b0 + (unsigned)A.iters, (unsigned)A.nbatch) : b0 + 1;  // otherwise iters == 1
    uint64_t nx[16];
    if constexpr (INV && !TP) {
        const uint64_t *src0 = A.in + voff(A.in_tab, A.in_bs, b0) + in_off;
#pragma unroll
        for (int k = 0; k < 16; k++) nx[k] = ldnt(&src0[nat_e<T>(k, tau)]);
    }
    for (unsigned bzi = b0; bzi < b1; bzi++) {
    uint64_t *__restrict__ dst = A.out + ((!INV && A.epi) ? (size_t)0 : voff(A.out_tab, A.out_bs, bzi)) + (size_t)ol * A.N + (size_t)row * N2;
    double x[16];
    if constexpr (!INV) {
        constexpr int sh0 = LOGB - 4;
        const uint64_t *__restrict__ src = A.in + voff(A.in_tab, A.in_bs, bzi) + in_off;
        if (A.flags & NTT_INPUT_F64) {  // doubles left by the basis extension (|x| < 2^53 through every stage: modup_f64_raw_ok)
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = __longlong_as_double((long long)ldnt(&src[(k << sh0) + tau]));
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                uint64_t v = ldnt(&src[(k << sh0) + tau]);
                if (A.flags & NTT_ADD_SCALAR) v += A.io_s[y];
                if (A.flags & NTT_REDUCE_INPUT) v = bred_add_lazy(v, mc.q, mc.brc0);
                x[k] = u52_to_f64(v);
            }
        }
        // (twiddles loaded inside the round: prefetching them across the exchange, as the inverse kernel and ntt_mac_f64 do,
        // measured 2 % slower here at four waves per SIMD)
#pragma unroll 1
        for (int rho = 0; rho < NR4; rho++) {
            const int s0 = 4 * rho, sh = LOGB - s0 - 4;
            if (rho > 0) rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, s0, sh, false);
            rows_round_f64<4, false>(x, tw, rowtw, s0, tau >> sh, tau, sh, q, qi);
            rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, s0, sh, true);
            rows_sync(sh);
        }
        if constexpr (GREM > 0) {
            constexpr int s0 = 4 * NR4;
            rows_lds_xfer_f64<LOGB, GREM>(x, lds, tau, s0, 0, false);
            rows_round_f64<GREM, false>(x, tw, rowtw, s0, 0, tau, 0, q, qi);
            rows_lds_xfer_f64<LOGB, GREM>(x, lds, tau, s0, 0, true);
            rows_sync(0);
        }
        if (A.epi) {
            // out = [w +] MRed(x + 2q - y, s) with x the transform: the subtraction and the product by s run in double
            // precision ((x - y) * s_plain mod q, s_plain = s * 2^-64 mod q), only the final CRed(w + .) is on integers so
            // that a lazy addend w gives the reference's word.  y (a key-switch accumulator) is canonical and below 2^47.
            const bool second = A.zsplit && (int)bzi >= A.zsplit;
            const size_t zz = second ? bzi - A.zsplit : bzi;
            const uint64_t *yp = (second ? A.epi_y2 + voff(A.epi_y2_tab, A.epi_y2_bs, zz) : A.epi_y + voff(A.epi_y_tab, A.epi_y_bs, zz)) + (size_t)ol * A.N + (size_t)row * N2;
            const uint64_t *wp = (second ? A.epi_w2 + voff(A.epi_w2_tab, A.epi_w2_bs, zz) : A.epi_w + voff(A.epi_w_tab, A.epi_w_bs, zz)) + (size_t)ol * A.N + (size_t)row * N2;
            uint64_t *op = (second ? A.out2 + voff(A.out2_tab, A.out2_bs, zz) : A.out + voff(A.out_tab, A.out_bs, zz)) + (size_t)ol * A.N + (size_t)row * N2;
            const bool addw = (second ? A.epi2 : A.epi) == 2;
            const double sp = (double)imform(A.epi_s[y], mc.q, mc.qinv);
            // all sixteen y (then w) loads are issued before the first use: with the format / addend tests inside the
            // element loop every load was followed by its own full wait (32 serialised HBM latencies per thread)
            double yv[16];
            if (A.epi_y_f64) {
#pragma unroll
                for (int k = 0; k < 16; k++) yv[k] = ldnt(&reinterpret_cast<const double *>(yp)[nat_e<T>(k, tau)]);
            } else if (A.epi_y_reduce) {  // caller-supplied y: any uint64
#pragma unroll
                for (int k = 0; k < 16; k++) yv[k] = u52_to_f64(bred_add_lazy(ldnt(&yp[nat_e<T>(k, tau)]), mc.q, mc.brc0));
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) yv[k] = u52_to_f64(ldnt(&yp[nat_e<T>(k, tau)]));
            }
            if (A.epi_tensor) {
                // addend = x y (ts 2^-128 mod q) from the product's inputs (caller words: reduced first), all in doubles; one
                // canonical reduction of addend + (transform - y) s gives the reference's word
                const size_t toff = (size_t)ol * A.N + (size_t)row * N2;
                const uint64_t *pa0 = A.ta0 + voff(A.ta0_tab, A.ta0_bs, zz) + toff, *pa1 = A.ta1 + voff(A.ta1_tab, A.ta1_bs, zz) + toff;
                const uint64_t *pb0 = A.tb0 + voff(A.tb0_tab, A.tb0_bs, zz) + toff, *pb1 = A.tb1 + voff(A.tb1_tab, A.tb1_bs, zz) + toff;
                const double tsp = (double)imform(imform(A.epi_ts[y], mc.q, mc.qinv), mc.q, mc.qinv);
                // caller words may be any 64-bit representative (as MRed accepts them); every pipeline of this library hands over
                // words below 2q, which convert as they are -- the Barrett reduction (a dozen integer instructions per word, up to
                // 64 words per thread) runs only for a wave that actually met a larger word
                const uint64_t twoq_u = mc.q << 1;
#ifndef HE_EPI_ALWAYS_REDUCE
#define HE_EPI_ALWAYS_REDUCE 0  // 1: the unconditional reduction of round 2 (A/B builds)
#endif
#ifndef HE_EPI_SHARED_FIRST
#define HE_EPI_SHARED_FIRST 1  // 0: round 2's order (component 1 reads a0, b1 then a1, b0)
#endif
                auto cvtn = [&](auto &w, auto &d) {
                    constexpr int n = (int)(sizeof(w) / sizeof(w[0]));
                    bool big = false;
#pragma unroll
                    for (int k = 0; k < n; k++) big = big || w[k] >= twoq_u;
                    if (HE_EPI_ALWAYS_REDUCE || __any(big)) {
#pragma unroll
                        for (int k = 0; k < n; k++) w[k] = bred_add_lazy(w[k], mc.q, mc.brc0);
                    }
#pragma unroll
                    for (int k = 0; k < n; k++) d[k] = u52_to_f64(w[k]);
                };
#if HE_EPI_SHARED_FIRST
                // Both components of an entry read a0 and b0 -- in two workgroups that the launch order puts on one XCD, a few
                // microseconds apart.  Component 1 used to read a0, b1 first and a1, b0 a phase later: by then the XCD's 4 MB L2
                // (turned over every ~6 us at this kernel's rate) had dropped b0, and 73 % of those second reads went to HBM
                // (PMC: 6.98 GB fetched per launch against 5.9 GB of distinct data).  Now both components issue the SAME a0 / b0
                // loads at the same point of the program, four coefficients at a time, and component 1 adds its a1 / b1 loads to
                // the same batch.
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    uint64_t r0[4], r1[4], r2[4], r3[4];
                    double u[4], v[4], u2[4], v2[4], wv[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) { const int e = nat_e<T>(4 * h + k, tau); r0[k] = pa0[e]; r1[k] = pb0[e]; }
                    if (second) {
#pragma unroll
                        for (int k = 0; k < 4; k++) { const int e = nat_e<T>(4 * h + k, tau); r2[k] = pa1[e]; r3[k] = pb1[e]; }
                    }
                    cvtn(r0, u); cvtn(r1, v);
                    if (second) {
                        cvtn(r2, u2); cvtn(r3, v2);
#pragma unroll
                        for (int k = 0; k < 4; k++) wv[k] = modmul_f64(u[k], v2[k], q, qi) + modmul_f64(u2[k], v[k], q, qi);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++) wv[k] = modmul_f64(u[k], v[k], q, qi);
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int e = nat_e<T>(4 * h + k, tau);
                        const double t = modmul_f64(wv[k], tsp, q, qi) + modmul_f64(lds[lds_phys(e)] - yv[4 * h + k], sp, q, qi);
                        stnt(&op[e], canon_f64(t, q, qi));
                    }
                }
#else
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    double u[8], v[8], wv[8];
                    uint64_t ur[8], vr[8];
                    const uint64_t *pv = second ? pb1 : pb0;
#pragma unroll
                    for (int k = 0; k < 8; k++) { const int e = nat_e<T>(8 * h + k, tau); ur[k] = pa0[e]; vr[k] = pv[e]; }
                    cvtn(ur, u); cvtn(vr, v);
#pragma unroll
                    for (int k = 0; k < 8; k++) wv[k] = modmul_f64(u[k], v[k], q, qi);
                    if (second) {
#pragma unroll
                        for (int k = 0; k < 8; k++) { const int e = nat_e<T>(8 * h + k, tau); ur[k] = pa1[e]; vr[k] = pb0[e]; }
                        cvtn(ur, u); cvtn(vr, v);
#pragma unroll
                        for (int k = 0; k < 8; k++) wv[k] += modmul_f64(u[k], v[k], q, qi);
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int e = nat_e<T>(8 * h + k, tau);
                        const double t = modmul_f64(wv[k], tsp, q, qi) + modmul_f64(lds[lds_phys(e)] - yv[8 * h + k], sp, q, qi);
                        stnt(&op[e], canon_f64(t, q, qi));
                    }
                }
#endif
            } else if (addw) {
                uint64_t wv[16];
#pragma unroll
                for (int k = 0; k < 16; k++) wv[k] = ldnt(&wp[nat_e<T>(k, tau)]);
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int e = nat_e<T>(k, tau);
                    const uint64_t v = canon_f64(modmul_f64(lds[lds_phys(e)] - yv[k], sp, q, qi), q, qi);
                    uint64_t *dp = &op[e];
                    if constexpr (SCAT) dp = op - (size_t)row * N2 + auto_dest((unsigned)(row * N2 + e), A.sc_ginv, A.sc_logN);
                    stnt(dp, cred(wv[k] + v, mc.q));
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int e = nat_e<T>(k, tau);
                    uint64_t *dp = &op[e];
                    if constexpr (SCAT) dp = op - (size_t)row * N2 + auto_dest((unsigned)(row * N2 + e), A.sc_ginv, A.sc_logN);
                    stnt(dp, canon_f64(modmul_f64(lds[lds_phys(e)] - yv[k], sp, q, qi), q, qi));
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int e = nat_e<T>(k, tau);
                dst[e] = canon_f64(lds[lds_phys(e)], q, qi);
            }
        }
    } else {
        if constexpr (TP) {
            // the transform's input is formed here: c2 = T(a1, b1) = MRed(MRed(a1, ts), b1) of the ciphertext product
            // (schemes/bgv/evaluator.go:634-647), canonical -- also written out (the digits' own limbs of the key inner product
            // read it) -- instead of a separate pass writing it and this one reading it back.  Caller words may be any 64-bit
            // representative; the Barrett reduction runs only for a wave that met one of 2q or above.
            const size_t off = (size_t)il * A.N + (size_t)row * N2;
            const uint64_t *pa = A.ta1 + voff(A.ta1_tab, A.ta1_bs, bzi) + off, *pb = A.tb1 + voff(A.tb1_tab, A.tb1_bs, bzi) + off;
            uint64_t *pc = A.out2 + (size_t)bzi * A.out2_bs + off;
            const double tsp = (double)imform(imform(A.epi_ts[y], mc.q, mc.qinv), mc.q, mc.qinv);
            const uint64_t twoq_u = mc.q << 1;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                uint64_t ua[8], ub[8];
#pragma unroll
                for (int k = 0; k < 8; k++) { const int e = nat_e<T>(8 * h + k, tau); ua[k] = ldnt(&pa[e]); ub[k] = ldnt(&pb[e]); }
                bool big = false;
#pragma unroll
                for (int k = 0; k < 8; k++) big = big || ua[k] >= twoq_u || ub[k] >= twoq_u;
                if (__any(big)) {
#pragma unroll
                    for (int k = 0; k < 8; k++) { ua[k] = bred_add_lazy(ua[k], mc.q, mc.brc0); ub[k] = bred_add_lazy(ub[k], mc.q, mc.brc0); }
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int e = nat_e<T>(8 * h + k, tau);
                    const double c = canon_f64d(modmul_f64(modmul_f64(u52_to_f64(ua[k]), u52_to_f64(ub[k]), q, qi), tsp, q, qi), q, qi);
                    stnt(&pc[e], f64_to_u52(c));
                    lds[lds_phys(e)] = c;
                }
            }
        } else {
        // caller-supplied words (any uint64): the Barrett reduction -- a dozen integer instructions per word -- runs only for a wave
        // that actually met a word of 2q or more (every pipeline of this library, and a caller that keeps its polynomials reduced,
        // hands over words below 2q, which convert as they are); the canonical result does not depend on the representative
        if (A.flags & NTT_REDUCE_INPUT) {
            const uint64_t twoq_u = mc.q << 1;
            bool big = false;
#pragma unroll
            for (int k = 0; k < 16; k++) big = big || nx[k] >= twoq_u;
            if (__any(big)) {
#pragma unroll
                for (int k = 0; k < 16; k++) nx[k] = bred_add_lazy(nx[k], mc.q, mc.brc0);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; k++) lds[lds_phys(nat_e<T>(k, tau))] = u52_to_f64(nx[k]);
        }
        if constexpr (PIPE) if (bzi + 1 < b1) {
            const uint64_t *srcn = A.in + voff(A.in_tab, A.in_bs, bzi + 1) + in_off;
#pragma unroll
            for (int k = 0; k < 16; k++) nx[k] = ldnt(&srcn[nat_e<T>(k, tau)]);
        }
        rows_sync(0);  // the wave filled its own 1024 coefficients: the first exchange stays inside the wave
        if constexpr (GREM > 0) {
            constexpr int s0 = 4 * NR4;
            rows_lds_xfer_f64<LOGB, GREM>(x, lds, tau, s0, 0, false);
            rows_round_f64<GREM, true>(x, tw, rowtw, s0, 0, tau, 0, q, qi);
            rows_lds_xfer_f64<LOGB, GREM>(x, lds, tau, s0, 0, true);
            rows_sync(GREM);
        }
        double t16[15];
        if constexpr (NR4 > 0) rows_tw16_f64(t16, tw, rowtw, 4 * (NR4 - 1), tau >> (LOGB - 4 * NR4));
#pragma unroll 1
        for (int rho = NR4 - 1; rho >= 0; rho--) {
            const int s0 = 4 * rho, sh = LOGB - s0 - 4;
            rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, s0, sh, false);
            // the sums X = U + V double per stage: bring everything back below q once per round
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = reduce_f64(x[k], q, qi);
            rows_round16_f64<true>(x, t16, q, qi);
            if (rho > 0) {
                rows_tw16_f64(t16, tw, rowtw, s0 - 4, tau >> (sh + 4));  // the next round's, in flight across the exchange
                rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, s0, sh, true);
                rows_sync(sh + 4);  // the consumers are the next round's groups of 2^(sh + 4) threads
            }
        }
        constexpr int sh0 = LOGB - 4;
        if (A.scale) {  // N^-1 (plain integer, exact in double)
            const double ninv = (double)imform(mc.ninv, mc.q, mc.qinv);
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = modmul_f64(reduce_f64(x[k], q, qi), ninv, q, qi);
        }
        if ((A.flags & NTT_ADD_SCALAR) && A.scale) {
            const uint64_t sadd = A.io_s[y];
#pragma unroll
            for (int k = 0; k < 16; k++) dst[(k << sh0) + tau] = cred(canon_f64(x[k], q, qi) + sadd, mc.q);
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) dst[(k << sh0) + tau] = canon_f64(x[k], q, qi);
        }
    }
    if (bzi + 1 < b1) __syncthreads();  // LDS is reused by the next entry
    }
}

// ------------------------------------------------------------------------------------
// ntt_mac_f64: forward row NTT of every non-own digit fused with the key-switch inner product (see kernels.h)
// ------------------------------------------------------------------------------------
#ifdef HE_MAC_STAMPS
__device__ uint64_t g_mac_stamps[2048 * 64];
extern "C" int he_debug_mac_stamps(uint64_t *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mac_stamps), (size_t)n * 8);
}
// diagnosis build (tools/build_variant.sh stamps "-DHE_MAC_STAMPS=1", tools/mac_timeline.py): s_memtime at the phase boundaries of 512
// mid-launch workgroups; the compiler may still sink arithmetic below a stamp, so a round and its exchange read as one phase
#define MAC_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (w >= 5000 && w < 5512 && (tau & 63) == 0) g_mac_stamps[(((w - 5000) * 4 + (tau >> 6)) * 64) + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MAC_STAMP(i) do {} while (0)
#endif
struct NttMacKArgs {
    const uint64_t *dec, *own;
    size_t dec_bs, own_bs;
    const size_t *own_tab;  // entry table of `own` (View::tab): the NTT-domain input of a coalesced key switch
    const double *keyd;
    uint64_t *o0Q, *o0P, *o1Q, *o1P;
    size_t oQ0_bs, oP0_bs, oQ1_bs, oP1_bs;
    const ModConst *mc;
    const double *twd;
    int N, a;
    NttMacArgs m;
    // entry tables (View::tab) of the four accumulator outputs when they are the caller's (GadgetProductLazy of a coalesced batch)
    const size_t *oQ0_tab = nullptr, *oP0_tab = nullptr, *oQ1_tab = nullptr, *oP1_tab = nullptr;
};
// QF64: every limb of the launch is a Q limb whose accumulators are written as doubles (NttMacArgs::q_out_f64)
template <int LOGB, bool QF64>
__global__ void __launch_bounds__((1 << LOGB) / 16 > 0 ? (1 << LOGB) / 16 : 1, 2) ntt_mac_f64_kernel(NttMacKArgs A) {
    constexpr int N2 = 1 << LOGB;
    constexpr int T = N2 / 16;
    constexpr int NR4 = LOGB / 4;
    constexpr int GREM = LOGB % 4;
    __shared__ double lds[N2 + N2 / 16];

    const int tau = threadIdx.x;
    // work list = (row, limb, batch) with the batch fastest: workgroups sharing the key / twiddle rows run together, on one XCD
    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    const unsigned w = xcd_swizzle(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nwg);
    const size_t bz = w % gridDim.x;
    const int l = (w / gridDim.x) % gridDim.y;
    const int row = w / (gridDim.x * gridDim.y);
    const int mi = A.m.mod[l];
    const ModConst mc = A.mc[mi];
    const double q = (double)mc.q, qi = mc.rq;
    const double *__restrict__ tw = A.twd + (size_t)mi * A.N;
    const int rowtw = (1 << A.a) + row;
    const size_t rowoff = (size_t)row * N2;
    const double *kbase = A.keyd + (size_t)A.m.key_limb[l] * A.N + rowoff;
    const int ql = A.m.dec_limb[l];  // Q-limb index when l < own_nq

    double acc0[16], acc1[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { acc0[k] = 0.0; acc1[k] = 0.0; }

    // the digit's own limb is the NTT-domain input itself (block-uniform); element k of a thread is word k*T + tau either way
    auto own_digit = [&](int d) -> bool {
        return A.m.own_alpha > 0 && A.m.out_view[l] == 0 && ql >= d * A.m.own_alpha && ql < (d + 1) * A.m.own_alpha;
    };
    auto digit_src = [&](int d) -> const uint64_t * {
        return own_digit(d) ? A.own + voff(A.own_tab, A.own_bs, bz) + (size_t)ql * A.N + rowoff
                            : A.dec + bz * A.dec_bs + (size_t)d * A.m.dec_dstride + (size_t)A.m.dec_limb[l] * A.N + rowoff;
    };
    double t16[15];  // twiddles of the coming radix-16 round
    if constexpr (NR4 > 0) rows_tw16_f64(t16, tw, rowtw, 0, tau >> (LOGB - 4));
    // software pipeline over the digits: the words of digit d + 1 are in flight while digit d is transformed and accumulated
    uint64_t nx[16];
    {
        const uint64_t *src = digit_src(0);
#pragma unroll
        for (int k = 0; k < 16; k++) nx[k] = ldnt(&src[k * T + tau]);
    }
    MAC_STAMP(0);
    for (int d = 0; d < A.m.beta; d++) {
        const bool is_own = own_digit(d);
        double x[16];
        MAC_STAMP(1 + d * 12 + 0);
        if (is_own && A.m.own_reduce) {  // caller-supplied words (any uint64, as MulCoeffsMontgomeryLazy accepts): bring them below 2^52
#pragma unroll
            for (int k = 0; k < 16; k++) nx[k] = bred_add_lazy(nx[k], mc.q, mc.brc0);
        }
        if (A.m.dec_f64 && !is_own) {  // the basis extension left doubles (launch_modup_fused, f64_raw)
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = __longlong_as_double((long long)nx[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = u52_to_f64(nx[k]);
        }
        if (d + 1 < A.m.beta) {
            const uint64_t *src = digit_src(d + 1);
#pragma unroll
            for (int k = 0; k < 16; k++) nx[k] = ldnt(&src[k * T + tau]);
        }
        if (!is_own) {
#pragma unroll 1
            for (int rho = 0; rho < NR4; rho++) {
                const int s0 = 4 * rho, sh = LOGB - s0 - 4;
                if (rho > 0) rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, s0, sh, false);
                MAC_STAMP(1 + d * 12 + 1 + rho * 3);
                rows_round16_f64<false>(x, t16, q, qi);
                MAC_STAMP(1 + d * 12 + 2 + rho * 3);
                // the next round's twiddles (round 0 of the next digit after the last one) are in flight across the exchange
                const int sn = rho + 1 < NR4 ? s0 + 4 : 0;
                rows_tw16_f64(t16, tw, rowtw, sn, tau >> (LOGB - sn - 4));
                rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, s0, sh, true);
                // the middle exchange of a 4096-row is local to 16 lanes (rows_sync); the last one feeds the cross-wave k T + tau
                // read below.  (The wave-local nat_e order of the plain row kernels costs this kernel 29 spilled registers.)
                if (rho + 1 < NR4) rows_sync(sh); else __syncthreads();
                MAC_STAMP(1 + d * 12 + 3 + rho * 3);
            }
            if constexpr (GREM > 0) {
                constexpr int s0 = 4 * NR4;
                rows_lds_xfer_f64<LOGB, GREM>(x, lds, tau, s0, 0, false);
                rows_round_f64<GREM, false>(x, tw, rowtw, s0, 0, tau, 0, q, qi);
                rows_lds_xfer_f64<LOGB, GREM>(x, lds, tau, s0, 0, true);
                __syncthreads();
            }
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = lds[lds_phys(k * T + tau)];
        }
        const double *k0 = kbase + (size_t)d * A.m.key_dstride, *k1 = k0 + A.m.key_kstride;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const double v = x[k];
            acc0[k] += modmul_f64(v, k0[k * T + tau], q, qi);
            acc1[k] += modmul_f64(v, k1[k * T + tau], q, qi);
        }
        MAC_STAMP(1 + d * 12 + 10);
        __syncthreads();  // LDS is reused by the next digit
        MAC_STAMP(1 + d * 12 + 11);
    }
    const int ol = A.m.out_limb[l];
    const bool isP = A.m.out_view[l] != 0;
    uint64_t *o0 = (isP ? A.o0P + voff(A.oP0_tab, A.oP0_bs, bz) : A.o0Q + voff(A.oQ0_tab, A.oQ0_bs, bz)) + (size_t)ol * A.N + rowoff;
    uint64_t *o1 = (isP ? A.o1P + voff(A.oP1_tab, A.oP1_bs, bz) : A.o1Q + voff(A.oQ1_tab, A.oQ1_bs, bz)) + (size_t)ol * A.N + rowoff;
    if constexpr (QF64) {  // the f64 ModDown epilogue reads these as doubles
        double *d0 = reinterpret_cast<double *>(o0), *d1 = reinterpret_cast<double *>(o1);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            stnt(&d0[k * T + tau], reduce_f64(acc0[k], q, qi));
            stnt(&d1[k * T + tau], reduce_f64(acc1[k], q, qi));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            o0[k * T + tau] = canon_f64(acc0[k], q, qi);
            o1[k * T + tau] = canon_f64(acc1[k], q, qi);
        }
    }
    MAC_STAMP(60);
}



// ------------------------------------------------------------------------------------
// ntt_mac_f64 at the production row sizes (4096 / 8192 coefficients): persistent workgroups, digits prefetched by LDS-DMA.
//
// What the plain kernel above loses (tools/mac_timeline.py, round 2): a fresh workgroup's first digit costs 30 k cycles against
// 13 k for the later ones (its first loads have nothing to hide behind); vector loads return in order, so every twiddle wait
// queued behind the prefetch of the next digit is a wait for that prefetch; and at 252 registers the compiler has no room to
// batch the key loads.  Here:
//  * the launch has as many workgroups as the chip holds at once (two per CU) and each walks its share of the work list; the
//    first digit of item i + 1 is the prefetch target of the last digit of item i, so only a workgroup's very first digit is cold.
//    XCD-aware order as before: the workgroups of one XCD walk one contiguous eighth of the (row, limb, batch) list side by
//    side, so those sharing a key / twiddle row use one L2;
//  * the next digit's words go from HBM straight into LDS (`global_load_lds_dwordx4`, no staging registers): every wave fetches
//    exactly the 16 x 64 words its own threads will read (lanes 0-31 the words of element k = 2j, lanes 32-63 those of 2j + 1,
//    per instruction j), into its own 8 KiB of LDS -- no workgroup barrier guards the buffer, only the wave's own vmcnt;
//  * round-0 twiddles are block-uniform (scalar registers), round-1 twiddles (16 groups x 15 per row) sit in LDS, so the only
//    ordinary vector loads of a digit are the round-2 twiddles and the two key rows, issued as three explicit batches in the
//    order they are needed (the DMA is issued after the first, and has the whole digit to land).
// Bit-identical to the plain kernel (same arithmetic on the same operands).
// ------------------------------------------------------------------------------------

// HE_MAC_R2 (round 6): the key inner product runs in the register order the LAST round of the row transform leaves (thread tau
// holds the sixteen consecutive coefficients 16 tau .. 16 tau + 15) instead of the coalesced order k T + tau: a transformed digit
// no longer stores its result back to the tile, waits for the workgroup and reads it again (one LDS exchange and one barrier of
// three fewer per digit).  The double-precision key copy is laid out to match (key_to_f64_kernel: position k T + t of a row
// holds coefficient 16 t + k, so the key rows are still read coalesced); the digit's own limb -- NTT-domain words in natural
// order -- takes one exchange through the tile; the accumulators take the exchange back where they leave the kernel (the
// ModDown epilogue subtracts first and transposes the difference: the same count as before).  0: round 3-5's order (A/B builds).
#ifndef HE_MAC_R2
#define HE_MAC_R2 1
#endif
#ifndef HE_MAC_K1_EARLY
#define HE_MAC_K1_EARLY 0
#endif
#ifndef HE_MAC_EPI_PREFETCH
#define HE_MAC_EPI_PREFETCH 1
#endif
// the tile positions of a thread's sixteen coefficients in that order (thread-private: no other thread touches them)
template <int LOGB>
__device__ __forceinline__ void mac_final_xfer(double (&x)[16], double *lds, int tau, bool store) {
    if constexpr (LOGB % 4 == 0) rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, LOGB - 4, 0, store);
    else rows_lds_xfer_f64<LOGB, LOGB % 4>(x, lds, tau, LOGB - LOGB % 4, 0, store);
}
typedef __attribute__((address_space(3))) void *he_lds_ptr;
__device__ __forceinline__ unsigned lds_byte_addr(const void *p) { return (unsigned)(uintptr_t)(he_lds_ptr)p; }
// one LDS-DMA instruction: lane i's 16 bytes at gsrc land at LDS byte lds_dst + 16 i (lds_dst wave-uniform).  hipcc does not
// count it: completion is the caller's own s_waitcnt vmcnt (MI355X_MICROARCH.md: only the issuing wave's vmcnt orders its reads)
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the ModDown epilogue of NttMacEpilogue in kernel form (strides in words; sp / tsp = IMForm(s), IMForm(IMForm(ts)) as doubles)
struct MacEpiK {
    const uint64_t *ext; size_t ext_bs;
    uint64_t *out0, *out1; size_t out0_bs, out1_bs;
    const uint64_t *w0, *w1; size_t w0_bs, w1_bs;
    const uint64_t *ta0, *ta1, *tb0, *tb1; size_t ta0_bs, ta1_bs, tb0_bs, tb1_bs;
    int ext_f64, tensor, has_w0, has_w1;
    double sp[kMaxLimbs], tsp[kMaxLimbs];
    // entry tables (View::tab) of the caller-facing operands, as ONE base pointer plus a row number per operand (eight separate
    // pointers cost this kernel seventy more spilled scalar registers): operand i's offsets are etab[row_i * nbatch + z], its
    // row the i-th nibble of etab_rows in the order out0, out1, w0, w1, ta0, ta1, tb0, tb1 (0xF: no table, z * bstride)
    const size_t *etab;
    unsigned etab_rows;
    unsigned sc_ginv;  // stores scattered by the automorphism of inverse Galois element sc_ginv (auto_dest); 0: none
    int sc_logN;
    // without the epilogue (EPI = false, SCAT = true): the giant step of a linear transformation (KsScatter): component 0 takes
    // the addend w0 (Q limbs) / w1 (P limbs) when has_w0, both accumulators go through auto_dest, gs_accum: the stores add
    int gs_accum;
};
enum { ME_OUT0 = 0, ME_OUT1, ME_W0, ME_W1, ME_TA0, ME_TA1, ME_TB0, ME_TB1 };
__device__ __forceinline__ size_t meoff(const MacEpiK &e, int which, size_t bs, size_t z, unsigned nbatch) {
    const unsigned row = (e.etab_rows >> (4 * which)) & 0xFu;
    return (e.etab && row != 0xFu) ? (size_t)ldc(reinterpret_cast<const uint64_t *>(e.etab), (size_t)row * nbatch + z) : z * bs;
}
struct NttMacDmaArgs {
    NttMacKArgs k;
    MacEpiK e;
    unsigned nbatch, nitems;  // work list = (row, limb, batch entry), batch fastest; nitems = rows * limbs * nbatch
    // per launch limb: mod | key_limb << 8 | dec_limb << 16 | out_limb << 24 | out_view << 32 -- ONE scalar load per work item (a
    // byte picked from the by-value arrays with a run-time index is a vector load, and waiting for it would drain the prefetch)
    uint64_t limb_info[kMaxLimbs];
};
#ifdef HE_MAC_STAMPS
// diagnosis build: s_memtime at the phase boundaries of every workgroup's item number HE_MAC_STAMP_ITEM (tools/mac_timeline.py --dma)
#ifndef HE_MAC_STAMP_ITEM
#define HE_MAC_STAMP_ITEM 10
#endif
#define MAC_STAMP2(i) do { __builtin_amdgcn_sched_barrier(0); if (t_cur == HE_MAC_STAMP_ITEM && g < 512u && (threadIdx.x & 63u) == 0) g_mac_stamps[((g * 4 + (threadIdx.x >> 6)) * 64) + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MAC_STAMP2(i) do {} while (0)
#endif
// EPI: the ModDown epilogue runs in this kernel (NttMacEpilogue): after the digits of an item, the rows of the two basis-extended
// accumulators' P parts arrive through the same prefetch chain as two more "digits", are transformed like them, and
// out_c = [w_c +] (NTT(ext_c) - acc_c) s is formed against the accumulator still in registers -- the Q accumulators are never
// written, and the separate forward-row + epilogue launch (HBM-bound, next to this latency-bound kernel) is gone.
// TEN (with EPI): the epilogue forms the tensor term (NttMacEpilogue::tensor) -- its own instantiation since round 6: with both
// epilogue forms in one kernel the 4096-row variant was 75 KiB of code for a 64 KiB instruction cache shared by two CUs
template <int LOGB, bool QF64, bool EPI = false, bool SCAT = false, bool TEN = false>
__global__ void __launch_bounds__((1 << LOGB) / 16, 2) ntt_mac_f64_dma_kernel(NttMacDmaArgs AA) {
    static_assert(LOGB == 12 || LOGB == 13, "production row sizes only");
    constexpr int N2 = 1 << LOGB;
    constexpr int T = N2 / 16;
    constexpr int GREM = LOGB % 4;
    const NttMacKArgs &A = AA.k;
    __shared__ double lds[N2 + N2 / 16];  // transform tile
    __shared__ double pbuf[N2];           // the coming digit: wave w owns words [1024 w, 1024 (w + 1)), word 64 k + lane = element k T + tau
    __shared__ double tw1s[16 * 15];      // round-1 twiddles of the item's row: [tau >> (LOGB - 8)][15]

    const unsigned tau = threadIdx.x, lane = tau & 63u;
    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(tau >> 6));
    const unsigned G = gridDim.x, g = blockIdx.x;
    const bool xcd = ((G | AA.nitems) & 7u) == 0;
    const unsigned span = xcd ? AA.nitems >> 3 : AA.nitems, stride = xcd ? G >> 3 : G;
    const unsigned first = xcd ? g >> 3 : g, base = xcd ? (g & 7u) * span : 0u;
    auto item_of = [&](unsigned t) -> unsigned { const unsigned i = first + t * stride; return i < span ? base + i : ~0u; };

    struct Item {
        size_t bz, rowoff;
        int ql, rowtw, out_limb, mi, l;
        bool isP;
        double q, qi;
        const double *tw, *kbase;
    };
    auto decode = [&](unsigned w) -> Item {
        Item it;
        it.bz = w % AA.nbatch;
        const int l = (int)((w / AA.nbatch) % (unsigned)A.m.nlimbs);
        const int row = (int)(w / (AA.nbatch * (unsigned)A.m.nlimbs));
        const uint64_t info = AA.limb_info[l];
        it.l = l;
        const int mi = (int)(info & 0xff);
        it.mi = mi;
        // modulus record through the scalar cache (ModConst: word 0 = q, word 8 = 1 / q as a double)
        const uint64_t *mcw = reinterpret_cast<const uint64_t *>(A.mc + mi);
        it.q = (double)ldc(mcw, 0); it.qi = __longlong_as_double((long long)ldc(mcw, 8));
        it.tw = A.twd + (size_t)mi * A.N;
        it.rowtw = (1 << A.a) + row;
        it.rowoff = (size_t)row * N2;
        it.kbase = A.keyd + (size_t)((info >> 8) & 0xff) * A.N + it.rowoff;
        it.ql = (int)((info >> 16) & 0xff);
        it.out_limb = (int)((info >> 24) & 0xff);
        it.isP = ((info >> 32) & 0xff) != 0;
        return it;
    };
    // the digit's own limb is the NTT-domain input itself (block-uniform)
    auto own_digit = [&](const Item &it, int d) -> bool {
        return A.m.own_alpha > 0 && !it.isP && it.ql >= d * A.m.own_alpha && it.ql < (d + 1) * A.m.own_alpha;
    };
    const int nd = A.m.beta + (EPI ? 2 : 0);  // EPI: "digits" beta, beta + 1 = the extension rows of components 0, 1
    auto digit_src = [&](const Item &it, int d) -> const uint64_t * {
        if (EPI && d >= A.m.beta)
            return AA.e.ext + ((size_t)(d - A.m.beta) * AA.nbatch + it.bz) * AA.e.ext_bs + (size_t)it.ql * A.N + it.rowoff;
        return own_digit(it, d) ? A.own + voff(A.own_tab, A.own_bs, it.bz) + (size_t)it.ql * A.N + it.rowoff
                                : A.dec + it.bz * A.dec_bs + (size_t)d * A.m.dec_dstride + (size_t)it.ql * A.N + it.rowoff;
    };
    // LDS-DMA of one digit row: instruction j moves elements {2j, 2j + 1} x (this wave's 64 columns)
    const unsigned pw_addr = lds_byte_addr(pbuf) + wv * 8192u;
    auto dma_digit = [&](const uint64_t *src, unsigned lane) {
        const uint64_t *gp = src + ((lane >> 5) * T + 64u * wv + 2u * (lane & 31u));
#pragma unroll
        for (int j = 0; j < 8; j++) glds16(gp + (size_t)j * 2 * T, pw_addr + (unsigned)j * 1024u);
    };

    unsigned t = 0;
    unsigned w = item_of(0);
    if (w == ~0u) return;
    Item cur = decode(w);
    dma_digit(digit_src(cur, 0), lane);
    for (;;) {
        const unsigned t_cur = t; (void)t_cur;
        MAC_STAMP2(0);
        double acc0[16], acc1[16];
#pragma unroll
        for (int k = 0; k < 16; k++) { acc0[k] = 0.0; acc1[k] = 0.0; }
        const double q = cur.q, qi = cur.qi;
        // round-1 twiddles of this row -> LDS (consumed after the first exchange barrier of the first transformed digit; the
        // previous item's readers are all past their last round 1: every wave crossed that digit's later barriers)
        if (tau < 240u) {
            const unsigned hi = tau / 15u, idx = tau - hi * 15u, u = 31u - (unsigned)__builtin_clz(idx + 1u), j = idx + 1u - (1u << u);
            tw1s[tau] = cur.tw[((unsigned)cur.rowtw << (4 + u)) + (hi << u) + j];  // its wait is the first digit's (below)
        }
        bool more = false;
        // forward row transform of x (element k T + tau in, the same order out), with the prefetch of `nsrc` and -- digits only --
        // the two key rows issued where they hide best
        auto transform = [&](int nk, auto read_back, double (&x)[16], unsigned tau, unsigned lane, const uint64_t *nsrc, const double *k0p,
                             const double *k1p, double (&kk0)[16], double (&kk1)[16], int early = 0) {
            // read_back = false: the result stays in the tile (element e at lds_phys(e)) for the caller to pick up
            // nk (block-uniform): rows of sixteen words fetched on the way -- 2 (k0p and k1p), 1 (k0p), 0
            // round-2 twiddles first, then the DMA: ordinary loads issued after it could only return after it
            double t2[15];
            rows_tw16_f64(t2, cur.tw, cur.rowtw, 8, tau >> (LOGB - 12));
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the buffer have returned
            if (nsrc) dma_digit(nsrc, lane);
            __builtin_amdgcn_sched_barrier(0);
            {   // round 0: the fifteen twiddles are the same for every thread of the workgroup -> scalar cache, scalar registers
                double t0[15];
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int j = 0; j < (1 << u); j++) t0[(1 << u) - 1 + j] = ldcd(cur.tw, (size_t)(((unsigned)cur.rowtw << u) + j));
                rows_round16_f64<false>(x, t0, q, qi);
            }
            __syncthreads();  // every wave is done with the tile (previous digit's last read) -- and tw1s is in place
            rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, 0, LOGB - 4, true);
            __syncthreads();
            rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, 4, LOGB - 8, false);
            {
                double t1[15];
                const double *tp = tw1s + (tau >> (LOGB - 8)) * 15u;
#pragma unroll
                for (int i = 0; i < 15; i++) t1[i] = tp[i];
                rows_round16_f64<false>(x, t1, q, qi);
            }
            rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, 4, LOGB - 8, true);
            rows_sync(LOGB - 8);
            rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, 8, LOGB - 12, false);
            __builtin_amdgcn_sched_barrier(0);
            if (nk >= 1) {
#pragma unroll
                for (int k = 0; k < 16; k++) kk0[k] = k0p[(unsigned)(k * T)];
            }
            // HE_MAC_K1_EARLY: nothing stands between round 2 and the products any more (HE_MAC_R2), so the second key row is
            // requested before the round as well where the registers allow it (4096-rows: 214 -> 2xx registers)
            constexpr bool k1_early = HE_MAC_R2 && HE_MAC_K1_EARLY && LOGB == 12;
            // early (the epilogue's transforms, HE_MAC_EPI_PREFETCH): both rows before the round -- held across the whole transform
            // they cost 176 spilled registers
            if ((k1_early || early != 0) && nk >= 2) {
#pragma unroll
                for (int k = 0; k < 16; k++) kk1[k] = k1p[(unsigned)(k * T)];
            }
            __builtin_amdgcn_sched_barrier(0);
            rows_round16_f64<false>(x, t2, q, qi);
            __builtin_amdgcn_sched_barrier(0);
            if (!(k1_early || early != 0) && nk >= 2) {
#pragma unroll
                for (int k = 0; k < 16; k++) kk1[k] = k1p[(unsigned)(k * T)];
            }
            __builtin_amdgcn_sched_barrier(0);
#if HE_MAC_R2
            // the result stays in the registers, in the last round's order (coefficients 16 tau .. 16 tau + 15)
            if constexpr (GREM > 0) {
                rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, 8, LOGB - 12, true);
                rows_sync(LOGB - 12);
                rows_lds_xfer_f64<LOGB, GREM>(x, lds, tau, 12, 0, false);
                rows_round_f64<GREM, false>(x, cur.tw, cur.rowtw, 12, 0, tau, 0, q, qi);
            }
            (void)read_back;
#else
            rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, 8, LOGB - 12, true);
            if constexpr (GREM > 0) {
                rows_sync(LOGB - 12);
                rows_lds_xfer_f64<LOGB, GREM>(x, lds, tau, 12, 0, false);
                rows_round_f64<GREM, false>(x, cur.tw, cur.rowtw, 12, 0, tau, 0, q, qi);
                rows_lds_xfer_f64<LOGB, GREM>(x, lds, tau, 12, 0, true);
            }
            __syncthreads();
            if constexpr (decltype(read_back)::value) {
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = lds[lds_phys(k * T + tau)];
            }
#endif
        };
        // the prefetch buffer's next content after source `d` of the current item (block-uniform): the item's next source, or the
        // next item's first digit (then `w`, `more` describe that item)
        auto next_src = [&](int d) -> const uint64_t * {
            if (d + 1 < nd) return digit_src(cur, d + 1);
            w = item_of(++t);
            more = w != ~0u;
            return more ? digit_src(decode(w), 0) : nullptr;
        };
        for (int d = 0; d < A.m.beta; d++) {
            const bool is_own = own_digit(cur, d);
            // loop-invariant address arithmetic is recomputed per digit from an opaque copy of the thread index: hoisted out of
            // the loops it costs more registers than the kernel has
            unsigned tau_d = threadIdx.x;
            asm volatile("" : "+v"(tau_d));
            const unsigned tau = tau_d, lane = tau & 63u;
            const double *pw = pbuf + wv * 1024u + lane;
            MAC_STAMP2(1 + d * 12 + 0);
            // the digit in flight has landed (this also retires every older vector load of the wave)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint64_t xi[16];
#pragma unroll
            for (int k = 0; k < 16; k++) xi[k] = (uint64_t)__double_as_longlong(pw[k * 64]);
            double x[16];
            if (is_own && A.m.own_reduce) {  // caller-supplied words (any uint64, as MulCoeffsMontgomeryLazy accepts): bring them below 2^52
#pragma unroll
                for (int k = 0; k < 16; k++) xi[k] = bred_add_lazy(xi[k], ldc(reinterpret_cast<const uint64_t *>(A.mc + cur.mi), 0), ldc(reinterpret_cast<const uint64_t *>(A.mc + cur.mi), 2));
            }
            if (A.m.dec_f64 && !is_own) {  // the basis extension left doubles (launch_modup_fused, f64_raw)
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = __longlong_as_double((long long)xi[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = u52_to_f64(xi[k]);
            }
            MAC_STAMP2(1 + d * 12 + 1);
            const uint64_t *nsrc = next_src(d);
            const double *k0p = cur.kbase + (size_t)d * A.m.key_dstride + tau, *k1p = k0p + A.m.key_kstride;
            double kk0[16], kk1[16];
            if (!is_own) {
                transform(2, std::true_type{}, x, tau, lane, nsrc, k0p, k1p, kk0, kk1);
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (nsrc) dma_digit(nsrc, lane);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 16; k++) kk0[k] = k0p[(unsigned)(k * T)];
#pragma unroll
                for (int k = 0; k < 16; k++) kk1[k] = k1p[(unsigned)(k * T)];
                __builtin_amdgcn_sched_barrier(0);
#if HE_MAC_R2
                // natural order -> the accumulators' order, through the tile (the key rows are in flight meanwhile)
                __syncthreads();  // every wave is done with the tile (the previous digit's / item's last cross-wave read)
                rows_lds_xfer_f64<LOGB, 4>(x, lds, tau, 0, LOGB - 4, true);  // element k T + tau
                __syncthreads();
                mac_final_xfer<LOGB>(x, lds, tau, false);
#endif
            }
            MAC_STAMP2(1 + d * 12 + 8);
#pragma unroll
            for (int k = 0; k < 16; k++) acc0[k] += modmul_f64(x[k], kk0[k], q, qi);
            MAC_STAMP2(1 + d * 12 + 9);
#pragma unroll
            for (int k = 0; k < 16; k++) acc1[k] += modmul_f64(x[k], kk1[k], q, qi);
            MAC_STAMP2(1 + d * 12 + 10);
        }
        if constexpr (EPI) {
            // the two extension rows, through the same prefetch chain as two more digits
            auto ext_pass = [&](auto cc) __attribute__((always_inline)) {
                const int c = cc;
                unsigned tau_d = threadIdx.x;
                asm volatile("" : "+v"(tau_d));
                const unsigned tau = tau_d, lane = tau & 63u;
                const double *pw = pbuf + wv * 1024u + lane;
                MAC_STAMP2(48 + c * 4 + 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                double x[16];
                if (AA.e.ext_f64) {  // doubles left by the basis extension (launch_modup_fused, f64_raw)
#pragma unroll
                    for (int k = 0; k < 16; k++) x[k] = pw[k * 64];
                } else {
#pragma unroll
                    for (int k = 0; k < 16; k++) x[k] = u52_to_f64((uint64_t)__double_as_longlong(pw[k * 64]));
                }
                const uint64_t *nsrc = next_src(A.m.beta + c);
                // x = NTT(ext_c) comes back in the accumulators' order (element k T + tau): the last op of ModDownQPtoQNTT against
                // the accumulator in registers (arithmetic of ntt_rows_f64_kernel's epilogue, word for word).  The operands it
                // needs from memory are requested before / while the transform runs, in the registers the key rows use in a digit.
                const bool second = c != 0;
                const size_t off = (size_t)cur.out_limb * A.N + cur.rowoff + tau;
                uint64_t *op = (second ? AA.e.out1 + meoff(AA.e, ME_OUT1, AA.e.out1_bs, cur.bz, AA.nbatch) : AA.e.out0 + meoff(AA.e, ME_OUT0, AA.e.out0_bs, cur.bz, AA.nbatch)) + off;
                const double sp = AA.e.sp[cur.l];
                const uint64_t *mcw = reinterpret_cast<const uint64_t *>(A.mc + cur.mi);
                const uint64_t qu = ldc(mcw, 0);
                // (4096-rows: compile-time; the 8192-row kernel keeps both forms behind a run-time flag -- split the same way its
                // Rotate variant measured 2 % slower, 4.35 -> 4.46 ms per c4 step of 128, with 40 KiB of code instead of 57)
                const bool tensor = LOGB == 12 ? TEN : AA.e.tensor != 0;
                const bool addw = !tensor && (second ? AA.e.has_w1 : AA.e.has_w0) != 0;
                double kd0[16], kd1[16];  // the key rows' registers: free in these transforms (no key rows on the way)
                MAC_STAMP2(48 + c * 4 + 1);
                // HE_MAC_EPI_PREFETCH (4096-rows): the operands the epilogue needs from memory -- a0, b0 (component 0) / a0, b1
                // (component 1) of the tensor term, or the addend row -- are requested at the head of the transform into those
                // registers, before its last round -- the place the key rows of a digit are requested (tools/mac_timeline.py,
                // round 6: the epilogue phases were 10.6 k and 17.1 k cycles per item for ~2.6 k and ~3.5 k cycles of arithmetic:
                // they waited for these loads)
                constexpr bool prefetch = HE_MAC_R2 && HE_MAC_EPI_PREFETCH && LOGB == 12;  // tensor + addend modes
                constexpr bool prefetch_w = false;  // the addend row alone at 8192-rows: measured equal to slightly slower (c4 2.35 -> 2.37-2.39 ms)
                const uint64_t *pa0 = nullptr, *pa1 = nullptr, *pb0 = nullptr, *pb1 = nullptr, *wp = nullptr;
                if (tensor) {
                    pa0 = AA.e.ta0 + meoff(AA.e, ME_TA0, AA.e.ta0_bs, cur.bz, AA.nbatch) + off; pa1 = AA.e.ta1 + meoff(AA.e, ME_TA1, AA.e.ta1_bs, cur.bz, AA.nbatch) + off;
                    pb0 = AA.e.tb0 + meoff(AA.e, ME_TB0, AA.e.tb0_bs, cur.bz, AA.nbatch) + off; pb1 = AA.e.tb1 + meoff(AA.e, ME_TB1, AA.e.tb1_bs, cur.bz, AA.nbatch) + off;
                } else if (addw) {
                    wp = (second ? AA.e.w1 + meoff(AA.e, ME_W1, AA.e.w1_bs, cur.bz, AA.nbatch) : AA.e.w0 + meoff(AA.e, ME_W0, AA.e.w0_bs, cur.bz, AA.nbatch)) + off;
                }
                constexpr int NB = LOGB == 12 ? 8 : 4;  // coefficients per batch (eight spill six registers in the 8192-row kernel).  Each batch waits ~2 000 cycles for its operands (tools/mac_timeline.py);
                                       // the next batch in flight as well measured equal (a batch's arithmetic covers a quarter of
                                       // that), and the registers that could hold a whole row early are what the transform runs on
                uint64_t A0[NB], A1[NB], A2[NB], A3[NB];
                if constexpr (prefetch) {
                    const double *e0 = reinterpret_cast<const double *>(tensor ? pa0 : wp);
                    const double *e1 = reinterpret_cast<const double *>(second ? pb1 : pb0);
                    transform(tensor ? 2 : (addw ? 1 : 0), std::true_type{}, x, tau, lane, nsrc, e0, e1, kd0, kd1, 1);
                } else if constexpr (prefetch_w) {
                    transform(addw ? 1 : 0, std::true_type{}, x, tau, lane, nsrc, reinterpret_cast<const double *>(wp), nullptr, kd0, kd1, 1);
                } else {
                    transform(0, std::true_type{}, x, tau, lane, nsrc, nullptr, nullptr, kd0, kd1);
                }
#if HE_MAC_R2
                // x - acc in the accumulators' order, then the transpose to the coalesced order of the operands and the stores
                // (the tile positions written are the thread's own: no barrier before the stores)
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] -= reduce_f64(acc0[k], q, qi);  // (|y| < q: x - y stays an exact integer below 2^53)
                mac_final_xfer<LOGB>(x, lds, tau, true);
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = lds[lds_phys(k * T + tau)];
#endif
                MAC_STAMP2(48 + c * 4 + 2);
                if (tensor) {
                    const double tsp = AA.e.tsp[cur.l];
                    const uint64_t twoq_u = qu << 1, brc0 = ldc(mcw, 2);
                    // caller words may be any 64-bit representative (as MRed accepts them); every pipeline of this library hands over
                    // words below 2q, which convert as they are -- the Barrett reduction runs only for a wave that met a larger one
                    auto cvtb = [&](uint64_t (&w)[NB], double (&dd)[NB]) {
                        bool big = false;
#pragma unroll
                        for (int k = 0; k < NB; k++) big = big || w[k] >= twoq_u;
                        if (__any(big)) {
#pragma unroll
                            for (int k = 0; k < NB; k++) w[k] = bred_add_lazy(w[k], qu, brc0);
                        }
#pragma unroll
                        for (int k = 0; k < NB; k++) dd[k] = u52_to_f64(w[k]);
                    };
                    auto issue = [&](auto hc, uint64_t (&r0)[NB], uint64_t (&r1)[NB], uint64_t (&r2)[NB], uint64_t (&r3)[NB]) {
                        constexpr int h = decltype(hc)::value;
#pragma unroll
                        for (int k = 0; k < NB; k++) { const unsigned e = (unsigned)((NB * h + k) * T); r0[k] = pa0[e]; r1[k] = pb0[e]; }
                        if (second) {
#pragma unroll
                            for (int k = 0; k < NB; k++) { const unsigned e = (unsigned)((NB * h + k) * T); r2[k] = pa1[e]; r3[k] = pb1[e]; }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    auto finish = [&](auto hc, uint64_t (&r0)[NB], uint64_t (&r1)[NB], uint64_t (&r2)[NB], uint64_t (&r3)[NB]) {
                        constexpr int h = decltype(hc)::value;
                        double u[NB], v[NB], u2[NB], v2[NB], wv[NB];
                        cvtb(r0, u); cvtb(r1, v);
                        if (second) {
                            cvtb(r2, u2); cvtb(r3, v2);
#pragma unroll
                            for (int k = 0; k < NB; k++) wv[k] = modmul_f64(u[k], v2[k], q, qi) + modmul_f64(u2[k], v[k], q, qi);
                        } else {
#pragma unroll
                            for (int k = 0; k < NB; k++) wv[k] = modmul_f64(u[k], v[k], q, qi);
                        }
#pragma unroll
                        for (int k = 0; k < NB; k++) {
                            const int i = NB * h + k;
#if HE_MAC_R2
                            const double tt = modmul_f64(wv[k], tsp, q, qi) + modmul_f64(x[i], sp, q, qi);
#else
                            const double yi = reduce_f64(acc0[i], q, qi);  // (|y| < q: x - y stays an exact integer below 2^53, as with the separate epilogue)
                            const double tt = modmul_f64(wv[k], tsp, q, qi) + modmul_f64(x[i] - yi, sp, q, qi);
#endif
                            stnt(&op[(unsigned)(i * T)], canon_f64(tt, q, qi));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
#define HE_H(n) std::integral_constant<int, n>{}
                    if constexpr (prefetch) {
                        // kd0 = a0, kd1 = b0 (component 0) / b1 (component 1), all sixteen coefficients, requested a transform ago
                        if (second) {  // the other two rows (b0, a1), all sixteen coefficients in one request: the accumulators are dead by now.
                                       // (Requested right after the transform instead, a transpose earlier: 2.37 -> 2.41 ms.)
#pragma unroll
                            for (int k = 0; k < NB; k++) {
                                const unsigned e = (unsigned)(k * T), e2 = (unsigned)((NB + k) * T);
                                A1[k] = pb0[e]; A2[k] = pa1[e]; A0[k] = pb0[e2]; A3[k] = pa1[e2];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        auto half = [&](auto hc, uint64_t (&vb0)[NB], uint64_t (&va1)[NB]) {
                            constexpr int h = decltype(hc)::value;
                            uint64_t w0[NB], w1[NB];
#pragma unroll
                            for (int k = 0; k < NB; k++) { w0[k] = (uint64_t)__double_as_longlong(kd0[NB * h + k]); w1[k] = (uint64_t)__double_as_longlong(kd1[NB * h + k]); }
                            // finish() forms u v (component 0) or u v2 + u2 v (component 1) from (r0, r1, r2, r3) = (u, v, u2, v2)
                            if (!second) finish(hc, w0, w1, w0, w1);
                            else finish(hc, w0, vb0, va1, w1);
                        };
                        half(HE_H(0), A1, A2);
                        half(HE_H(1), A0, A3);
                    } else if (NB == 8 && !second) {
                        // component 0 has two operand rows, not four: all sixteen coefficients' words in ONE request (the second
                        // eight in the registers component 1 uses for a1, b1)
#pragma unroll
                        for (int k = 0; k < NB; k++) {
                            const unsigned e = (unsigned)(k * T), e2 = (unsigned)((NB + k) * T);
                            A0[k] = pa0[e]; A1[k] = pb0[e]; A2[k] = pa0[e2]; A3[k] = pb0[e2];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        finish(HE_H(0), A0, A1, A0, A1);
                        finish(HE_H(1), A2, A3, A2, A3);
                    } else {
                        issue(HE_H(0), A0, A1, A2, A3); finish(HE_H(0), A0, A1, A2, A3);
                        issue(HE_H(1), A0, A1, A2, A3); finish(HE_H(1), A0, A1, A2, A3);
                    }
                    if constexpr (NB < 8) {
                        issue(HE_H(2), A0, A1, A2, A3); finish(HE_H(2), A0, A1, A2, A3);
                        issue(HE_H(3), A0, A1, A2, A3); finish(HE_H(3), A0, A1, A2, A3);
                    }
#undef HE_H
                } else if (addw) {
                    uint64_t wv[16];
                    if constexpr (prefetch || prefetch_w) {
#pragma unroll
                        for (int k = 0; k < 16; k++) wv[k] = (uint64_t)__double_as_longlong(kd0[k]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 16; k++) wv[k] = ldnt(&wp[(unsigned)(k * T)]);
                    }
#pragma unroll
                    for (int k = 0; k < 16; k++) {
#if HE_MAC_R2
                        const uint64_t v = canon_f64(modmul_f64(x[k], sp, q, qi), q, qi);
#else
                        const uint64_t v = canon_f64(modmul_f64(x[k] - reduce_f64(acc0[k], q, qi), sp, q, qi), q, qi);
#endif
                        uint64_t *dp = &op[(unsigned)(k * T)];
                        if constexpr (SCAT) dp = op - (cur.rowoff + tau) + auto_dest((unsigned)(cur.rowoff + tau) + (unsigned)(k * T), AA.e.sc_ginv, AA.e.sc_logN);
                        stnt(dp, cred(wv[k] + v, qu));
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        uint64_t *dp = &op[(unsigned)(k * T)];
                        if constexpr (SCAT) dp = op - (cur.rowoff + tau) + auto_dest((unsigned)(cur.rowoff + tau) + (unsigned)(k * T), AA.e.sc_ginv, AA.e.sc_logN);
#if HE_MAC_R2
                        stnt(dp, canon_f64(modmul_f64(x[k], sp, q, qi), q, qi));
#else
                        stnt(dp, canon_f64(modmul_f64(x[k] - reduce_f64(acc0[k], q, qi), sp, q, qi), q, qi));
#endif
                    }
                }
                MAC_STAMP2(48 + c * 4 + 3);
                // the second pass works on the other accumulator through the same names (a run-time choice between the two
                // arrays inside the pass would put both in scratch memory)
#pragma unroll
                for (int k = 0; k < 16; k++) acc0[k] = acc1[k];
            };
            if constexpr (HE_MAC_R2 && HE_MAC_EPI_PREFETCH && LOGB == 12) {
                // two copies of the pass: the second accumulator's registers are free in the second one (as ONE loop body the
                // prefetched operand rows cost 180 spilled registers)
                ext_pass(std::integral_constant<int, 0>{});
                ext_pass(std::integral_constant<int, 1>{});
            } else {
#pragma unroll 1
                for (int c = 0; c < 2; c++) ext_pass(c);
            }
        }
        const int ol = cur.out_limb;
        uint64_t *o0 = nullptr, *o1 = nullptr;
        // an opaque copy of the thread index for the item's tail, as in the digits: with the kernel-scope one the tile addresses
        // of the exchanges below are hoisted out of the persistent loop -- and spilled (the giant-step variant: 54-89 registers,
        // every reload followed by a vmcnt(0))
        unsigned tau_t = threadIdx.x;
        asm volatile("" : "+v"(tau_t));
        const unsigned tau = tau_t;
        if constexpr (!EPI) {
            o0 = (cur.isP ? A.o0P + voff(A.oP0_tab, A.oP0_bs, cur.bz) : A.o0Q + voff(A.oQ0_tab, A.oQ0_bs, cur.bz)) + (size_t)ol * A.N + cur.rowoff + tau;
            o1 = (cur.isP ? A.o1P + voff(A.oP1_tab, A.oP1_bs, cur.bz) : A.o1Q + voff(A.oQ1_tab, A.oQ1_bs, cur.bz)) + (size_t)ol * A.N + cur.rowoff + tau;
        }
        // giant step (KsScatter::plain / accumulate): the addend row and the two destination rows are requested BEFORE the
        // accumulators' exchanges (requested after them the launch was 29 % slower: 16.4 -> 21.1 ms per c5 step)
        [[maybe_unused]] uint64_t gs_av[16], gs_p0[16], gs_p1[16];
        [[maybe_unused]] const uint64_t *gs_ap = nullptr;
        if constexpr (!EPI && SCAT) {
            __builtin_amdgcn_sched_barrier(0);  // (not above the last digit's products: the key rows' registers are free only now)
            const size_t src = cur.rowoff + tau;
            if (AA.e.has_w0) gs_ap = (cur.isP ? AA.e.w1 + meoff(AA.e, ME_W1, AA.e.w1_bs, cur.bz, AA.nbatch) : AA.e.w0 + meoff(AA.e, ME_W0, AA.e.w0_bs, cur.bz, AA.nbatch)) + (size_t)ol * A.N + src;
            if (gs_ap) {
#pragma unroll
                for (int k = 0; k < 16; k++) gs_av[k] = ldnt(&gs_ap[(unsigned)(k * T)]);
            }
            if (AA.e.gs_accum) {
                const uint64_t *b0p = o0 - src;  // limb base
#pragma unroll
                for (int k = 0; k < 16; k++) gs_p0[k] = b0p[auto_dest((unsigned)src + (unsigned)(k * T), AA.e.sc_ginv, AA.e.sc_logN)];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#if HE_MAC_R2
        if constexpr (!EPI) {
            // the accumulators leave in the coalesced order: one exchange each (the tile positions a thread writes are its own, and
            // the other waves' last tile accesses were to theirs: no barrier before the first store)
            auto to_natural = [&](double (&a)[16]) {
                mac_final_xfer<LOGB>(a, lds, tau, true);
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 16; k++) a[k] = lds[lds_phys(k * T + tau)];
            };
            to_natural(acc0);
            if constexpr (SCAT) {
                // giant step: component 0 leaves now (its operands were requested before the exchange), component 1's destination
                // row is requested meanwhile -- all three rows in flight at once cost the 8192-row kernel 89 spilled registers
                const uint64_t qu = ldc(reinterpret_cast<const uint64_t *>(A.mc + cur.mi), 0);
                const size_t src = cur.rowoff + tau;
                uint64_t *b0p = o0 - src;
                const uint64_t *b1p = o1 - src;
                const bool accum = AA.e.gs_accum != 0;
                if (accum) {
#pragma unroll
                    for (int k = 0; k < 16; k++) gs_p1[k] = b1p[auto_dest((unsigned)src + (unsigned)(k * T), AA.e.sc_ginv, AA.e.sc_logN)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    uint64_t v0 = canon_f64(acc0[k], q, qi);
                    if (gs_ap) v0 = cred(v0 + gs_av[k], qu);   // ringQP.Add of canonical words
                    if (accum) v0 += gs_p0[k];                   // ...ThenAddLazy: no reduction
                    stnt(&b0p[auto_dest((unsigned)src + (unsigned)(k * T), AA.e.sc_ginv, AA.e.sc_logN)], v0);
                }
            }
            __syncthreads();  // every wave has read acc0 before acc1 lands on it
            to_natural(acc1);
        }
#endif
        if constexpr (EPI) {
            (void)o0; (void)o1; (void)ol;  // the epilogue wrote the final outputs
        } else if constexpr (SCAT) {
            // giant step (KsScatter::plain / accumulate): out_c[auto_dest(e)] (+)= CRed(acc_c[e] [+ add[e]]); component 0 left above
            const size_t src = cur.rowoff + tau;
            uint64_t *b1p = o1 - src;
            const bool accum = AA.e.gs_accum != 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                uint64_t v1 = canon_f64(acc1[k], q, qi);
                if (accum) v1 += gs_p1[k];
                stnt(&b1p[auto_dest((unsigned)src + (unsigned)(k * T), AA.e.sc_ginv, AA.e.sc_logN)], v1);
            }
        } else if constexpr (QF64) {  // the f64 ModDown epilogue reads these as doubles
            double *d0 = reinterpret_cast<double *>(o0), *d1 = reinterpret_cast<double *>(o1);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                stnt(&d0[(unsigned)(k * T)], reduce_f64(acc0[k], q, qi));
                stnt(&d1[(unsigned)(k * T)], reduce_f64(acc1[k], q, qi));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                o0[(unsigned)(k * T)] = canon_f64(acc0[k], q, qi);
                o1[(unsigned)(k * T)] = canon_f64(acc1[k], q, qi);
            }
        }
        MAC_STAMP2(60);
        if (!more) break;
        cur = decode(w);
    }
}

// workgroups the chip holds at once at this kernel's occupancy (two per CU with 256 threads, one with 512)
static unsigned mac_resident_workgroups(int logb) {
    static const unsigned cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return (unsigned)n;
    }();
    static const unsigned forced = getenv("HERING_MAC_WGS") ? (unsigned)atoi(getenv("HERING_MAC_WGS")) : 0u;
    return forced ? forced : (logb >= 13 ? cus : 2u * cus);
}

bool ntt_prod_in_supported(int logN) {
    const int b = ntt_row_bits(logN);
    return b == 12 || b == 13;
}
bool epilogue_scatter_supported(int logN) {
    const int b = ntt_row_bits(logN);
    return b == 12 || b == 13;
}
bool ntt_mac_epilogue_supported(int logN) {
    const int b = ntt_row_bits(logN);
    return b == 12 || b == 13;
}
bool ntt_mac_giant_supported(int logN) {
    const int b = ntt_row_bits(logN);
    return HE_MAC_R2 && (b == 12 || b == 13);
}
hipError_t launch_ntt_mac_f64(const RingDev &r, const NttMacArgs &a, View dec, View own, const double *keyd, View out0Q,
                              View out0P, View out1Q, View out1P, int batch, hipStream_t s, const NttMacEpilogue *epi,
                              const KsScatter *giant) {
    if (a.nlimbs <= 0 || batch <= 0) return hipSuccess;
    if (!r.twd_fwd || !keyd) return hipErrorInvalidValue;
    if (giant && (epi || a.q_out_f64 || !giant->ginv || !ntt_mac_giant_supported(r.logN))) return hipErrorInvalidValue;
    if (dec.tab || (epi && epi->ext.tab) || (a.q_out_f64 && !no_tab({out0Q, out1Q}))) return hipErrorInvalidValue;
    const int b = ntt_row_bits(r.logN), aa = r.logN - b;
    if (epi && !ntt_mac_epilogue_supported(r.logN)) return hipErrorInvalidValue;
    NttMacKArgs A{};
    A.dec = dec.p; A.dec_bs = dec.bstride; A.own = own.p; A.own_bs = own.bstride; A.own_tab = own.tab; A.keyd = keyd;
    A.o0Q = out0Q.p; A.o0P = out0P.p; A.o1Q = out1Q.p; A.o1P = out1P.p;
    A.oQ0_bs = out0Q.bstride; A.oP0_bs = out0P.bstride; A.oQ1_bs = out1Q.bstride; A.oP1_bs = out1P.bstride;
    A.oQ0_tab = out0Q.tab; A.oP0_tab = out0P.tab; A.oQ1_tab = out1Q.tab; A.oP1_tab = out1P.tab;
    A.mc = r.mc; A.twd = r.twd_fwd; A.N = r.N; A.a = aa; A.m = a;
    // beta digits in (an own digit is the input limb itself), two key rows per digit shared by the batch, two accumulators out
    double mac_bytes = ((double)a.beta * batch + 2.0 * a.beta + 2.0 * batch) * a.nlimbs * (double)r.N * 8.0;
    // with the epilogue: + the two extension rows, + the addends (w0 / w1) or the four inputs of the product, per entry and limb
    if (epi) mac_bytes += (2.0 + (epi->tensor ? 4.0 : (epi->has_w0 ? 1.0 : 0.0) + (epi->has_w1 ? 1.0 : 0.0))) * batch * a.nlimbs * (double)r.N * 8.0;
    static const bool plain_only = env_flag("HERING_MAC_PLAIN") && !HE_MAC_R2;  // (the plain kernel reads the key rows in natural order)
    if (epi || giant || b == 13 || (b == 12 && !plain_only)) {  // (the plain kernel has no 8192-row instantiation: it would spill)
        NttMacDmaArgs D;
        D.k = A;
        D.e = MacEpiK{};
        if (epi) {
            D.e.ext = epi->ext.p; D.e.ext_bs = epi->ext.bstride; D.e.ext_f64 = epi->ext_f64 ? 1 : 0;
            D.e.out0 = epi->out0.p; D.e.out0_bs = epi->out0.bstride; D.e.out1 = epi->out1.p; D.e.out1_bs = epi->out1.bstride;
            D.e.has_w0 = epi->has_w0 ? 1 : 0; D.e.has_w1 = epi->has_w1 ? 1 : 0;
            D.e.w0 = epi->w0.p; D.e.w0_bs = epi->w0.bstride; D.e.w1 = epi->w1.p; D.e.w1_bs = epi->w1.bstride;
            D.e.tensor = epi->tensor ? 1 : 0;
            if (epi->scatter_ginv && epi->tensor) return hipErrorInvalidValue;
            D.e.sc_ginv = epi->scatter_ginv; D.e.sc_logN = r.logN;
            D.e.ta0 = epi->ta0.p; D.e.ta0_bs = epi->ta0.bstride; D.e.ta1 = epi->ta1.p; D.e.ta1_bs = epi->ta1.bstride;
            D.e.tb0 = epi->tb0.p; D.e.tb0_bs = epi->tb0.bstride; D.e.tb1 = epi->tb1.p; D.e.tb1_bs = epi->tb1.bstride;
            {   // the operands' entry tables must be rows of ONE table with `batch` entries per row (api.cpp builds them so)
                const View *vs[8] = {&epi->out0, &epi->out1, &epi->w0, &epi->w1, &epi->ta0, &epi->ta1, &epi->tb0, &epi->tb1};
                const size_t *base = nullptr;
                for (const View *v : vs) if (v->tab && (!base || v->tab < base)) base = v->tab;
                D.e.etab = base;
                D.e.etab_rows = 0xFFFFFFFFu;
                for (int i = 0; i < 8 && base; i++) {
                    if (!vs[i]->tab) continue;
                    const size_t d = (size_t)(vs[i]->tab - base);
                    if (d % (size_t)batch != 0 || d / (size_t)batch >= 15) return hipErrorInvalidValue;
                    D.e.etab_rows = (D.e.etab_rows & ~(0xFu << (4 * i))) | ((unsigned)(d / (size_t)batch) << (4 * i));
                }
            }
            for (int i = 0; i < a.nlimbs; i++) { D.e.sp[i] = epi->sp[i]; D.e.tsp[i] = epi->tsp[i]; }
        }
        if (giant) {
            D.e.sc_ginv = giant->ginv; D.e.sc_logN = r.logN;
            D.e.has_w0 = giant->add0.p ? 1 : 0;
            D.e.w0 = giant->add0.p; D.e.w0_bs = giant->add0.bstride; D.e.w1 = giant->add0P.p; D.e.w1_bs = giant->add0P.bstride;
            D.e.gs_accum = giant->accumulate ? 1 : 0;
            // entry tables of the addend: rows of the request's table, like the epilogue's operands
            const View *vs[2] = {&giant->add0, &giant->add0P};
            const size_t *base = nullptr;
            for (const View *v : vs) if (v->tab && (!base || v->tab < base)) base = v->tab;
            D.e.etab = base;
            D.e.etab_rows = 0xFFFFFFFFu;
            for (int i = 0; i < 2 && base; i++) {
                if (!vs[i]->tab) continue;
                const size_t d = (size_t)(vs[i]->tab - base);
                if (d % (size_t)batch != 0 || d / (size_t)batch >= 15) return hipErrorInvalidValue;
                const int which = i == 0 ? ME_W0 : ME_W1;
                D.e.etab_rows = (D.e.etab_rows & ~(0xFu << (4 * which))) | ((unsigned)(d / (size_t)batch) << (4 * which));
            }
            // + the addend row and, accumulating, the two destination rows read
            mac_bytes += ((giant->add0.p ? 1.0 : 0.0) + (giant->accumulate ? 2.0 : 0.0)) * batch * a.nlimbs * (double)r.N * 8.0;
        }
        D.nbatch = (unsigned)batch;
        D.nitems = (unsigned)batch * (unsigned)a.nlimbs * (1u << aa);
        for (int i = 0; i < a.nlimbs; i++)
            D.limb_info[i] = (uint64_t)a.mod[i] | (uint64_t)a.key_limb[i] << 8 | (uint64_t)a.dec_limb[i] << 16 |
                             (uint64_t)a.out_limb[i] << 24 | (uint64_t)a.out_view[i] << 32;
        // as many workgroups as run at once, every one with the same number of items (rounded up)
        unsigned G = mac_resident_workgroups(b);
        if (D.nitems <= G) G = D.nitems;
        else {
            const unsigned per = (D.nitems + G - 1) / G;
            G = (D.nitems + per - 1) / per;
            if ((D.nitems & 7u) == 0) G = (G + 7u) & ~7u;
        }
        ProfScope ps(K_NTT_MAC_F64, s, mac_bytes);
        if (epi && D.e.sc_ginv) {
            if (b == 12) hipLaunchKernelGGL((ntt_mac_f64_dma_kernel<12, false, true, true>), dim3(G), dim3(256), 0, s, D);
            else hipLaunchKernelGGL((ntt_mac_f64_dma_kernel<13, false, true, true>), dim3(G), dim3(512), 0, s, D);
        } else if (epi && D.e.tensor && b == 12) {
            hipLaunchKernelGGL((ntt_mac_f64_dma_kernel<12, false, true, false, true>), dim3(G), dim3(256), 0, s, D);
        } else if (epi) {
            if (b == 12) hipLaunchKernelGGL((ntt_mac_f64_dma_kernel<12, false, true>), dim3(G), dim3(256), 0, s, D);
            else hipLaunchKernelGGL((ntt_mac_f64_dma_kernel<13, false, true>), dim3(G), dim3(512), 0, s, D);
        } else if (giant) {
            if (b == 12) hipLaunchKernelGGL((ntt_mac_f64_dma_kernel<12, false, false, true>), dim3(G), dim3(256), 0, s, D);
            else hipLaunchKernelGGL((ntt_mac_f64_dma_kernel<13, false, false, true>), dim3(G), dim3(512), 0, s, D);
        } else if (b == 12) {
            if (a.q_out_f64) hipLaunchKernelGGL((ntt_mac_f64_dma_kernel<12, true>), dim3(G), dim3(256), 0, s, D);
            else hipLaunchKernelGGL((ntt_mac_f64_dma_kernel<12, false>), dim3(G), dim3(256), 0, s, D);
        } else {
            if (a.q_out_f64) hipLaunchKernelGGL((ntt_mac_f64_dma_kernel<13, true>), dim3(G), dim3(512), 0, s, D);
            else hipLaunchKernelGGL((ntt_mac_f64_dma_kernel<13, false>), dim3(G), dim3(512), 0, s, D);
        }
        return hipGetLastError();
    }
    dim3 grid(batch, a.nlimbs, 1u << aa);
#ifdef HE_MAC_STAMPS
    {   // diagnosis build only: HERING_MAC_ABL makes the key (1, 8), input (2) and output (4) streams cache-resident
        static const int abl = getenv("HERING_MAC_ABL") ? atoi(getenv("HERING_MAC_ABL")) : 0;
        if (abl & 1) { A.m.key_dstride = 0; A.m.key_kstride = 0; }
        if (abl & 2) { A.dec_bs = 0; A.own_bs = 0; A.m.dec_dstride = 0; }
        if (abl & 4) { A.oQ0_bs = A.oP0_bs = A.oQ1_bs = A.oP1_bs = 0; }
        if (abl & 8) { for (int i = 0; i < a.nlimbs; i++) A.m.key_limb[i] = 0; }
    }
#endif
    ProfScope ps(K_NTT_MAC_F64, s, mac_bytes);
#define HE_MAC_CASE(B)                                                                                      \
    case B:                                                                                                 \
        if (a.q_out_f64) hipLaunchKernelGGL((ntt_mac_f64_kernel<B, true>), grid, dim3((1 << B) / 16), 0, s, A);  \
        else hipLaunchKernelGGL((ntt_mac_f64_kernel<B, false>), grid, dim3((1 << B) / 16), 0, s, A);             \
        break;
    switch (b) {
        HE_MAC_CASE(4) HE_MAC_CASE(5) HE_MAC_CASE(6) HE_MAC_CASE(7) HE_MAC_CASE(8) HE_MAC_CASE(9) HE_MAC_CASE(10)
        HE_MAC_CASE(11) HE_MAC_CASE(12)
        default: return hipErrorInvalidValue;
    }
#undef HE_MAC_CASE
    return hipGetLastError();
}

struct KeyF64Args {
    const uint64_t *key;
    double *keyd;
    const ModConst *mc;
    int N, nlimbs;
    int rowbits;  // > 0: rows of 2^rowbits coefficients are stored in the NTT + MAC kernel's accumulator order (HE_MAC_R2)
    uint8_t mod[kMaxLimbs];
};
__global__ void __launch_bounds__(256) key_to_f64_kernel(KeyF64Args A) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= A.N) return;
    const int l = blockIdx.y;
    const size_t base = ((size_t)blockIdx.z * A.nlimbs + l) * A.N;
    const ModConst m = A.mc[A.mod[l]];
    // coefficient e = 16 t + k of a row goes to position k T + t (T = row / 16 threads): thread t of the kernel reads its k-th key
    // word at k T + t -- coalesced -- and finds the coefficient its k-th register holds after the last round of the transform
    size_t dst = (size_t)x;
    if (A.rowbits > 0) {
        const unsigned e = (unsigned)x & ((1u << A.rowbits) - 1u), T = 1u << (A.rowbits - 4);
        dst = (size_t)((unsigned)x - e) + (e & 15u) * T + (e >> 4);
    }
    A.keyd[base + dst] = (m.q >> kF64Bits) == 0 ? (double)imform(A.key[base + x], m.q, m.qinv) : 0.0;
}
hipError_t launch_key_to_f64(const RingDev &r, const uint64_t *key, double *keyd, int nblocks, const uint8_t *limb_mod_host,
                             int nlimbs, hipStream_t s) {
    KeyF64Args A{};
    A.key = key; A.keyd = keyd; A.mc = r.mc; A.N = r.N; A.nlimbs = nlimbs;
#if HE_MAC_R2
    A.rowbits = ntt_row_bits(r.logN) >= 12 ? ntt_row_bits(r.logN) : 0;  // the persistent kernel's rows (launch_ntt_mac_f64)
#else
    A.rowbits = 0;
#endif
    for (int i = 0; i < nlimbs; i++) A.mod[i] = limb_mod_host[i];
    dim3 grid((unsigned)((r.N + 255) / 256), nlimbs, nblocks), block(256);
    hipLaunchKernelGGL(key_to_f64_kernel, grid, block, 0, s, A);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// ntt_cols: LOGA strided stages in registers.  With cb = 0 they are the LOGA outermost stages of the transform, on elements
// strided by N / 2^LOGA.  With cb > 0 (logN >= 19, where the a = logN - 13 column stages no longer fit one thread's registers)
// an outer pass has done stages 0 .. cb-1 and this one does stages cb .. cb+LOGA-1: inside each of the 2^cb blocks of N / 2^cb
// coefficients, block j playing the part of "row" 2^cb + j of the twiddle table exactly as a row does in ntt_rows.
// grid = (N / 2^LOGA / 256, limbs, batch), block = 256.
// ------------------------------------------------------------------------------------
// INNER = false is the outer (or only) pass: its twiddle indices are compile-time constants, i.e. scalar loads.
template <int LOGA, bool INV, bool INNER = false>
__global__ void __launch_bounds__(256) ntt_cols_kernel(NttArgs A) {
    constexpr int R = 1 << LOGA;
    const int cbits = INNER ? A.cb : 0;
    const int N2 = (A.N >> cbits) >> LOGA;    // stride of a thread's coefficients
    const int cg = blockIdx.x * blockDim.x + threadIdx.x;
    if (cg >= (A.N >> LOGA)) return;
    const int blk = INNER ? cg >> __builtin_ctz((unsigned)N2) : 0, c = INNER ? cg & (N2 - 1) : cg;  // (N2 is a power of two)
    const int rowtw = INNER ? (1 << cbits) + blk : 1;
    const int y = blockIdx.y;
    const int il = A.tab.in_limb[y], ol = A.tab.out_limb[y], mi = A.tab.mod[y];
    const ModConst mc = A.mc[mi];
    const uint64_t q = mc.q, qinv = mc.qinv, twoq = mc.q << 1;
    const uint64_t *__restrict__ tw = A.tw + (size_t)mi * A.N;
    const size_t e0 = (size_t)blk * ((size_t)N2 << LOGA) + c;
    const uint64_t *__restrict__ src = A.in + voff(A.in_tab, A.in_bs, blockIdx.z) + (size_t)il * A.N + e0;
    uint64_t *__restrict__ dst = A.out + voff(A.out_tab, A.out_bs, blockIdx.z) + (size_t)ol * A.N + e0;

    uint64_t x[R];
#pragma unroll
    for (int r = 0; r < R; r++) x[r] = src[(size_t)r * N2];
    if (!INV && (A.flags & NTT_ADD_SCALAR)) {
        const uint64_t sadd = A.io_s[y];
#pragma unroll
        for (int r = 0; r < R; r++) x[r] += sadd;
    }
    if (A.flags & NTT_REDUCE_INPUT) {
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = bred_add_lazy(x[r], q, mc.brc0);
    }
    if constexpr (!INV) {
        const bool nc = (q >> kNoCorrBits) == 0;
#pragma unroll
        for (int s = 0; s < LOGA; s++) {
            const int d = 1 << (LOGA - 1 - s);
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (r & d) continue;
                if (nc) bfly_fwd_nc(x[r], x[r + d], tw[(rowtw << s) + (r >> (LOGA - s))], q, qinv);
                else bfly_fwd(x[r], x[r + d], tw[(rowtw << s) + (r >> (LOGA - s))], q, twoq, qinv);
            }
        }
    } else {
#pragma unroll
        for (int s = LOGA - 1; s >= 0; s--) {
            const int d = 1 << (LOGA - 1 - s);
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (r & d) continue;
                const uint64_t wv = tw[(rowtw << s) + (r >> (LOGA - s))];
                if (s == 0 && A.scale) {
                    const uint64_t wn = mred(wv, mc.ninv, q, qinv);
                    bfly_inv_scaled(x[r], x[r + d], wn, mc.ninv, q, twoq, qinv);
                } else {
                    bfly_inv(x[r], x[r + d], wv, q, twoq, qinv);
                }
            }
        }
    }
    if (INV && (A.flags & NTT_ADD_SCALAR) && A.scale) {  // canonical outputs of the finished inverse
        const uint64_t sadd = A.io_s[y];
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = cred(x[r] + sadd, q);
    }
#pragma unroll
    for (int r = 0; r < R; r++) dst[(size_t)r * N2] = x[r];
}

// algorithmic bytes of one row-pass launch: in + out per (entry, limb), plus the epilogue's streams -- y, the addend w or, in
// tensor mode, the four inputs of the product shared by the two components of an entry (two per component)
static double rows_bytes(dim3 grid, const NttArgs &A, int logb) {
    const double entries = A.epi_tensor ? 2.0 * A.zsplit : (A.nbatch_prof > 0 ? (double)A.nbatch_prof : (double)grid.x);
    double streams = A.tprod ? 4.0 : 2.0;  // (product prologue: two inputs read, the product and the transform written)
    if (A.epi) streams += 1.0 + (A.epi_tensor ? 2.0 : (A.epi == 2 || (A.zsplit && A.epi2 == 2)) ? 1.0 : 0.0);
    return entries * grid.y * grid.z * (double)(1u << logb) * 8.0 * streams;
}
template <bool INV, bool NC>
static hipError_t launch_rows_nc(int logb, dim3 grid, const NttArgs &A, hipStream_t s) {
    static const bool no_lean = env_flag("HERING_NO_LEAN_INV_ROWS");
    // the production row sizes without the N^-1 fold: the lean inverse variant (see ntt_rows_kernel)
    if constexpr (INV) {
        if (!no_lean && (logb == 12 || logb == 13) && !A.scale) {
            ProfScope ps(K_NTT_ROWS_INV, s, rows_bytes(grid, A, logb));
            if (logb == 12) hipLaunchKernelGGL((ntt_rows_kernel<12, true, false, true>), grid, dim3(256), 0, s, A);
            else hipLaunchKernelGGL((ntt_rows_kernel<13, true, false, true>), grid, dim3(512), 0, s, A);
            return hipGetLastError();
        }
    }
    if constexpr (!INV) {
        if (A.sc_ginv) {  // scattered epilogue stores (NttEpilogue::scatter_ginv): production row sizes only
            if ((logb != 12 && logb != 13) || !A.epi) return hipErrorInvalidValue;
            ProfScope ps(K_NTT_ROWS_FWD, s, rows_bytes(grid, A, logb));
            if (logb == 12) hipLaunchKernelGGL((ntt_rows_kernel<12, false, NC, false, true>), grid, dim3(256), 0, s, A);
            else hipLaunchKernelGGL((ntt_rows_kernel<13, false, NC, false, true>), grid, dim3(512), 0, s, A);
            return hipGetLastError();
        }
    }
#define HE_ROWS_CASE(B)                                                                           \
    case B:                                                                                       \
        { ProfScope ps(INV ? K_NTT_ROWS_INV : K_NTT_ROWS_FWD, s, rows_bytes(grid, A, B));                                 \
        hipLaunchKernelGGL((ntt_rows_kernel<B, INV, NC>), grid, dim3((1 << B) / 16), 0, s, A); }   \
        break;
    switch (logb) {
        HE_ROWS_CASE(4) HE_ROWS_CASE(5) HE_ROWS_CASE(6) HE_ROWS_CASE(7) HE_ROWS_CASE(8) HE_ROWS_CASE(9)
        HE_ROWS_CASE(10) HE_ROWS_CASE(11) HE_ROWS_CASE(12) HE_ROWS_CASE(13)
        default: return hipErrorInvalidValue;
    }
#undef HE_ROWS_CASE
    return hipGetLastError();
}
template <bool INV>
static hipError_t launch_rows_f64(int logb, dim3 grid, const NttArgs &A, hipStream_t s) {
    if constexpr (INV) {
        if (A.tprod) {  // product prologue (NttProdIn): production row sizes, one entry per workgroup
            if (logb != 12 && logb != 13) return hipErrorInvalidValue;
            ProfScope ps(K_NTT_ROWS_INV_F64, s, rows_bytes(grid, A, logb));
            if (logb == 12) hipLaunchKernelGGL((ntt_rows_f64_kernel<12, true, true>), grid, dim3(256), 0, s, A);
            else hipLaunchKernelGGL((ntt_rows_f64_kernel<13, true, true>), grid, dim3(512), 0, s, A);
            return hipGetLastError();
        }
    }
    if constexpr (!INV) {
        if (A.sc_ginv) {
            if ((logb != 12 && logb != 13) || !A.epi) return hipErrorInvalidValue;
            ProfScope ps(K_NTT_ROWS_FWD_F64, s, rows_bytes(grid, A, logb));
            if (logb == 12) hipLaunchKernelGGL((ntt_rows_f64_kernel<12, false, false, true>), grid, dim3(256), 0, s, A);
            else hipLaunchKernelGGL((ntt_rows_f64_kernel<13, false, false, true>), grid, dim3(512), 0, s, A);
            return hipGetLastError();
        }
    }
#define HE_ROWSF_CASE(B)                                                                          \
    case B:                                                                                       \
        { ProfScope ps(INV ? K_NTT_ROWS_INV_F64 : K_NTT_ROWS_FWD_F64, s, rows_bytes(grid, A, B));                         \
        hipLaunchKernelGGL((ntt_rows_f64_kernel<B, INV>), grid, dim3((1 << B) / 16), 0, s, A); }   \
        break;
    switch (logb) {
        HE_ROWSF_CASE(4) HE_ROWSF_CASE(5) HE_ROWSF_CASE(6) HE_ROWSF_CASE(7) HE_ROWSF_CASE(8) HE_ROWSF_CASE(9)
        HE_ROWSF_CASE(10) HE_ROWSF_CASE(11) HE_ROWSF_CASE(12) HE_ROWSF_CASE(13)
        default: return hipErrorInvalidValue;
    }
#undef HE_ROWSF_CASE
    return hipGetLastError();
}
// Launches are split by modulus size (cls[] per modulus: 2 = below 2^47 -> double-precision kernel,
// 1 = below 2^58 -> correction-free integer butterflies (forward only), 0 = Harvey [0,4q) form).
template <bool INV>
static hipError_t launch_rows(int logb, dim3 grid, const NttArgs &A, const uint8_t *cls, hipStream_t s) {
    if (!cls) return launch_rows_nc<INV, false>(logb, grid, A, s);
    NttArgs P[3] = {A, A, A};
    for (auto &p : P) p.tab.n = 0;
    for (int i = 0; i < A.tab.n; i++) {
        int c = cls[A.tab.mod[i]];
        if (c == 2 && !A.twd) c = 1;
        if (INV && c == 1) c = 0;
        NttArgs &D = P[c];
        D.tab.in_limb[D.tab.n] = A.tab.in_limb[i];
        D.tab.out_limb[D.tab.n] = A.tab.out_limb[i];
        D.tab.mod[D.tab.n] = A.tab.mod[i];
        D.epi_s[D.tab.n] = A.epi_s[i];
        D.io_s[D.tab.n] = A.io_s[i];
        D.epi_ts[D.tab.n] = A.epi_ts[i];
        D.tab.n++;
    }
    hipError_t e = hipSuccess;
    if (P[2].tab.n) {
        // entries per workgroup (inverse only): 2 while the launch still has well over the ~1500 workgroups that fill the chip
        static const int forced = getenv("HERING_ROWS_ITERS") ? atoi(getenv("HERING_ROWS_ITERS")) : 0;
        const size_t wgs = (size_t)grid.x * P[2].tab.n * grid.z;
        int iters = (!INV || logb > 12 || P[2].tprod) ? 1 : forced > 0 ? forced : (wgs >= 6144 ? 2 : 1);
        if (iters > (int)grid.x) iters = (int)grid.x;
        P[2].nbatch = (int)grid.x; P[2].iters = iters; P[2].nbatch_prof = (int)grid.x;
        dim3 g2((grid.x + iters - 1) / iters, P[2].tab.n, grid.z);
        e = launch_rows_f64<INV>(logb, g2, P[2], s);
    }
    P[0].tprod = P[1].tprod = 0;  // (the integer kernels read their input as it is)
    if (e == hipSuccess && P[1].tab.n) { dim3 g2(grid.x, P[1].tab.n, grid.z); e = launch_rows_nc<INV, true>(logb, g2, P[1], s); }
    if (e == hipSuccess && P[0].tab.n) { dim3 g2(grid.x, P[0].tab.n, grid.z); e = launch_rows_nc<INV, false>(logb, g2, P[0], s); }
    return e;
}
template <bool INV>
static hipError_t launch_cols(int loga, dim3 grid, const NttArgs &A, hipStream_t s) {
#define HE_COLS_CASE(Av)                                                                 \
    case Av:                                                                             \
        { ProfScope ps(INV ? K_NTT_COLS_INV : K_NTT_COLS_FWD, s, 2.0 * A.tab.n * grid.z * (double)A.N * 8.0);                        \
        hipLaunchKernelGGL((ntt_cols_kernel<Av, INV>), grid, dim3(256), 0, s, A); }       \
        break;
    if (A.cb > 0) {  // inner pass of a two-pass column stage (logN = 19, 20): always four stages
        if (loga != 4) return hipErrorInvalidValue;
        ProfScope ps(INV ? K_NTT_COLS_INV : K_NTT_COLS_FWD, s, 2.0 * A.tab.n * grid.z * (double)A.N * 8.0);
        hipLaunchKernelGGL((ntt_cols_kernel<4, INV, true>), grid, dim3(256), 0, s, A);
        return hipGetLastError();
    }
    switch (loga) {
        HE_COLS_CASE(1) HE_COLS_CASE(2) HE_COLS_CASE(3) HE_COLS_CASE(4) HE_COLS_CASE(5)
        default: return hipErrorInvalidValue;
    }
#undef HE_COLS_CASE
    return hipGetLastError();
}

static void set_epilogue(NttArgs &A, const NttEpilogue &epi, int n) {
    A.epi_y_f64 = epi.y_small_f64 ? 1 : 0;
    A.epi_y_reduce = epi.y_reduce ? 1 : 0;
    A.epi = epi.has_w ? 2 : 1;
    A.epi_y = epi.y.p; A.epi_y_bs = epi.y.bstride; A.epi_y_tab = epi.y.tab;
    A.epi_w = epi.w.p; A.epi_w_bs = epi.w.bstride; A.epi_w_tab = epi.w.tab;
    for (int i = 0; i < n; i++) A.epi_s[i] = epi.s[i];
    A.epi_tensor = epi.tensor ? 1 : 0;
    A.sc_ginv = epi.tensor ? 0u : epi.scatter_ginv;
    A.sc_logN = 0;
    for (int nn = A.N; nn > 1; nn >>= 1) A.sc_logN++;
    if (epi.tensor) {
        A.ta0 = epi.ta0.p; A.ta1 = epi.ta1.p; A.tb0 = epi.tb0.p; A.tb1 = epi.tb1.p;
        A.ta0_bs = epi.ta0.bstride; A.ta1_bs = epi.ta1.bstride; A.tb0_bs = epi.tb0.bstride; A.tb1_bs = epi.tb1.bstride;
        A.ta0_tab = epi.ta0.tab; A.ta1_tab = epi.ta1.tab; A.tb0_tab = epi.tb0.tab; A.tb1_tab = epi.tb1.tab;
        for (int i = 0; i < n; i++) A.epi_ts[i] = epi.ts[i];
    }
    if (epi.zsplit > 0) {
        A.zsplit = epi.zsplit; A.epi2 = epi.has_w2 ? 2 : 1;
        A.out2 = epi.out2.p; A.out2_bs = epi.out2.bstride; A.out2_tab = epi.out2.tab;
        A.epi_y2 = epi.y2.p; A.epi_y2_bs = epi.y2.bstride; A.epi_y2_tab = epi.y2.tab;
        A.epi_w2 = epi.w2.p; A.epi_w2_bs = epi.w2.bstride; A.epi_w2_tab = epi.w2.tab;
    }
}
hipError_t launch_ntt(const RingDev &r, const LimbTab &tab, View in, View out, int batch, bool inverse, int flags,
                      hipStream_t s, const uint64_t *io_scalar, const NttEpilogue *epi) {
    if (tab.n <= 0 || batch <= 0) return hipSuccess;
    const int n = r.logN;
    if (n < 4 || n > kMaxLogN) return hipErrorInvalidValue;
    if (epi && (inverse || epi->zsplit > 0)) return hipErrorInvalidValue;
    const int b = ntt_row_bits(n), a = n - b;
    // column stages: one pass of up to five (32 coefficients per thread); logN = 19, 20 (a = 6, 7) take an outer pass of a - 4
    // and an inner pass of 4 inside the 2^(a-4) blocks the outer pass leaves (ntt_cols_kernel, NttArgs::cb)
    const int a_out = a > 5 ? a - 4 : a, a_in = a - a_out;
    NttArgs A{};  // (zero-filled: the per-limb arrays are copied by launch_rows whether or not an epilogue set them)
    A.mc = r.mc;
    A.N = r.N;
    A.a = a;
    A.tab = tab;
    A.epi = 0; A.epi_y = A.epi_w = nullptr; A.epi_y_bs = A.epi_w_bs = 0;
    A.zsplit = 0; A.epi2 = 0; A.out2 = nullptr; A.out2_bs = 0; A.epi_y2 = A.epi_w2 = nullptr; A.epi_y2_bs = A.epi_w2_bs = 0;
    A.epi_y_f64 = 0; A.epi_y_reduce = 0; A.epi_tensor = 0;
    const int sflag = io_scalar ? NTT_ADD_SCALAR : 0;
    for (int i = 0; i < tab.n; i++) A.io_s[i] = io_scalar ? io_scalar[i] : 0;
    dim3 grows(batch, tab.n, 1u << a);
    dim3 gcols((unsigned)(((r.N >> a_out) + 255) / 256), tab.n, batch);
    dim3 gcols_in((unsigned)(((r.N >> (a_in > 0 ? a_in : 1)) + 255) / 256), tab.n, batch);
    hipError_t e;
    if (!inverse) {
        A.tw = r.tw_fwd;
        A.twd = r.twd_fwd;
        A.scale = 0;
        if (a > 0) {
            A.in = in.p; A.in_bs = in.bstride; A.in_tab = in.tab; A.out = out.p; A.out_bs = out.bstride; A.out_tab = out.tab;
            A.flags = (flags & NTT_REDUCE_INPUT) | sflag;
            if ((e = launch_cols<false>(a_out, gcols, A, s)) != hipSuccess) return e;
            // second pass in place on `out`: limbs are now addressed by out_limb
            NttArgs B = A;
            for (int i = 0; i < tab.n; i++) B.tab.in_limb[i] = tab.out_limb[i];
            B.in = out.p; B.in_bs = out.bstride; B.in_tab = out.tab;
            if (a_in > 0) {
                B.flags = 0; B.cb = a_out;
                if ((e = launch_cols<false>(a_in, gcols_in, B, s)) != hipSuccess) return e;
                B.cb = 0;
            }
            B.flags = flags & NTT_LAZY_OUT;
            if (epi) {
                set_epilogue(B, *epi, tab.n);
                if (epi->has_dst) { B.out = epi->dst.p; B.out_bs = epi->dst.bstride; B.out_tab = epi->dst.tab; }
            }
            return launch_rows<false>(b, grows, B, r.host_small, s);
        }
        A.in = in.p; A.in_bs = in.bstride; A.in_tab = in.tab; A.out = out.p; A.out_bs = out.bstride; A.out_tab = out.tab;
        A.flags = flags | sflag;
        if (epi) {
            set_epilogue(A, *epi, tab.n);
            if (epi->has_dst) { A.out = epi->dst.p; A.out_bs = epi->dst.bstride; A.out_tab = epi->dst.tab; }
        }
        return launch_rows<false>(b, grows, A, r.host_small, s);
    }
    A.tw = r.tw_inv;
    A.twd = r.twd_inv;
    A.in = in.p; A.in_bs = in.bstride; A.in_tab = in.tab; A.out = out.p; A.out_bs = out.bstride; A.out_tab = out.tab;
    A.flags = (flags & NTT_REDUCE_INPUT) | (a == 0 ? sflag : 0);
    A.scale = (a == 0);
    if ((e = launch_rows<true>(b, grows, A, r.host_small, s)) != hipSuccess) return e;
    if (a > 0) {
        NttArgs B = A;
        for (int i = 0; i < tab.n; i++) B.tab.in_limb[i] = tab.out_limb[i];
        B.in = out.p; B.in_bs = out.bstride; B.in_tab = out.tab;
        if (a_in > 0) {
            B.flags = 0; B.scale = 0; B.cb = a_out;
            if ((e = launch_cols<true>(a_in, gcols_in, B, s)) != hipSuccess) return e;
            B.cb = 0;
        }
        B.flags = sflag;
        B.scale = 1;
        return launch_cols<true>(a_out, gcols, B, s);
    }
    return hipSuccess;
}

hipError_t launch_ntt_rows(const RingDev &r, const LimbTab &tab, View in, View out, int batch, bool inverse, int flags,
                           hipStream_t s, const NttEpilogue *epi, const NttProdIn *prod) {
    if (tab.n <= 0 || batch <= 0) return hipSuccess;
    const int n = r.logN;
    if (n < 4 || n > 17) return hipErrorInvalidValue;
    if (epi && inverse) return hipErrorInvalidValue;
    if (prod && (!inverse || !r.twd_inv || !ntt_prod_in_supported(n))) return hipErrorInvalidValue;
    // entry tables: every operand but the product prologue's copy of c2 (always a scratch batch) takes one
    // (forward tensor-mode epilogue: the workgroup -> entry map of that launch shape is not the input table's)
    if ((in.tab && epi && epi->tensor) || (prod && prod->c.tab)) return hipErrorInvalidValue;
    const int b = ntt_row_bits(n), a = n - b;
    NttArgs A{};
    A.mc = r.mc; A.N = r.N; A.a = a; A.tab = tab;
    A.epi = 0; A.epi_y = A.epi_w = nullptr; A.epi_y_bs = A.epi_w_bs = 0;
    A.zsplit = 0; A.epi2 = 0; A.out2 = nullptr; A.out2_bs = 0; A.epi_y2 = A.epi_w2 = nullptr; A.epi_y2_bs = A.epi_w2_bs = 0;
    A.epi_y_f64 = 0; A.epi_y_reduce = 0;
    for (int i = 0; i < tab.n; i++) A.io_s[i] = 0;
    A.epi_tensor = 0;
    if (epi) {
        if (epi->tensor && (epi->zsplit <= 0 || batch != 2 * epi->zsplit)) return hipErrorInvalidValue;
        set_epilogue(A, *epi, tab.n);
    }
    A.in = in.p; A.in_bs = in.bstride; A.in_tab = in.tab; A.out = out.p; A.out_bs = out.bstride; A.out_tab = out.tab;
    A.flags = flags;
    if (prod) {
        A.tprod = 1;
        A.ta1 = prod->a.p; A.ta1_bs = prod->a.bstride; A.tb1 = prod->b.p; A.tb1_bs = prod->b.bstride;
        A.ta1_tab = prod->a.tab; A.tb1_tab = prod->b.tab;
        A.out2 = prod->c.p; A.out2_bs = prod->c.bstride;
        for (int i = 0; i < tab.n; i++) A.epi_ts[i] = prod->ts[i];
    }
    // tensor mode: both components of 8 entries per 16 consecutive workgroups (see NttEpilogue::tensor)
    dim3 grows(A.epi_tensor ? (unsigned)((epi->zsplit + 7) / 8 * 16) : (unsigned)batch, tab.n, 1u << a);
    if (!inverse) {
        A.tw = r.tw_fwd; A.twd = r.twd_fwd; A.scale = 0;
        return launch_rows<false>(b, grows, A, r.host_small, s);
    }
    A.tw = r.tw_inv; A.twd = r.twd_inv; A.scale = (a == 0);
    return launch_rows<true>(b, grows, A, r.host_small, s);
}

// ------------------------------------------------------------------------------------
// conjugate-invariant fold around the standard butterfly network (see kernels.h)
// ------------------------------------------------------------------------------------
struct CiFoldArgs {
    const uint64_t *in;
    uint64_t *out;
    size_t in_bs, out_bs;
    const ModConst *mc;
    int N, inverse, reduce_input;
    uint8_t in_limb[kMaxLimbs], out_limb[kMaxLimbs], mod[kMaxLimbs];
};
__global__ void __launch_bounds__(256) ci_fold_kernel(CiFoldArgs A) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. N/2
    if (j > A.N / 2) return;
    const ModConst m = A.mc[A.mod[blockIdx.y]];
    const uint64_t q = m.q, qinv = m.qinv, twoq = m.q << 1;
    const uint64_t *in = A.in + (size_t)blockIdx.z * A.in_bs + (size_t)A.in_limb[blockIdx.y] * A.N;
    uint64_t *out = A.out + (size_t)blockIdx.z * A.out_bs + (size_t)A.out_limb[blockIdx.y] * A.N;
    const int jy = (A.N - j) & (A.N - 1);  // partner; j = 0 and j = N/2 pair with themselves
    uint64_t a = in[j], b = in[jy];
    if (A.reduce_input) { a = bred_add_lazy(a, q, m.brc0); b = bred_add_lazy(b, q, m.brc0); }
    if (!A.inverse) {
        const uint64_t F = m.pad0;
        if (j == 0) { out[0] = a; return; }
        const uint64_t oa = a + twoq - mred_lazy(b, F, q, qinv), ob = b + twoq - mred_lazy(a, F, q, qinv);
        out[j] = oa;
        if (jy != j) out[jy] = ob;
    } else {
        const uint64_t F = m.pad1;
        if (j == 0) { out[0] = cred(a << 1, q); return; }
        const uint64_t oa = cred(a + q - mred(b, F, q, qinv), q), ob = cred(b + q - mred(a, F, q, qinv), q);
        out[j] = oa;
        if (jy != j) out[jy] = ob;
    }
}
hipError_t launch_ci_fold(const RingDev &r, const LimbTab &tab, View in, View out, int batch, bool inverse, bool reduce_input,
                          hipStream_t s) {
    if (!no_tab({in, out})) return hipErrorInvalidValue;  // no entry tables here (View::tab)
    if (tab.n <= 0 || batch <= 0) return hipSuccess;
    CiFoldArgs A{};
    A.in = in.p; A.out = out.p; A.in_bs = in.bstride; A.out_bs = out.bstride; A.mc = r.mc; A.N = r.N;
    A.inverse = inverse; A.reduce_input = reduce_input;
    for (int i = 0; i < tab.n; i++) { A.in_limb[i] = tab.in_limb[i]; A.out_limb[i] = tab.out_limb[i]; A.mod[i] = tab.mod[i]; }
    dim3 grid((unsigned)((r.N / 2 + 1 + 255) / 256), tab.n, batch), block(256);
    ProfScope ps(K_CI_FOLD, s, 2.0 * tab.n * batch * (double)r.N * 8.0);
    hipLaunchKernelGGL(ci_fold_kernel, grid, block, 0, s, A);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// INTTConjugateInvariantLazy with the reference's exact lazy words (ring/ntt.go:1104-1152 + the NInv pass :728-737).
// DivRoundByLastModulusNTT / DivFloorByLastModulusNTT feed that LAZY output of the top limb into the other moduli
// (ring/scaling.go:15,110), so its representative -- not only its residue class -- is observable there: a word r + q_L instead
// of r moves the quotient by one.  One launch per Gentleman-Sande stage, every butterfly exactly the reference's invbutterfly
// (X = U + V, minus 2q when >= 2q; Y = MRedLazy(U + 4q - V, F)), then the fold with roots[1] and MRedLazy by NInv.  Off the hot
// path: only the rescale of a conjugate-invariant ring takes it, for one limb.
// ------------------------------------------------------------------------------------
struct CiRefArgs {
    const uint64_t *in;
    uint64_t *out;
    size_t in_bs, out_bs;
    ModConst mc;
    const uint64_t *tw;  // remapped backward table of the modulus: the stage with h blocks uses tw[h + i]
    int N, t;
};
__global__ void __launch_bounds__(256) ci_ref_inv_stage_kernel(CiRefArgs A) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.N / 2) return;
    const int i = b / A.t, j = b - i * A.t, jx = i * 2 * A.t + j, jy = jx + A.t, h = A.N / (2 * A.t);
    const uint64_t q = A.mc.q, twoq = q << 1, fourq = q << 2;
    const uint64_t *src = A.in + (size_t)blockIdx.z * A.in_bs;
    uint64_t *dst = A.out + (size_t)blockIdx.z * A.out_bs;
    const uint64_t U = src[jx], V = src[jy], F = A.tw[h + i];
    uint64_t X = U + V;
    if (X >= twoq) X -= twoq;
    dst[jx] = X;
    dst[jy] = mred_lazy(U + fourq - V, F, q, A.mc.qinv);
}
__global__ void __launch_bounds__(256) ci_ref_inv_fold_kernel(CiRefArgs A) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. N/2
    if (j > A.N / 2) return;
    const uint64_t q = A.mc.q, qinv = A.mc.qinv, twoq = q << 1, F = A.mc.pad1, ninv = A.mc.ninv;
    uint64_t *p = A.out + (size_t)blockIdx.z * A.out_bs;
    if (j == 0) { p[0] = mred_lazy(cred(p[0] << 1, q), ninv, q, qinv); return; }
    const int jy = A.N - j;
    const uint64_t a = p[j], b = p[jy];
    p[j] = mred_lazy(a + twoq - mred_lazy(b, F, q, qinv), ninv, q, qinv);
    if (jy != j) p[jy] = mred_lazy(b + twoq - mred_lazy(a, F, q, qinv), ninv, q, qinv);
}
hipError_t launch_ci_intt_lazy_ref(const RingDev &r, const ModConst &mc_host, int mod, View in, View out, int batch, hipStream_t s) {
    if (!no_tab({in, out})) return hipErrorInvalidValue;  // no entry tables here (View::tab)
    if (batch <= 0) return hipSuccess;
    CiRefArgs A{};
    A.in = in.p; A.in_bs = in.bstride; A.out = out.p; A.out_bs = out.bstride; A.mc = mc_host;
    A.tw = r.tw_inv + (size_t)mod * r.N; A.N = r.N;
    ProfScope ps(K_CI_FOLD, s, 2.0 * (r.logN + 1) * batch * (double)r.N * 8.0);  // one pass per stage (off the hot path)
    for (int t = 1; t < r.N; t <<= 1) {
        A.t = t;
        hipLaunchKernelGGL(ci_ref_inv_stage_kernel, dim3((unsigned)((r.N / 2 + 255) / 256), 1, batch), dim3(256), 0, s, A);
        A.in = out.p; A.in_bs = out.bstride;  // the first stage reads p1, the others run in place on p2
    }
    hipLaunchKernelGGL(ci_ref_inv_fold_kernel, dim3((unsigned)((r.N / 2 + 1 + 255) / 256), 1, batch), dim3(256), 0, s, A);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// coefficient-wise kernels (ring/vec_ops.go): one launch for all limbs x batch.
// Each thread handles two adjacent coefficients (16-byte accesses).
// ------------------------------------------------------------------------------------
struct EwArgs {
    const uint64_t *x, *y, *w;
    uint64_t *z;
    size_t x_bs, y_bs, z_bs, w_bs;
    const size_t *x_tab, *y_tab, *z_tab, *w_tab;  // entry tables (View::tab)
    const ModConst *mc;
    int N;
    int n;
    uint8_t x_limb[kMaxLimbs], y_limb[kMaxLimbs], z_limb[kMaxLimbs], mod[kMaxLimbs];
    uint64_t s[kMaxLimbs], s2[kMaxLimbs];
    int dbl;  // double-scalar form: s for coefficients < N/2, s2 for the others
};

template <int OP>
__device__ __forceinline__ uint64_t ew_apply(uint64_t x, uint64_t y, uint64_t z, const ModConst &m, uint64_t s, uint64_t s2) {
    const uint64_t q = m.q, qinv = m.qinv;
    if constexpr (OP == EW_ADD) return cred(x + y, q);
    else if constexpr (OP == EW_ADD_LAZY) return x + y;
    else if constexpr (OP == EW_SUB) return cred((x + q) - y, q);
    else if constexpr (OP == EW_SUB_LAZY) return x + q - y;
    else if constexpr (OP == EW_MUL_BARRETT) return bred(x, y, q, m.brc0, m.brc1);
    else if constexpr (OP == EW_MUL_BARRETT_LAZY) return bred_lazy(x, y, q, m.brc0, m.brc1);
    else if constexpr (OP == EW_MUL_BARRETT_THEN_ADD) return cred(z + bred(x, y, q, m.brc0, m.brc1), q);
    else if constexpr (OP == EW_MUL_BARRETT_THEN_ADD_LAZY) return z + bred(x, y, q, m.brc0, m.brc1);
    else if constexpr (OP == EW_MUL_MONT) return mred(x, y, q, qinv);
    else if constexpr (OP == EW_MUL_MONT_LAZY) return mred_lazy(x, y, q, qinv);
    else if constexpr (OP == EW_MUL_MONT_LAZY_THEN_NEG) return (q << 1) - mred_lazy(x, y, q, qinv);
    else if constexpr (OP == EW_MUL_MONT_THEN_ADD) return cred(z + mred(x, y, q, qinv), q);
    else if constexpr (OP == EW_MUL_MONT_THEN_ADD_LAZY) return z + mred(x, y, q, qinv);
    else if constexpr (OP == EW_MUL_MONT_LAZY_THEN_ADD_LAZY) return z + mred_lazy(x, y, q, qinv);
    else if constexpr (OP == EW_MUL_MONT_THEN_SUB) return cred(z + (q - mred(x, y, q, qinv)), q);
    else if constexpr (OP == EW_MUL_MONT_THEN_SUB_LAZY) return z + (q - mred(x, y, q, qinv));
    else if constexpr (OP == EW_MUL_MONT_LAZY_THEN_SUB_LAZY) return z + ((q << 1) - mred_lazy(x, y, q, qinv));
    else if constexpr (OP == EW_NEG) return q - x;
    else if constexpr (OP == EW_REDUCE) return bred_add(x, q, m.brc0);
    else if constexpr (OP == EW_REDUCE_LAZY) return bred_add_lazy(x, q, m.brc0);
    else if constexpr (OP == EW_MFORM) return mform(x, q, m.brc0, m.brc1);
    else if constexpr (OP == EW_MFORM_LAZY) return mform_lazy(x, q, m.brc0, m.brc1);
    else if constexpr (OP == EW_IMFORM) return imform(x, q, qinv);
    else if constexpr (OP == EW_COPY) return x;
    else if constexpr (OP == EW_ZERO) return 0;
    else if constexpr (OP == EW_ADD_SCALAR) return cred(x + s, q);
    else if constexpr (OP == EW_SUB_SCALAR) return cred(x + q - s, q);
    else if constexpr (OP == EW_MUL_SCALAR_MONT) return mred(x, s, q, qinv);
    else if constexpr (OP == EW_MUL_SCALAR_MONT_THEN_ADD) return cred(z + mred(x, s, q, qinv), q);
    else if constexpr (OP == EW_ADD_SCALAR_LAZY) return x + s;
    else if constexpr (OP == EW_SUB_THEN_MUL_SCALAR_MONT_2Q) return mred((q << 1) - y + x, s, q, qinv);
    else if constexpr (OP == EW_DIVROUND_COEFF) return mred(x + (s2 + (q << 1) - y), s, q, qinv);
    else if constexpr (OP == EW_SUBMUL2Q_THEN_ADD) return cred(z + mred((q << 1) - y + x, s, q, qinv), q);
    else return 0;
}
template <int OP>
constexpr bool ew_reads_y() {
    return (OP >= 0 && OP < 100) || OP == EW_SUB_THEN_MUL_SCALAR_MONT_2Q || OP == EW_DIVROUND_COEFF || OP == EW_SUBMUL2Q_THEN_ADD;
}
template <int OP>
constexpr bool ew_reads_z() {
    return OP == EW_MUL_BARRETT_THEN_ADD || OP == EW_MUL_BARRETT_THEN_ADD_LAZY || OP == EW_MUL_MONT_THEN_ADD ||
           OP == EW_MUL_MONT_THEN_ADD_LAZY || OP == EW_MUL_MONT_LAZY_THEN_ADD_LAZY || OP == EW_MUL_MONT_THEN_SUB ||
           OP == EW_MUL_MONT_THEN_SUB_LAZY || OP == EW_MUL_MONT_LAZY_THEN_SUB_LAZY || OP == EW_MUL_SCALAR_MONT_THEN_ADD ||
           OP == EW_SUBMUL2Q_THEN_ADD;
}

template <int OP>
__global__ void __launch_bounds__(256) ew_kernel(EwArgs A) {
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (j >= A.N) return;
    const int yy = blockIdx.y;
    ModConst m{};
    if constexpr (OP != EW_COPY && OP != EW_ZERO) m = A.mc[A.mod[yy]];  // (a copy needs no modulus: he_poly_copy launches it without a ring)
    const uint64_t s2 = A.s2[yy], s = (A.dbl && j >= (A.N >> 1)) ? s2 : A.s[yy];
    const size_t bz = blockIdx.z;
    ulonglong2 xv = make_ulonglong2(0, 0);
    if constexpr (OP != EW_ZERO) xv = ldnt2(A.x + voff(A.x_tab, A.x_bs, bz) + (size_t)A.x_limb[yy] * A.N + j);
    ulonglong2 yv = make_ulonglong2(0, 0), zv = make_ulonglong2(0, 0);
    if constexpr (ew_reads_y<OP>())
        yv = ldnt2(A.y + voff(A.y_tab, A.y_bs, bz) + (size_t)A.y_limb[yy] * A.N + j);
    uint64_t *zp = A.z + voff(A.z_tab, A.z_bs, bz) + (size_t)A.z_limb[yy] * A.N + j;
    if constexpr (ew_reads_z<OP>())
        zv = ldnt2(A.w + voff(A.w_tab, A.w_bs, bz) + (size_t)A.z_limb[yy] * A.N + j);
    ulonglong2 o;
    o.x = ew_apply<OP>(xv.x, yv.x, zv.x, m, s, s2);
    o.y = ew_apply<OP>(xv.y, yv.y, zv.y, m, s, s2);
    *reinterpret_cast<ulonglong2 *>(zp) = o;
}

static hipError_t launch_ew_impl(const RingDev &r, const LimbTab &tab, int op, View x, View y, View w, View z, int batch,
                                 const ScalarTab *sc, const uint8_t *x_limb_override, hipStream_t s, int dbl = 0);
hipError_t launch_ew_double(const RingDev &r, const LimbTab &tab, int op, View x, View z, int batch, const ScalarTab *sc,
                            hipStream_t s) {
    if (r.N < 4) return hipErrorInvalidValue;  // two coefficients per thread must not straddle N/2
    return launch_ew_impl(r, tab, op, x, x, z, z, batch, sc, nullptr, s, 1);
}
hipError_t launch_ew(const RingDev &r, const LimbTab &tab, int op, View x, View y, View z, int batch,
                     const ScalarTab *sc, const uint8_t *x_limb_override, hipStream_t s) {
    return launch_ew_impl(r, tab, op, x, y, z, z, batch, sc, x_limb_override, s);
}
hipError_t launch_ew_w(const RingDev &r, const LimbTab &tab, int op, View x, View y, View w, View z, int batch,
                       const ScalarTab *sc, hipStream_t s) {
    return launch_ew_impl(r, tab, op, x, y, w, z, batch, sc, nullptr, s);
}
static hipError_t launch_ew_impl(const RingDev &r, const LimbTab &tab, int op, View x, View y, View w, View z, int batch,
                                 const ScalarTab *sc, const uint8_t *x_limb_override, hipStream_t s, int dbl) {
    if (tab.n <= 0 || batch <= 0) return hipSuccess;
    EwArgs A{};
    A.dbl = dbl;
    A.x = x.p; A.y = y.p; A.z = z.p; A.w = w.p;
    A.x_bs = x.bstride; A.y_bs = y.bstride; A.z_bs = z.bstride; A.w_bs = w.bstride;
    A.x_tab = x.tab; A.y_tab = y.tab; A.z_tab = z.tab; A.w_tab = w.tab;
    A.mc = r.mc; A.N = r.N; A.n = tab.n;
    for (int i = 0; i < tab.n; i++) {
        A.x_limb[i] = x_limb_override ? x_limb_override[i] : tab.in_limb[i];
        A.y_limb[i] = tab.in_limb[i];
        A.z_limb[i] = tab.out_limb[i];
        A.mod[i] = tab.mod[i];
        A.s[i] = sc ? sc->s[i] : 0;
        A.s2[i] = sc ? sc->s2[i] : 0;
    }
    dim3 grid((unsigned)((r.N / 2 + 255) / 256), tab.n, batch), block(256);
    {   // HERING_EW_STATS=1: launches and limb-units per element-wise op, printed at exit (diagnosis of driver-level traces)
        static const bool stats = env_flag("HERING_EW_STATS");
        if (stats) {
            static std::mutex mu;
            static std::unordered_map<int, std::pair<long, double>> cnt;
            static const int reg = atexit([] { for (auto &kv : cnt) fprintf(stderr, "ew op %d: %ld launches, %.0f limb-entries\n", kv.first, kv.second.first, kv.second.second); });
            (void)reg;
            std::lock_guard<std::mutex> lk(mu);
            auto &c = cnt[op];
            c.first++; c.second += (double)tab.n * batch;
        }
    }
    // x in, z out, plus y and the addend where the formula has them
    const bool ew_y = (op >= 0 && op < 100) || op == EW_SUB_THEN_MUL_SCALAR_MONT_2Q || op == EW_DIVROUND_COEFF || op == EW_SUBMUL2Q_THEN_ADD;
    const bool ew_z = op == EW_MUL_BARRETT_THEN_ADD || op == EW_MUL_BARRETT_THEN_ADD_LAZY || op == EW_MUL_MONT_THEN_ADD ||
                      op == EW_MUL_MONT_THEN_ADD_LAZY || op == EW_MUL_MONT_LAZY_THEN_ADD_LAZY || op == EW_MUL_MONT_THEN_SUB ||
                      op == EW_MUL_MONT_THEN_SUB_LAZY || op == EW_MUL_MONT_LAZY_THEN_SUB_LAZY || op == EW_MUL_SCALAR_MONT_THEN_ADD ||
                      op == EW_SUBMUL2Q_THEN_ADD;
    ProfScope ps(K_EW, s, ((op == EW_ZERO ? 1.0 : 2.0) + ew_y + ew_z) * tab.n * batch * (double)r.N * 8.0);
#define HE_EW_CASE(O) \
    case O: hipLaunchKernelGGL((ew_kernel<O>), grid, block, 0, s, A); break;
    switch (op) {
        HE_EW_CASE(EW_ADD) HE_EW_CASE(EW_ADD_LAZY) HE_EW_CASE(EW_SUB) HE_EW_CASE(EW_SUB_LAZY)
        HE_EW_CASE(EW_MUL_BARRETT) HE_EW_CASE(EW_MUL_BARRETT_LAZY) HE_EW_CASE(EW_MUL_BARRETT_THEN_ADD)
        HE_EW_CASE(EW_MUL_BARRETT_THEN_ADD_LAZY) HE_EW_CASE(EW_MUL_MONT) HE_EW_CASE(EW_MUL_MONT_LAZY)
        HE_EW_CASE(EW_MUL_MONT_LAZY_THEN_NEG) HE_EW_CASE(EW_MUL_MONT_THEN_ADD) HE_EW_CASE(EW_MUL_MONT_THEN_ADD_LAZY)
        HE_EW_CASE(EW_MUL_MONT_LAZY_THEN_ADD_LAZY) HE_EW_CASE(EW_MUL_MONT_THEN_SUB)
        HE_EW_CASE(EW_MUL_MONT_THEN_SUB_LAZY) HE_EW_CASE(EW_MUL_MONT_LAZY_THEN_SUB_LAZY)
        HE_EW_CASE(EW_NEG) HE_EW_CASE(EW_REDUCE) HE_EW_CASE(EW_REDUCE_LAZY) HE_EW_CASE(EW_MFORM)
        HE_EW_CASE(EW_MFORM_LAZY) HE_EW_CASE(EW_IMFORM) HE_EW_CASE(EW_COPY) HE_EW_CASE(EW_ZERO)
        HE_EW_CASE(EW_ADD_SCALAR) HE_EW_CASE(EW_SUB_SCALAR) HE_EW_CASE(EW_MUL_SCALAR_MONT)
        HE_EW_CASE(EW_MUL_SCALAR_MONT_THEN_ADD) HE_EW_CASE(EW_ADD_SCALAR_LAZY)
        HE_EW_CASE(EW_SUB_THEN_MUL_SCALAR_MONT_2Q) HE_EW_CASE(EW_DIVROUND_COEFF) HE_EW_CASE(EW_SUBMUL2Q_THEN_ADD)
        default: return hipErrorInvalidValue;
    }
#undef HE_EW_CASE
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// automorphism
// ------------------------------------------------------------------------------------
struct GatherArgs {
    const uint64_t *in;
    uint64_t *out;
    size_t in_bs, out_bs;
    const size_t *in_tab, *out_tab;  // entry tables (View::tab)
    const uint32_t *index;
    int N;
    uint8_t in_limb[kMaxLimbs], out_limb[kMaxLimbs];
};
// out[l][j] (+)= in[l][index[j]]   (ring/automorphism.go:50-109); index shared by all limbs
template <bool ADD>
__global__ void __launch_bounds__(256) gather_kernel(GatherArgs A) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= A.N) return;
    const uint32_t src = A.index[j];
    const uint64_t *in = A.in + voff(A.in_tab, A.in_bs, blockIdx.z) + (size_t)A.in_limb[blockIdx.y] * A.N;
    uint64_t *out = A.out + voff(A.out_tab, A.out_bs, blockIdx.z) + (size_t)A.out_limb[blockIdx.y] * A.N;
    const uint64_t v = ldnt(&in[src]);
    if (ADD) out[j] += v; else out[j] = v;
}
hipError_t launch_gather(const RingDev &r, const LimbTab &tab, View in, const uint32_t *index, View out, int batch,
                         bool then_add, hipStream_t s) {
    if (tab.n <= 0 || batch <= 0) return hipSuccess;
    GatherArgs A{};
    A.in = in.p; A.out = out.p; A.in_bs = in.bstride; A.out_bs = out.bstride; A.in_tab = in.tab; A.out_tab = out.tab; A.index = index; A.N = r.N;
    for (int i = 0; i < tab.n; i++) { A.in_limb[i] = tab.in_limb[i]; A.out_limb[i] = tab.out_limb[i]; }
    dim3 grid((unsigned)((r.N + 255) / 256), tab.n, batch), block(256);
    ProfScope ps(K_GATHER, s, (then_add ? 3.0 : 2.0) * tab.n * batch * (double)r.N * 8.0);
    if (then_add) hipLaunchKernelGGL((gather_kernel<true>), grid, block, 0, s, A);
    else hipLaunchKernelGGL((gather_kernel<false>), grid, block, 0, s, A);
    return hipGetLastError();
}

// index[i] = bitrev(((gal*(2*bitrev(i)+1) & (NthRoot-1)) - 1) >> 1), bit reversals over lognth = log2(NthRoot) - 1 bits
// (ring/automorphism.go:12-34): NthRoot = 2N (lognth = logN) for the standard ring, 4N (lognth = logN + 1) for the
// conjugate-invariant one (ring/ring.go:254,261)
__global__ void build_index_kernel(int logN, int lognth, uint64_t gal, uint32_t *index) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t N = 1u << logN;
    if (i >= N) return;
    const uint64_t mask = (2ull << lognth) - 1;
    const uint64_t t1 = 2 * (__brevll((uint64_t)i) >> (64 - lognth)) + 1;
    const uint64_t t2 = (((gal * t1) & mask) - 1) >> 1;
    index[i] = (uint32_t)(__brevll(t2) >> (64 - lognth));
}
hipError_t launch_build_automorphism_index(int logN, int lognth, uint64_t gal, uint32_t *index, hipStream_t s) {
    const unsigned N = 1u << logN;
    ProfScope ps(K_INDEX, s);
    hipLaunchKernelGGL(build_index_kernel, dim3((N + 255) / 256), dim3(256), 0, s, logN, lognth, gal, index);
    return hipGetLastError();
}

struct AutoCoeffArgs {
    const uint64_t *in;
    uint64_t *out;
    size_t in_bs, out_bs;
    const size_t *in_tab, *out_tab;  // entry tables (View::tab)
    const ModConst *mc;
    int N, logN;
    uint64_t gal, ginv;  // ginv = gal^-1 mod 2N (conjugate-invariant form only)
    uint8_t in_limb[kMaxLimbs], out_limb[kMaxLimbs], mod[kMaxLimbs];
};
// out[i*gal mod N] = +-in[i]; a negated zero is stored as q, exactly as the reference's
// `(q - c) * tmp` does (ring/automorphism.go:170)
__global__ void __launch_bounds__(256) automorphism_coeff_kernel(AutoCoeffArgs A) {
    const uint64_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)A.N) return;
    const uint64_t raw = i * A.gal, idx = raw & (uint64_t)(A.N - 1), neg = (raw >> A.logN) & 1;
    const uint64_t q = A.mc[A.mod[blockIdx.y]].q;
    const uint64_t c = (A.in + voff(A.in_tab, A.in_bs, blockIdx.z) + (size_t)A.in_limb[blockIdx.y] * A.N)[i];
    (A.out + voff(A.out_tab, A.out_bs, blockIdx.z) + (size_t)A.out_limb[blockIdx.y] * A.N)[idx] = neg ? q - c : c;
}
// Conjugate-invariant ring Z[X + X^-1]/(X^2N + 1) (ring/automorphism.go:122-151): the reference walks i over [0, 2N) and keeps
// the images i * gal mod 2N that fall below N, reading coefficient i (or 2N - i, negated, for i >= N).  gal is odd, so every
// image below N has exactly one source: i = x * gal^-1 mod 2N -- the same map as a gather.
__global__ void __launch_bounds__(256) automorphism_coeff_ci_kernel(AutoCoeffArgs A) {
    const uint64_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (uint64_t)A.N) return;
    const uint64_t N = (uint64_t)A.N, mask2 = 2 * N - 1;
    const uint64_t i = (x * A.ginv) & mask2;
    uint64_t neg = ((i * A.gal) >> (A.logN + 1)) & 1;
    uint64_t idx = i;
    if (i >= N) { idx = 2 * N - i; neg ^= 1; }
    const uint64_t q = A.mc[A.mod[blockIdx.y]].q;
    const uint64_t c = (A.in + voff(A.in_tab, A.in_bs, blockIdx.z) + (size_t)A.in_limb[blockIdx.y] * A.N)[idx];
    (A.out + voff(A.out_tab, A.out_bs, blockIdx.z) + (size_t)A.out_limb[blockIdx.y] * A.N)[x] = neg ? q - c : c;
}
hipError_t launch_automorphism_coeff(const RingDev &r, const LimbTab &tab, View in, uint64_t gal, View out, int batch,
                                     hipStream_t s, bool conjugate_invariant) {
    if (tab.n <= 0 || batch <= 0) return hipSuccess;
    AutoCoeffArgs A{};
    A.in = in.p; A.out = out.p; A.in_bs = in.bstride; A.out_bs = out.bstride; A.in_tab = in.tab; A.out_tab = out.tab; A.mc = r.mc; A.N = r.N; A.logN = r.logN;
    A.gal = gal; A.ginv = 0;
    for (int i = 0; i < tab.n; i++) { A.in_limb[i] = tab.in_limb[i]; A.out_limb[i] = tab.out_limb[i]; A.mod[i] = tab.mod[i]; }
    dim3 grid((unsigned)((r.N + 255) / 256), tab.n, batch), block(256);
    ProfScope ps(K_AUTO_COEFF, s, 2.0 * tab.n * batch * (double)r.N * 8.0);
    if (conjugate_invariant) {
        const uint64_t m = 2ull * r.N;  // inverse of the odd gal modulo the power of two 2N (Newton iteration)
        uint64_t inv = gal;
        for (int i = 0; i < 6; i++) inv *= 2 - gal * inv;
        A.ginv = inv & (m - 1);
        A.gal = gal & (2 * m - 1);  // only i * gal mod 4N matters (the sign bit is bit log2(2N))
        hipLaunchKernelGGL(automorphism_coeff_ci_kernel, grid, block, 0, s, A);
    } else {
        hipLaunchKernelGGL(automorphism_coeff_kernel, grid, block, 0, s, A);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// basis extension: ModUpExact = reconstructRNS + multSum per coefficient
// (ring/basis_extension.go:282-308, :550-673).  One thread per coefficient; the y_i stay
// in registers, the destination limbs are split over blockIdx.y chunks.
// The float64 term v = trunc(sum_i fl(fl(y_i)/fl(q_i))) is accumulated sequentially in
// source-limb order with IEEE round-to-nearest division and addition (no contraction),
// which is what the Go code does.
// ------------------------------------------------------------------------------------
struct ModUpKArgs {
    const uint64_t *src;
    uint64_t *dstA, *dstB;
    size_t src_bs, dstA_bs, dstB_bs;
    const size_t *src_tab, *dstA_tab, *dstB_tab;  // entry tables (View::tab)
    const ModConst *mc;
    const uint64_t *a, *T, *vt;
    int N, nchunk;
    ModUpArgs m;
};

template <int NSRC>
__global__ void __launch_bounds__(64) modup_kernel(ModUpKArgs A) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= A.N) return;
    const size_t bz = blockIdx.z;
    const int nsrc = NSRC > 0 ? NSRC : A.m.nsrc;
    uint64_t yv[NSRC > 0 ? NSRC : 32];
    double vi = 0.0;
    const uint64_t *src = A.src + voff(A.src_tab, A.src_bs, bz) + x;
#pragma unroll
    for (int i = 0; i < (NSRC > 0 ? NSRC : 32); i++) {
        if (i < nsrc) {
            const ModConst mq = A.mc[A.m.src_mod[i]];
            uint64_t v = src[(size_t)A.m.src_limb[i] * A.N];
            const uint64_t h = A.m.src_half[i];
            if (h) v = cred(v + h, mq.q);
            const uint64_t yi = mred(v, A.a[i], mq.q, mq.qinv);
            yv[i] = yi;
            vi = __dadd_rn(vi, __ddiv_rn(__ull2double_rn(yi), __ull2double_rn(mq.q)));
        }
    }
    const uint64_t v = (uint64_t)vi;
    const int per = (A.m.ndst + A.nchunk - 1) / A.nchunk;
    const int j0 = blockIdx.y * per, j1 = min(A.m.ndst, j0 + per);
    for (int j = j0; j < j1; j++) {
        const ModConst mp = A.mc[A.m.dst_mod[j]];
        const int row = A.m.dst_row[j];
        const uint64_t *Tr = A.T + (size_t)row * nsrc;
        u128 acc = (u128)yv[0] * Tr[0];
#pragma unroll
        for (int i = 1; i < (NSRC > 0 ? NSRC : 32); i++)
            if (i < nsrc) acc += (u128)yv[i] * Tr[i];
        const uint64_t rlo = (uint64_t)acc, rhi = (uint64_t)(acc >> 64);
        uint64_t res = rhi - mulhi64(rlo * mp.qinv, mp.q) + mp.q + A.vt[(size_t)row * (nsrc + 1) + v];
        res = cred(res + mp.q - A.m.dst_half[j], mp.q);   // SubScalar, vec_ops.go:653
        uint64_t *dst = A.m.dst_view[j] ? (A.dstB + voff(A.dstB_tab, A.dstB_bs, bz)) : (A.dstA + voff(A.dstA_tab, A.dstA_bs, bz));
        dst[(size_t)A.m.dst_limb[j] * A.N + x] = res;
    }
}

hipError_t launch_modup(const RingDev &r, const ModUpDev &c, const ModUpArgs &a, View src, View dstA, View dstB,
                        int batch, hipStream_t s) {
    if (a.ndst <= 0 || batch <= 0) return hipSuccess;
    ModUpKArgs A{};
    A.src = src.p; A.dstA = dstA.p; A.dstB = dstB.p;
    A.src_bs = src.bstride; A.dstA_bs = dstA.bstride; A.dstB_bs = dstB.bstride;
    A.src_tab = src.tab; A.dstA_tab = dstA.tab; A.dstB_tab = dstB.tab;
    A.mc = r.mc; A.a = c.a; A.T = c.T; A.vt = c.vt; A.N = r.N; A.m = a;
    const int bx = (r.N + 63) / 64;
    int nchunk = 2048 / (bx * batch);
    if (nchunk < 1) nchunk = 1;
    if (nchunk > a.ndst) nchunk = a.ndst;
    if (nchunk > 4) nchunk = 4;
    A.nchunk = nchunk;
    dim3 grid(bx, nchunk, batch), block(64);
    ProfScope ps(K_MODUP, s, (double)(a.nsrc + a.ndst) * batch * (double)r.N * 8.0);
    switch (a.nsrc) {
        case 1: hipLaunchKernelGGL((modup_kernel<1>), grid, block, 0, s, A); break;
        case 2: hipLaunchKernelGGL((modup_kernel<2>), grid, block, 0, s, A); break;
        case 3: hipLaunchKernelGGL((modup_kernel<3>), grid, block, 0, s, A); break;
        case 4: hipLaunchKernelGGL((modup_kernel<4>), grid, block, 0, s, A); break;
        case 5: hipLaunchKernelGGL((modup_kernel<5>), grid, block, 0, s, A); break;
        case 6: hipLaunchKernelGGL((modup_kernel<6>), grid, block, 0, s, A); break;
        case 7: hipLaunchKernelGGL((modup_kernel<7>), grid, block, 0, s, A); break;
        case 8: hipLaunchKernelGGL((modup_kernel<8>), grid, block, 0, s, A); break;
        default: hipLaunchKernelGGL((modup_kernel<0>), grid, block, 0, s, A); break;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// fused basis extension (see kernels.h): one thread owns the 2^LOGA coefficients
// {c + r*N/2^LOGA} of every source / destination limb.
// ------------------------------------------------------------------------------------
struct ModUpFusedArgs {
    const ModUpDesc *desc;
    const uint64_t *src;
    uint64_t *dstA, *dstB;
    size_t src_bs, dstA_bs, dstB_bs;
    const size_t *src_tab, *dstA_tab, *dstB_tab;  // entry tables (View::tab)
    const ModConst *mc;
    const uint64_t *tw_fwd, *tw_inv;
    const double *twd_fwd, *twd_inv;
    const uint64_t *tws_fwd;
    int N;
    int nchunk;   // small grids: the destinations of a digit are shared out over nchunk workgroups (blockIdx.y = digit * nchunk + chunk),
                  // each redoing the source stage -- a launch of a few hundred workgroups is the latency of ONE wave walking all
                  // its destinations, and the chip is idle anyway
    int f64_raw;  // double-precision destinations are stored as the doubles they are (unreduced, bounded by modup_f64_raw_ok): the consumer is a
                  // double-precision row kernel told so (NttMacArgs::dec_f64 / NTT_INPUT_F64), six instructions per word saved here
};

// DSTF64 = false: destinations in 64-bit integer arithmetic (any modulus).
// DSTF64 = true : only destination moduli below 2^47, the mat-vec and the column stages in exact double-precision
//                 integer arithmetic (see ntt_rows_f64_kernel); same canonical results.
#ifndef HE_MODUP_ASM
#define HE_MODUP_ASM 1  // integer source stages, y_i and lean-destination butterflies through mred_lazy_col_asm (0: compiler forms)
#endif
#ifndef HE_MODUP_WAVES
#define HE_MODUP_WAVES 3  // waves per SIMD the register allocation aims at (the LDS footprint allows three)
#endif
// The all-integer variant keeps the residues in registers (2 R NSRC of them, beside NSRC matrix entries per destination): only
// the shapes that fit the three-wave register budget without spilling are instantiated (modup_int_light); every other shape
// takes the LDS-parking variant, whatever the classes of its destination moduli (it handles integer destinations too)
constexpr bool modup_int_light(int nsrc, int loga) { return 2 * (1 << loga) * nsrc + 4 * nsrc <= 48; }
template <int NSRC, int LOGA, bool DSTF64>
__global__ void __launch_bounds__(128, HE_MODUP_WAVES) modup_fused_kernel(ModUpFusedArgs A) {
    constexpr int R = 1 << LOGA;
    const int N2 = A.N >> LOGA;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    // the digit's descriptor goes to LDS first: read from global memory, its byte fields cost a vector load and a full wait at
    // every use (the compiler cannot move them across the destination stores), 24 of them per destination limb
    __shared__ ModUpDesc Ds;
    {
        const uint32_t *g = reinterpret_cast<const uint32_t *>(A.desc + (A.nchunk > 1 ? blockIdx.y / (unsigned)A.nchunk : blockIdx.y));
        uint32_t *l = reinterpret_cast<uint32_t *>(&Ds);
        for (int w = threadIdx.x; w < (int)(sizeof(ModUpDesc) / 4); w += blockDim.x) l[w] = g[w];
    }
    __syncthreads();
    if (c >= N2) return;
    const ModUpDesc &D = Ds;
    auto U = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };  // block-uniform -> SGPR
    auto U64 = [&](uint64_t v) -> uint64_t { return ((uint64_t)U((uint32_t)(v >> 32)) << 32) | U((uint32_t)v); };
    const bool single = U(D.single) != 0, reduce_out = U(D.reduce_out) != 0;
    const int ndst = (int)U(D.ndst);
    const uint64_t *Da = reinterpret_cast<const uint64_t *>(U64((uint64_t)D.a));
    const uint64_t *DT = reinterpret_cast<const uint64_t *>(U64((uint64_t)D.T));
    const uint64_t *Dvt = reinterpret_cast<const uint64_t *>(U64((uint64_t)D.vt));
    const double *DTd = reinterpret_cast<const double *>(U64((uint64_t)D.Td));
    const double *Dvtd = reinterpret_cast<const double *>(U64((uint64_t)D.vtd));
    const uint64_t *Dfc = reinterpret_cast<const uint64_t *>(U64((uint64_t)D.fc));
    const size_t dst_off = U64(D.dst_off);
    uint32_t splitmask = 0;  // bit i: source residue i is split at kYSplitBits
#pragma unroll
    for (int i = 0; i < NSRC; i++) splitmask |= (D.src_split[i] != 0 ? 1u : 0u) << i;
    splitmask = single ? 0u : U(splitmask);
    const size_t bz = blockIdx.z;
    const uint64_t *src = A.src + voff(A.src_tab, A.src_bs, bz) + c;
    const size_t dstA_off = voff(A.dstA_tab, A.dstA_bs, bz), dstB_off = voff(A.dstB_tab, A.dstB_bs, bz);

    // mixed variant: the integer residues live in LDS (read back only for the few large destination moduli)
    // Long digits (NSRC > 3 with three column stages: 8 residues per source and thread) keep the first KREG sources' residues in
    // registers and park only the rest: with all of them in LDS a 128-thread workgroup needs NSRC KiB x 8 (40 KiB at the
    // bootstrapping shape, NSRC = 5) and a CU holds three of them -- 1.5 waves per SIMD for an instruction-bound kernel.
#ifndef HE_MODUP_KREG_MAX
#define HE_MODUP_KREG_MAX 3
#endif
    constexpr int KREG = (DSTF64 && LOGA == 3 && NSRC > 3) ? (NSRC - 3 < HE_MODUP_KREG_MAX ? NSRC - 3 : HE_MODUP_KREG_MAX) : 0;
    __shared__ uint64_t ylds[DSTF64 ? (NSRC - KREG) * R : 1][128];
    uint64_t yreg[KREG ? R : 1][KREG ? KREG : 1];
    auto ypark = [&](int r, int i, uint64_t v) {  // (r, i are compile-time constants after unrolling)
        if (i < KREG) yreg[r][i] = v; else ylds[(i - KREG) * R + r][threadIdx.x] = v;
    };
    auto ytake = [&](int r, int i) -> uint64_t { return i < KREG ? yreg[r][i] : ylds[(i - KREG) * R + r][threadIdx.x]; };
    uint64_t y[DSTF64 ? 1 : R][DSTF64 ? 1 : NSRC];   // integer variant
    // Short digits (NSRC <= 3, up to 8 coefficients per thread) also keep every residue that fits a double (source modulus
    // below 2^51: no split) as a double in registers: the double-precision destinations -- most of them at the headline
    // shape -- then take their operand without an LDS read and a conversion per (term, destination).  48 registers; the kernel
    // stays within the three-wave budget.
#ifndef HE_MODUP_YD
#define HE_MODUP_YD 1
#endif
    constexpr bool YD = DSTF64 && HE_MODUP_YD && NSRC <= 3 && LOGA <= 3 && KREG == 0;
    double ydreg[YD ? R : 1][YD ? NSRC : 1];
    double vi[R];
    uint32_t negmask = 0;  // centred-copy path only: bit r = coefficient r was negated
#pragma unroll
    for (int r = 0; r < R; r++) vi[r] = 0.0;
#pragma unroll
    for (int i = 0; i < NSRC; i++) {
        const int mi = (int)U(D.src_mod[i]);
        const ModConst mq = A.mc[mi];
        const uint64_t q = mq.q, qinv = mq.qinv, twoq = mq.q << 1;
        uint64_t x[R];
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = ldnt(&src[(size_t)U(D.src_limb[i]) * A.N + (size_t)r * N2]);
        const bool src_small = DSTF64 && (q >> kF64Bits) == 0 && A.twd_inv != nullptr;  // block-uniform
        double xd[R];  // src_small: the coefficients as doubles, not yet scaled by N^-1 (LOGA > 0) nor reduced (|xd| < 16q)
        const double ninv = (src_small && LOGA > 0) ? (double)imform(mq.ninv, q, qinv) : 1.0;
        if (src_small) {
            // source modulus below 2^47: the inverse column stages, N^-1 and y_i = x*c_i in exact double arithmetic
            const double qd = (double)q, qid = mq.rq;
#pragma unroll
            for (int r = 0; r < R; r++) xd[r] = u52_to_f64(x[r]);
            if constexpr (LOGA > 0) {
                const double *tw = A.twd_inv + (size_t)mi * A.N;
#pragma unroll
                for (int s = LOGA - 1; s >= 0; s--) {
                    const int d = 1 << (LOGA - 1 - s);
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (r & d) continue;
                        const double U = xd[r], V = xd[r + d];
                        xd[r] = U + V;
                        xd[r + d] = modmul_f64(U - V, tw[(1 << s) + (r >> (LOGA - s))], qd, qid);
                    }
                }
            }
            if (single) {  // the centred value needs the canonical coefficient itself
#pragma unroll
                for (int r = 0; r < R; r++) x[r] = canon_f64(LOGA > 0 ? modmul_f64(xd[r], ninv, qd, qid) : xd[r], qd, qid);
            }
        } else if constexpr (LOGA > 0) {  // finish the inverse NTT: the LOGA strided stages, N^-1 folded into the last
            const uint64_t *tw = A.tw_inv + (size_t)mi * A.N;
#pragma unroll
            for (int s = LOGA - 1; s >= 0; s--) {
                const int d = 1 << (LOGA - 1 - s);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (r & d) continue;
                    const uint64_t wv = tw[(1 << s) + (r >> (LOGA - s))];
#if HE_MODUP_ASM
                    if (s == 0) bfly_inv_scaled_asm(x[r], x[r + d], mred(wv, mq.ninv, q, qinv), mq.ninv, q, twoq, qinv);
                    else bfly_inv_asm(x[r], x[r + d], wv, q, twoq, qinv);
#else
                    if (s == 0) bfly_inv_scaled(x[r], x[r + d], mred(wv, mq.ninv, q, qinv), mq.ninv, q, twoq, qinv);
                    else bfly_inv(x[r], x[r + d], wv, q, twoq, qinv);
#endif
                }
            }
        }
        if (single) {  // one-limb digit: centred value (ring/basis_extension.go:402-436)
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint64_t cv = x[r];
                const bool ng = cv >= (q >> 1);
                negmask |= (uint32_t)ng << r;
                cv = ng ? q - cv : cv;
                if constexpr (DSTF64) ypark(r, i, cv);
                else y[r][i] = cv;
            }
        } else {
            const uint64_t h = U64(D.src_half[i]), ai = Da[i];
            const double rq = mq.rq;
            const double qd = (double)q, qid = mq.rq, apl = src_small ? (double)imform(ai, q, qinv) : 0.0;
            uint64_t yi[R];
            double yd[R];
            if (src_small) {
                // y = (x N^-1 + h) a = x (N^-1 a) + h a: one product per coefficient, the two block-uniform constants once
                const double c1 = LOGA > 0 ? canon_f64d(modmul_f64(ninv, apl, qd, qid), qd, qid) : apl;
                const double c2 = canon_f64d(modmul_f64((double)h, apl, qd, qid), qd, qid);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    yd[r] = canon_f64d(modmul_f64(xd[r], c1, qd, qid) + c2, qd, qid);
                    yi[r] = f64_to_u52(yd[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
#if HE_MODUP_ASM
                    yi[r] = cred(mred_lazy_col_asm(cred(x[r] + h, q), ai, q, qinv), q);
#else
                    yi[r] = mred(cred(x[r] + h, q), ai, q, qinv);
#endif
                    yd[r] = __ull2double_rn(yi[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                // fast estimate of fl(y/q); the exact IEEE division is redone below only when the sum lands within
                // 2^-40 of an integer (|estimate - exact sum| < 2^-42 for <= 32 terms)
                vi[r] = __dadd_rn(vi[r], yd[r] * rq);
                if constexpr (DSTF64) {
                    ypark(r, i, yi[r]);
                    if constexpr (YD) ydreg[r][i] = yd[r];  // (used only when the source is not split)
                } else {
                    y[r][i] = yi[r];
                }
            }
        }
    }
    if (!single) {  // exact v = trunc(sum_i fl(fl(y_i)/fl(q_i))) where the estimate is not conclusive (rare)
#pragma unroll
        for (int r = 0; r < R; r++) {
            const double fr = vi[r] - floor(vi[r]);
            if (fr < 0x1p-40 || fr > 1.0 - 0x1p-40) {
                double e = 0.0;
#pragma unroll
                for (int i = 0; i < NSRC; i++) {
                    uint64_t yi;
                    if constexpr (DSTF64) yi = ytake(r, i);
                    else yi = y[r][i];
                    e = __dadd_rn(e, __ddiv_rn(__ull2double_rn(yi), __ull2double_rn(A.mc[U(D.src_mod[i])].q)));
                }
                vi[r] = e;
            }
        }
    }
    uint32_t v[R];
    double vd[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        v[r] = (uint32_t)vi[r];  // 0 <= vi < 33: one v_cvt_u32_f64 (truncating)
        vd[r] = floor(vi[r]);
    }

    int jlo = 0, jhi = ndst;
    if (A.nchunk > 1) {
        const int per = (ndst + A.nchunk - 1) / A.nchunk, ch = (int)(blockIdx.y % (unsigned)A.nchunk);
        jlo = ch * per;
        jhi = min(ndst, jlo + per);
    }
    for (int j = jlo; j < jhi; j++) {
        if constexpr (KREG > 0) {
            // the register-resident residues are made opaque per destination: otherwise their conversions (to double, to 26- and
            // 30-bit halves) are hoisted out of this loop and kept live -- 80 registers instead of 16
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int i = 0; i < KREG; i++) asm volatile("" : "+v"(yreg[r][i]));
        }
        const int mi = (int)U(D.dst_mod[j]);
        const ModConst mp = A.mc[mi];
        const uint64_t p = mp.q, pinv = mp.qinv, twop = mp.q << 1;
        const bool small = (p >> kF64Bits) == 0;  // block-uniform
        uint64_t *dst = (U(D.dst_view[j]) ? (A.dstB + dstB_off) : (A.dstA + dstA_off)) + dst_off +
                        (size_t)U(D.dst_limb[j]) * A.N + c;
        // residue y_i of coefficient r as an integer (the mixed variant keeps them as doubles)
        auto Y = [&](int r, int i) -> uint64_t {
            if constexpr (DSTF64) return ytake(r, i);
            else return y[r][i];
        };
        bool done = false;
        if constexpr (DSTF64) if (small) {
            done = true;
            const double pd = (double)p, pid = mp.rq;
            double o[R];
            if (single) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint64_t t = bred_add(ytake(r, 0), p, mp.brc0);
                    o[r] = u52_to_f64(((negmask >> r) & 1) ? p - t : t);
                }
            } else {
                const int row = (int)U(D.dst_row[j]);
                const double *Tr = DTd + (size_t)row * NSRC * 2;
                const double vt1 = ldcd(Dvtd, (size_t)row * (NSRC + 1) + 1);  // vt[v] = v * vt[1] mod p: the v-correction is one exact fma (v <= NSRC, v * vt1 < 2^50)
                const double hd = (double)U64(D.dst_half[j]);
#pragma unroll
                for (int r = 0; r < R; r++) o[r] = __fma_rn(vd[r], vt1, -hd);
#if HE_MODUP_MAGIC
                // Residues that do not fit a double (source modulus of 2^51 and above: the special primes in ModDown, q0 in the
                // decomposition) are split y = yh 2^29 + yl and their 2 x (split sources) products with {T, T 2^29 mod p} are
                // summed EXACTLY before one reduction: H runs in the binade [2^84, 2^85) (every product is below 2^79, at most
                // sixteen of them), so each fma rounds the running sum to a multiple of 2^32; the part it dropped,
                // l = a w - (H' - H), is an integer below 2^31 recovered exactly by a second fma and summed in L.  Four
                // operations per product and one reduction per coefficient, against seven per product for modmul_f64.
                if (splitmask) {
                    constexpr double C = 0x1p84;
                    constexpr uint32_t ML = (1u << kYSplitBits) - 1u;
                    double Tl[NSRC], Th[NSRC];
#pragma unroll
                    for (int i = 0; i < NSRC; i++) { Tl[i] = ldcd(Tr, 2 * i); Th[i] = ldcd(Tr, 2 * i + 1); }
                    auto piece = [](double a, double w, double &H, double &L) {
                        const double Hn = __fma_rn(a, w, H);
                        L += __fma_rn(a, w, -(Hn - H));
                        H = Hn;
                    };
                    auto finish = [&](int r, double H, double L) {
                        const double Hs = H - C;
                        o[r] += __fma_rn(-rint(Hs * pid), pd, Hs) + L;
                    };
                    if (splitmask == (1u << NSRC) - 1u) {
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            double H = C, L = 0.0;
#pragma unroll
                            for (int i = 0; i < NSRC; i++) {
                                const uint64_t yy = ytake(r, i);
                                piece((double)((uint32_t)yy & ML), Tl[i], H, L);
                                piece((double)(uint32_t)(yy >> kYSplitBits), Th[i], H, L);
                            }
                            finish(r, H, L);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            double H = C, L = 0.0;
#pragma unroll
                            for (int i = 0; i < NSRC; i++) {
                                if (!((splitmask >> i) & 1)) continue;
                                const uint64_t yy = ytake(r, i);
                                piece((double)((uint32_t)yy & ML), Tl[i], H, L);
                                piece((double)(uint32_t)(yy >> kYSplitBits), Th[i], H, L);
                            }
                            finish(r, H, L);
                        }
                    }
                }
#endif
#pragma unroll
                for (int i = 0; i < NSRC; i++) {
                    const double Tl = ldcd(Tr, 2 * i);  // block-uniform: scalar loads
                    [[maybe_unused]] const double Th = ldcd(Tr, 2 * i + 1);
                    if ((splitmask >> i) & 1) {
#if !HE_MODUP_MAGIC
                        // y >= 2^51 possible: y = yh 2^26 + yl, two exact products
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            const uint64_t yy = ytake(r, i);
                            o[r] += modmul_f64(u52_to_f64(yy & ((1ull << kYSplitBits) - 1)), Tl, pd, pid);
                            o[r] += modmul_f64(u52_to_f64(yy >> kYSplitBits), Th, pd, pid);
                        }
#endif
                    } else if constexpr (YD) {
#pragma unroll
                        for (int r = 0; r < R; r++) o[r] += modmul_f64(ydreg[r][i], Tl, pd, pid);
                    } else {
#pragma unroll
                        for (int r = 0; r < R; r++) o[r] += modmul_f64(u52_to_f64(ytake(r, i)), Tl, pd, pid);
                    }
                }  // |o| < (2 + 5*NSRC) p + NSRC 2^32 (the split sources' low parts L are added unreduced; modup_f64_raw_ok counts them)
            }
            if constexpr (LOGA > 0) {
                const double *tw = A.twd_fwd + (size_t)mi * A.N;
#pragma unroll
                for (int s = 0; s < LOGA; s++) {
                    const int d = 1 << (LOGA - 1 - s);
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (r & d) continue;
                        const double t = modmul_f64(o[r + d], ldcd(tw, (size_t)((1 << s) + (r >> (LOGA - s)))), pd, pid);
                        const double U = o[r];
                        o[r] = U + t;
                        o[r + d] = U - t;
                    }
                }
            }
            if (A.f64_raw) {
#pragma unroll
                for (int r = 0; r < R; r++) stnt(&dst[(size_t)r * N2], (uint64_t)__double_as_longlong(o[r]));
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) stnt(&dst[(size_t)r * N2], f64_to_u52(reduce_f64(o[r], pd, pid) + pd));  // (0, 2p)
            }
        }
        if (!done && !single && U(D.dst_fast[j]) == 2) {
            // The same folding for the remaining destination moduli (up to 2^61: the 60/61-bit q0 and special primes of the CKKS
            // chains): acc = C0m + sum_i y_i Tm_i + v V1m in 128 bits, one Montgomery reduction -> [0, 2p); Shoup column
            // butterflies in the Harvey range ([0, 4p): U is brought below 2p first).
            done = true;
            const int row = (int)U(D.dst_row[j]);
            uint64_t Tm[NSRC + 1];
#pragma unroll
            for (int i = 0; i < NSRC; i++) Tm[i] = ldc(DT, (size_t)row * NSRC + i);
            Tm[NSRC] = ldc(Dfc, 2 * (size_t)row);
            const uint64_t C0 = ldc(Dfc, 2 * (size_t)row + 1);
            uint64_t o[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                u128 acc = (u128)C0 + (u128)(uint64_t)v[r] * Tm[NSRC];
#pragma unroll
                for (int i = 0; i < NSRC; i++) acc += (u128)Y(r, i) * Tm[i];
                o[r] = (uint64_t)(acc >> 64) - mulhi64((uint64_t)acc * pinv, p) + p;  // (0, 2p)
            }
            if constexpr (LOGA > 0) {
                [[maybe_unused]] const uint64_t *ts = A.tws_fwd + (size_t)mi * 32;
                [[maybe_unused]] const uint64_t *twm = A.tw_fwd + (size_t)mi * A.N;
#pragma unroll
                for (int s = 0; s < LOGA; s++) {
                    const int d = 1 << (LOGA - 1 - s);
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (r & d) continue;
                        const size_t ix = (size_t)((1 << s) + (r >> (LOGA - s)));
                        const uint64_t V = o[r + d];
#if HE_MODUP_ASM
                        const uint64_t rr = mred_lazy_col_asm(V, ldc(twm, ix), p, pinv);  // [0, 2p)
#else
                        const uint64_t w = ldc(ts, 2 * ix), ws = ldc(ts, 2 * ix + 1);
                        const uint64_t rr = V * w - mulhi64(V, ws) * p;  // [0, 2p)
#endif
                        uint64_t Uu = o[r];
                        Uu = Uu >= twop ? Uu - twop : Uu;
                        o[r] = Uu + rr;
                        o[r + d] = Uu + twop - rr;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) stnt(&dst[(size_t)r * N2], o[r]);  // [0, 4p)
        }
        if (!done && !single && U(D.dst_fast[j]) != 0) {
            // Lean integer path for destination moduli below 2^58 (the 55-bit limbs of the headline chain).  Only the residue
            // class of the result matters here (the row NTT that follows accepts any word below 10p), so:
            //  * the v correction and the centring constant join the sum as one more term / the initial value, in Montgomery
            //    form: acc = sum_i y_i Tm_i + v V1m + C0m, one reduction, no selects, no conditional subtractions;
            //  * operands are split at 30 bits (y = x1 2^30 + x0, Tm = t1 2^30 + t0) and the partial products are summed by
            //    column in three 64-bit registers without carries (the host checks (nsrc+1)(p + max q + 2^31) < 2^64), 4
            //    v_mad_u64_u32 per term instead of a 128-bit multiply-add;
            //  * the matrix row, the constants and the twiddles are block-uniform and come through the scalar cache;
            //  * the column butterflies use Shoup products (w, floor(w 2^64 / p)): r = V w - mulhi(V, w') p in [0, 2p) for any
            //    64-bit V, X = U + r, Y = U + 2p - r -- each stage adds at most 2p to the bound.
            //  * dst_fast = 3 (HE_MODUP_R60; the host checks the column bounds): the three columns are reduced where they are, at
            //    radix 2^30 -- m0 = L (-p^-1) mod 2^30 clears the low column, its carry joins the middle one, m1 clears that, and
            //    the high column is the result (sum + (m0 + m1 2^30) p) / 2^60 in (0, 2p), the constants being in 2^60-Montgomery
            //    form: 4 v_mad_u64_u32 + 2 v_mul_lo_u32 + shifts, instead of assembling a 128-bit sum (two 64-bit shifts, two
            //    128-bit additions) and reducing it with a 64 x 64 low and a 64 x 64 high product.
            done = true;
            const int row = (int)U(D.dst_row[j]);
            constexpr uint32_t M30 = (1u << 30) - 1;
            const bool r60 = HE_MODUP_R60 && DSTF64 && U(D.dst_fast[j]) == 3;  // block-uniform (the all-integer variant has no register room for both forms: it keeps the 128-bit sum)
            const uint64_t *Dt60 = reinterpret_cast<const uint64_t *>(U64((uint64_t)D.t60));
            uint32_t t0[NSRC + 1], t1[NSRC + 1];
#pragma unroll
            for (int i = 0; i < NSRC; i++) {
                const uint64_t Tm = r60 ? ldc(Dt60, (size_t)row * (NSRC + 2) + i) : ldc(DT, (size_t)row * NSRC + i);
                t0[i] = (uint32_t)Tm & M30; t1[i] = (uint32_t)(Tm >> 30);
            }
            const uint64_t V1 = r60 ? ldc(Dt60, (size_t)row * (NSRC + 2) + NSRC) : ldc(Dfc, 2 * (size_t)row);
            const uint64_t C0 = r60 ? ldc(Dt60, (size_t)row * (NSRC + 2) + NSRC + 1) : ldc(Dfc, 2 * (size_t)row + 1);
            const uint32_t p0 = (uint32_t)p & M30, p1 = (uint32_t)(p >> 30), nq30 = (uint32_t)(0 - pinv) & M30;
            t0[NSRC] = (uint32_t)V1 & M30; t1[NSRC] = (uint32_t)(V1 >> 30);
            const uint64_t L0 = C0 & M30, M0 = C0 >> 30;
            uint64_t o[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint64_t Lc = L0, Mc = M0, Hc = 0;
#pragma unroll
                for (int i = 0; i < NSRC; i++) {
                    const uint64_t yy = Y(r, i);
                    const uint32_t x0 = (uint32_t)yy & M30, x1 = (uint32_t)(yy >> 30);
                    Lc = mad32(x0, t0[i], Lc);
                    Mc = mad32(x0, t1[i], Mc);
                    Mc = mad32(x1, t0[i], Mc);
                    Hc = mad32(x1, t1[i], Hc);
                }
                Lc = mad32(v[r], t0[NSRC], Lc);
                Mc = mad32(v[r], t1[NSRC], Mc);
                if (r60) {
                    const uint32_t m0 = ((uint32_t)Lc * nq30) & M30;
                    Lc = mad32(m0, p0, Lc);
                    Mc = mad32(m0, p1, Mc) + (Lc >> 30);
                    const uint32_t m1 = ((uint32_t)Mc * nq30) & M30;
                    Mc = mad32(m1, p0, Mc);
                    o[r] = mad32(m1, p1, Hc) + (Mc >> 30);  // (0, 2p)
                } else {
                    const u128 acc = (u128)Lc + ((u128)Mc << 30) + ((u128)Hc << 60);
                    o[r] = (uint64_t)(acc >> 64) - mulhi64((uint64_t)acc * pinv, p) + p;  // (0, 2p)
                }
            }
            if constexpr (LOGA > 0) {
                [[maybe_unused]] const uint64_t *ts = A.tws_fwd + (size_t)mi * 32;
                [[maybe_unused]] const uint64_t *twm = A.tw_fwd + (size_t)mi * A.N;
#pragma unroll
                for (int s = 0; s < LOGA; s++) {
                    const int d = 1 << (LOGA - 1 - s);
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (r & d) continue;
                        const size_t ix = (size_t)((1 << s) + (r >> (LOGA - s)));
                        const uint64_t V = o[r + d];
#if HE_MODUP_ASM
                        const uint64_t rr = mred_lazy_col_asm(V, ldc(twm, ix), p, pinv);
#else
                        const uint64_t w = ldc(ts, 2 * ix), ws = ldc(ts, 2 * ix + 1);
                        const uint64_t rr = V * w - mulhi64(V, ws) * p;
#endif
                        const uint64_t Uu = o[r];
                        o[r] = Uu + rr;
                        o[r + d] = Uu + twop - rr;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) stnt(&dst[(size_t)r * N2], o[r]);
        }
        if (!done) {
            uint64_t o[R];
            if (single) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint64_t t = bred_add(Y(r, 0), p, mp.brc0);
                    o[r] = ((negmask >> r) & 1) ? p - t : t;
                }
            } else {
                const int row = (int)U(D.dst_row[j]);
                const uint64_t *Tr = DT + (size_t)row * NSRC;
                const uint64_t *vtr = Dvt + (size_t)row * (NSRC + 1);
                const uint64_t hd = U64(D.dst_half[j]);
                uint64_t Tv[NSRC], vts[NSRC + 1];
#pragma unroll
                for (int i = 0; i < NSRC; i++) Tv[i] = Tr[i];
#pragma unroll
                for (int i = 0; i <= NSRC; i++) vts[i] = vtr[i];  // block-uniform: picked by selects below, not a dependent load per r
#pragma unroll
                for (int r = 0; r < R; r++) {
                    u128 acc = (u128)Y(r, 0) * Tv[0];
#pragma unroll
                    for (int i = 1; i < NSRC; i++) acc += (u128)Y(r, i) * Tv[i];
                    uint64_t vsel = vts[0];
#pragma unroll
                    for (int i = 1; i <= NSRC; i++) vsel = v[r] == (uint32_t)i ? vts[i] : vsel;
                    uint64_t res = (uint64_t)(acc >> 64) - mulhi64((uint64_t)acc * pinv, p) + p + vsel;
                    res = cred(res + p - hd, p);
                    if (reduce_out) res = bred_add_lazy(res, p, mp.brc0);
                    o[r] = res;
                }
            }
            if constexpr (LOGA > 0) {  // start the forward NTT: the LOGA strided stages
                const uint64_t *tw = A.tw_fwd + (size_t)mi * A.N;
                const bool nc = (p >> kNoCorrBits) == 0;
#pragma unroll
                for (int s = 0; s < LOGA; s++) {
                    const int d = 1 << (LOGA - 1 - s);
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (r & d) continue;
                        if (nc) bfly_fwd_nc(o[r], o[r + d], tw[(1 << s) + (r >> (LOGA - s))], p, pinv);
                        else bfly_fwd(o[r], o[r + d], tw[(1 << s) + (r >> (LOGA - s))], p, twop, pinv);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) stnt(&dst[(size_t)r * N2], o[r]);
        }
    }
}

// source limbs per digit: up to 8 (the descriptor's arrays) for up to three fused column stages, up to 5 with four (logN = 17:
// sixteen coefficients per thread; beyond that the residues no longer fit the registers / LDS and the unfused path is used)
bool modup_fused_supported(int logN, int nsrc) {
    const int a = logN - ntt_row_bits(logN);
    return nsrc >= 1 && a >= 0 && ((a <= 3 && nsrc <= 8) || (a == 4 && nsrc <= 5));
}

template <bool F64>
static void launch_modup_fused_variant(int a, int nsrc, dim3 grid, dim3 block, const ModUpFusedArgs &A, hipStream_t s) {
#define HE_MF(NS, LA) do { if constexpr (F64 || modup_int_light(NS, LA)) hipLaunchKernelGGL((modup_fused_kernel<NS, LA, F64>), grid, block, 0, s, A); } while (0)
#define HE_MF_A(NS)                                  \
    switch (a) {                                     \
        case 0: HE_MF(NS, 0); break;                 \
        case 1: HE_MF(NS, 1); break;                 \
        case 2: HE_MF(NS, 2); break;                 \
        case 3: HE_MF(NS, 3); break;                 \
        default: HE_MF(NS, 4); break;                \
    }
#define HE_MF_B(NS)                                  \
    switch (a) {                                     \
        case 0: HE_MF(NS, 0); break;                 \
        case 1: HE_MF(NS, 1); break;                 \
        case 2: HE_MF(NS, 2); break;                 \
        default: HE_MF(NS, 3); break;                \
    }
    switch (nsrc) {
        case 1: HE_MF_A(1); break;
        case 2: HE_MF_A(2); break;
        case 3: HE_MF_A(3); break;
        case 4: HE_MF_A(4); break;
        case 5: HE_MF_A(5); break;
        case 6: HE_MF_B(6); break;
        case 7: HE_MF_B(7); break;
        default: HE_MF_B(8); break;
    }
#undef HE_MF_A
#undef HE_MF_B
#undef HE_MF
}

bool modup_f64_raw_ok(int logN, int nsrc, uint64_t max_small_modulus) {
    // |o| < (2 + 5 nsrc) p + nsrc 2^32 after the matrix-vector sum (the second term: the exact low parts L of the split residues'
    // running sum, two per split source and below 2^31 each, join o unreduced -- HE_MODUP_MAGIC), + 2p per forward stage (column
    // and row): everything must stay below 2^53
    const long double bound = (long double)(2 + 5 * nsrc + 2 * logN) * (long double)max_small_modulus + (long double)nsrc * 0x1p32L;
    return bound < 0x1p53L;
}
hipError_t launch_modup_fused(const RingDev &r, const ModUpDesc *descs_dev, int ndesc, int nsrc, int dst_classes, View src,
                              View dstA, View dstB, int batch, hipStream_t s, bool f64_raw, int total_limbs) {
    if (ndesc <= 0 || batch <= 0) return hipSuccess;
    if (!modup_fused_supported(r.logN, nsrc)) return hipErrorInvalidValue;
    const int a = r.logN - ntt_row_bits(r.logN);
    ModUpFusedArgs A{};
    A.f64_raw = f64_raw ? 1 : 0;
    A.desc = descs_dev; A.src = src.p; A.dstA = dstA.p; A.dstB = dstB.p;
    A.src_bs = src.bstride; A.dstA_bs = dstA.bstride; A.dstB_bs = dstB.bstride;
    A.src_tab = src.tab; A.dstA_tab = dstA.tab; A.dstB_tab = dstB.tab;
    A.mc = r.mc; A.tw_fwd = r.tw_fwd; A.tw_inv = r.tw_inv; A.twd_fwd = r.twd_fwd; A.twd_inv = r.twd_inv; A.N = r.N;
    A.tws_fwd = r.tws_fwd;
    const bool use_f64 = ((dst_classes & 2) && r.twd_fwd != nullptr) || !modup_int_light(nsrc, a);
    const int n2 = r.N >> a;
    // fewer than four workgroups per CU: split the destinations (up to four ways) instead of leaving the chip idle
    static const int force_chunk = getenv("HERING_MODUP_NCHUNK") ? atoi(getenv("HERING_MODUP_NCHUNK")) : 0;
    const long wgs = (long)((n2 + 127) / 128) * ndesc * batch;
    int nchunk = force_chunk > 0 ? force_chunk : (wgs >= 1024 ? 1 : (int)std::min<long>(4, 1024 / std::max<long>(wgs, 1)));
    if (nchunk < 1) nchunk = 1;
    A.nchunk = nchunk;
    dim3 grid((unsigned)((n2 + 127) / 128), ndesc * nchunk, batch), block(128);
    ProfScope ps(K_MODUP, s, (double)total_limbs * batch * (double)r.N * 8.0);
    if (use_f64) launch_modup_fused_variant<true>(a, nsrc, grid, block, A, s);   // mixed: f64 for small destinations
    else launch_modup_fused_variant<false>(a, nsrc, grid, block, A, s);
    return hipGetLastError();
}

// single-limb digit (ring/basis_extension.go:402-436): centred value reduced into every
// destination limb:  c >= q/2 ? q_dst - BRedAdd(q - c) : BRedAdd(c)
struct CenterArgs {
    const uint64_t *src;
    uint64_t *dstA, *dstB;
    size_t src_bs, dstA_bs, dstB_bs;
    const size_t *src_tab, *dstA_tab, *dstB_tab;  // entry tables (View::tab)
    const ModConst *mc;
    int N, strict;
    ModUpArgs m;
};
__global__ void __launch_bounds__(256) center_copy_kernel(CenterArgs A) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= A.N) return;
    const size_t bz = blockIdx.z;
    const uint64_t qs = A.mc[A.m.src_mod[0]].q;
    uint64_t c = (A.src + voff(A.src_tab, A.src_bs, bz))[(size_t)A.m.src_limb[0] * A.N + x];
    const bool neg = (A.strict & 1) ? c > (qs >> 1) : c >= (qs >> 1);
    if (neg) c = qs - c;
    const int j = blockIdx.y;
    const ModConst mp = A.mc[A.m.dst_mod[j]];
    // bit 1: small-norm form, no reduction of |c| (ring/ringqp/operations.go:325-349)
    const uint64_t t = (A.strict & 2) ? c : bred_add(c, mp.q, mp.brc0);
    uint64_t *dst = A.m.dst_view[j] ? (A.dstB + voff(A.dstB_tab, A.dstB_bs, bz)) : (A.dstA + voff(A.dstA_tab, A.dstA_bs, bz));
    dst[(size_t)A.m.dst_limb[j] * A.N + x] = neg ? mp.q - t : t;
}
hipError_t launch_center_copy(const RingDev &r, const ModUpArgs &a, View src, View dstA, View dstB, int batch,
                              hipStream_t s, int strict) {
    if (a.ndst <= 0 || batch <= 0) return hipSuccess;
    CenterArgs A{};
    A.src = src.p; A.dstA = dstA.p; A.dstB = dstB.p;
    A.src_bs = src.bstride; A.dstA_bs = dstA.bstride; A.dstB_bs = dstB.bstride;
    A.src_tab = src.tab; A.dstA_tab = dstA.tab; A.dstB_tab = dstB.tab;
    A.mc = r.mc; A.N = r.N; A.m = a; A.strict = strict;
    dim3 grid((unsigned)((r.N + 255) / 256), a.ndst, batch), block(256);
    ProfScope ps(K_CENTER, s, (double)(1 + a.ndst) * batch * (double)r.N * 8.0);
    hipLaunchKernelGGL(center_copy_kernel, grid, block, 0, s, A);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// base-2 gadget windows
// ------------------------------------------------------------------------------------
struct MaskSpreadKArgs {
    const uint64_t *src;
    uint64_t *dec;
    size_t src_bs, dec_bs, dec_ds;
    int N;
    MaskSpreadArgs m;
};
__global__ void __launch_bounds__(256) mask_spread_kernel(MaskSpreadKArgs A) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= A.N) return;
    const int b = blockIdx.y;
    const size_t bz = blockIdx.z;
    const uint64_t v = (A.src[bz * A.src_bs + (size_t)A.m.blk_limb[b] * A.N + x] >> A.m.blk_shift[b]) & A.m.mask;
    uint64_t *dst = A.dec + bz * A.dec_bs + (size_t)b * A.dec_ds + x;
    for (int l = 0; l < A.m.ndst; l++) dst[(size_t)A.m.dst_limb[l] * A.N] = v;
}
hipError_t launch_mask_spread(const RingDev &r, const MaskSpreadArgs &a, View src, uint64_t *dec, size_t dec_bs, size_t dec_ds,
                              int batch, hipStream_t s) {
    if (!no_tab({src})) return hipErrorInvalidValue;  // no entry tables here (View::tab)
    if (a.nblk <= 0 || batch <= 0) return hipSuccess;
    MaskSpreadKArgs A{};
    A.src = src.p; A.src_bs = src.bstride; A.dec = dec; A.dec_bs = dec_bs; A.dec_ds = dec_ds; A.N = r.N; A.m = a;
    dim3 grid((unsigned)((r.N + 255) / 256), a.nblk, batch), block(256);
    ProfScope ps(K_MASK_SPREAD, s, ((double)a.nblk * a.ndst + a.ndst) * batch * (double)r.N * 8.0);
    hipLaunchKernelGGL(mask_spread_kernel, grid, block, 0, s, A);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// key-switch inner product.  Per (limb, coefficient): acc_k = sum_d key[d][k] * dec[d]
// accumulated exactly in 128 bits (high word kept below q), ONE Montgomery reduction at
// the end.  The canonical result equals the reference's lazy MRedLazy accumulation after
// its final Reduce (core/rlwe/evaluator_gadget_product.go:160-200).  Key words are loaded
// once per thread and reused across BB batch entries.
// ------------------------------------------------------------------------------------
struct KsKArgs {
    const uint64_t *own;
    size_t own_bs;
    const size_t *own_tab;  // entry table of `own` (View::tab)
    const uint64_t *dec;
    const uint64_t *key;
    uint64_t *o0Q, *o0P, *o1Q, *o1P;
    size_t dec_bs, oQ0_bs, oP0_bs, oQ1_bs, oP1_bs;
    const size_t *dec_tab, *oQ0_tab, *oP0_tab, *oQ1_tab, *oP1_tab, *add0_tab;  // entry tables (View::tab)
    const ModConst *mc;
    int N, batch;
    KsArgs k;
    // SCAT (AutomorphismHoistedLazy in one launch, core/rlwe/evaluator_automorphism.go:104-165): the accumulators are stored
    // through the automorphism (auto_dest), and component 0 of the Q limbs first takes the addend MRed(add0, add_s[limb]) --
    // ctIn[0] * P -- at the SOURCE position (the permutation is applied to the sum)
    unsigned sc_ginv;
    int sc_logN;
    const uint64_t *add0;
    size_t add0_bs;
    uint64_t add_s[kMaxLimbs];
    // the giant step of a linear transformation (KsScatter::plain / accumulate): plain addend on Q and P limbs, stores that add
    int add_plain, accum;
    const uint64_t *add0P;
    size_t add0P_bs;
    const size_t *add0P_tab;
};

template <int BB, bool SCAT = false>
__global__ void __launch_bounds__(256) ks_inner_kernel(KsKArgs A) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= A.N) return;
    const int l = blockIdx.y;
    const int b0 = blockIdx.z * BB;
    const ModConst m = A.mc[A.k.mod[l]];
    const uint64_t q = m.q;
    uint64_t hi0[BB], lo0[BB], hi1[BB], lo1[BB];
#pragma unroll
    for (int b = 0; b < BB; b++) { hi0[b] = lo0[b] = hi1[b] = lo1[b] = 0; }
    const uint64_t *kp = A.key + (size_t)A.k.key_limb[l] * A.N + x;
    const uint64_t *dp = A.dec + (size_t)A.k.dec_limb[l] * A.N + x;
    const uint64_t *op = A.own + (size_t)A.k.dec_limb[l] * A.N + x;
    // the digit's own Q limbs come from the NTT-domain input itself (block-uniform choice); entries past the batch re-read the
    // block's first entry so that every load is unconditional, and digit d + 1 is in flight while digit d is accumulated
    const int ql = A.k.dec_limb[l];  // Q-limb index for Q limbs
    size_t boff_own[BB], boff_dec[BB];
#pragma unroll
    for (int b = 0; b < BB; b++) {
        const size_t bb = (size_t)(b0 + b < A.batch ? b0 + b : b0);
        boff_own[b] = voff(A.own_tab, A.own_bs, bb);
        boff_dec[b] = voff(A.dec_tab, A.dec_bs, bb);
    }
    auto is_own = [&](int d) -> bool {
        return A.k.own_alpha > 0 && A.k.out_view[l] == 0 && ql >= d * A.k.own_alpha && ql < (d + 1) * A.k.own_alpha;
    };
    [[maybe_unused]] uint64_t addv[BB], prev0[BB], prev1[BB];
    if constexpr (SCAT) {
        const bool onP = A.k.out_view[l] != 0;
        if (A.add0 && (!onP || A.add_plain)) {  // block-uniform; in flight over the whole digit loop
            const uint64_t *ap = onP ? A.add0P : A.add0;
#pragma unroll
            for (int b = 0; b < BB; b++) {
                const size_t bb = (size_t)(b0 + b < A.batch ? b0 + b : b0);
                addv[b] = ldnt(&ap[(onP ? voff(A.add0P_tab, A.add0P_bs, bb) : voff(A.add0_tab, A.add0_bs, bb)) + (size_t)A.k.out_limb[l] * A.N + x]);
            }
        }
        if (A.accum) {  // the words the stores will increase, requested now
            const size_t pos = (size_t)A.k.out_limb[l] * A.N + auto_dest((unsigned)x, A.sc_ginv, A.sc_logN);
#pragma unroll
            for (int b = 0; b < BB; b++) {
                const size_t bb = (size_t)(b0 + b < A.batch ? b0 + b : b0);
                prev0[b] = (onP ? A.o0P + voff(A.oP0_tab, A.oP0_bs, bb) : A.o0Q + voff(A.oQ0_tab, A.oQ0_bs, bb))[pos];
                prev1[b] = (onP ? A.o1P + voff(A.oP1_tab, A.oP1_bs, bb) : A.o1Q + voff(A.oQ1_tab, A.oQ1_bs, bb))[pos];
            }
        }
    }
    uint64_t cn[BB], kn0, kn1;
    auto fetch = [&](int d) {
        kn0 = kp[(size_t)d * A.k.key_dstride];
        kn1 = kp[(size_t)d * A.k.key_dstride + A.k.key_kstride];
        if (is_own(d)) {
#pragma unroll
            for (int b = 0; b < BB; b++) cn[b] = ldnt(&op[boff_own[b]]);
        } else {
#pragma unroll
            for (int b = 0; b < BB; b++) cn[b] = ldnt(&dp[boff_dec[b] + (size_t)d * A.k.dec_dstride]);
        }
    };
    fetch(0);
    for (int d = 0; d < A.k.beta; d++) {
        const uint64_t k0 = kn0, k1 = kn1;
        uint64_t c[BB];
#pragma unroll
        for (int b = 0; b < BB; b++) c[b] = cn[b];
        if (d + 1 < A.k.beta) fetch(d + 1);
#pragma unroll
        for (int b = 0; b < BB; b++) {
            uint64_t ph, pl;
            mul64wide(c[b], k0, ph, pl);
            lo0[b] += pl; hi0[b] += ph + (lo0[b] < pl);
            hi0[b] = hi0[b] >= q ? hi0[b] - q : hi0[b];
            mul64wide(c[b], k1, ph, pl);
            lo1[b] += pl; hi1[b] += ph + (lo1[b] < pl);
            hi1[b] = hi1[b] >= q ? hi1[b] - q : hi1[b];
        }
    }
    const int ol = A.k.out_limb[l];
    const bool isP = A.k.out_view[l] != 0;
#pragma unroll
    for (int b = 0; b < BB; b++) {
        if (b0 + b < A.batch) {
            uint64_t r0 = cred(mred128_lazy(hi0[b], lo0[b], q, m.qinv), q);
            const uint64_t r1 = cred(mred128_lazy(hi1[b], lo1[b], q, m.qinv), q);
            uint64_t *o0 = isP ? A.o0P + voff(A.oP0_tab, A.oP0_bs, (size_t)(b0 + b)) : A.o0Q + voff(A.oQ0_tab, A.oQ0_bs, (size_t)(b0 + b));
            uint64_t *o1 = isP ? A.o1P + voff(A.oP1_tab, A.oP1_bs, (size_t)(b0 + b)) : A.o1Q + voff(A.oQ1_tab, A.oQ1_bs, (size_t)(b0 + b));
            size_t pos = (size_t)x;
            uint64_t s0 = r0, s1 = r1;
            if constexpr (SCAT) {
                if (A.add0 && A.add_plain) r0 = cred(r0 + addv[b], q);  // ringQP.Add of canonical words
                else if (A.add0 && !isP) r0 = cred(r0 + mred(addv[b], A.add_s[l], q, m.qinv), q);
                pos = auto_dest((unsigned)x, A.sc_ginv, A.sc_logN);
                s0 = r0;
                if (A.accum) { s0 = prev0[b] + r0; s1 = prev1[b] + r1; }  // AutomorphismNTTWithIndexThenAddLazy: no reduction
            }
            stnt(&o0[(size_t)ol * A.N + pos], s0);
            stnt(&o1[(size_t)ol * A.N + pos], s1);
        }
    }
}

hipError_t launch_ks_inner(const RingDev &r, const KsArgs &a, View dec, View own, const uint64_t *key, View out0Q,
                           View out0P, View out1Q, View out1P, int batch, hipStream_t s, const KsScatter *sc) {
    if (a.nlimbs <= 0 || batch <= 0) return hipSuccess;
    KsKArgs A{};
    A.own = own.p; A.own_bs = own.bstride; A.own_tab = own.tab;
    A.dec = dec.p; A.dec_bs = dec.bstride; A.key = key;
    A.o0Q = out0Q.p; A.o0P = out0P.p; A.o1Q = out1Q.p; A.o1P = out1P.p;
    A.oQ0_bs = out0Q.bstride; A.oP0_bs = out0P.bstride; A.oQ1_bs = out1Q.bstride; A.oP1_bs = out1P.bstride;
    A.dec_tab = dec.tab; A.oQ0_tab = out0Q.tab; A.oP0_tab = out0P.tab; A.oQ1_tab = out1Q.tab; A.oP1_tab = out1P.tab; A.add0_tab = nullptr;
    A.mc = r.mc; A.N = r.N; A.batch = batch; A.k = a;
    A.sc_ginv = 0; A.sc_logN = r.logN; A.add0 = nullptr; A.add0_bs = 0;
    A.add_plain = 0; A.accum = 0; A.add0P = nullptr; A.add0P_bs = 0; A.add0P_tab = nullptr;
    const int bb = batch >= 4 ? 4 : (batch >= 2 ? 2 : 1);
    dim3 grid((unsigned)((r.N + 255) / 256), a.nlimbs, (batch + bb - 1) / bb), block(256);
    // beta digits in, two key rows per digit shared by the batch, two accumulators out
    double ks_bytes = ((double)a.beta * batch + 2.0 * a.beta + 2.0 * batch) * a.nlimbs * (double)r.N * 8.0;
    if (sc && sc->ginv) {
        A.sc_ginv = sc->ginv;
        int nadd = 0;
        if (sc->add0.p) {
            A.add0 = sc->add0.p; A.add0_bs = sc->add0.bstride; A.add0_tab = sc->add0.tab;
            A.add_plain = sc->plain ? 1 : 0;
            A.add0P = sc->add0P.p; A.add0P_bs = sc->add0P.bstride; A.add0P_tab = sc->add0P.tab;
            for (int i = 0; i < a.nlimbs; i++) { A.add_s[i] = sc->add_s[i]; nadd += a.out_view[i] == 0 || sc->plain; }
        }
        A.accum = sc->accumulate ? 1 : 0;
        if (A.accum) nadd += 2 * a.nlimbs;  // the two destination rows are read as well
        ProfScope ps(K_KS_INNER, s, ks_bytes + (double)nadd * batch * (double)r.N * 8.0);
        if (bb == 4) hipLaunchKernelGGL((ks_inner_kernel<4, true>), grid, block, 0, s, A);
        else if (bb == 2) hipLaunchKernelGGL((ks_inner_kernel<2, true>), grid, block, 0, s, A);
        else hipLaunchKernelGGL((ks_inner_kernel<1, true>), grid, block, 0, s, A);
        return hipGetLastError();
    }
    ProfScope ps(K_KS_INNER, s, ks_bytes);
    if (bb == 4) hipLaunchKernelGGL((ks_inner_kernel<4>), grid, block, 0, s, A);
    else if (bb == 2) hipLaunchKernelGGL((ks_inner_kernel<2>), grid, block, 0, s, A);
    else hipLaunchKernelGGL((ks_inner_kernel<1>), grid, block, 0, s, A);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Ring.Shift / Ring.MultByMonomial (coefficient permutations, see kernels.h)
// ------------------------------------------------------------------------------------
struct ShiftArgs {
    const uint64_t *in;
    uint64_t *out;
    size_t in_bs, out_bs;
    const size_t *in_tab, *out_tab;  // entry tables (View::tab)
    const ModConst *mc;
    int N, k, monomial;
    uint8_t in_limb[kMaxLimbs], out_limb[kMaxLimbs], mod[kMaxLimbs];
};
__global__ void __launch_bounds__(256) shift_kernel(ShiftArgs A) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= A.N) return;
    const int l = blockIdx.y;
    const uint64_t *in = A.in + voff(A.in_tab, A.in_bs, blockIdx.z) + (size_t)A.in_limb[l] * A.N;
    uint64_t *out = A.out + voff(A.out_tab, A.out_bs, blockIdx.z) + (size_t)A.out_limb[l] * A.N;
    if (!A.monomial) {
        int src = j + A.k;
        if (src >= A.N) src -= A.N;
        out[j] = in[src];
        return;
    }
    // p1 * X^shift: first negate everything when shift >= N (tmp = q - p1), then rotate by shift mod N with the wrapped
    // part negated again (ring/operations.go:326-358)
    const uint64_t q = A.mc[A.mod[l]].q;
    const bool flip = A.k >= A.N;
    const int sh = flip ? A.k - A.N : A.k;
    if (sh == 0) {  // k = N: only the negation (k = 0 is a copy)
        out[j] = flip ? q - in[j] : in[j];
        return;
    }
    if (j < sh) {
        const uint64_t t = in[A.N - sh + j];
        out[j] = q - (flip ? q - t : t);
    } else {
        const uint64_t t = in[j - sh];
        out[j] = flip ? q - t : t;
    }
}
static hipError_t launch_shift_impl(const RingDev &r, const LimbTab &tab, View in, int k, View out, int batch, int monomial,
                                    hipStream_t s) {
    if (tab.n <= 0 || batch <= 0) return hipSuccess;
    ShiftArgs A{};
    A.in = in.p; A.out = out.p; A.in_bs = in.bstride; A.out_bs = out.bstride; A.in_tab = in.tab; A.out_tab = out.tab; A.mc = r.mc; A.N = r.N; A.k = k; A.monomial = monomial;
    for (int i = 0; i < tab.n; i++) { A.in_limb[i] = tab.in_limb[i]; A.out_limb[i] = tab.out_limb[i]; A.mod[i] = tab.mod[i]; }
    dim3 grid((unsigned)((r.N + 255) / 256), tab.n, batch), block(256);
    ProfScope ps(K_GATHER, s, 2.0 * tab.n * batch * (double)r.N * 8.0);
    hipLaunchKernelGGL(shift_kernel, grid, block, 0, s, A);
    return hipGetLastError();
}
hipError_t launch_shift(const RingDev &r, const LimbTab &tab, View in, int k, View out, int batch, hipStream_t s) {
    return launch_shift_impl(r, tab, in, k, out, batch, 0, s);
}
hipError_t launch_mult_by_monomial(const RingDev &r, const LimbTab &tab, View in, int shift, View out, int batch, hipStream_t s) {
    return launch_shift_impl(r, tab, in, shift, out, batch, 1, s);
}

// ------------------------------------------------------------------------------------
// plaintext-diagonal x ciphertext multiply-accumulate (see kernels.h)
// ------------------------------------------------------------------------------------
struct DiagMacKArgs {
    DiagMacArgs a;
    uint64_t *o0, *o1;
    size_t o0_bs, o1_bs;
    const size_t *o0_tab, *o1_tab;  // entry tables (View::tab) of the outputs; the terms' are in DiagMacArgs
    const ModConst *mc;
    int N, batch;
};
template <int BB>
__global__ void __launch_bounds__(256) diag_mac_kernel(const DiagMacKArgs A) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= A.N) return;
    const int l = blockIdx.y;
    const int b0 = blockIdx.z * BB;
    const ModConst m = A.mc[A.a.mod0 + l];
    const uint64_t q = m.q;
    uint64_t hi0[BB], lo0[BB], hi1[BB], lo1[BB];
#pragma unroll
    for (int b = 0; b < BB; b++) { hi0[b] = lo0[b] = hi1[b] = lo1[b] = 0; }
    const size_t lo = (size_t)l * A.N;
    for (int i = 0; i < A.a.n; i++) {
        if (A.a.c0[i] == nullptr) continue;  // block-uniform
        const int xi = A.a.index[i] ? (int)A.a.index[i][x] : x;
        const uint64_t *pp = A.a.pt[i] + lo + x;
        const uint64_t *p0 = A.a.c0[i] + lo + xi, *p1 = A.a.c1[i] + lo + xi;
        const size_t pbs = A.a.pt_bs[i], bs0 = A.a.c0_bs[i], bs1 = A.a.c1_bs[i];
        // entry tables of the terms (a coalesced batch of single-ciphertext callers): rows term_rows * i + {0, 1, 2} of term_tab hold
        // the word offsets of term i's plaintext, c0 and c1 per entry, `batch` entries per row
        const size_t *tt = A.a.term_tab ? A.a.term_tab + (size_t)A.a.term_rows * i * A.batch : nullptr;
        uint64_t w = pp[0];
#pragma unroll
        for (int b = 0; b < BB; b++) {
            if (b0 + b < A.batch) {
                const size_t zb = (size_t)(b0 + b);
                if (tt) w = pp[ldc(reinterpret_cast<const uint64_t *>(tt), zb)];
                else if (pbs != 0) w = pp[zb * pbs];
                uint64_t ph, pl;
                mul64wide(ldnt(&p0[tt ? (size_t)ldc(reinterpret_cast<const uint64_t *>(tt), (size_t)A.batch + zb) : zb * bs0]), w, ph, pl);
                lo0[b] += pl; hi0[b] += ph + (lo0[b] < pl);
                hi0[b] = hi0[b] >= q ? hi0[b] - q : hi0[b];
                mul64wide(ldnt(&p1[tt ? (size_t)ldc(reinterpret_cast<const uint64_t *>(tt), (size_t)2 * A.batch + zb) : zb * bs1]), w, ph, pl);
                lo1[b] += pl; hi1[b] += ph + (lo1[b] < pl);
                hi1[b] = hi1[b] >= q ? hi1[b] - q : hi1[b];
            }
        }
    }
#pragma unroll
    for (int b = 0; b < BB; b++) {
        if (b0 + b < A.batch) {
            uint64_t r0 = cred(mred128_lazy(hi0[b], lo0[b], q, m.qinv), q);
            uint64_t r1 = cred(mred128_lazy(hi1[b], lo1[b], q, m.qinv), q);
            uint64_t *o0 = A.o0 + voff(A.o0_tab, A.o0_bs, (size_t)(b0 + b)) + lo + x, *o1 = A.o1 + voff(A.o1_tab, A.o1_bs, (size_t)(b0 + b)) + lo + x;
            if (A.a.accumulate) {
                r0 = cred(r0 + bred_add(*o0, q, m.brc0), q);
                r1 = cred(r1 + bred_add(*o1, q, m.brc0), q);
            }
            *o0 = r0;
            *o1 = r1;
        }
    }
}

hipError_t launch_diag_mac(const RingDev &r, const DiagMacArgs &a, View out0, View out1, int batch, hipStream_t s) {
    if (a.nlimbs <= 0 || batch <= 0) return hipSuccess;
    if (a.n < 0 || a.n > kMaxDiag) return hipErrorInvalidValue;
    DiagMacKArgs A{};
    A.a = a;
    A.o0 = out0.p; A.o1 = out1.p; A.o0_bs = out0.bstride; A.o1_bs = out1.bstride; A.o0_tab = out0.tab; A.o1_tab = out1.tab;
    A.mc = r.mc; A.N = r.N; A.batch = batch;
    const int bb = batch >= 4 ? 4 : (batch >= 2 ? 2 : 1);
    dim3 grid((unsigned)((r.N + 255) / 256), a.nlimbs, (batch + bb - 1) / bb), block(256);
    // per term the plaintext diagonal (shared by the batch unless it has a batch stride) and the two ciphertext components; the two
    // accumulators out (and in, when accumulating)
    double dm_limbs = (a.accumulate ? 4.0 : 2.0) * batch;
    for (int i = 0; i < a.n; i++) {
        if (!a.c0[i] && !a.c1[i]) continue;
        dm_limbs += ((a.pt_bs[i] || a.term_tab) ? (double)batch : 1.0) + (a.c0[i] ? batch : 0) + (a.c1[i] ? batch : 0);
    }
    ProfScope ps(K_DIAG_MAC, s, dm_limbs * a.nlimbs * (double)r.N * 8.0);
    if (bb == 4) hipLaunchKernelGGL((diag_mac_kernel<4>), grid, block, 0, s, A);
    else if (bb == 2) hipLaunchKernelGGL((diag_mac_kernel<2>), grid, block, 0, s, A);
    else hipLaunchKernelGGL((diag_mac_kernel<1>), grid, block, 0, s, A);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// degree-1 x degree-1 tensor product, fused: 4 inputs -> 3 outputs in one pass.
// ------------------------------------------------------------------------------------
struct TensorArgs {
    const uint64_t *a0, *a1, *b0, *b1;
    uint64_t *c0, *c1, *c2;
    size_t a0_bs, a1_bs, b0_bs, b1_bs, c0_bs, c1_bs, c2_bs;
    const size_t *a0_tab, *a1_tab, *b0_tab, *b1_tab, *c0_tab, *c1_tab, *c2_tab;  // entry tables (View::tab)
    const ModConst *mc;
    int N;
    uint8_t in_limb[kMaxLimbs], out_limb[kMaxLimbs], mod[kMaxLimbs];
    uint64_t s[kMaxLimbs];
};
__global__ void __launch_bounds__(256) tensor_kernel(TensorArgs A) {
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (j >= A.N) return;
    const int yy = blockIdx.y;
    const ModConst m = A.mc[A.mod[yy]];
    const uint64_t q = m.q, qinv = m.qinv, sc = A.s[yy];
    const size_t bz = blockIdx.z, io = (size_t)A.in_limb[yy] * A.N + j, oo = (size_t)A.out_limb[yy] * A.N + j;
    if (!A.c0) {  // c2 only (the fused MulRelin forms c0 / c1 in the ModDown epilogue, NttEpilogue::tensor)
        const ulonglong2 a1 = ldnt2(A.a1 + voff(A.a1_tab, A.a1_bs, bz) + io), b1 = ldnt2(A.b1 + voff(A.b1_tab, A.b1_bs, bz) + io);
        ulonglong2 c2;
        c2.x = mred(mred(a1.x, sc, q, qinv), b1.x, q, qinv);
        c2.y = mred(mred(a1.y, sc, q, qinv), b1.y, q, qinv);
        *reinterpret_cast<ulonglong2 *>(A.c2 + voff(A.c2_tab, A.c2_bs, bz) + oo) = c2;
        return;
    }
    const ulonglong2 a0 = ldnt2(A.a0 + voff(A.a0_tab, A.a0_bs, bz) + io);
    const ulonglong2 a1 = ldnt2(A.a1 + voff(A.a1_tab, A.a1_bs, bz) + io);
    const ulonglong2 b0 = ldnt2(A.b0 + voff(A.b0_tab, A.b0_bs, bz) + io);
    const ulonglong2 b1 = ldnt2(A.b1 + voff(A.b1_tab, A.b1_bs, bz) + io);
    ulonglong2 c0, c1, c2;
    {
        const uint64_t t0 = mred(a0.x, sc, q, qinv), t1 = mred(a1.x, sc, q, qinv);
        c0.x = mred(t0, b0.x, q, qinv);
        c2.x = mred(t1, b1.x, q, qinv);
        c1.x = cred(mred(t0, b1.x, q, qinv) + mred(t1, b0.x, q, qinv), q);
    }
    {
        const uint64_t t0 = mred(a0.y, sc, q, qinv), t1 = mred(a1.y, sc, q, qinv);
        c0.y = mred(t0, b0.y, q, qinv);
        c2.y = mred(t1, b1.y, q, qinv);
        c1.y = cred(mred(t0, b1.y, q, qinv) + mred(t1, b0.y, q, qinv), q);
    }
    *reinterpret_cast<ulonglong2 *>(A.c0 + voff(A.c0_tab, A.c0_bs, bz) + oo) = c0;
    *reinterpret_cast<ulonglong2 *>(A.c1 + voff(A.c1_tab, A.c1_bs, bz) + oo) = c1;
    *reinterpret_cast<ulonglong2 *>(A.c2 + voff(A.c2_tab, A.c2_bs, bz) + oo) = c2;
}
hipError_t launch_tensor(const RingDev &r, const LimbTab &tab, const uint64_t *scalar, View a0, View a1, View b0, View b1,
                         View c0, View c1, View c2, int batch, hipStream_t s) {
    if (tab.n <= 0 || batch <= 0) return hipSuccess;
    TensorArgs A{};
    A.a0 = a0.p; A.a1 = a1.p; A.b0 = b0.p; A.b1 = b1.p; A.c0 = c0.p; A.c1 = c1.p; A.c2 = c2.p;
    A.a0_bs = a0.bstride; A.a1_bs = a1.bstride; A.b0_bs = b0.bstride; A.b1_bs = b1.bstride;
    A.c0_bs = c0.bstride; A.c1_bs = c1.bstride; A.c2_bs = c2.bstride;
    A.a0_tab = a0.tab; A.a1_tab = a1.tab; A.b0_tab = b0.tab; A.b1_tab = b1.tab; A.c0_tab = c0.tab; A.c1_tab = c1.tab; A.c2_tab = c2.tab;
    A.mc = r.mc; A.N = r.N;
    for (int i = 0; i < tab.n; i++) {
        A.in_limb[i] = tab.in_limb[i]; A.out_limb[i] = tab.out_limb[i]; A.mod[i] = tab.mod[i]; A.s[i] = scalar[i];
    }
    dim3 grid((unsigned)((r.N / 2 + 255) / 256), tab.n, batch), block(256);
    // a1, b1 -> c2 always; a0, b0 -> c0, c1 when those outputs exist
    ProfScope ps(K_TENSOR, s, (c0.p ? 7.0 : 3.0) * tab.n * batch * (double)r.N * 8.0);
    hipLaunchKernelGGL(tensor_kernel, grid, block, 0, s, A);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// entry tables of coalesced calls (View::tab): the values travel as kernel arguments, so nothing on the host has to outlive the
// enqueue and the fill is ordered on the stream like every other launch (one table per evaluator is enough: the fill of the
// next batch runs after the last kernel of the previous one)
// ------------------------------------------------------------------------------------
constexpr int kTabFillMax = 448;
struct TabFillArgs {
    size_t *dst;
    int n;
    size_t v[kTabFillMax];
};
__global__ void __launch_bounds__(64) tab_fill_kernel(TabFillArgs A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < A.n) A.dst[i] = A.v[i];
}
hipError_t launch_tab_fill(size_t *dst, const size_t *vals, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += kTabFillMax) {
        TabFillArgs A{};
        A.dst = dst + i0;
        A.n = n - i0 < kTabFillMax ? n - i0 : kTabFillMax;
        for (int i = 0; i < A.n; i++) A.v[i] = vals[i0 + i];
        hipLaunchKernelGGL(tab_fill_kernel, dim3((unsigned)((A.n + 63) / 64)), dim3(64), 0, s, A);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------------------------
// modular-multiply throughput probe (bench.py --microbench): `iters` dependent MRedLazy
// per element, 4 independent chains per thread.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) modmul_probe_kernel(uint64_t *buf, size_t n, int iters, uint64_t q, uint64_t qinv) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 4 + 3 >= n) return;
    uint64_t a0 = buf[i * 4], a1 = buf[i * 4 + 1], a2 = buf[i * 4 + 2], a3 = buf[i * 4 + 3];
    const uint64_t w = (a0 >> 4) | 1;  // below q (the probe's modulus has 61 bits): inside the sequence's domain
    // the production sequence of the row kernels (mred_lazy_col_asm, 16 instructions): the compiler's own rendering of the same
    // product is ~15 % slower and would flatter every fraction measured against it
    for (int k = 0; k < iters; k++) {
        a0 = mred_lazy_col_asm(a0, w, q, qinv); a1 = mred_lazy_col_asm(a1, w, q, qinv);
        a2 = mred_lazy_col_asm(a2, w, q, qinv); a3 = mred_lazy_col_asm(a3, w, q, qinv);
    }
    buf[i * 4] = a0; buf[i * 4 + 1] = a1; buf[i * 4 + 2] = a2; buf[i * 4 + 3] = a3;
}
// the same probe for the double-precision exact product (modmul_f64: the arithmetic of every limb below 2^47)
__global__ void __launch_bounds__(256) modmul_f64_probe_kernel(double *buf, size_t n, int iters, double q, double qi) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 4 + 3 >= n) return;
    double a0 = buf[i * 4], a1 = buf[i * 4 + 1], a2 = buf[i * 4 + 2], a3 = buf[i * 4 + 3];
    const double w = a0;
    for (int k = 0; k < iters; k++) {
        a0 = modmul_f64(a0, w, q, qi); a1 = modmul_f64(a1, w, q, qi);
        a2 = modmul_f64(a2, w, q, qi); a3 = modmul_f64(a3, w, q, qi);
    }
    buf[i * 4] = a0; buf[i * 4 + 1] = a1; buf[i * 4 + 2] = a2; buf[i * 4 + 3] = a3;
}
hipError_t launch_modmul_f64_probe(double *buf, size_t n, int iters, double q, hipStream_t s) {
    const size_t threads = n / 4;
    hipLaunchKernelGGL(modmul_f64_probe_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, buf, n, iters, q, 1.0 / q);
    return hipGetLastError();
}
hipError_t launch_modmul_probe(uint64_t *buf, size_t n, int iters, uint64_t q, uint64_t qinv, hipStream_t s) {
    const size_t threads = n / 4;
    hipLaunchKernelGGL(modmul_probe_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, buf, n, iters, q, qinv);
    return hipGetLastError();
}

}  // namespace he
