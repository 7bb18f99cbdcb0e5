/*
 * lattigo_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see lattigo_oracle.h).
 *
 * Scalar C restatement of tuneinsight/lattigo v6.2.0 `ring` + `core/rlwe`
 * key-switch arithmetic.  Written from the behaviour of the cited reference
 * lines; no reference source is copied.  The reference is scalar Go (manually
 * unrolled by 8); this file is scalar C with plain loops -- same word-level
 * arithmetic, same evaluation order wherever the order is observable (lazy
 * representatives, float64 accumulation).
 */
#define _GNU_SOURCE /* pthread_setaffinity_np, CPU_SET (the timed CPU baseline at the end of the file) */
#include "lattigo_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

static __thread char lo_err[256];
const char *lo_last_error(void) { return lo_err; }
#define LO_FAIL(...) do { snprintf(lo_err, sizeof lo_err, __VA_ARGS__); } while (0)

static inline uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) >> 64); }

/* ========================================================================== */
/* Scratch pool                                                                */
/* ========================================================================== */
/* The reference draws every temporary polynomial from sync.Pool-backed buffer pools
 * (ring/pool.go:17, core/rlwe/pool.go:17): a steady-state evaluator call allocates
 * nothing.  Same here: per-thread free lists keyed by size, so that the timed CPU
 * baseline (bench.py) does not spend its time in mmap/munmap and page faults. */
#define POOL_SLOTS 96
typedef struct { size_t sz, pad; } pool_hdr;
static __thread struct { size_t sz; void *p; } pool_free_list[POOL_SLOTS];
static void *pool_get(size_t bytes) {
    for (int i = 0; i < POOL_SLOTS; i++)
        if (pool_free_list[i].p && pool_free_list[i].sz == bytes) {
            void *p = pool_free_list[i].p;
            pool_free_list[i].p = NULL;
            return p;
        }
    pool_hdr *h = (pool_hdr *)malloc(sizeof(pool_hdr) + bytes);
    if (!h) return NULL;
    h->sz = bytes;
    return h + 1;
}
static void pool_put(void *p) {
    if (!p) return;
    pool_hdr *h = (pool_hdr *)p - 1;
    for (int i = 0; i < POOL_SLOTS; i++)
        if (!pool_free_list[i].p) { pool_free_list[i].p = p; pool_free_list[i].sz = h->sz; return; }
    free(h);
}
/* returns the calling thread's cached scratch to the allocator */
void lo_pool_release(void) {
    for (int i = 0; i < POOL_SLOTS; i++)
        if (pool_free_list[i].p) { free((pool_hdr *)pool_free_list[i].p - 1); pool_free_list[i].p = NULL; }
}

/* ========================================================================== */
/* Scalars: ring/modular_reduction.go                                          */
/* ========================================================================== */

/* MForm, ring/modular_reduction.go:11-35 */
uint64_t lo_mform(uint64_t a, uint64_t q, const uint64_t brc[2]) {
    uint64_t mhi = mulhi64(a, brc[1]);
    uint64_t r = (uint64_t)(0 - (a * brc[0] + mhi)) * q;
    if (r >= q) r -= q;
    return r;
}
/* MFormLazy, :40-45 */
uint64_t lo_mform_lazy(uint64_t a, uint64_t q, const uint64_t brc[2]) {
    uint64_t mhi = mulhi64(a, brc[1]);
    return (uint64_t)(0 - (a * brc[0] + mhi)) * q;
}
/* IMForm, :49-56 */
uint64_t lo_imform(uint64_t a, uint64_t q, uint64_t qinv) {
    uint64_t r = mulhi64(a * qinv, q);
    r = q - r;
    if (r >= q) r -= q;
    return r;
}
/* IMFormLazy, :61-65 */
uint64_t lo_imform_lazy(uint64_t a, uint64_t q, uint64_t qinv) {
    return q - mulhi64(a * qinv, q);
}
/* GenMRedConstant, :68-75 */
uint64_t lo_gen_mred_constant(uint64_t q) {
    uint64_t m = 1;
    for (int i = 0; i < 63; i++) { m *= q; q *= q; }
    return m;
}
/* GenBRedConstant, :99-107: floor(2^128 / q) split into (hi, lo). */
void lo_gen_bred_constant(uint64_t q, uint64_t brc[2]) {
    /* 2^128 / q by schoolbook long division of the 3-word number {1,0,0}. */
    u128 rem = 1;                       /* top word */
    rem = (rem << 64);                  /* {1,0} */
    uint64_t hi = (uint64_t)(rem / q);
    rem = rem % q;
    rem = (rem << 64);                  /* bring down the last zero word */
    uint64_t lo = (uint64_t)(rem / q);
    brc[0] = hi; brc[1] = lo;
}
/* MRed, :78-86 */
uint64_t lo_mred(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv) {
    u128 m = (u128)x * y;
    uint64_t mhi = (uint64_t)(m >> 64), mlo = (uint64_t)m;
    uint64_t hhi = mulhi64(mlo * qinv, q);
    uint64_t r = mhi - hhi + q;
    if (r >= q) r -= q;
    return r;
}
/* MRedLazy, :90-95 */
uint64_t lo_mred_lazy(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv) {
    u128 m = (u128)x * y;
    uint64_t ahi = (uint64_t)(m >> 64), alo = (uint64_t)m;
    uint64_t H = mulhi64(alo * qinv, q);
    return ahi - H + q;
}
/* BRedAdd, :110-117 */
uint64_t lo_bred_add(uint64_t a, uint64_t q, const uint64_t brc[2]) {
    uint64_t mhi = mulhi64(a, brc[0]);
    uint64_t r = a - mhi * q;
    if (r >= q) r -= q;
    return r;
}
/* BRedAddLazy, :121-124 */
uint64_t lo_bred_add_lazy(uint64_t a, uint64_t q, const uint64_t brc[2]) {
    return a - mulhi64(a, brc[0]) * q;
}
/* BRedLazy, :166-196 (word-for-word carry chain) */
uint64_t lo_bred_lazy(uint64_t x, uint64_t y, uint64_t q, const uint64_t brc[2]) {
    u128 m = (u128)x * y;
    uint64_t mhi = (uint64_t)(m >> 64), mlo = (uint64_t)m;
    uint64_t r = mhi * brc[0];
    u128 h = (u128)mlo * brc[0];
    uint64_t hhi = (uint64_t)(h >> 64), hlo = (uint64_t)h;
    r += hhi;
    uint64_t lhi = mulhi64(mlo, brc[1]);
    uint64_t s0 = hlo + lhi;
    uint64_t carry = s0 < hlo;
    r += carry;
    h = (u128)mhi * brc[1];
    hhi = (uint64_t)(h >> 64); hlo = (uint64_t)h;
    r += hhi;
    uint64_t s1 = hlo + s0;
    carry = s1 < hlo;
    r += carry;
    return mlo - r * q;
}
/* BRed, :127-162 */
uint64_t lo_bred(uint64_t x, uint64_t y, uint64_t q, const uint64_t brc[2]) {
    uint64_t r = lo_bred_lazy(x, y, q, brc);
    if (r >= q) r -= q;
    return r;
}
/* CRed, :200-205 */
uint64_t lo_cred(uint64_t a, uint64_t q) { return a >= q ? a - q : a; }

/* ModExp, ring/utils.go:30-41 */
uint64_t lo_modexp(uint64_t x, uint64_t e, uint64_t p) {
    uint64_t brc[2];
    lo_gen_bred_constant(p, brc);
    uint64_t result = 1;
    for (uint64_t i = e; i > 0; i >>= 1) {
        if (i & 1) result = lo_bred(result, x, p, brc);
        x = lo_bred(x, x, p, brc);
    }
    return result;
}
/* ModexpMontgomery, ring/utils.go:58-69 */
static uint64_t modexp_montgomery(uint64_t x, uint64_t e, uint64_t q, uint64_t qinv, const uint64_t brc[2]) {
    uint64_t result = lo_mform(1, q, brc);
    for (uint64_t i = e; i > 0; i >>= 1) {
        if (i & 1) result = lo_mred(result, x, q, qinv);
        x = lo_mred(x, x, q, qinv);
    }
    return result;
}

/* ---- primality / factoring (replaces math/big ProbablyPrime and
 *      utils/factorization.GetFactors; both are exact on 64-bit inputs, so any
 *      exact method yields identical results) --------------------------------- */
static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)(((u128)a * b) % m); }
static uint64_t powmod(uint64_t a, uint64_t e, uint64_t m) {
    uint64_t r = 1; a %= m;
    while (e) { if (e & 1) r = mulmod(r, a, m); a = mulmod(a, a, m); e >>= 1; }
    return r;
}
/* IsPrime, ring/primes.go:11 (deterministic Miller-Rabin is exact below 2^64) */
int lo_is_prime(uint64_t n) {
    static const uint64_t bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return 0;
    for (int i = 0; i < 12; i++) { if (n % bases[i] == 0) return n == bases[i]; }
    uint64_t d = n - 1; int s = 0;
    while ((d & 1) == 0) { d >>= 1; s++; }
    for (int i = 0; i < 12; i++) {
        uint64_t x = powmod(bases[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int r = 1; r < s; r++) { x = mulmod(x, x, n); if (x == n - 1) { comp = 0; break; } }
        if (comp) return 0;
    }
    return 1;
}
static uint64_t gcd64(uint64_t a, uint64_t b) { while (b) { uint64_t t = a % b; a = b; b = t; } return a; }
static uint64_t pollard_rho(uint64_t n) {
    if ((n & 1) == 0) return 2;
    for (uint64_t c = 1;; c++) {
        uint64_t x = 2, y = 2, d = 1;
        while (d == 1) {
            x = (mulmod(x, x, n) + c) % n;
            y = (mulmod(y, y, n) + c) % n;
            y = (mulmod(y, y, n) + c) % n;
            d = gcd64(x > y ? x - y : y - x, n);
        }
        if (d != n) return d;
    }
}
static void factor_rec(uint64_t n, uint64_t *out, int *cnt) {
    if (n == 1) return;
    if (lo_is_prime(n)) {
        for (int i = 0; i < *cnt; i++) if (out[i] == n) return;
        out[(*cnt)++] = n;
        return;
    }
    uint64_t d = pollard_rho(n);
    factor_rec(d, out, cnt);
    factor_rec(n / d, out, cnt);
}
static int unique_factors(uint64_t n, uint64_t *out) {
    int cnt = 0;
    for (uint64_t p = 2; p < 1000 && p * p <= n; p++) {
        if (n % p == 0) { out[cnt++] = p; while (n % p == 0) n /= p; }
    }
    factor_rec(n, out, &cnt);
    return cnt;
}

/* ========================================================================== */
/* SubRing / Ring                                                               */
/* ========================================================================== */

static int bitlen64(uint64_t x) { int n = 0; while (x) { n++; x >>= 1; } return n; }
/* utils.BitReverse64, utils/utils.go:34-36 */
static uint64_t bitrev64(uint64_t x, int bits) {
    uint64_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

/* NewSubRingWithCustomNTT (ring/subring.go:46-80) + generateNTTConstants
 * (:99-159) + PrimitiveRoot (:163-196), standard ring (NthRoot = 2N). */
lo_subring *lo_subring_new(int N, uint64_t q) { return lo_subring_new_nthroot(N, q, 2 * (uint64_t)N); }
/* NthRoot = 2N: standard ring; NthRoot = 4N: conjugate-invariant ring (ring/ring.go:260-261) */
lo_subring *lo_subring_new_nthroot(int N, uint64_t q, uint64_t nthroot) {
    if (N < 8 || (N & (N - 1)) != 0) { LO_FAIL("invalid ring degree: must be a power of 2 greater than 8"); return NULL; }
    if (!lo_is_prime(q)) { LO_FAIL("invalid modulus: %llu is not prime)", (unsigned long long)q); return NULL; }
    if ((q & (nthroot - 1)) != 1) { LO_FAIL("invalid modulus: %llu != 1 mod NthRoot)", (unsigned long long)q); return NULL; }
    lo_subring *s = (lo_subring *)calloc(1, sizeof *s);
    s->N = N; s->q = q; s->nthroot = nthroot;
    s->mask = ((uint64_t)1 << bitlen64(q - 1)) - 1;
    lo_gen_bred_constant(q, s->brc);
    s->qinv = lo_gen_mred_constant(q);
    s->nfactors = unique_factors(q - 1, s->factors);
    /* smallest g >= 3 that is a generator, subring.go:181-193 */
    uint64_t g = 2; int notfound = 1;
    while (notfound) {
        g++;
        for (int i = 0; i < s->nfactors; i++) {
            if (lo_modexp(g, (q - 1) / s->factors[i], q) == 1) { notfound = 1; break; }
            notfound = 0;
        }
    }
    s->primroot = g;
    int lognth = bitlen64(nthroot >> 1) - 1;
    s->ninv = lo_mform(lo_modexp(nthroot >> 1, q - 2, q), q, s->brc);
    uint64_t psi = lo_mform(lo_modexp(g, (q - 1) / nthroot, q), q, s->brc);
    uint64_t psiinv = lo_mform(lo_modexp(g, q - ((q - 1) / nthroot) - 1, q), q, s->brc);
    uint64_t n = nthroot >> 1;
    s->roots_fwd = (uint64_t *)malloc(n * sizeof(uint64_t));
    s->roots_bwd = (uint64_t *)malloc(n * sizeof(uint64_t));
    s->roots_fwd[0] = lo_mform(1, q, s->brc);
    s->roots_bwd[0] = lo_mform(1, q, s->brc);
    for (uint64_t j = 1; j < n; j++) {
        uint64_t prev = bitrev64(j - 1, lognth), next = bitrev64(j, lognth);
        s->roots_fwd[next] = lo_mred(s->roots_fwd[prev], psi, q, s->qinv);
        s->roots_bwd[next] = lo_mred(s->roots_bwd[prev], psiinv, q, s->qinv);
    }
    return s;
}
void lo_subring_free(lo_subring *s) {
    if (!s) return;
    free(s->roots_fwd); free(s->roots_bwd); free(s);
}

/* NewRing (ring/ring.go:207-320) + rewRescaleConstants (:329-346) */
lo_ring *lo_ring_new(int N, const uint64_t *moduli, int nmod) { return lo_ring_new_type(N, moduli, nmod, 0); }
/* NewRingFromType, ring/ring.go:267-279: type 0 = Standard, 1 = ConjugateInvariant */
lo_ring *lo_ring_new_type(int N, const uint64_t *moduli, int nmod, int type) {
    if (nmod <= 0) { LO_FAIL("invalid ModuliChain (must be a non-empty []uint64)"); return NULL; }
    for (int i = 0; i < nmod; i++)
        for (int j = i + 1; j < nmod; j++)
            if (moduli[i] == moduli[j]) { LO_FAIL("invalid ModuliChain (moduli are not distinct)"); return NULL; }
    lo_ring *r = (lo_ring *)calloc(1, sizeof *r);
    r->N = N; r->nmod = nmod;
    r->s = (lo_subring **)calloc(nmod, sizeof(lo_subring *));
    for (int i = 0; i < nmod; i++) {
        r->s[i] = lo_subring_new_nthroot(N, moduli[i], (type == 1 ? 4 : 2) * (uint64_t)N);
        if (!r->s[i]) { lo_ring_free(r); return NULL; }
    }
    r->rescale = (uint64_t **)calloc(nmod > 1 ? nmod - 1 : 1, sizeof(uint64_t *));
    for (int j = nmod - 1; j > 0; j--) {
        uint64_t qj = r->s[j]->q;
        r->rescale[j - 1] = (uint64_t *)malloc(j * sizeof(uint64_t));
        for (int i = 0; i < j; i++) {
            uint64_t qi = r->s[i]->q;
            r->rescale[j - 1][i] = lo_mform(qi - lo_modexp(qj, qi - 2, qi), qi, r->s[i]->brc);
        }
    }
    return r;
}
void lo_ring_free(lo_ring *r) {
    if (!r) return;
    if (r->s) { for (int i = 0; i < r->nmod; i++) lo_subring_free(r->s[i]); free(r->s); }
    if (r->rescale) { for (int i = 0; i < r->nmod - 1; i++) free(r->rescale[i]); free(r->rescale); }
    free(r);
}
const uint64_t *lo_ring_roots_fwd(const lo_ring *r, int i) { return r->s[i]->roots_fwd; }
const uint64_t *lo_ring_roots_bwd(const lo_ring *r, int i) { return r->s[i]->roots_bwd; }
void lo_ring_constants(const lo_ring *r, int i, uint64_t out[7]) {
    const lo_subring *s = r->s[i];
    out[0] = s->q; out[1] = s->qinv; out[2] = s->brc[0]; out[3] = s->brc[1];
    out[4] = s->ninv; out[5] = s->primroot; out[6] = s->mask;
}
uint64_t lo_ring_rescale_constant(const lo_ring *r, int j, int i) { return r->rescale[j - 1][i]; }

/* ---- NTTFriendlyPrimesGenerator (ring/primes.go:16-229) -------------------- */
typedef struct {
    double size; uint64_t next, prev, nthroot; int check_next, check_prev;
} primegen;
static void primegen_init(primegen *g, uint64_t bitsize, uint64_t nthroot) {
    g->check_next = 1; g->check_prev = 1;
    g->next = ((uint64_t)1 << bitsize) + 1;
    g->prev = ((uint64_t)1 << bitsize) + 1;
    if (g->next > 0xffffffffffffffffULL - nthroot) g->check_next = 0;
    if (g->prev < nthroot) g->check_prev = 0;
    g->prev -= nthroot;
    g->nthroot = nthroot; g->size = (double)bitsize;
}
/* NextDownstreamPrime, primes.go:118-150 */
static int primegen_down(primegen *g, uint64_t *out) {
    uint64_t prev = g->prev;
    if (!g->check_prev) return -1;
    for (;;) {
        if (g->size - log2((double)prev) >= 0.5 || prev < g->nthroot) { g->check_prev = 0; return -1; }
        if (lo_is_prime(prev)) { g->prev = prev - g->nthroot; *out = prev; return 0; }
        prev -= g->nthroot;
    }
}
/* NextAlternatingPrime, primes.go:152-229 */
static int primegen_alt(primegen *g, uint64_t *out) {
    uint64_t next = g->next, prev = g->prev;
    int cn = g->check_next, cp = g->check_prev;
    for (;;) {
        if (!(cn || cp)) return -1;
        if (cn) {
            if (log2((double)next) - g->size >= 0.5 || next > 0xffffffffffffffffULL - g->nthroot) {
                cn = 0;
            } else {
                if (lo_is_prime(next)) {
                    g->next = next + g->nthroot; g->prev = prev; g->check_next = cn; g->check_prev = cp;
                    *out = next; return 0;
                }
                next += g->nthroot;
            }
        }
        if (cp) {
            if (g->size - log2((double)prev) >= 0.5 || prev < g->nthroot) {
                cp = 0;
            } else {
                if (lo_is_prime(prev)) {
                    g->next = next; g->prev = prev - g->nthroot; g->check_next = cn; g->check_prev = cp;
                    *out = prev; return 0;
                }
                prev -= g->nthroot;
            }
        }
    }
}
/* GenModuli, core/rlwe/params.go:811-862 */
int lo_gen_moduli(int log_nth_root, const int *logq, int nq, const int *logp, int np,
                  uint64_t *q_out, uint64_t *p_out) {
    int count[64] = {0};
    uint64_t *primes[64] = {0};
    int used[64] = {0};
    for (int i = 0; i < nq; i++) { if (logq[i] <= 0 || logq[i] > 61) { LO_FAIL("logQ out of range"); return -1; } count[logq[i]]++; }
    for (int i = 0; i < np; i++) { if (logp[i] <= 0 || logp[i] > 61) { LO_FAIL("logP out of range"); return -1; } count[logp[i]]++; }
    int rc = 0;
    for (int b = 1; b < 64 && rc == 0; b++) {
        if (!count[b]) continue;
        primes[b] = (uint64_t *)malloc(count[b] * sizeof(uint64_t));
        primegen g; primegen_init(&g, (uint64_t)b, (uint64_t)1 << log_nth_root);
        for (int k = 0; k < count[b]; k++) {
            int e = (b == 61) ? primegen_down(&g, &primes[b][k]) : primegen_alt(&g, &primes[b][k]);
            if (e) { LO_FAIL("cannot GenModuli: failed to generate %d primes of bit-size=%d", count[b], b); rc = -1; break; }
        }
    }
    if (rc == 0) {
        for (int i = 0; i < nq; i++) q_out[i] = primes[logq[i]][used[logq[i]]++];
        for (int i = 0; i < np; i++) p_out[i] = primes[logp[i]][used[logp[i]]++];
    }
    for (int b = 0; b < 64; b++) free(primes[b]);
    return rc;
}

/* ========================================================================== */
/* NTT: ring/ntt.go                                                             */
/* ========================================================================== */

/* nttCoreLazy (:209-221): nttLazy (:223-257) for N<16, else the schedule of
 * nttUnrolled16Lazy (:258-552): first stage without the 4q correction, then a
 * stage applies `U >= 4q -> U -= 4q` iff bitlen(m) is odd (:318) or t == 1
 * (:502-517).  Output in [0, 6q-2]. */
static void ntt_core_lazy(const uint64_t *p1, uint64_t *p2, int N, uint64_t Q, uint64_t qinv, const uint64_t *roots) {
    uint64_t fourQ = 4 * Q, twoQ = 2 * Q;
    int t = N >> 1;
    uint64_t F = roots[1];
    int unrolled = N >= 16;
    for (int jx = 0, jy = t; jx < t; jx++, jy++) {
        uint64_t U = p1[jx];
        if (!unrolled && U >= fourQ) U -= fourQ;
        uint64_t V = lo_mred_lazy(p1[jy], F, Q, qinv);
        p2[jx] = U + V; p2[jy] = U + twoQ - V;
    }
    for (int m = 2; m < N; m <<= 1) {
        t >>= 1;
        int reduce = unrolled ? ((bitlen64((uint64_t)m) & 1) == 1 || t == 1) : 1;
        for (int i = 0; i < m; i++) {
            int j1 = (i * t) << 1, j2 = j1 + t;
            F = roots[m + i];
            for (int jx = j1, jy = j1 + t; jx < j2; jx++, jy++) {
                uint64_t U = p2[jx];
                if (reduce && U >= fourQ) U -= fourQ;
                uint64_t V = lo_mred_lazy(p2[jy], F, Q, qinv);
                p2[jx] = U + V; p2[jy] = U + twoQ - V;
            }
        }
    }
}
/* inttCoreLazy (:554-568): inttLazy (:570-606) == inttLazyUnrolled16 (:608-714)
 * arithmetic; invbutterfly (:164-171). */
static void intt_core_lazy(const uint64_t *p1, uint64_t *p2, int N, uint64_t Q, uint64_t qinv, const uint64_t *roots) {
    uint64_t twoQ = Q << 1, fourQ = Q << 2;
    int t = 1, h = N >> 1;
    for (int i = 0, j1 = 0; i < h; i++, j1 += 2 * t) {
        uint64_t F = roots[h + i];
        for (int jx = j1, jy = j1 + t; jx < j1 + t; jx++, jy++) {
            uint64_t U = p1[jx], V = p1[jy];
            uint64_t X = U + V; if (X >= twoQ) X -= twoQ;
            p2[jx] = X; p2[jy] = lo_mred_lazy(U + fourQ - V, F, Q, qinv);
        }
    }
    t <<= 1;
    for (int m = N >> 1; m > 1; m >>= 1) {
        h = m >> 1;
        for (int i = 0, j1 = 0; i < h; i++, j1 += 2 * t) {
            uint64_t F = roots[h + i];
            for (int jx = j1, jy = j1 + t; jx < j1 + t; jx++, jy++) {
                uint64_t U = p2[jx], V = p2[jy];
                uint64_t X = U + V; if (X >= twoQ) X -= twoQ;
                p2[jx] = X; p2[jy] = lo_mred_lazy(U + fourQ - V, F, Q, qinv);
            }
        }
        t <<= 1;
    }
}
/* nttConjugateInvariantLazy, ring/ntt.go:757-786 (the unrolled form :788-1090 is the same arithmetic
 * with a sparser 4q-correction schedule, i.e. other lazy representatives of the same residues) */
static void ntt_ci_core_lazy(const uint64_t *p1, uint64_t *p2, int N, uint64_t Q, uint64_t qinv, const uint64_t *roots) {
    uint64_t fourQ = 4 * Q, twoQ = 2 * Q;
    int t = N;
    uint64_t F = roots[1];
    for (int jx = 1, jy = N - 1; jx < (N >> 1); jx++, jy--) {
        uint64_t a = p1[jx], b = p1[jy];
        p2[jx] = a + twoQ - lo_mred_lazy(b, F, Q, qinv);
        p2[jy] = b + twoQ - lo_mred_lazy(a, F, Q, qinv);
    }
    p2[N >> 1] = p1[N >> 1] + twoQ - lo_mred_lazy(p1[N >> 1], F, Q, qinv);
    p2[0] = p1[0];
    for (int m = 2; m < 2 * N; m <<= 1) {
        t >>= 1;
        int h = m >> 1;
        for (int i = 0, j1 = 0; i < h; i++, j1 += 2 * t) {
            F = roots[m + i];
            for (int jx = j1, jy = j1 + t; jx < j1 + t; jx++, jy++) {
                uint64_t U = p2[jx];
                if (U >= fourQ) U -= fourQ;
                uint64_t V = lo_mred_lazy(p2[jy], F, Q, qinv);
                p2[jx] = U + V; p2[jy] = U + twoQ - V;
            }
        }
    }
}
/* inttConjugateInvariantLazy, ring/ntt.go:1104-1152 */
static void intt_ci_core_lazy(const uint64_t *p1, uint64_t *p2, int N, uint64_t Q, uint64_t qinv, const uint64_t *roots) {
    uint64_t twoQ = Q << 1, fourQ = Q << 2;
    int t = 1, h = N >> 1;
    for (int i = 0, j1 = 0; i < h; i++, j1 += 2 * t) {
        uint64_t F = roots[N + i];
        for (int jx = j1, jy = j1 + t; jx < j1 + t; jx++, jy++) {
            uint64_t U = p1[jx], V = p1[jy];
            uint64_t X = U + V; if (X >= twoQ) X -= twoQ;
            p2[jx] = X; p2[jy] = lo_mred_lazy(U + fourQ - V, F, Q, qinv);
        }
    }
    t <<= 1;
    for (int m = N >> 1; m > 1; m >>= 1) {
        h = m >> 1;
        for (int i = 0, j1 = 0; i < h; i++, j1 += 2 * t) {
            uint64_t F = roots[m + i];
            for (int jx = j1, jy = j1 + t; jx < j1 + t; jx++, jy++) {
                uint64_t U = p2[jx], V = p2[jy];
                uint64_t X = U + V; if (X >= twoQ) X -= twoQ;
                p2[jx] = X; p2[jy] = lo_mred_lazy(U + fourQ - V, F, Q, qinv);
            }
        }
        t <<= 1;
    }
    uint64_t F = roots[1];
    for (int jx = 1, jy = N - 1; jx < (N >> 1); jx++, jy--) {
        uint64_t a = p2[jx], b = p2[jy];
        p2[jx] = a + twoQ - lo_mred_lazy(b, F, Q, qinv);
        p2[jy] = b + twoQ - lo_mred_lazy(a, F, Q, qinv);
    }
    p2[N >> 1] = p2[N >> 1] + twoQ - lo_mred_lazy(p2[N >> 1], F, Q, qinv);
    p2[0] = lo_cred(p2[0] << 1, Q);
}
/* NTTStandard / NTTStandardLazy, ring/ntt.go:174-183; NTTConjugateInvariant[Lazy] :717-725 */
void lo_subring_ntt(const lo_subring *s, const uint64_t *p1, uint64_t *p2, int lazy) {
    if (s->nthroot == 4 * (uint64_t)s->N) {
        if (p1 == p2) {  /* the fold reads p1[N-j] after p2[j] is written: go through a copy like a caller would */
            uint64_t *tmp = (uint64_t *)pool_get((size_t)s->N * 8);
            memcpy(tmp, p1, (size_t)s->N * 8);
            ntt_ci_core_lazy(tmp, p2, s->N, s->q, s->qinv, s->roots_fwd);
            pool_put(tmp);
        } else ntt_ci_core_lazy(p1, p2, s->N, s->q, s->qinv, s->roots_fwd);
        if (!lazy) for (int i = 0; i < s->N; i++) p2[i] = lo_bred_add(p2[i], s->q, s->brc);
        return;
    }
    ntt_core_lazy(p1, p2, s->N, s->q, s->qinv, s->roots_fwd);
    if (!lazy) for (int i = 0; i < s->N; i++) p2[i] = lo_bred_add(p2[i], s->q, s->brc);
}
/* INTTStandard / INTTStandardLazy, ring/ntt.go:185-207 */
void lo_subring_intt(const lo_subring *s, const uint64_t *p1, uint64_t *p2, int lazy) {
    if (s->nthroot == 4 * (uint64_t)s->N) {  /* INTTConjugateInvariant[Lazy], ring/ntt.go:728-737 */
        intt_ci_core_lazy(p1, p2, s->N, s->q, s->qinv, s->roots_bwd);
        for (int i = 0; i < s->N; i++)
            p2[i] = lazy ? lo_mred_lazy(p2[i], s->ninv, s->q, s->qinv) : lo_mred(p2[i], s->ninv, s->q, s->qinv);
        return;
    }
    intt_core_lazy(p1, p2, s->N, s->q, s->qinv, s->roots_bwd);
    if (s->N < 16 && lazy) { for (int i = 0; i < s->N; i++) p2[i] = lo_mred_lazy(p2[i], s->ninv, s->q, s->qinv); }
    else { for (int i = 0; i < s->N; i++) p2[i] = lo_mred(p2[i], s->ninv, s->q, s->qinv); }
}
/* Ring.NTT/NTTLazy/INTT/INTTLazy, ring/ntt.go:127-152 */
void lo_ntt(const lo_ring *r, int level, const uint64_t *p1, uint64_t *p2) {
    for (int i = 0; i <= level; i++) lo_subring_ntt(r->s[i], p1 + (size_t)i * r->N, p2 + (size_t)i * r->N, 0);
}
void lo_ntt_lazy(const lo_ring *r, int level, const uint64_t *p1, uint64_t *p2) {
    for (int i = 0; i <= level; i++) lo_subring_ntt(r->s[i], p1 + (size_t)i * r->N, p2 + (size_t)i * r->N, 1);
}
void lo_intt(const lo_ring *r, int level, const uint64_t *p1, uint64_t *p2) {
    for (int i = 0; i <= level; i++) lo_subring_intt(r->s[i], p1 + (size_t)i * r->N, p2 + (size_t)i * r->N, 0);
}
void lo_intt_lazy(const lo_ring *r, int level, const uint64_t *p1, uint64_t *p2) {
    for (int i = 0; i <= level; i++) lo_subring_intt(r->s[i], p1 + (size_t)i * r->N, p2 + (size_t)i * r->N, 1);
}

/* ========================================================================== */
/* Coefficient-wise ops: ring/vec_ops.go, per limb via ring/operations.go       */
/* ========================================================================== */

void lo_binop(const lo_ring *r, int level, int op, const uint64_t *p1, const uint64_t *p2, uint64_t *p3) {
    int N = r->N;
    for (int l = 0; l <= level; l++) {
        const lo_subring *s = r->s[l];
        uint64_t q = s->q, qinv = s->qinv, twoq = q << 1;
        const uint64_t *brc = s->brc;
        const uint64_t *x = p1 + (size_t)l * N, *y = p2 + (size_t)l * N;
        uint64_t *z = p3 + (size_t)l * N;
        for (int j = 0; j < N; j++) {
            switch (op) {
            case LO_ADD: z[j] = lo_cred(x[j] + y[j], q); break;                               /* vec_ops.go:20 */
            case LO_ADD_LAZY: z[j] = x[j] + y[j]; break;                                       /* :44 */
            case LO_SUB: z[j] = lo_cred((x[j] + q) - y[j], q); break;                          /* :68 */
            case LO_SUB_LAZY: z[j] = x[j] + q - y[j]; break;                                   /* :92 */
            case LO_MUL_BARRETT: z[j] = lo_bred(x[j], y[j], q, brc); break;                    /* :230 */
            case LO_MUL_BARRETT_LAZY: z[j] = lo_bred_lazy(x[j], y[j], q, brc); break;          /* :254 */
            case LO_MUL_BARRETT_THEN_ADD: z[j] = lo_cred(z[j] + lo_bred(x[j], y[j], q, brc), q); break; /* :278 */
            case LO_MUL_BARRETT_THEN_ADD_LAZY: z[j] += lo_bred(x[j], y[j], q, brc); break;     /* :302 */
            case LO_MUL_MONT: z[j] = lo_mred(x[j], y[j], q, qinv); break;                      /* :325 */
            case LO_MUL_MONT_LAZY: z[j] = lo_mred_lazy(x[j], y[j], q, qinv); break;            /* :349 */
            case LO_MUL_MONT_LAZY_THEN_NEG: z[j] = twoq - lo_mred_lazy(x[j], y[j], q, qinv); break; /* :518 */
            case LO_MUL_MONT_THEN_ADD: z[j] = lo_cred(z[j] + lo_mred(x[j], y[j], q, qinv), q); break; /* :372 */
            case LO_MUL_MONT_THEN_ADD_LAZY: z[j] += lo_mred(x[j], y[j], q, qinv); break;       /* :396 */
            case LO_MUL_MONT_LAZY_THEN_ADD_LAZY: z[j] += lo_mred_lazy(x[j], y[j], q, qinv); break; /* :420 */
            case LO_MUL_MONT_THEN_SUB: z[j] = lo_cred(z[j] + (q - lo_mred(x[j], y[j], q, qinv)), q); break; /* :444 */
            case LO_MUL_MONT_THEN_SUB_LAZY: z[j] += (q - lo_mred(x[j], y[j], q, qinv)); break; /* :468 */
            case LO_MUL_MONT_LAZY_THEN_SUB_LAZY: z[j] += twoq - lo_mred_lazy(x[j], y[j], q, qinv); break; /* :493 */
            default: break;
            }
        }
    }
}
void lo_unop(const lo_ring *r, int level, int op, const uint64_t *p1, uint64_t *p2) {
    int N = r->N;
    for (int l = 0; l <= level; l++) {
        const lo_subring *s = r->s[l];
        uint64_t q = s->q, qinv = s->qinv;
        const uint64_t *brc = s->brc;
        const uint64_t *x = p1 + (size_t)l * N;
        uint64_t *z = p2 + (size_t)l * N;
        for (int j = 0; j < N; j++) {
            switch (op) {
            case LO_NEG: z[j] = q - x[j]; break;                                 /* vec_ops.go:114 */
            case LO_REDUCE: z[j] = lo_bred_add(x[j], q, brc); break;             /* :136 */
            case LO_REDUCE_LAZY: z[j] = lo_bred_add_lazy(x[j], q, brc); break;   /* :158 */
            case LO_MFORM: z[j] = lo_mform(x[j], q, brc); break;                 /* :789 */
            case LO_MFORM_LAZY: z[j] = lo_mform_lazy(x[j], q, brc); break;       /* :811 */
            case LO_IMFORM: z[j] = lo_imform(x[j], q, qinv); break;              /* :833 */
            default: break;
            }
        }
    }
}
/* AddScalar/SubScalar/MulScalar/MulScalarThenAdd/MulScalarThenSub, ring/operations.go:151-229 */
void lo_scalarop(const lo_ring *r, int level, int op, const uint64_t *p1, uint64_t scalar, uint64_t *p2) {
    int N = r->N;
    for (int l = 0; l <= level; l++) {
        const lo_subring *s = r->s[l];
        uint64_t q = s->q, qinv = s->qinv;
        const uint64_t *x = p1 + (size_t)l * N;
        uint64_t *z = p2 + (size_t)l * N;
        uint64_t sm = 0;
        if (op == LO_MUL_SCALAR || op == LO_MUL_SCALAR_THEN_ADD) sm = lo_mform(scalar, q, s->brc);
        if (op == LO_MUL_SCALAR_THEN_SUB) sm = lo_mform(q - lo_bred_add(scalar, q, s->brc), q, s->brc);
        for (int j = 0; j < N; j++) {
            switch (op) {
            case LO_ADD_SCALAR: z[j] = lo_cred(x[j] + scalar, q); break;          /* vec_ops.go:586 */
            case LO_SUB_SCALAR: z[j] = lo_cred(x[j] + q - scalar, q); break;      /* :653 */
            case LO_MUL_SCALAR: z[j] = lo_mred(x[j], sm, q, qinv); break;         /* :675 */
            case LO_MUL_SCALAR_THEN_ADD:
            case LO_MUL_SCALAR_THEN_SUB: z[j] = lo_cred(z[j] + lo_mred(x[j], sm, q, qinv), q); break; /* :719 */
            default: break;
            }
        }
    }
}
/* MulRNSScalarMontgomery, ring/operations.go:216-220 */
void lo_mul_rns_scalar_montgomery(const lo_ring *r, int level, const uint64_t *p1, const uint64_t *scalar, uint64_t *p2) {
    int N = r->N;
    for (int l = 0; l <= level; l++) {
        const lo_subring *s = r->s[l];
        for (int j = 0; j < N; j++) p2[(size_t)l * N + j] = lo_mred(p1[(size_t)l * N + j], scalar[l], s->q, s->qinv);
    }
}
/* big.Int.Mod(scalar, q) for a little-endian multi-word scalar */
static uint64_t words_mod(const uint64_t *w, int n, uint64_t q) {
    u128 rem = 0;
    for (int i = n - 1; i >= 0; i--) rem = ((rem << 64) | w[i]) % q;
    return (uint64_t)rem;
}
/* AddScalarBigint / SubScalarBigint / MulScalarBigint, ring/operations.go:158,193,231 */
void lo_add_scalar_bigint(const lo_ring *r, int level, const uint64_t *p1, const uint64_t *words, int nwords, uint64_t *p2) {
    int N = r->N;
    for (int l = 0; l <= level; l++) {
        uint64_t q = r->s[l]->q, sc = words_mod(words, nwords, q);
        for (int j = 0; j < N; j++) p2[(size_t)l * N + j] = lo_cred(p1[(size_t)l * N + j] + sc, q);
    }
}
void lo_sub_scalar_bigint(const lo_ring *r, int level, const uint64_t *p1, const uint64_t *words, int nwords, uint64_t *p2) {
    int N = r->N;
    for (int l = 0; l <= level; l++) {
        uint64_t q = r->s[l]->q, sc = words_mod(words, nwords, q);
        for (int j = 0; j < N; j++) p2[(size_t)l * N + j] = lo_cred(p1[(size_t)l * N + j] + q - sc, q);
    }
}
void lo_mul_scalar_bigint(const lo_ring *r, int level, const uint64_t *p1, const uint64_t *words, int nwords, uint64_t *p2) {
    int N = r->N;
    for (int l = 0; l <= level; l++) {
        const lo_subring *s = r->s[l];
        uint64_t sm = lo_mform(words_mod(words, nwords, s->q), s->q, s->brc);
        for (int j = 0; j < N; j++) p2[(size_t)l * N + j] = lo_mred(p1[(size_t)l * N + j], sm, s->q, s->qinv);
    }
}

/* ========================================================================== */
/* Rescale: ring/scaling.go                                                     */
/* ========================================================================== */

/* subthenmulscalarmontgomeryTwoModulusvec, ring/vec_ops.go:752-775: z = MRed(2q - y + x, s) */
static void sub_then_mul_scalar_mont_2q(const lo_subring *s, const uint64_t *x, const uint64_t *y, uint64_t sc, uint64_t *z) {
    uint64_t twoq = s->q << 1;
    for (int j = 0; j < s->N; j++) z[j] = lo_mred(twoq - y[j] + x[j], sc, s->q, s->qinv);
}
/* DivFloorByLastModulusNTT, ring/scaling.go:6-23 */
void lo_div_floor_by_last_modulus_ntt(const lo_ring *r, int level, const uint64_t *p0, uint64_t *p1) {
    int N = r->N;
    uint64_t *b0 = (uint64_t *)pool_get(N * 8), *b1 = (uint64_t *)pool_get(N * 8);
    lo_subring_intt(r->s[level], p0 + (size_t)level * N, b0, 1);
    for (int i = 0; i < level; i++) {
        lo_subring_ntt(r->s[i], b0, b1, 1);
        sub_then_mul_scalar_mont_2q(r->s[i], b1, p0 + (size_t)i * N, r->rescale[level - 1][i], p1 + (size_t)i * N);
    }
    pool_put(b0); pool_put(b1);
}
/* DivFloorByLastModulus, :26-34 */
void lo_div_floor_by_last_modulus(const lo_ring *r, int level, const uint64_t *p0, uint64_t *p1) {
    int N = r->N;
    for (int i = 0; i < level; i++)
        sub_then_mul_scalar_mont_2q(r->s[i], p0 + (size_t)level * N, p0 + (size_t)i * N, r->rescale[level - 1][i], p1 + (size_t)i * N);
}
/* DivRoundByLastModulusNTT, :101-122 */
void lo_div_round_by_last_modulus_ntt(const lo_ring *r, int level, const uint64_t *p0, uint64_t *p1) {
    int N = r->N;
    uint64_t *b0 = (uint64_t *)pool_get(N * 8), *b1 = (uint64_t *)pool_get(N * 8);
    const lo_subring *sl = r->s[level];
    lo_subring_intt(sl, p0 + (size_t)level * N, b0, 1);
    uint64_t phalf = (sl->q - 1) >> 1;
    for (int j = 0; j < N; j++) b0[j] = lo_cred(b0[j] + phalf, sl->q);
    for (int i = 0; i < level; i++) {
        const lo_subring *s = r->s[i];
        uint64_t sc = s->q - lo_bred_add(phalf, s->q, s->brc);
        for (int j = 0; j < N; j++) b1[j] = b0[j] + sc;                       /* AddScalarLazy */
        lo_subring_ntt(s, b1, b1, 1);
        sub_then_mul_scalar_mont_2q(s, b1, p0 + (size_t)i * N, r->rescale[level - 1][i], p1 + (size_t)i * N);
    }
    pool_put(b0); pool_put(b1);
}
/* DivRoundByLastModulus, :126-144 */
void lo_div_round_by_last_modulus(const lo_ring *r, int level, const uint64_t *p0, uint64_t *p1) {
    int N = r->N;
    uint64_t *b0 = (uint64_t *)pool_get(N * 8);
    const lo_subring *sl = r->s[level];
    uint64_t phalf = (sl->q - 1) >> 1;
    for (int j = 0; j < N; j++) b0[j] = lo_cred(p0[(size_t)level * N + j] + phalf, sl->q);
    for (int i = 0; i < level; i++) {
        const lo_subring *s = r->s[i];
        uint64_t sc = s->q - lo_bred_add(phalf, s->q, s->brc);
        uint64_t twoq = s->q << 1, rc = r->rescale[level - 1][i];
        for (int j = 0; j < N; j++) {
            uint64_t b1 = sc + twoq - p0[(size_t)i * N + j];                    /* vec_ops.go:631 */
            p1[(size_t)i * N + j] = lo_mred(b0[j] + b1, rc, s->q, s->qinv);     /* vec_ops.go:542 */
        }
    }
    pool_put(b0);
}
/* DivRoundByLastModulusManyNTT, :148-174 */
void lo_div_round_by_last_modulus_many_ntt(const lo_ring *r, int level, int nb, const uint64_t *p0, uint64_t *p1) {
    int N = r->N;
    if (nb == 0) { if (p0 != p1) memcpy(p1, p0, (size_t)(level + 1) * N * 8); return; }
    if (nb > 1) {
        uint64_t *buff = (uint64_t *)pool_get((size_t)(level + 1) * N * 8);
        lo_intt(r, level, p0, buff);
        int lv = level;
        for (int i = 0; i < nb; i++) { lo_div_round_by_last_modulus(r, lv, buff, buff); lv--; }
        lo_ntt(r, lv, buff, p1);
        pool_put(buff);
    } else {
        lo_div_round_by_last_modulus_ntt(r, level, p0, p1);
    }
}
/* DivRoundByLastModulusMany, :177-212 */
void lo_div_round_by_last_modulus_many(const lo_ring *r, int level, int nb, const uint64_t *p0, uint64_t *p1) {
    int N = r->N;
    if (nb == 0) { if (p0 != p1) memcpy(p1, p0, (size_t)(level + 1) * N * 8); return; }
    if (nb > 1) {
        uint64_t *buff = (uint64_t *)pool_get((size_t)(level + 1) * N * 8);
        int lv = level;
        lo_div_round_by_last_modulus(r, lv, p0, buff); lv--;
        for (int i = 1; i < nb; i++) {
            if (i == nb - 1) lo_div_round_by_last_modulus(r, lv, buff, p1);
            else lo_div_round_by_last_modulus(r, lv, buff, buff);
            lv--;
        }
        pool_put(buff);
    } else {
        lo_div_round_by_last_modulus(r, level, p0, p1);
    }
}
/* DivFloorByLastModulusManyNTT, :37-62 */
void lo_div_floor_by_last_modulus_many_ntt(const lo_ring *r, int level, int nb, const uint64_t *p0, uint64_t *p1) {
    int N = r->N;
    if (nb == 0) { if (p0 != p1) memcpy(p1, p0, (size_t)(level + 1) * N * 8); return; }
    uint64_t *buff = (uint64_t *)pool_get((size_t)(level + 1) * N * 8);
    lo_intt(r, level, p0, buff);
    int lv = level;
    for (int i = 0; i < nb; i++) { lo_div_floor_by_last_modulus(r, lv, buff, buff); lv--; }
    lo_ntt(r, lv, buff, p1);
    pool_put(buff);
}
/* DivFloorByLastModulusMany, :65-99 */
void lo_div_floor_by_last_modulus_many(const lo_ring *r, int level, int nb, const uint64_t *p0, uint64_t *p1) {
    int N = r->N;
    if (nb == 0) { if (p0 != p1) memcpy(p1, p0, (size_t)(level + 1) * N * 8); return; }
    if (nb > 1) {
        uint64_t *buff = (uint64_t *)pool_get((size_t)(level + 1) * N * 8);
        int lv = level;
        lo_div_floor_by_last_modulus(r, lv, p0, buff); lv--;
        for (int i = 1; i < nb; i++) {
            if (i == nb - 1) lo_div_floor_by_last_modulus(r, lv, buff, p1);
            else lo_div_floor_by_last_modulus(r, lv, buff, buff);
            lv--;
        }
        pool_put(buff);
    } else {
        lo_div_floor_by_last_modulus(r, level, p0, p1);
    }
}

/* ========================================================================== */
/* Automorphism: ring/automorphism.go                                           */
/* ========================================================================== */

/* AutomorphismNTTIndex, :12-34 */
void lo_automorphism_ntt_index(int N, uint64_t nthroot, uint64_t galel, uint64_t *index) {
    int lognth = bitlen64(nthroot - 1) - 1;
    uint64_t mask = nthroot - 1;
    for (int i = 0; i < N; i++) {
        uint64_t tmp1 = 2 * bitrev64((uint64_t)i, lognth) + 1;
        uint64_t tmp2 = (((galel * tmp1) & mask) - 1) >> 1;
        index[i] = bitrev64(tmp2, lognth);
    }
}
/* AutomorphismNTTWithIndex, :50-77 */
void lo_automorphism_ntt_with_index(const lo_ring *r, int level, const uint64_t *pin, const uint64_t *index, uint64_t *pout) {
    int N = r->N;
    for (int i = 0; i <= level; i++)
        for (int j = 0; j < N; j++) pout[(size_t)i * N + j] = pin[(size_t)i * N + index[j]];
}
/* AutomorphismNTTWithIndexThenAddLazy, :82-109 */
void lo_automorphism_ntt_with_index_then_add_lazy(const lo_ring *r, int level, const uint64_t *pin, const uint64_t *index, uint64_t *pout) {
    int N = r->N;
    for (int i = 0; i <= level; i++)
        for (int j = 0; j < N; j++) pout[(size_t)i * N + j] += pin[(size_t)i * N + index[j]];
}
/* Automorphism (coefficient domain), :113-176: conjugate-invariant branch :122-151, standard ring :153-174 */
void lo_automorphism(const lo_ring *r, int level, const uint64_t *pin, uint64_t galel, uint64_t *pout) {
    uint64_t N = (uint64_t)r->N, mask = N - 1;
    if (r->s[0]->nthroot == 4 * N) {  /* Z[X + X^-1]/(X^2N + 1): i runs over [0, 2N), only images below N are kept */
        mask = 2 * N - 1;
        int logN2 = bitlen64(mask);
        for (uint64_t i = 0; i < 2 * N; i++) {
            uint64_t raw = i * galel, index = raw & mask, tmp = (raw >> logN2) & 1;
            if (index < N) {
                uint64_t idx = i;
                if (idx >= N) { idx = 2 * N - idx; tmp ^= 1; }
                for (int j = 0; j <= level; j++) {
                    uint64_t c = pin[(size_t)j * N + idx];
                    pout[(size_t)j * N + index] = (c * (tmp ^ 1)) | ((r->s[j]->q - c) * tmp);
                }
            }
        }
        return;
    }
    int logN = bitlen64(mask);
    for (uint64_t i = 0; i < N; i++) {
        uint64_t raw = i * galel, index = raw & mask, tmp = (raw >> logN) & 1;
        for (int j = 0; j <= level; j++) {
            uint64_t c = pin[(size_t)j * N + i];
            pout[(size_t)j * N + index] = (c * (tmp ^ 1)) | ((r->s[j]->q - c) * tmp);
        }
    }
}

/* ========================================================================== */
/* Basis extension: ring/basis_extension.go                                     */
/* ========================================================================== */

/* GenModUpConstants, :101-172 */
lo_modup_constants *lo_gen_modup_constants(const uint64_t *Q, int nq, const uint64_t *P, int np) {
    lo_modup_constants *c = (lo_modup_constants *)calloc(1, sizeof *c);
    c->nq = nq; c->np = np;
    c->qoverqiinvqi = (uint64_t *)calloc(nq, 8);
    c->qoverqimodp = (uint64_t *)calloc((size_t)np * nq, 8);
    c->vtimesqmodp = (uint64_t *)calloc((size_t)np * (nq + 1), 8);
    uint64_t (*bredQ)[2] = (uint64_t (*)[2])malloc(sizeof(uint64_t[2]) * nq);
    uint64_t (*bredP)[2] = (uint64_t (*)[2])malloc(sizeof(uint64_t[2]) * (np ? np : 1));
    uint64_t *mredQ = (uint64_t *)malloc(8 * nq), *mredP = (uint64_t *)malloc(8 * (np ? np : 1));
    for (int i = 0; i < nq; i++) { lo_gen_bred_constant(Q[i], bredQ[i]); mredQ[i] = lo_gen_mred_constant(Q[i]); }
    for (int i = 0; i < np; i++) { lo_gen_bred_constant(P[i], bredP[i]); mredP[i] = lo_gen_mred_constant(P[i]); }
    for (int i = 0; i < nq; i++) {
        uint64_t qi = Q[i];
        uint64_t qistar = lo_mform(1, qi, bredQ[i]);
        for (int j = 0; j < nq; j++)
            if (j != i) qistar = lo_mred(qistar, lo_mform(Q[j], qi, bredQ[i]), qi, mredQ[i]);
        c->qoverqiinvqi[i] = modexp_montgomery(qistar, qi - 2, qi, mredQ[i], bredQ[i]);
        for (int j = 0; j < np; j++) {
            uint64_t pj = P[j];
            qistar = 1;
            for (int u = 0; u < nq; u++)
                if (u != i) qistar = lo_mred(qistar, lo_mform(Q[u], pj, bredP[j]), pj, mredP[j]);
            c->qoverqimodp[(size_t)j * nq + i] = lo_mform(qistar, pj, bredP[j]);
        }
    }
    for (int j = 0; j < np; j++) {
        uint64_t pj = P[j], qmodp = 1;
        for (int i = 0; i < nq; i++) qmodp = lo_mred(qmodp, lo_mform(Q[i], pj, bredP[j]), pj, mredP[j]);
        uint64_t v = pj - qmodp;
        uint64_t *row = c->vtimesqmodp + (size_t)j * (nq + 1);
        row[0] = 0;
        for (int i = 1; i < nq + 1; i++) row[i] = lo_cred(row[i - 1] + v, pj);
    }
    free(bredQ); free(bredP); free(mredQ); free(mredP);
    return c;
}
void lo_modup_constants_free(lo_modup_constants *c) {
    if (!c) return;
    free(c->qoverqiinvqi); free(c->qoverqimodp); free(c->vtimesqmodp); free(c);
}

/* genmodDownConstants, :25-49 */
static uint64_t **gen_moddown_constants(const lo_ring *ringQ, const lo_ring *ringP) {
    uint64_t **c = (uint64_t **)calloc(ringP->nmod, sizeof(uint64_t *));
    for (int j = 0; j < ringP->nmod; j++) {
        uint64_t pj = ringP->s[j]->q;
        c[j] = (uint64_t *)malloc(8 * ringQ->nmod);
        for (int i = 0; i < ringQ->nmod; i++) {
            const lo_subring *sq = ringQ->s[i];
            uint64_t v = lo_modexp(pj, sq->q - 2, sq->q);
            v = lo_mform(v, sq->q, sq->brc);
            if (j > 0) v = lo_mred(v, c[j - 1][i], sq->q, sq->qinv);
            c[j][i] = v;
        }
    }
    return c;
}

static void ring_moduli(const lo_ring *r, uint64_t *out) { for (int i = 0; i < r->nmod; i++) out[i] = r->s[i]->q; }

/* NewBasisExtender, :52-89 */
lo_basis_extender *lo_basis_extender_new(lo_ring *ringQ, lo_ring *ringP) {
    lo_basis_extender *be = (lo_basis_extender *)calloc(1, sizeof *be);
    be->ringQ = ringQ; be->ringP = ringP; be->nQ = ringQ->nmod; be->nP = ringP->nmod;
    uint64_t *Q = (uint64_t *)malloc(8 * ringQ->nmod), *P = (uint64_t *)malloc(8 * ringP->nmod);
    ring_moduli(ringQ, Q); ring_moduli(ringP, P);
    be->qtop = (lo_modup_constants **)calloc(ringQ->nmod, sizeof(void *));
    for (int i = 0; i < ringQ->nmod; i++) be->qtop[i] = lo_gen_modup_constants(Q, i + 1, P, ringP->nmod);
    be->ptoq = (lo_modup_constants **)calloc(ringP->nmod, sizeof(void *));
    for (int i = 0; i < ringP->nmod; i++) be->ptoq[i] = lo_gen_modup_constants(P, i + 1, Q, ringQ->nmod);
    be->moddown_ptoq = gen_moddown_constants(ringQ, ringP);
    be->moddown_qtop = gen_moddown_constants(ringP, ringQ);
    free(Q); free(P);
    return be;
}
void lo_basis_extender_free(lo_basis_extender *be) {
    if (!be) return;
    for (int i = 0; i < be->nQ; i++) { lo_modup_constants_free(be->qtop[i]); free(be->moddown_qtop[i]); }
    for (int i = 0; i < be->nP; i++) { lo_modup_constants_free(be->ptoq[i]); free(be->moddown_ptoq[i]); }
    free(be->qtop); free(be->ptoq); free(be->moddown_ptoq); free(be->moddown_qtop); free(be);
}

/* ModUpExact (:282-308) = reconstructRNS (:550-594) + multSum (:597-673) per
 * coefficient.  p1: source limbs [nsrc][N] (row stride N); p2: targets.
 * The float64 accumulation runs in increasing source-limb order with one
 * rounding per division and per addition, exactly as the Go code.           */
static void modup_exact(const uint64_t *p1, int nsrc, uint64_t *p2, int ndst, int N,
                        const lo_ring *ringQ, const lo_ring *ringP, const lo_modup_constants *muc) {
    uint64_t y[32];
    for (int x = 0; x < N; x++) {
        double vi = 0.0;
        for (int i = 0; i < nsrc; i++) {
            const lo_subring *s = ringQ->s[i];
            y[i] = lo_mred(p1[(size_t)i * N + x], muc->qoverqiinvqi[i], s->q, s->qinv);
            vi += (double)y[i] / (double)s->q;
        }
        uint64_t v = (uint64_t)vi;
        for (int j = 0; j < ndst; j++) {
            const lo_subring *sp = ringP->s[j];
            const uint64_t *qq = muc->qoverqimodp + (size_t)j * muc->nq;
            u128 acc = (u128)y[0] * qq[0];
            for (int i = 1; i < nsrc; i++) acc += (u128)y[i] * qq[i];
            uint64_t rlo = (uint64_t)acc, rhi = (uint64_t)(acc >> 64);
            uint64_t hhi = mulhi64(rlo * sp->qinv, sp->q);
            p2[(size_t)j * N + x] = rhi - hhi + sp->q + muc->vtimesqmodp[(size_t)j * (muc->nq + 1) + v];
        }
    }
}

/* little-endian multiword product of moduli (ring.ModulusAtLevel) then >> 1 */
static int moduli_product_half(const uint64_t *mods, int n, uint64_t *words /* >= n */) {
    int nw = 1; words[0] = 1;
    for (int i = 0; i < n; i++) {
        uint64_t carry = 0;
        for (int k = 0; k < nw; k++) {
            u128 t = (u128)words[k] * mods[i] + carry;
            words[k] = (uint64_t)t; carry = (uint64_t)(t >> 64);
        }
        if (carry) words[nw++] = carry;
    }
    for (int k = 0; k < nw; k++) {
        words[k] >>= 1;
        if (k + 1 < nw) words[k] |= words[k + 1] << 63;
    }
    return nw;
}

/* ModUpQtoP, :177-190 */
void lo_modup_q_to_p(const lo_basis_extender *be, int levelQ, int levelP, const uint64_t *polQ, uint64_t *polP) {
    int N = be->ringQ->N;
    uint64_t mods[64], half[64];
    ring_moduli(be->ringQ, mods);
    int nw = moduli_product_half(mods, levelQ + 1, half);
    uint64_t *buff = (uint64_t *)pool_get((size_t)(levelQ + 1) * N * 8);
    lo_add_scalar_bigint(be->ringQ, levelQ, polQ, half, nw, buff);
    modup_exact(buff, levelQ + 1, polP, levelP + 1, N, be->ringQ, be->ringP, be->qtop[levelQ]);
    lo_sub_scalar_bigint(be->ringP, levelP, polP, half, nw, polP);
    pool_put(buff);
}
/* ModUpPtoQ, :195-210 */
void lo_modup_p_to_q(const lo_basis_extender *be, int levelP, int levelQ, const uint64_t *polP, uint64_t *polQ) {
    int N = be->ringQ->N;
    uint64_t mods[64], half[64];
    ring_moduli(be->ringP, mods);
    int nw = moduli_product_half(mods, levelP + 1, half);
    uint64_t *buff = (uint64_t *)pool_get((size_t)(levelP + 1) * N * 8);
    lo_add_scalar_bigint(be->ringP, levelP, polP, half, nw, buff);
    modup_exact(buff, levelP + 1, polQ, levelQ + 1, N, be->ringP, be->ringQ, be->ptoq[levelP]);
    lo_sub_scalar_bigint(be->ringQ, levelQ, polQ, half, nw, polQ);
    pool_put(buff);
}
/* ModDownQPtoQ, :215-230 */
void lo_moddown_qp_to_q(const lo_basis_extender *be, int levelQ, int levelP, const uint64_t *p1Q, const uint64_t *p1P, uint64_t *p2Q) {
    int N = be->ringQ->N;
    uint64_t *buffQ = (uint64_t *)pool_get((size_t)(levelQ + 1) * N * 8);
    lo_modup_p_to_q(be, levelP, levelQ, p1P, buffQ);
    for (int i = 0; i <= levelQ; i++) {
        const lo_subring *s = be->ringQ->s[i];
        sub_then_mul_scalar_mont_2q(s, buffQ + (size_t)i * N, p1Q + (size_t)i * N, s->q - be->moddown_ptoq[levelP][i], p2Q + (size_t)i * N);
    }
    pool_put(buffQ);
}
/* ModDownQPtoQNTT, :235-256 */
void lo_moddown_qp_to_q_ntt(const lo_basis_extender *be, int levelQ, int levelP, const uint64_t *p1Q, const uint64_t *p1P, uint64_t *p2Q) {
    int N = be->ringQ->N;
    uint64_t *buffP = (uint64_t *)pool_get((size_t)(levelP + 1) * N * 8);
    uint64_t *buffQ = (uint64_t *)pool_get((size_t)(levelQ + 1) * N * 8);
    lo_intt_lazy(be->ringP, levelP, p1P, buffP);
    lo_modup_p_to_q(be, levelP, levelQ, buffP, buffQ);
    lo_ntt_lazy(be->ringQ, levelQ, buffQ, buffQ);
    for (int i = 0; i <= levelQ; i++) {
        const lo_subring *s = be->ringQ->s[i];
        sub_then_mul_scalar_mont_2q(s, buffQ + (size_t)i * N, p1Q + (size_t)i * N, s->q - be->moddown_ptoq[levelP][i], p2Q + (size_t)i * N);
    }
    pool_put(buffP); pool_put(buffQ);
}
/* ModDownQPtoP, :262-277 */
void lo_moddown_qp_to_p(const lo_basis_extender *be, int levelQ, int levelP, const uint64_t *p1Q, const uint64_t *p1P, uint64_t *p2P) {
    int N = be->ringQ->N;
    uint64_t *buffP = (uint64_t *)pool_get((size_t)(levelP + 1) * N * 8);
    lo_modup_q_to_p(be, levelQ, levelP, p1Q, buffP);
    for (int i = 0; i <= levelP; i++) {
        const lo_subring *s = be->ringP->s[i];
        sub_then_mul_scalar_mont_2q(s, buffP + (size_t)i * N, p1P + (size_t)i * N, s->q - be->moddown_qtop[levelQ][i], p2P + (size_t)i * N);
    }
    pool_put(buffP);
}

/* NewDecomposer, :320-377 */
lo_decomposer *lo_decomposer_new(lo_ring *ringQ, lo_ring *ringP) {
    lo_decomposer *d = (lo_decomposer *)calloc(1, sizeof *d);
    d->ringQ = ringQ; d->ringP = ringP;
    if (!ringP) return d;
    int nQ = ringQ->nmod, nP = ringP->nmod;
    uint64_t Q[64], P[64];
    ring_moduli(ringQ, Q); ring_moduli(ringP, P);
    d->nlvlP = nP - 1;                                   /* ringP.MaxLevel() */
    d->ndigits = (int *)calloc(d->nlvlP ? d->nlvlP : 1, sizeof(int));
    d->nconst = (int **)calloc(d->nlvlP ? d->nlvlP : 1, sizeof(int *));
    d->muc = (lo_modup_constants ****)calloc(d->nlvlP ? d->nlvlP : 1, sizeof(void *));
    for (int lvlP = 0; lvlP < d->nlvlP; lvlP++) {
        int nbPi = lvlP + 2;
        int nd = (nQ + nbPi - 1) / nbPi;                 /* ceil(len(Q)/nbPi) */
        d->ndigits[lvlP] = nd;
        d->nconst[lvlP] = (int *)calloc(nd, sizeof(int));
        d->muc[lvlP] = (lo_modup_constants ***)calloc(nd, sizeof(void *));
        for (int i = 0; i < nd; i++) {
            int xnb = nbPi;
            if (i == nd - 1 && nQ % nbPi != 0) xnb = nQ % nbPi;
            d->nconst[lvlP][i] = xnb - 1;
            d->muc[lvlP][i] = (lo_modup_constants **)calloc(xnb > 1 ? xnb - 1 : 1, sizeof(void *));
            for (int j = 0; j < xnb - 1; j++) {
                uint64_t Qi[64], Pi[128];
                for (int k = 0; k < j + 2; k++) Qi[k] = Q[i * nbPi + k];
                for (int k = 0; k < nQ; k++) Pi[k] = Q[k];
                for (int k = 0; k < nbPi; k++) Pi[nQ + k] = P[k];
                d->muc[lvlP][i][j] = lo_gen_modup_constants(Qi, j + 2, Pi, nQ + nbPi);
            }
        }
    }
    return d;
}
void lo_decomposer_free(lo_decomposer *d) {
    if (!d) return;
    for (int l = 0; l < d->nlvlP; l++) {
        for (int i = 0; i < d->ndigits[l]; i++) {
            for (int j = 0; j < d->nconst[l][i]; j++) lo_modup_constants_free(d->muc[l][i][j]);
            free(d->muc[l][i]);
        }
        free(d->muc[l]); free(d->nconst[l]);
    }
    free(d->muc); free(d->nconst); free(d->ndigits); free(d);
}

/* DecomposeAndSplit, :381-502 (+ reconstructRNSCentered :504-548, multSum :597-673).
 * p0Q: [levelQ+1][N] coefficient domain; p1Q: [levelQ+1][N]; p1P: [levelP+1][N].
 * As in the reference, the digit's own limbs of p1Q are NOT produced by the
 * extension (the reference leaves scratch there and DecomposeSingleNTT then
 * overwrites them); here they are left untouched. */
void lo_decompose_and_split(const lo_decomposer *d, int levelQ, int levelP, int nbPi, int digit,
                            const uint64_t *p0Q, uint64_t *p1Q, uint64_t *p1P) {
    const lo_ring *ringQ = d->ringQ, *ringP = d->ringP;
    int N = ringQ->N;
    int lvlQStart = digit * nbPi;
    int decompLvl;
    if (levelQ > nbPi * (digit + 1) - 1) decompLvl = nbPi - 2;
    else decompLvl = (levelQ % nbPi) - 1;

    if (decompLvl < 0) {
        uint64_t qs = ringQ->s[lvlQStart]->q;
        for (int j = 0; j < N; j++) {
            uint64_t coeff = p0Q[(size_t)lvlQStart * N + j];
            uint64_t pos = 1, neg = 0;
            if (coeff >= (qs >> 1)) { coeff = qs - coeff; pos = 0; neg = 1; }
            for (int i = 0; i <= levelQ; i++) {
                const lo_subring *s = ringQ->s[i];
                uint64_t tmp = lo_bred_add(coeff, s->q, s->brc);
                p1Q[(size_t)i * N + j] = tmp * pos + (s->q - tmp) * neg;
            }
            for (int i = 0; i <= levelP; i++) {
                const lo_subring *s = ringP->s[i];
                uint64_t tmp = lo_bred_add(coeff, s->q, s->brc);
                p1P[(size_t)i * N + j] = tmp * pos + (s->q - tmp) * neg;
            }
        }
        return;
    }

    int p0idxst = digit * nbPi, p0idxed = p0idxst + nbPi;
    if (p0idxed > levelQ + 1) p0idxed = levelQ + 1;
    const lo_modup_constants *muc = d->muc[nbPi - 2][digit][decompLvl];
    int nQ = ringQ->nmod;

    uint64_t mods[64], half[64], halfmod[64];
    for (int i = p0idxst; i < p0idxed; i++) mods[i - p0idxst] = ringQ->s[i]->q;
    int nw = moduli_product_half(mods, p0idxed - p0idxst, half);
    for (int i = 0, j = p0idxst; j < p0idxed; i++, j++) halfmod[i] = words_mod(half, nw, ringQ->s[j]->q);

    uint64_t y[32];
    int nsrc = decompLvl + 2;   /* multSum(level = decompLvl+1) sums level+1 terms */
    for (int x = 0; x < N; x++) {
        double vi = 0.0;
        for (int i = 0, j = p0idxst; j < p0idxed; i++, j++) {
            const lo_subring *s = ringQ->s[j];
            y[i] = lo_mred(p0Q[(size_t)j * N + x] + halfmod[i], muc->qoverqiinvqi[i], s->q, s->qinv);
            vi += (double)y[i] / (double)s->q;
        }
        uint64_t v = (uint64_t)vi;
        for (int j = 0; j <= levelQ; j++) {
            if (j >= p0idxst && j < p0idxed) continue;
            const lo_subring *sp = ringQ->s[j];
            const uint64_t *qq = muc->qoverqimodp + (size_t)j * muc->nq;
            u128 acc = (u128)y[0] * qq[0];
            for (int i = 1; i < nsrc; i++) acc += (u128)y[i] * qq[i];
            uint64_t hhi = mulhi64((uint64_t)acc * sp->qinv, sp->q);
            p1Q[(size_t)j * N + x] = (uint64_t)(acc >> 64) - hhi + sp->q + muc->vtimesqmodp[(size_t)j * (muc->nq + 1) + v];
        }
        for (int j = 0, u = nQ; j <= levelP; j++, u++) {
            const lo_subring *sp = ringP->s[j];
            const uint64_t *qq = muc->qoverqimodp + (size_t)u * muc->nq;
            u128 acc = (u128)y[0] * qq[0];
            for (int i = 1; i < nsrc; i++) acc += (u128)y[i] * qq[i];
            uint64_t hhi = mulhi64((uint64_t)acc * sp->qinv, sp->q);
            p1P[(size_t)j * N + x] = (uint64_t)(acc >> 64) - hhi + sp->q + muc->vtimesqmodp[(size_t)u * (muc->nq + 1) + v];
        }
    }
    /* ringQ.SubScalarBigint(p1Q, QHalf, p1Q); ringP.SubScalarBigint(p1P, QHalf, p1P)  (:500-501) */
    for (int l = 0; l <= levelQ; l++) {
        if (l >= p0idxst && l < p0idxed) continue;
        uint64_t q = ringQ->s[l]->q, sc = words_mod(half, nw, q);
        for (int j = 0; j < N; j++) p1Q[(size_t)l * N + j] = lo_cred(p1Q[(size_t)l * N + j] + q - sc, q);
    }
    for (int l = 0; l <= levelP; l++) {
        uint64_t q = ringP->s[l]->q, sc = words_mod(half, nw, q);
        for (int j = 0; j < N; j++) p1P[(size_t)l * N + j] = lo_cred(p1P[(size_t)l * N + j] + q - sc, q);
    }
}

/* ========================================================================== */
/* core/rlwe evaluator                                                          */
/* ========================================================================== */

lo_evaluator *lo_evaluator_new(lo_ring *ringQ, lo_ring *ringP) {
    lo_evaluator *e = (lo_evaluator *)calloc(1, sizeof *e);
    e->ringQ = ringQ; e->ringP = ringP;
    if (!ringP) return e;  /* parameters without special primes (P = nil): levelP = -1 everywhere, no basis extension */
    e->be = lo_basis_extender_new(ringQ, ringP);
    e->dec = lo_decomposer_new(ringQ, ringP);
    return e;
}
void lo_evaluator_free(lo_evaluator *e) {
    if (!e) return;
    lo_basis_extender_free(e->be); lo_decomposer_free(e->dec); free(e);
}
/* BaseRNSDecompositionVectorSize, core/rlwe/params.go:543-550 */
int lo_base_rns_decomposition_vector_size(int levelQ, int levelP) {
    if (levelP == -1) return levelQ + 1;
    return (levelQ + levelP + 1) / (levelP + 1);
}
/* QiOverflowMargin / PiOverflowMargin, core/rlwe/params.go:554-568 */
static int overflow_margin(const lo_ring *r, int level) {
    uint64_t mx = 0;
    for (int i = 0; i <= level; i++) if (r->s[i]->q > mx) mx = r->s[i]->q;
    return (int)(exp2(64) / (double)mx);
}

/* DecomposeSingleNTT, core/rlwe/evaluator_gadget_product.go:485-510 */
static void decompose_single_ntt(const lo_evaluator *e, int levelQ, int levelP, int nbPi, int digit,
                                 const uint64_t *c2NTT, const uint64_t *c2InvNTT, uint64_t *outQ, uint64_t *outP) {
    int N = e->ringQ->N;
    lo_decompose_and_split(e->dec, levelQ, levelP, nbPi, digit, c2InvNTT, outQ, outP);
    int p0idxst = digit * nbPi, p0idxed = p0idxst + nbPi;
    for (int x = 0; x <= levelQ; x++) {
        if (p0idxst <= x && x < p0idxed) memcpy(outQ + (size_t)x * N, c2NTT + (size_t)x * N, (size_t)N * 8);
        else lo_subring_ntt(e->ringQ->s[x], outQ + (size_t)x * N, outQ + (size_t)x * N, 0);
    }
    lo_ntt(e->ringP, levelP, outP, outP);
}
/* DecomposeNTT, :459-483 */
void lo_decompose_ntt(const lo_evaluator *e, int levelQ, int levelP, int nbPi, const uint64_t *c2, int c2_is_ntt,
                      uint64_t *decompQ, uint64_t *decompP) {
    int N = e->ringQ->N;
    size_t szQ = (size_t)(levelQ + 1) * N, szP = (size_t)(levelP + 1) * N;
    uint64_t *buff = (uint64_t *)pool_get(szQ * 8);
    const uint64_t *polyNTT, *polyInvNTT;
    if (c2_is_ntt) { polyNTT = c2; lo_intt(e->ringQ, levelQ, c2, buff); polyInvNTT = buff; }
    else { polyInvNTT = c2; lo_ntt(e->ringQ, levelQ, c2, buff); polyNTT = buff; }
    int beta = lo_base_rns_decomposition_vector_size(levelQ, levelP);
    for (int i = 0; i < beta; i++)
        decompose_single_ntt(e, levelQ, levelP, nbPi, i, polyNTT, polyInvNTT, decompQ + i * szQ, decompP + i * szP);
    pool_put(buff);
}

static void reduce_poly(const lo_ring *r, int level, uint64_t *p) { lo_unop(r, level, LO_REDUCE, p, p); }

/* accumulate one digit: ct[k] (+)= MRedLazy(evk[d][k], decomp) on Q and P,
 * ringqp MulCoeffsMontgomeryLazy[ThenAddLazy] (ring/ringqp/operations.go) */
static void evk_mac(const lo_evaluator *e, int levelQ, int levelP, const lo_evk *evk, int d, int first,
                    const uint64_t *dq, const uint64_t *dp, uint64_t *ctQ, uint64_t *ctP) {
    int N = e->ringQ->N;
    size_t szQ = (size_t)(levelQ + 1) * N, szP = (size_t)(levelP + 1) * N;
    int op = first ? LO_MUL_MONT_LAZY : LO_MUL_MONT_LAZY_THEN_ADD_LAZY;
    for (int k = 0; k < 2; k++) {
        const uint64_t *kq = evk->q + ((size_t)(d * 2 + k) * evk->nQk) * N;
        const uint64_t *kp = evk->p + ((size_t)(d * 2 + k) * evk->nPk) * N;
        lo_binop(e->ringQ, levelQ, op, kq, dq, ctQ + k * szQ);
        lo_binop(e->ringP, levelP, op, kp, dp, ctP + k * szP);
    }
}
static void periodic_reduce(const lo_evaluator *e, int levelQ, int levelP, int reduce, int QiOverF, int PiOverF,
                            uint64_t *ctQ, uint64_t *ctP) {
    int N = e->ringQ->N;
    size_t szQ = (size_t)(levelQ + 1) * N, szP = (size_t)(levelP + 1) * N;
    if (reduce % QiOverF == QiOverF - 1) { reduce_poly(e->ringQ, levelQ, ctQ); reduce_poly(e->ringQ, levelQ, ctQ + szQ); }
    /* `if ringP != nil` (:309-312) */
    if (e->ringP && reduce % PiOverF == PiOverF - 1) { reduce_poly(e->ringP, levelP, ctP); reduce_poly(e->ringP, levelP, ctP + szP); }
}
static void final_reduce(const lo_evaluator *e, int levelQ, int levelP, int reduce, int QiOverF, int PiOverF,
                         uint64_t *ctQ, uint64_t *ctP) {
    int N = e->ringQ->N;
    size_t szQ = (size_t)(levelQ + 1) * N, szP = (size_t)(levelP + 1) * N;
    if (reduce % QiOverF != 0) { reduce_poly(e->ringQ, levelQ, ctQ); reduce_poly(e->ringQ, levelQ, ctQ + szQ); }
    /* `if ringP != nil` (:330-336) */
    if (e->ringP && reduce % PiOverF != 0) { reduce_poly(e->ringP, levelP, ctP); reduce_poly(e->ringP, levelP, ctP + szP); }
}

/* gadgetProductMultiplePLazy, :129-201 (ctQP.IsNTT = true) */
static void gadget_product_multiple_p_lazy(const lo_evaluator *e, int levelQ, const uint64_t *cx, const lo_evk *evk,
                                           uint64_t *ctQ, uint64_t *ctP) {
    int N = e->ringQ->N, levelP = evk->nPk - 1;
    size_t szQ = (size_t)(levelQ + 1) * N, szP = (size_t)(levelP + 1) * N;
    uint64_t *cxInv = (uint64_t *)pool_get(szQ * 8);
    uint64_t *c2Q = (uint64_t *)pool_get(szQ * 8), *c2P = (uint64_t *)pool_get(szP * 8);
    lo_intt(e->ringQ, levelQ, cx, cxInv);
    int beta = lo_base_rns_decomposition_vector_size(levelQ, levelP);
    int QiOverF = overflow_margin(e->ringQ, levelQ) >> 1, PiOverF = overflow_margin(e->ringP, levelP) >> 1;
    int reduce = 0;
    for (int i = 0; i < beta; i++) {
        decompose_single_ntt(e, levelQ, levelP, levelP + 1, i, cx, cxInv, c2Q, c2P);
        evk_mac(e, levelQ, levelP, evk, i, i == 0, c2Q, c2P, ctQ, ctP);
        periodic_reduce(e, levelQ, levelP, reduce, QiOverF, PiOverF, ctQ, ctP);
        reduce++;
    }
    final_reduce(e, levelQ, levelP, reduce, QiOverF, PiOverF, ctQ, ctP);
    pool_put(cxInv); pool_put(c2Q); pool_put(c2P);
}
/* gadgetProductSinglePAndBitDecompLazy, :203-338 (ctQP.IsNTT = true): one RNS digit per Q-limb, and for a
 * base-2 gadget (BaseTwoDecomposition = pw2 != 0) one bit window (x >> j*pw2) & mask per stored block */
static void gadget_product_single_p_lazy(const lo_evaluator *e, int levelQ, const uint64_t *cx, const lo_evk *evk,
                                         uint64_t *ctQ, uint64_t *ctP) {
    int N = e->ringQ->N, levelP = evk->nPk - 1;
    size_t szQ = (size_t)(levelQ + 1) * N, szP = (size_t)(levelP + 1) * N;
    uint64_t *cxInv = (uint64_t *)pool_get(szQ * 8);
    uint64_t *c2Q = (uint64_t *)pool_get(szQ * 8), *c2P = (uint64_t *)pool_get(szP * 8);
    uint64_t *cw = (uint64_t *)pool_get((size_t)N * 8), *cwNTT = (uint64_t *)pool_get((size_t)N * 8);
    lo_intt(e->ringQ, levelQ, cx, cxInv);
    int pw2 = evk->pw2;
    uint64_t mask = pw2 ? (((uint64_t)1 << pw2) - 1) : 0;
    /* levelP = -1 (ringP == nil, :284,:302): the P loops below do not run and PiOverF is never looked at */
    int QiOverF = overflow_margin(e->ringQ, levelQ) >> 1, PiOverF = e->ringP ? overflow_margin(e->ringP, levelP) >> 1 : 1;
    int reduce = 0, blk = 0;
    for (int i = 0; i < levelQ + 1; i++) {
        if (mask == 0) lo_decompose_and_split(e->dec, levelQ, levelP, levelP + 1, i, cxInv, c2Q, c2P);
        int nj = pw2 ? evk->nj[i] : 1;
        for (int j = 0; j < nj; j++, blk++) {
            if (mask != 0)   /* ring.MaskVec, ring/vec_ops.go:870 */
                for (int x = 0; x < N; x++) cw[x] = (cxInv[(size_t)i * N + x] >> (j * pw2)) & mask;
            int first = (i == 0 && j == 0);
            for (int u = 0; u <= levelQ; u++) {
                const lo_subring *s = e->ringQ->s[u];
                lo_subring_ntt(s, mask == 0 ? c2Q + (size_t)u * N : cw, cwNTT, 1);
                for (int k = 0; k < 2; k++) {
                    const uint64_t *kq = evk->q + (((size_t)(blk * 2 + k) * evk->nQk) + u) * N;
                    uint64_t *z = ctQ + k * szQ + (size_t)u * N;
                    if (first) for (int x = 0; x < N; x++) z[x] = lo_mred_lazy(kq[x], cwNTT[x], s->q, s->qinv);
                    else for (int x = 0; x < N; x++) z[x] += lo_mred_lazy(kq[x], cwNTT[x], s->q, s->qinv);
                }
            }
            for (int u = 0; u <= levelP; u++) {
                const lo_subring *s = e->ringP->s[u];
                lo_subring_ntt(s, mask == 0 ? c2P + (size_t)u * N : cw, cwNTT, 1);
                for (int k = 0; k < 2; k++) {
                    const uint64_t *kp = evk->p + (((size_t)(blk * 2 + k) * evk->nPk) + u) * N;
                    uint64_t *z = ctP + k * szP + (size_t)u * N;
                    if (first) for (int x = 0; x < N; x++) z[x] = lo_mred_lazy(kp[x], cwNTT[x], s->q, s->qinv);
                    else for (int x = 0; x < N; x++) z[x] += lo_mred_lazy(kp[x], cwNTT[x], s->q, s->qinv);
                }
            }
            periodic_reduce(e, levelQ, levelP, reduce, QiOverF, PiOverF, ctQ, ctP);
            reduce++;
        }
    }
    final_reduce(e, levelQ, levelP, reduce, QiOverF, PiOverF, ctQ, ctP);
    pool_put(cxInv); pool_put(c2Q); pool_put(c2P); pool_put(cw); pool_put(cwNTT);
}
/* GadgetProductLazy, :108-127 */
void lo_gadget_product_lazy(const lo_evaluator *e, int levelQ, const uint64_t *cx, const lo_evk *evk,
                            uint64_t *ctQ, uint64_t *ctP) {
    if (evk->nPk - 1 > 0) gadget_product_multiple_p_lazy(e, levelQ, cx, evk, ctQ, ctP);
    else gadget_product_single_p_lazy(e, levelQ, cx, evk, ctQ, ctP);
}
/* gadgetProductMultiplePLazyHoisted, :401-453 */
void lo_gadget_product_hoisted_lazy(const lo_evaluator *e, int levelQ, const uint64_t *decompQ, const uint64_t *decompP,
                                    const lo_evk *evk, uint64_t *ctQ, uint64_t *ctP) {
    int N = e->ringQ->N, levelP = evk->nPk - 1;
    size_t szQ = (size_t)(levelQ + 1) * N, szP = (size_t)(levelP + 1) * N;
    int beta = lo_base_rns_decomposition_vector_size(levelQ, levelP);
    int QiOverF = overflow_margin(e->ringQ, levelQ) >> 1, PiOverF = overflow_margin(e->ringP, levelP) >> 1;
    int reduce = 0;
    for (int i = 0; i < beta; i++) {
        evk_mac(e, levelQ, levelP, evk, i, i == 0, decompQ + i * szQ, decompP + i * szP, ctQ, ctP);
        periodic_reduce(e, levelQ, levelP, reduce, QiOverF, PiOverF, ctQ, ctP);
        reduce++;
    }
    final_reduce(e, levelQ, levelP, reduce, QiOverF, PiOverF, ctQ, ctP);
}
/* ModDown, :39-97, branch ctQP.IsNTT && ct.IsNTT, levelP != -1 */
void lo_moddown_ntt(const lo_evaluator *e, int levelQ, int levelP, const uint64_t *ctQ, const uint64_t *ctP, uint64_t *ct) {
    int N = e->ringQ->N;
    size_t szQ = (size_t)(levelQ + 1) * N, szP = (size_t)(levelP + 1) * N;
    if (levelP == -1) {  /* :76-81: ctQP.Value[k].Q.CopyLvl(levelQ, ct.Value[k]) */
        memcpy(ct, ctQ, 2 * szQ * 8);
        return;
    }
    lo_moddown_qp_to_q_ntt(e->be, levelQ, levelP, ctQ, ctP, ct);
    lo_moddown_qp_to_q_ntt(e->be, levelQ, levelP, ctQ + szQ, ctP + szP, ct + szQ);
}
/* GadgetProduct, :16-36 */
void lo_gadget_product(const lo_evaluator *e, int levelQ, const uint64_t *cx, const lo_evk *evk, uint64_t *ct) {
    int N = e->ringQ->N, levelP = evk->nPk - 1;
    if (levelQ > evk->nQk - 1) levelQ = evk->nQk - 1;
    size_t szQ = (size_t)(levelQ + 1) * N, szP = (size_t)(levelP + 1) * N;
    uint64_t *ctQ = (uint64_t *)pool_get(2 * szQ * 8), *ctP = (uint64_t *)pool_get(2 * szP * 8);
    lo_gadget_product_lazy(e, levelQ, cx, evk, ctQ, ctP);
    lo_moddown_ntt(e, levelQ, levelP, ctQ, ctP, ct);
    pool_put(ctQ); pool_put(ctP);
}
/* GadgetProductHoisted, :348-368 */
void lo_gadget_product_hoisted(const lo_evaluator *e, int levelQ, const uint64_t *decompQ, const uint64_t *decompP,
                               const lo_evk *evk, uint64_t *ct) {
    int N = e->ringQ->N, levelP = evk->nPk - 1;
    size_t szQ = (size_t)(levelQ + 1) * N, szP = (size_t)(levelP + 1) * N;
    uint64_t *ctQ = (uint64_t *)pool_get(2 * szQ * 8), *ctP = (uint64_t *)pool_get(2 * szP * 8);
    lo_gadget_product_hoisted_lazy(e, levelQ, decompQ, decompP, evk, ctQ, ctP);
    lo_moddown_ntt(e, levelQ, levelP, ctQ, ctP, ct);
    pool_put(ctQ); pool_put(ctP);
}
/* Relinearize, core/rlwe/evaluator_evaluationkey.go:117-148 */
void lo_relinearize(const lo_evaluator *e, int level, const uint64_t *ct_in, const lo_evk *rlk, uint64_t *ct_out) {
    int N = e->ringQ->N;
    size_t sz = (size_t)(level + 1) * N;
    uint64_t *tmp = (uint64_t *)pool_get(2 * sz * 8);
    lo_gadget_product(e, level, ct_in + 2 * sz, rlk, tmp);
    lo_binop(e->ringQ, level, LO_ADD, ct_in, tmp, ct_out);
    lo_binop(e->ringQ, level, LO_ADD, ct_in + sz, tmp + sz, ct_out + sz);
    pool_put(tmp);
}
/* Automorphism, core/rlwe/evaluator_automorphism.go:13-54 (ctIn.IsNTT) */
void lo_automorphism_ct(const lo_evaluator *e, int level, const uint64_t *ct_in, uint64_t galel, const lo_evk *gk, uint64_t *ct_out) {
    int N = e->ringQ->N;
    size_t sz = (size_t)(level + 1) * N;
    uint64_t *tmp = (uint64_t *)pool_get(2 * sz * 8);
    uint64_t *index = (uint64_t *)pool_get((size_t)N * 8);
    lo_automorphism_ntt_index(N, e->ringQ->s[0]->nthroot, galel, index);
    lo_gadget_product(e, level, ct_in + sz, gk, tmp);
    lo_binop(e->ringQ, level, LO_ADD, tmp, ct_in, tmp);
    lo_automorphism_ntt_with_index(e->ringQ, level, tmp, index, ct_out);
    lo_automorphism_ntt_with_index(e->ringQ, level, tmp + sz, index, ct_out + sz);
    pool_put(tmp); pool_put(index);
}
/* AutomorphismHoisted, :60-100 */
void lo_automorphism_hoisted(const lo_evaluator *e, int level, const uint64_t *ct_in, const uint64_t *decompQ,
                             const uint64_t *decompP, uint64_t galel, const lo_evk *gk, uint64_t *ct_out) {
    int N = e->ringQ->N;
    size_t sz = (size_t)(level + 1) * N;
    uint64_t *tmp = (uint64_t *)pool_get(2 * sz * 8);
    uint64_t *index = (uint64_t *)pool_get((size_t)N * 8);
    lo_automorphism_ntt_index(N, e->ringQ->s[0]->nthroot, galel, index);
    lo_gadget_product_hoisted(e, level, decompQ, decompP, gk, tmp);
    lo_binop(e->ringQ, level, LO_ADD, tmp, ct_in, tmp);
    lo_automorphism_ntt_with_index(e->ringQ, level, tmp, index, ct_out);
    lo_automorphism_ntt_with_index(e->ringQ, level, tmp + sz, index, ct_out + sz);
    pool_put(tmp); pool_put(index);
}

/* AutomorphismHoistedLazy, core/rlwe/evaluator_automorphism.go:104-165 (ctQP.IsNTT).
 * ct_in0: [level+1][N] (ctIn.Value[0]); outQ: [2][levelQ+1][N], outP: [2][levelP+1][N]. */
void lo_automorphism_hoisted_lazy(const lo_evaluator *e, int levelQ, const uint64_t *ct_in0, const uint64_t *decompQ,
                                  const uint64_t *decompP, uint64_t galel, const lo_evk *gk, uint64_t *outQ, uint64_t *outP) {
    int N = e->ringQ->N, levelP = gk->nPk - 1;
    size_t szQ = (size_t)(levelQ + 1) * N, szP = (size_t)(levelP + 1) * N;
    uint64_t *tQ = (uint64_t *)pool_get(2 * szQ * 8), *tP = (uint64_t *)pool_get(2 * szP * 8);
    uint64_t *index = (uint64_t *)pool_get((size_t)N * 8);
    lo_automorphism_ntt_index(N, e->ringQ->s[0]->nthroot, galel, index);
    lo_gadget_product_hoisted_lazy(e, levelQ, decompQ, decompP, gk, tQ, tP);
    lo_automorphism_ntt_with_index(e->ringQ, levelQ, tQ + szQ, index, outQ + szQ);
    lo_automorphism_ntt_with_index(e->ringP, levelP, tP + szP, index, outP + szP);
    {   /* ringQ.MulScalarBigint(ctIn.Value[0], ringP.ModulusAtLevel[levelP], ctTmp.Value[1].Q) */
        uint64_t mods[64], words[64];
        ring_moduli(e->ringP, mods);
        int nw = 1; words[0] = 1;
        for (int i = 0; i <= levelP; i++) {
            uint64_t carry = 0;
            for (int k = 0; k < nw; k++) { u128 t = (u128)words[k] * mods[i] + carry; words[k] = (uint64_t)t; carry = (uint64_t)(t >> 64); }
            if (carry) words[nw++] = carry;
        }
        lo_mul_scalar_bigint(e->ringQ, levelQ, ct_in0, words, nw, tQ + szQ);
    }
    lo_binop(e->ringQ, levelQ, LO_ADD, tQ, tQ + szQ, tQ);
    lo_automorphism_ntt_with_index(e->ringQ, levelQ, tQ, index, outQ);
    lo_automorphism_ntt_with_index(e->ringP, levelP, tP, index, outP);
    pool_put(tQ); pool_put(tP); pool_put(index);
}

/* ========================================================================== */
/* Scheme glue                                                                  */
/* ========================================================================== */

/* shared tensor body: schemes/ckks/evaluator.go:807-838 and
 * schemes/bgv/evaluator.go:634-666 (regular, non-squaring case) */
static void tensor(const lo_evaluator *e, int level, const uint64_t *c00, const uint64_t *c01,
                   const uint64_t *op1, const lo_evk *rlk, int relin, uint64_t *out) {
    int N = e->ringQ->N;
    size_t sz = (size_t)(level + 1) * N;
    uint64_t *c0 = out, *c1 = out + sz;
    uint64_t *c2 = relin ? (uint64_t *)pool_get(sz * 8) : out + 2 * sz;
    lo_binop(e->ringQ, level, LO_MUL_MONT, c00, op1, c0);
    lo_binop(e->ringQ, level, LO_MUL_MONT, c01, op1 + sz, c2);
    lo_binop(e->ringQ, level, LO_MUL_MONT, c00, op1 + sz, c1);
    lo_binop(e->ringQ, level, LO_MUL_MONT_THEN_ADD, c01, op1, c1);
    if (relin) {
        uint64_t *tmp = (uint64_t *)pool_get(2 * sz * 8);
        lo_gadget_product(e, level, c2, rlk, tmp);
        lo_binop(e->ringQ, level, LO_ADD, c0, tmp, c0);
        lo_binop(e->ringQ, level, LO_ADD, c1, tmp + sz, c1);
        pool_put(tmp); pool_put(c2);
    }
}
/* CKKS mulRelin, schemes/ckks/evaluator.go:764-872 (degree-1 x degree-1) */
void lo_ckks_mul_relin(const lo_evaluator *e, int level, const uint64_t *op0, const uint64_t *op1,
                       const lo_evk *rlk, int relin, uint64_t *out) {
    int N = e->ringQ->N;
    size_t sz = (size_t)(level + 1) * N;
    uint64_t *c00 = (uint64_t *)pool_get(sz * 8), *c01 = (uint64_t *)pool_get(sz * 8);
    lo_unop(e->ringQ, level, LO_MFORM, op0, c00);
    lo_unop(e->ringQ, level, LO_MFORM, op0 + sz, c01);
    tensor(e, level, c00, c01, op1, rlk, relin, out);
    pool_put(c00); pool_put(c01);
}
/* BGV tensorStandard, schemes/bgv/evaluator.go:592-685; tMontgomery :59-62 */
void lo_bgv_mul_relin(const lo_evaluator *e, int level, uint64_t t, const uint64_t *op0, const uint64_t *op1,
                      const lo_evk *rlk, int relin, uint64_t *out) {
    int N = e->ringQ->N;
    size_t sz = (size_t)(level + 1) * N;
    uint64_t tmont[64];
    for (int i = 0; i <= level; i++) {
        const lo_subring *s = e->ringQ->s[i];
        uint64_t w[2] = {0, t};                               /* t << 64 */
        tmont[i] = lo_mform(words_mod(w, 2, s->q), s->q, s->brc);
    }
    uint64_t *c00 = (uint64_t *)pool_get(sz * 8), *c01 = (uint64_t *)pool_get(sz * 8);
    lo_mul_rns_scalar_montgomery(e->ringQ, level, op0, tmont, c00);
    lo_mul_rns_scalar_montgomery(e->ringQ, level, op0 + sz, tmont, c01);
    tensor(e, level, c00, c01, op1, rlk, relin, out);
    pool_put(c00); pool_put(c01);
}
/* CKKS Rescale (schemes/ckks/evaluator.go:477-515) / BGV Rescale
 * (schemes/bgv/evaluator.go:1363-1393): per poly DivRoundByLastModulusManyNTT */
void lo_rescale(const lo_ring *r, int level, int degree, int nb, const uint64_t *in, uint64_t *out) {
    int N = r->N;
    size_t szin = (size_t)(level + 1) * N, szout = (size_t)(level + 1 - nb) * N;
    uint64_t *tmp = (uint64_t *)pool_get(szin * 8);
    for (int i = 0; i <= degree; i++) {
        lo_div_round_by_last_modulus_many_ntt(r, level, nb, in + i * szin, tmp);
        memcpy(out + i * szout, tmp, szout * 8);
    }
    pool_put(tmp);
}

/* ========================================================================== */
/* Timed CPU baseline (bench.py's cpu_baseline leg)                             */
/* ========================================================================== */
/* `nthreads` OS threads each repeat one operation on the shared (read-only) evaluator until `seconds` have elapsed -- the shape
 * of the reference's own parallel benchmarks (b.RunParallel over one evaluator, schemes/ckks/ckks_benchmarks_test.go:95-325;
 * evaluator methods are safe for concurrent callers since 6.2.0 because scratch comes from pools).
 * kind 0: BGV MulRelin(op0, op1, key)        (schemes/bgv/evaluator.go:592-667)
 * kind 1: Automorphism(op0, gal, key)        (core/rlwe/evaluator_automorphism.go:13-51; CKKS Rotate)
 * kind 2: CKKS Mul (degree 2) + Rescale      (schemes/ckks/evaluator.go:764-872, :477-515), key unused
 * Placement (round 3): every thread is pinned to one CPU of the process's affinity set and works on its OWN first-touched copy
 * of the inputs and of the key -- with one shared copy, first touched by the calling thread, every other NUMA node read the
 * key over the fabric and the rate FELL beyond 16 threads.  The evaluator's tables (twiddles, basis-extension constants:
 * a few MB, read-only) stay shared.  counts[i] = ops finished by thread i; returns the elapsed wall time in seconds. */
#include <pthread.h>
#include <sched.h>
#include <time.h>
typedef struct {
    int kind, cpu, private_copy;
    const lo_evaluator *e; int level; uint64_t t, gal; const uint64_t *op0, *op1; const lo_evk *key;
    double deadline; uint64_t count;
    pthread_barrier_t *start;
} bench_arg;
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static uint64_t *dup_words(const uint64_t *src, size_t n) {
    uint64_t *d = (uint64_t *)malloc(n * 8);
    memcpy(d, src, n * 8);  /* first touch by the calling (pinned) thread */
    return d;
}
static void *bench_worker(void *vp) {
    bench_arg *a = (bench_arg *)vp;
    if (a->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(a->cpu, &set);
        pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    }
    const int N = a->e->ringQ->N;
    const size_t sz = (size_t)(a->level + 1) * N;
    const uint64_t *op0 = a->op0, *op1 = a->op1;
    lo_evk key;
    if (a->key) key = *a->key;
    uint64_t *own[4] = {NULL, NULL, NULL, NULL};
    if (a->private_copy) {
        op0 = own[0] = dup_words(a->op0, 2 * sz);
        if (a->op1) op1 = own[1] = dup_words(a->op1, 2 * sz);
        if (a->key) {
            key.q = own[2] = dup_words(a->key->q, (size_t)a->key->beta * 2 * a->key->nQk * N);
            if (a->key->p) key.p = own[3] = dup_words(a->key->p, (size_t)a->key->beta * 2 * a->key->nPk * N);
        }
    }
    uint64_t *out = (uint64_t *)malloc(3 * sz * 8), *out2 = (uint64_t *)malloc(3 * sz * 8);
    /* one untimed call: page in the scratch pools and the outputs */
    if (a->kind == 0) lo_bgv_mul_relin(a->e, a->level, a->t, op0, op1, &key, 1, out);
    else if (a->kind == 1) lo_automorphism_ct(a->e, a->level, op0, a->gal, &key, out);
    else { lo_ckks_mul_relin(a->e, a->level, op0, op1, NULL, 0, out); lo_rescale(a->e->ringQ, a->level, 2, 1, out, out2); }
    pthread_barrier_wait(a->start);
    a->deadline += now_s();
    do {
        if (a->kind == 0) lo_bgv_mul_relin(a->e, a->level, a->t, op0, op1, &key, 1, out);
        else if (a->kind == 1) lo_automorphism_ct(a->e, a->level, op0, a->gal, &key, out);
        else { lo_ckks_mul_relin(a->e, a->level, op0, op1, NULL, 0, out); lo_rescale(a->e->ringQ, a->level, 2, 1, out, out2); }
        a->count++;
    } while (now_s() < a->deadline);
    free(out); free(out2);
    for (int i = 0; i < 4; i++) free(own[i]);
    lo_pool_release();
    return NULL;
}
double lo_bench_op(const lo_evaluator *e, int kind, int level, uint64_t t, uint64_t gal, const uint64_t *op0, const uint64_t *op1,
                   const lo_evk *key, int nthreads, double seconds, int pin, int private_copy, uint64_t *counts) {
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    bench_arg *args = (bench_arg *)calloc(nthreads, sizeof(bench_arg));
    /* the CPUs this process may run on, in order: thread i goes to the i-th of them (round robin) */
    cpu_set_t allowed;
    int cpus[CPU_SETSIZE], ncpu = 0;
    if (pin && sched_getaffinity(0, sizeof allowed, &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed)) cpus[ncpu++] = c;
    pthread_barrier_t start;
    pthread_barrier_init(&start, NULL, (unsigned)nthreads + 1);
    for (int i = 0; i < nthreads; i++) {
        args[i] = (bench_arg){kind, ncpu > 0 ? cpus[i % ncpu] : -1, private_copy, e, level, t, gal, op0, op1, key, seconds, 0, &start};
        pthread_create(&th[i], NULL, bench_worker, &args[i]);
    }
    pthread_barrier_wait(&start);  /* every thread has its copies and has run once */
    const double t0 = now_s();
    for (int i = 0; i < nthreads; i++) { pthread_join(th[i], NULL); counts[i] = args[i].count; }
    const double dt = now_s() - t0;
    pthread_barrier_destroy(&start);
    free(th); free(args);
    return dt;
}
double lo_bench_bgv_mul_relin(const lo_evaluator *e, int level, uint64_t t, const uint64_t *op0, const uint64_t *op1,
                              const lo_evk *rlk, int nthreads, double seconds, uint64_t *counts) {
    return lo_bench_op(e, 0, level, t, 0, op0, op1, rlk, nthreads, seconds, 1, 1, counts);
}

/* Checker for whole batches (bench.py / tests verify EVERY entry of a timed batch): entry b of `nb` independent inputs through
 * the same operation kinds as lo_bench_op, entries dealt round-robin to `nthreads` OS threads.  op0 / op1: [nb][2][level+1][N];
 * out: [nb][ncomp][level_out+1][N] with (ncomp, level_out) = (2, level) for kinds 0 / 1 and (3, level - 1) for kind 2. */
typedef struct {
    int kind, tid, nthreads, nb;
    const lo_evaluator *e; int level; uint64_t t, gal; const uint64_t *op0, *op1; const lo_evk *key; uint64_t *out;
} batch_arg;
static void *batch_worker(void *vp) {
    batch_arg *a = (batch_arg *)vp;
    const int N = a->e->ringQ->N;
    const size_t sz = (size_t)(a->level + 1) * N;
    const size_t osz = a->kind == 2 ? 3 * (size_t)a->level * N : 2 * sz;
    uint64_t *tmp = a->kind == 2 ? (uint64_t *)malloc(3 * sz * 8) : NULL;
    for (int b = a->tid; b < a->nb; b += a->nthreads) {
        const uint64_t *x0 = a->op0 + (size_t)b * 2 * sz, *x1 = a->op1 ? a->op1 + (size_t)b * 2 * sz : NULL;
        uint64_t *o = a->out + (size_t)b * osz;
        if (a->kind == 0) lo_bgv_mul_relin(a->e, a->level, a->t, x0, x1, a->key, 1, o);
        else if (a->kind == 1) lo_automorphism_ct(a->e, a->level, x0, a->gal, a->key, o);
        else { lo_ckks_mul_relin(a->e, a->level, x0, x1, NULL, 0, tmp); lo_rescale(a->e->ringQ, a->level, 2, 1, tmp, o); }
    }
    free(tmp);
    lo_pool_release();
    return NULL;
}
void lo_batch_op(const lo_evaluator *e, int kind, int level, uint64_t t, uint64_t gal, const uint64_t *op0, const uint64_t *op1,
                 const lo_evk *key, int nb, int nthreads, uint64_t *out) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nb) nthreads = nb;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    batch_arg *args = (batch_arg *)calloc(nthreads, sizeof(batch_arg));
    if (!th || !args) { fprintf(stderr, "lo_batch_op: out of memory\n"); abort(); }
    for (int i = 0; i < nthreads; i++) {
        args[i] = (batch_arg){kind, i, nthreads, nb, e, level, t, gal, op0, op1, key, out};
        if (pthread_create(&th[i], NULL, batch_worker, &args[i]) != 0) { fprintf(stderr, "lo_batch_op: pthread_create failed\n"); abort(); }
    }
    for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    free(th); free(args);
}
