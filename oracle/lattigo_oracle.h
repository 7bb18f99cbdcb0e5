/*
 * lattigo_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, never shipped, never
 * on the product path).
 *
 * Plain-C restatement of the pure-Go arithmetic of tuneinsight/lattigo v6.2.0's
 * `ring` package and of the `core/rlwe` key-switch built on it. Every function
 * cites the reference file:line it follows (paths relative to the reference
 * tree).  Parity is PINNED: the restatement reproduces all 12 known-answer
 * vectors of ring/ntt_test.go:17-88 (tests/golden/ntt_kat.json) and the
 * big-integer property checks of ring/ring_test.go (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this code -- as the checker / the timed CPU baseline, never as
 * the thing shipped.
 *
 * Data model: a polynomial is a contiguous row-major array [limbs][N] of
 * uint64 (the reference's Poly.Coeffs [][]uint64, ring/poly.go:13-15, made
 * contiguous).  "level" always means "limbs 0..level are in use".
 */
#ifndef LATTIGO_ORACLE_H
#define LATTIGO_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalars: ring/modular_reduction.go ---------------------------------- */
uint64_t lo_mform(uint64_t a, uint64_t q, const uint64_t brc[2]);
uint64_t lo_mform_lazy(uint64_t a, uint64_t q, const uint64_t brc[2]);
uint64_t lo_imform(uint64_t a, uint64_t q, uint64_t qinv);
uint64_t lo_imform_lazy(uint64_t a, uint64_t q, uint64_t qinv);
uint64_t lo_gen_mred_constant(uint64_t q);
void     lo_gen_bred_constant(uint64_t q, uint64_t brc[2]);
uint64_t lo_mred(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv);
uint64_t lo_mred_lazy(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv);
uint64_t lo_bred_add(uint64_t a, uint64_t q, const uint64_t brc[2]);
uint64_t lo_bred_add_lazy(uint64_t a, uint64_t q, const uint64_t brc[2]);
uint64_t lo_bred(uint64_t x, uint64_t y, uint64_t q, const uint64_t brc[2]);
uint64_t lo_bred_lazy(uint64_t x, uint64_t y, uint64_t q, const uint64_t brc[2]);
uint64_t lo_cred(uint64_t a, uint64_t q);
uint64_t lo_modexp(uint64_t x, uint64_t e, uint64_t p);
int      lo_is_prime(uint64_t x);

/* ---- SubRing / Ring: ring/subring.go, ring/ring.go ------------------------ */
typedef struct lo_subring {
    int      N;
    uint64_t q;            /* Modulus */
    uint64_t mask;
    uint64_t brc[2];       /* BRedConstant */
    uint64_t qinv;         /* MRedConstant */
    uint64_t nthroot;
    uint64_t primroot;     /* PrimitiveRoot */
    uint64_t ninv;         /* NInv (Montgomery) */
    uint64_t *roots_fwd;   /* RootsForward[N]  */
    uint64_t *roots_bwd;   /* RootsBackward[N] */
    int      nfactors;
    uint64_t factors[64];
} lo_subring;

typedef struct lo_ring {
    int          N;
    int          nmod;
    lo_subring **s;
    uint64_t   **rescale;  /* RescaleConstants[j-1][i], i<j (ring/ring.go:329) */
} lo_ring;

lo_subring *lo_subring_new(int N, uint64_t q);             /* NULL on error */
lo_subring *lo_subring_new_nthroot(int N, uint64_t q, uint64_t nthroot);
lo_ring    *lo_ring_new_type(int N, const uint64_t *moduli, int nmod, int type); /* 1 = ConjugateInvariant */
void        lo_subring_free(lo_subring *s);
lo_ring    *lo_ring_new(int N, const uint64_t *moduli, int nmod);
void        lo_ring_free(lo_ring *r);
const char *lo_last_error(void);

/* table / constant accessors for the Python side */
const uint64_t *lo_ring_roots_fwd(const lo_ring *r, int i);
const uint64_t *lo_ring_roots_bwd(const lo_ring *r, int i);
void lo_ring_constants(const lo_ring *r, int i, uint64_t out[7]); /* q,qinv,brc0,brc1,ninv,primroot,mask */
uint64_t lo_ring_rescale_constant(const lo_ring *r, int j, int i);

/* prime generation: ring/primes.go + core/rlwe/params.go:811 */
int lo_gen_moduli(int log_nth_root, const int *logq, int nq, const int *logp, int np,
                  uint64_t *q_out, uint64_t *p_out);

/* ---- NTT: ring/ntt.go ----------------------------------------------------- */
void lo_ntt(const lo_ring *r, int level, const uint64_t *p1, uint64_t *p2);
void lo_ntt_lazy(const lo_ring *r, int level, const uint64_t *p1, uint64_t *p2);
void lo_intt(const lo_ring *r, int level, const uint64_t *p1, uint64_t *p2);
void lo_intt_lazy(const lo_ring *r, int level, const uint64_t *p1, uint64_t *p2);
void lo_subring_ntt(const lo_subring *s, const uint64_t *p1, uint64_t *p2, int lazy);
void lo_subring_intt(const lo_subring *s, const uint64_t *p1, uint64_t *p2, int lazy);

/* ---- coefficient-wise ops: ring/vec_ops.go via ring/operations.go ---------- */
enum lo_binop {
    LO_ADD = 0, LO_ADD_LAZY, LO_SUB, LO_SUB_LAZY,
    LO_MUL_BARRETT, LO_MUL_BARRETT_LAZY, LO_MUL_BARRETT_THEN_ADD, LO_MUL_BARRETT_THEN_ADD_LAZY,
    LO_MUL_MONT, LO_MUL_MONT_LAZY, LO_MUL_MONT_LAZY_THEN_NEG,
    LO_MUL_MONT_THEN_ADD, LO_MUL_MONT_THEN_ADD_LAZY, LO_MUL_MONT_LAZY_THEN_ADD_LAZY,
    LO_MUL_MONT_THEN_SUB, LO_MUL_MONT_THEN_SUB_LAZY, LO_MUL_MONT_LAZY_THEN_SUB_LAZY,
    LO_BINOP_COUNT
};
enum lo_unop {
    LO_NEG = 0, LO_REDUCE, LO_REDUCE_LAZY, LO_MFORM, LO_MFORM_LAZY, LO_IMFORM, LO_UNOP_COUNT
};
enum lo_scalarop {
    LO_ADD_SCALAR = 0, LO_SUB_SCALAR, LO_MUL_SCALAR, LO_MUL_SCALAR_THEN_ADD, LO_MUL_SCALAR_THEN_SUB,
    LO_SCALAROP_COUNT
};
void lo_binop(const lo_ring *r, int level, int op, const uint64_t *p1, const uint64_t *p2, uint64_t *p3);
void lo_unop(const lo_ring *r, int level, int op, const uint64_t *p1, uint64_t *p2);
void lo_scalarop(const lo_ring *r, int level, int op, const uint64_t *p1, uint64_t scalar, uint64_t *p2);
/* scalar given per limb, already in Montgomery form (ring/operations.go:216) */
void lo_mul_rns_scalar_montgomery(const lo_ring *r, int level, const uint64_t *p1, const uint64_t *scalar, uint64_t *p2);
/* scalar given as big integer, little-endian 64-bit words */
void lo_add_scalar_bigint(const lo_ring *r, int level, const uint64_t *p1, const uint64_t *words, int nwords, uint64_t *p2);
void lo_sub_scalar_bigint(const lo_ring *r, int level, const uint64_t *p1, const uint64_t *words, int nwords, uint64_t *p2);
void lo_mul_scalar_bigint(const lo_ring *r, int level, const uint64_t *p1, const uint64_t *words, int nwords, uint64_t *p2);

/* ---- rescale: ring/scaling.go ---------------------------------------------- */
void lo_div_floor_by_last_modulus_ntt(const lo_ring *r, int level, const uint64_t *p0, uint64_t *p1);
void lo_div_floor_by_last_modulus(const lo_ring *r, int level, const uint64_t *p0, uint64_t *p1);
void lo_div_round_by_last_modulus_ntt(const lo_ring *r, int level, const uint64_t *p0, uint64_t *p1);
void lo_div_round_by_last_modulus(const lo_ring *r, int level, const uint64_t *p0, uint64_t *p1);
void lo_div_round_by_last_modulus_many_ntt(const lo_ring *r, int level, int nb, const uint64_t *p0, uint64_t *p1);
void lo_div_round_by_last_modulus_many(const lo_ring *r, int level, int nb, const uint64_t *p0, uint64_t *p1);
void lo_div_floor_by_last_modulus_many_ntt(const lo_ring *r, int level, int nb, const uint64_t *p0, uint64_t *p1);
void lo_div_floor_by_last_modulus_many(const lo_ring *r, int level, int nb, const uint64_t *p0, uint64_t *p1);

/* ---- automorphism: ring/automorphism.go ----------------------------------- */
void lo_automorphism_ntt_index(int N, uint64_t nthroot, uint64_t galel, uint64_t *index);
void lo_automorphism_ntt_with_index(const lo_ring *r, int level, const uint64_t *pin, const uint64_t *index, uint64_t *pout);
void lo_automorphism_ntt_with_index_then_add_lazy(const lo_ring *r, int level, const uint64_t *pin, const uint64_t *index, uint64_t *pout);
void lo_automorphism(const lo_ring *r, int level, const uint64_t *pin, uint64_t galel, uint64_t *pout);

/* ---- basis extension: ring/basis_extension.go ------------------------------ */
typedef struct lo_modup_constants {
    int nq, np;
    uint64_t *qoverqiinvqi;  /* [nq]        */
    uint64_t *qoverqimodp;   /* [np][nq]    */
    uint64_t *vtimesqmodp;   /* [np][nq+1]  */
} lo_modup_constants;

typedef struct lo_basis_extender {
    lo_ring *ringQ, *ringP;
    lo_modup_constants **qtop;   /* [nQ] source = Q[:i+1], target = P */
    lo_modup_constants **ptoq;   /* [nP] source = P[:i+1], target = Q */
    uint64_t **moddown_ptoq;     /* [nP][nQ] */
    uint64_t **moddown_qtop;     /* [nQ][nP] */
    int nQ, nP;                  /* table sizes (so that free does not need the rings) */
} lo_basis_extender;

lo_modup_constants *lo_gen_modup_constants(const uint64_t *Q, int nq, const uint64_t *P, int np);
void lo_modup_constants_free(lo_modup_constants *c);
lo_basis_extender *lo_basis_extender_new(lo_ring *ringQ, lo_ring *ringP);
void lo_basis_extender_free(lo_basis_extender *be);
void lo_modup_q_to_p(const lo_basis_extender *be, int levelQ, int levelP, const uint64_t *polQ, uint64_t *polP);
void lo_modup_p_to_q(const lo_basis_extender *be, int levelP, int levelQ, const uint64_t *polP, uint64_t *polQ);
void lo_moddown_qp_to_q(const lo_basis_extender *be, int levelQ, int levelP, const uint64_t *p1Q, const uint64_t *p1P, uint64_t *p2Q);
void lo_moddown_qp_to_q_ntt(const lo_basis_extender *be, int levelQ, int levelP, const uint64_t *p1Q, const uint64_t *p1P, uint64_t *p2Q);
void lo_moddown_qp_to_p(const lo_basis_extender *be, int levelQ, int levelP, const uint64_t *p1Q, const uint64_t *p1P, uint64_t *p2P);

typedef struct lo_decomposer {
    lo_ring *ringQ, *ringP;
    int nlvlP;                       /* ringP.MaxLevel() */
    int *ndigits;                    /* per lvlP */
    int **nconst;                    /* [lvlP][digit] = xnbPi-1 */
    lo_modup_constants ****muc;      /* [lvlP][digit][j] */
} lo_decomposer;
lo_decomposer *lo_decomposer_new(lo_ring *ringQ, lo_ring *ringP);
void lo_decomposer_free(lo_decomposer *d);
void lo_decompose_and_split(const lo_decomposer *d, int levelQ, int levelP, int nbPi, int digit,
                            const uint64_t *p0Q, uint64_t *p1Q, uint64_t *p1P);

/* ---- core/rlwe evaluator --------------------------------------------------- */
/* Evaluation key (GadgetCiphertext, core/rlwe/gadgetciphertext.go:19-42) with
 * BaseTwoDecomposition = 0: value[d][k] = {Q:[nQk][N], P:[nPk][N]}, NTT+Montgomery.
 * Flat layout: q[((d*2+k)*nQk + limb)*N + j], p[((d*2+k)*nPk + limb)*N + j]. */
typedef struct lo_evk {
    int beta;      /* number of stored (RNS digit, bit window) blocks: sum_i nj[i] */
    int nQk, nPk;  /* limbs of the key (levelQ+1, levelP+1) */
    const uint64_t *q, *p;
    int pw2;       /* BaseTwoDecomposition (0 = none)                                       */
    int nj[64];    /* BaseTwoDecompositionVectorSize()[i]: bit windows of RNS digit i;       */
                   /* block (i, j) is stored at index sum_{i'<i} nj[i'] + j (all 1 if pw2=0) */
} lo_evk;

typedef struct lo_evaluator {
    lo_ring *ringQ, *ringP;          /* ringP may be NULL (levelP = -1 unsupported here) */
    lo_basis_extender *be;
    lo_decomposer *dec;
} lo_evaluator;
lo_evaluator *lo_evaluator_new(lo_ring *ringQ, lo_ring *ringP);
void lo_evaluator_free(lo_evaluator *e);
int  lo_base_rns_decomposition_vector_size(int levelQ, int levelP);
/* decompQ: [beta][levelQ+1][N], decompP: [beta][levelP+1][N] */
void lo_decompose_ntt(const lo_evaluator *e, int levelQ, int levelP, int nbPi, const uint64_t *c2, int c2_is_ntt,
                      uint64_t *decompQ, uint64_t *decompP);
/* ctQ: [2][levelQ+1][N], ctP: [2][levelP+1][N]; NTT-domain in/out (IsNTT = true) */
void lo_gadget_product_lazy(const lo_evaluator *e, int levelQ, const uint64_t *cx, const lo_evk *evk,
                            uint64_t *ctQ, uint64_t *ctP);
void lo_gadget_product_hoisted_lazy(const lo_evaluator *e, int levelQ, const uint64_t *decompQ, const uint64_t *decompP,
                                    const lo_evk *evk, uint64_t *ctQ, uint64_t *ctP);
void lo_gadget_product(const lo_evaluator *e, int levelQ, const uint64_t *cx, const lo_evk *evk, uint64_t *ct);
void lo_gadget_product_hoisted(const lo_evaluator *e, int levelQ, const uint64_t *decompQ, const uint64_t *decompP,
                               const lo_evk *evk, uint64_t *ct);
void lo_moddown_ntt(const lo_evaluator *e, int levelQ, int levelP, const uint64_t *ctQ, const uint64_t *ctP, uint64_t *ct);
/* ct_in: [3][level+1][N] -> ct_out: [2][level+1][N] */
void lo_relinearize(const lo_evaluator *e, int level, const uint64_t *ct_in, const lo_evk *rlk, uint64_t *ct_out);
/* ct_in/out: [2][level+1][N], NTT domain */
void lo_automorphism_ct(const lo_evaluator *e, int level, const uint64_t *ct_in, uint64_t galel, const lo_evk *gk, uint64_t *ct_out);
void lo_automorphism_hoisted(const lo_evaluator *e, int level, const uint64_t *ct_in, const uint64_t *decompQ,
                             const uint64_t *decompP, uint64_t galel, const lo_evk *gk, uint64_t *ct_out);

void lo_automorphism_hoisted_lazy(const lo_evaluator *e, int levelQ, const uint64_t *ct_in0, const uint64_t *decompQ,
                                  const uint64_t *decompP, uint64_t galel, const lo_evk *gk, uint64_t *outQ, uint64_t *outP);

/* ---- scheme glue: schemes/ckks/evaluator.go, schemes/bgv/evaluator.go ------- */
/* op0, op1: [2][level+1][N]; out: [3] (relin=0) or [2] (relin=1) */
void lo_ckks_mul_relin(const lo_evaluator *e, int level, const uint64_t *op0, const uint64_t *op1,
                       const lo_evk *rlk, int relin, uint64_t *out);
void lo_bgv_mul_relin(const lo_evaluator *e, int level, uint64_t t, const uint64_t *op0, const uint64_t *op1,
                      const lo_evk *rlk, int relin, uint64_t *out);
/* in: [degree+1][level+1][N] -> out: [degree+1][level+1-nb][N] */
void lo_rescale(const lo_ring *r, int level, int degree, int nb_rescales, const uint64_t *in, uint64_t *out);

/* per-thread scratch pool (the role of ring/pool.go, core/rlwe/pool.go) and the threaded timing loop of bench.py's
 * cpu_baseline leg (b.RunParallel, schemes/ckks/ckks_benchmarks_test.go:95-325) */
void lo_pool_release(void);
double lo_bench_bgv_mul_relin(const lo_evaluator *e, int level, uint64_t t, const uint64_t *op0, const uint64_t *op1,
                              const lo_evk *rlk, int nthreads, double seconds, uint64_t *counts);
/* every entry of a batch through one operation kind of lo_bench_op, on nthreads OS threads (checker for timed batches):
 * op0 / op1 [nb][2][level+1][N] -> out [nb][2][level+1][N] (kinds 0, 1) or [nb][3][level][N] (kind 2) */
void lo_batch_op(const lo_evaluator *e, int kind, int level, uint64_t t, uint64_t gal, const uint64_t *op0, const uint64_t *op1,
                 const lo_evk *key, int nb, int nthreads, uint64_t *out);
/* generalised form: kind 0 BGV MulRelin, 1 Automorphism (Rotate), 2 CKKS Mul + Rescale; pin = one CPU per thread,
 * private_copy = per-thread first-touched copies of the inputs and the key (see lattigo_oracle.c) */
double lo_bench_op(const lo_evaluator *e, int kind, int level, uint64_t t, uint64_t gal, const uint64_t *op0, const uint64_t *op1,
                   const lo_evk *key, int nthreads, double seconds, int pin, int private_copy, uint64_t *counts);

#ifdef __cplusplus
}
#endif
#endif
