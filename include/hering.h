/*
 * hering.h -- C ABI of libhering.so: MI355X (gfx950) ring-arithmetic backend for
 * Lattigo-style RNS homomorphic encryption.
 *
 * This is the drop-in boundary.  The reference (tuneinsight/lattigo v6.2.0) is
 * pure Go and has no FFI; these entry points are what a cgo binding of its
 * `ring` / `core/rlwe` operator layer binds (INTEGRATION.md shows the stub).
 * Every entry cites the reference method it replaces (paths relative to the
 * reference tree).
 *
 * Conventions
 *  - plain C: opaque uint64 handles, raw pointers + sizes, no C++/torch types;
 *  - every function returns 0 on success, <0 on error (HE_E*); he_last_error()
 *    returns a thread-local message (the Go shim wraps it into an `error`);
 *    the library never aborts the process;
 *  - host buffers are borrowed for the duration of the call only (cgo's
 *    no-retained-pointer rule);
 *  - polynomials live in device HBM: a poly handle is [batch][limbs][N] uint64,
 *    limb-major like ring.Poly.Coeffs (ring/poly.go:13-15); `level` = "limbs
 *    0..level in use" as in Ring.AtLevel (ring/ring.go:186);
 *  - all work of a context is enqueued on its HIP stream; results become
 *    visible to the host after he_ctx_sync / he_poly_download;
 *  - thread-safe: handles may be used from any OS thread (per-call
 *    hipSetDevice); concurrent calls on ONE context serialize on its stream.
 *  - outputs are caller-allocated and come last; in-place aliasing is allowed
 *    exactly where the reference allows it (coefficient-wise ops, NTT), not for
 *    automorphisms (ring/automorphism.go:37).
 */
#ifndef HERING_H
#define HERING_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t he_handle;

enum {
    HE_OK = 0,
    HE_EINVAL = -1,   /* bad argument / shape mismatch                         */
    HE_EHANDLE = -2,  /* unknown or wrong-type handle                          */
    HE_EDEVICE = -3,  /* HIP runtime error                                     */
    HE_EPARAM = -4,   /* invalid ring parameters (not prime, not 1 mod 2N ...) */
    HE_ENOMEM = -5
};

const char *he_last_error(void);
const char *he_version(void);

/* ---- context ---------------------------------------------------------------- */
int he_device_count(int *n);  /* HIP devices visible to the process (0 without a driver or a device) */
int he_ctx_create(int device_id, he_handle *ctx);
int he_ctx_destroy(he_handle ctx);
int he_ctx_sync(he_handle ctx);
/* HIP-event timing on the context's stream (bench.py's timed region). */
int he_timer_start(he_handle ctx);
int he_timer_stop(he_handle ctx, float *elapsed_ms);
/* device facts for reports: out[0]=CU count, out[1]=LDS bytes/CU, out[2]=clock kHz, out[3]=total HBM bytes */
int he_device_info(he_handle ctx, uint64_t out[4]);

/* ---- replayable launch sequences (hipGraph).  No counterpart in the reference (a Go loop has no launch queue to shorten):
 * between he_graph_begin and he_graph_end the calls made on `ctx` are recorded instead of executed, with the handles they were
 * given; he_graph_launch replays the whole sequence as ONE enqueue on the context's stream -- for latency-bound chains of small
 * calls (a single-ciphertext MulRelin is eleven launches, a bootstrap several thousand).  Contract:
 *  - run the sequence once before capturing it (plans, index tables and the scratch arena are built on first use);
 *  - captured calls must be device work only: he_poly_upload / he_poly_download / he_ctx_sync fail while capturing;
 *  - the replay reads and writes the SAME polynomials: every handle passed to a captured call must outlive the graph (the
 *    temporaries created AND released during the capture are kept for it until he_graph_destroy, and so is the context's
 *    scratch arena); scalars passed by value are frozen at their captured values. */
int he_graph_begin(he_handle ctx);
int he_graph_end(he_handle ctx, he_handle *graph);
int he_graph_launch(he_handle graph);
int he_graph_nodes(he_handle graph, int *nodes);  /* kernel / memset / copy nodes recorded */
int he_graph_destroy(he_handle graph);

/* ---- ring: ring.NewRing (ring/ring.go:207), SubRing tables (ring/subring.go:99-159),
 *      RescaleConstants (ring/ring.go:329).  Standard (negacyclic) type, NthRoot = 2N.
 *      logN in [4, 20] (the reference's MaxLogN, core/rlwe/params.go:21); NTT-friendly prime moduli below 2^61 (the range in
 *      which the reference's own lazy butterflies are exact, ring/ntt.go:169; its GenModuli draws 61-bit primes downstream of
 *      2^61 only, params.go:838).  The fused key-switch pipelines cover logN <= 17; larger rings run the same operations through
 *      the generic passes (three-pass NTT from logN = 19, unfused basis extension), bit-identical results. */
int he_ring_create(he_handle ctx, int logN, const uint64_t *moduli, int n_moduli, he_handle *ring);
/* ring.NewRingFromType (ring/ring.go:267): ring_type 0 = Standard, 1 = ConjugateInvariant
 * (Z[X+X^-1]/(X^2N+1), NthRoot = 4N; NTT of ring/ntt.go:716-1311).  Conjugate-invariant rings are accepted everywhere a
 * standard ring is: ring-level ops, rescale (the NTT variants reproduce the reference's lazy INTT words, which are observable
 * there), automorphisms (index table over NthRoot = 4N, coefficient-domain form of ring/automorphism.go:122-151), basis
 * extenders and evaluators (Q and P of the same type; they take the unfused launches). */
int he_ring_create_type(he_handle ctx, int logN, int ring_type, const uint64_t *moduli, int n_moduli, he_handle *ring);
int he_ring_destroy(he_handle ring);
/* which = 0: Modulus, 1: MRedConstant, 2: BRedConstant[0], 3: BRedConstant[1], 4: NInv, 5: PrimitiveRoot */
int he_ring_constant(he_handle ring, int limb, int which, uint64_t *out);
/* download RootsForward (dir=0) / RootsBackward (dir=1) of one limb: N words */
int he_ring_roots(he_handle ring, int limb, int dir, uint64_t *out);

/* ---- polynomials: ring.Poly (ring/poly.go:13) as device-resident batches -------- */
int he_poly_alloc(he_handle ring, int n_limbs, int batch, he_handle *poly);
/* same shape, contents unspecified: for results and temporaries the next operation overwrites in full (saves the
 * zero-fill launch ring.NewPoly's semantics require; the reference draws its temporaries from a recycling
 * buffer pool without clearing them either, core/rlwe/pool.go:12-60, core/rlwe/evaluator.go:20).  HERING_POISON=1 fills them with a pattern. */
int he_poly_alloc_scratch(he_handle ring, int n_limbs, int batch, he_handle *poly);
int he_poly_free(he_handle poly);
int he_poly_shape(he_handle poly, int *n_limbs, int *batch, int *N);
/* whole-batch transfers of the contiguous [batch][limbs][N] image */
int he_poly_upload(he_handle poly, const uint64_t *src, size_t n_words);
int he_poly_download(he_handle poly, uint64_t *dst, size_t n_words);
/* one limb of one batch entry (what a [][]uint64 row maps to) */
int he_poly_upload_limb(he_handle poly, int b, int limb, const uint64_t *src);
int he_poly_download_limb(he_handle poly, int b, int limb, uint64_t *dst);
int he_poly_copy(he_handle dst, he_handle src, int level);      /* Poly.CopyLvl */
/* limbs 0..level of the batch entries [src_b0, src_b0 + nb) of src -> entries [dst_b0, dst_b0 + nb) of dst (regrouping of
 * independent ciphertexts into one batch, e.g. the real and imaginary halves before EvalMod) */
int he_poly_copy_batch(he_handle dst, int dst_b0, he_handle src, int src_b0, int nb, int level);
int he_poly_zero(he_handle poly);
/* the polynomial's device storage ([batch][n_limbs][N] words) for transports that move device memory themselves (an RCCL
 * all-reduce of partial key-switch accumulators); drains the context's stream first */
int he_poly_device_buffer(he_handle poly, void **ptr, size_t *bytes);

/* ---- NTT: Ring.NTT / NTTLazy / INTT / INTTLazy (ring/ntt.go:127-152) ------------- */
int he_ntt(he_handle ring, int level, he_handle p1, he_handle p2);
int he_ntt_lazy(he_handle ring, int level, he_handle p1, he_handle p2);   /* returns values in [0, 2q) */
int he_intt(he_handle ring, int level, he_handle p1, he_handle p2);
int he_intt_lazy(he_handle ring, int level, he_handle p1, he_handle p2);

/* ring.NumberTheoreticTransformer (ring/ntt.go:17-22: Forward/ForwardLazy/Backward/BackwardLazy(p1, p2 []uint64)),
 * the per-limb host-slice plug point of ring.NewRingWithCustomNTT (ring/ring.go:284): N words in, N words out,
 * one H2D + D2H round trip per call (BASELINE config 1 plumbing; the device-resident entries above are the fast path). */
int he_subring_ntt_host(he_handle ring, int limb, int backward, int lazy, const uint64_t *p1, uint64_t *p2);

/* ---- coefficient-wise ops (ring/operations.go:11-377 over ring/vec_ops.go) --------
 * The numeric codes are the op tables of the reference: one entry per method.    */
enum he_binop {
    HE_ADD = 0,                               /* Ring.Add                          operations.go:11  */
    HE_ADD_LAZY,                              /* Ring.AddLazy                      :18  */
    HE_SUB,                                   /* Ring.Sub                          :25  */
    HE_SUB_LAZY,                              /* Ring.SubLazy                      :32  */
    HE_MUL_COEFFS_BARRETT,                    /* Ring.MulCoeffsBarrett             :60  */
    HE_MUL_COEFFS_BARRETT_LAZY,               /* Ring.MulCoeffsBarrettLazy         :67  */
    HE_MUL_COEFFS_BARRETT_THEN_ADD,           /* Ring.MulCoeffsBarrettThenAdd      :74  */
    HE_MUL_COEFFS_BARRETT_THEN_ADD_LAZY,      /* Ring.MulCoeffsBarrettThenAddLazy  :81  */
    HE_MUL_COEFFS_MONTGOMERY,                 /* Ring.MulCoeffsMontgomery          :88  */
    HE_MUL_COEFFS_MONTGOMERY_LAZY,            /* Ring.MulCoeffsMontgomeryLazy      :95  */
    HE_MUL_COEFFS_MONTGOMERY_LAZY_THEN_NEG,   /* Ring.MulCoeffsMontgomeryLazyThenNeg :102 */
    HE_MUL_COEFFS_MONTGOMERY_THEN_ADD,        /* Ring.MulCoeffsMontgomeryThenAdd   :109 */
    HE_MUL_COEFFS_MONTGOMERY_THEN_ADD_LAZY,   /* Ring.MulCoeffsMontgomeryThenAddLazy :116 */
    HE_MUL_COEFFS_MONTGOMERY_LAZY_THEN_ADD_LAZY, /* Ring.MulCoeffsMontgomeryLazyThenAddLazy :123 */
    HE_MUL_COEFFS_MONTGOMERY_THEN_SUB,        /* Ring.MulCoeffsMontgomeryThenSub   :130 */
    HE_MUL_COEFFS_MONTGOMERY_THEN_SUB_LAZY,   /* Ring.MulCoeffsMontgomeryThenSubLazy :137 */
    HE_MUL_COEFFS_MONTGOMERY_LAZY_THEN_SUB_LAZY, /* Ring.MulCoeffsMontgomeryLazyThenSubLazy :144 */
    HE_BINOP_COUNT
};
enum he_unop {
    HE_NEG = 0,       /* Ring.Neg        operations.go:39  */
    HE_REDUCE,        /* Ring.Reduce     :46  */
    HE_REDUCE_LAZY,   /* Ring.ReduceLazy :53  */
    HE_MFORM,         /* Ring.MForm      :285 */
    HE_MFORM_LAZY,    /* Ring.MFormLazy  :292 */
    HE_IMFORM,        /* Ring.IMForm     :299 */
    HE_UNOP_COUNT
};
enum he_scalarop {
    HE_ADD_SCALAR = 0,        /* Ring.AddScalar        operations.go:151 */
    HE_SUB_SCALAR,            /* Ring.SubScalar        :186 */
    HE_MUL_SCALAR,            /* Ring.MulScalar        :201 */
    HE_MUL_SCALAR_THEN_ADD,   /* Ring.MulScalarThenAdd :208 */
    HE_MUL_SCALAR_THEN_SUB,   /* Ring.MulScalarThenSub :223 */
    HE_SCALAROP_COUNT
};
/* p3 = op(p1, p2 [, p3]) on limbs 0..level of every batch entry */
int he_binop(he_handle ring, int level, int op, he_handle p1, he_handle p2, he_handle p3);
int he_unop(he_handle ring, int level, int op, he_handle p1, he_handle p2);
int he_scalarop(he_handle ring, int level, int op, he_handle p1, uint64_t scalar, he_handle p2);
/* Ring.MulRNSScalarMontgomery (operations.go:216): scalar[i] per limb, Montgomery form */
int he_mul_rns_scalar_montgomery(he_handle ring, int level, he_handle p1, const uint64_t *scalar, he_handle p2);
/* Ring.{Add,Sub,Mul}ScalarBigint (operations.go:158,193,231): little-endian 64-bit words */
int he_add_scalar_bigint(he_handle ring, int level, he_handle p1, const uint64_t *words, int n_words, he_handle p2);
int he_sub_scalar_bigint(he_handle ring, int level, he_handle p1, const uint64_t *words, int n_words, he_handle p2);
int he_mul_scalar_bigint(he_handle ring, int level, he_handle p1, const uint64_t *words, int n_words, he_handle p2);
/* Ring.MulScalarBigintThenAdd (operations.go:240) */
int he_mul_scalar_bigint_then_add(he_handle ring, int level, he_handle p1, const uint64_t *words, int n_words, he_handle p2);
/* Ring.AddDoubleRNSScalar / SubDoubleRNSScalar / MulDoubleRNSScalar / MulDoubleRNSScalarThenAdd (operations.go:166,176,
 * 249,260): scalar0[i] applies to the coefficients [0, N/2) of limb i, scalar1[i] to [N/2, N); plain residues (the
 * Montgomery form of the Mul forms is taken inside, as the reference).  op: 0 add, 1 sub, 2 mul, 3 mul-then-add. */
int he_double_rns_scalarop(he_handle ring, int level, int op, he_handle p1, const uint64_t *scalar0, const uint64_t *scalar1,
                           he_handle p2);
/* Ring.Shift (operations.go:279): p2[i][j] = p1[i][(j + k) mod N]; p2 may be p1 */
int he_shift(he_handle ring, int level, he_handle p1, int k, he_handle p2);
/* Ring.MultByMonomial (operations.go:307): p2 = p1 * X^k, coefficient domain; p2 may be p1 */
int he_mult_by_monomial(he_handle ring, int level, he_handle p1, int k, he_handle p2);
/* Ring.MulByVectorMontgomery / MulByVectorMontgomeryThenAddLazy (operations.go:363,370): every limb of p1 times the
 * same N-word vector (limb 0 of the batch-1 polynomial `vector`) */
int he_mul_by_vector_montgomery(he_handle ring, int level, he_handle p1, he_handle vector, int then_add_lazy, he_handle p2);

/* named wrappers, one per reference method that the key-switch path calls */
int he_add(he_handle ring, int level, he_handle p1, he_handle p2, he_handle p3);
int he_sub(he_handle ring, int level, he_handle p1, he_handle p2, he_handle p3);
int he_neg(he_handle ring, int level, he_handle p1, he_handle p2);
int he_reduce(he_handle ring, int level, he_handle p1, he_handle p2);
int he_mform(he_handle ring, int level, he_handle p1, he_handle p2);
int he_imform(he_handle ring, int level, he_handle p1, he_handle p2);
int he_mul_coeffs_montgomery(he_handle ring, int level, he_handle p1, he_handle p2, he_handle p3);
int he_mul_coeffs_montgomery_then_add(he_handle ring, int level, he_handle p1, he_handle p2, he_handle p3);
int he_mul_coeffs_montgomery_lazy(he_handle ring, int level, he_handle p1, he_handle p2, he_handle p3);
int he_mul_coeffs_montgomery_lazy_then_add_lazy(he_handle ring, int level, he_handle p1, he_handle p2, he_handle p3);

/* ---- rescale: ring/scaling.go --------------------------------------------------- */
/* p1 receives limbs 0..level-nb.  In-place (p0 == p1) is allowed as in the reference. */
int he_div_round_by_last_modulus_ntt(he_handle ring, int level, he_handle p0, he_handle p1);              /* :101 */
int he_div_round_by_last_modulus(he_handle ring, int level, he_handle p0, he_handle p1);                  /* :126 */
int he_div_floor_by_last_modulus_ntt(he_handle ring, int level, he_handle p0, he_handle p1);              /* :6   */
int he_div_floor_by_last_modulus(he_handle ring, int level, he_handle p0, he_handle p1);                  /* :26  */
int he_div_round_by_last_modulus_many_ntt(he_handle ring, int level, int nb, he_handle p0, he_handle p1); /* :148 */
int he_div_round_by_last_modulus_many(he_handle ring, int level, int nb, he_handle p0, he_handle p1);     /* :177 */
int he_div_floor_by_last_modulus_many_ntt(he_handle ring, int level, int nb, he_handle p0, he_handle p1); /* :37  */
int he_div_floor_by_last_modulus_many(he_handle ring, int level, int nb, he_handle p0, he_handle p1);     /* :65  */
/* Evaluator.Rescale's loop over the polynomials of ONE ciphertext (schemes/ckks/evaluator.go:503-507, schemes/bgv/evaluator.go:
 * 1385-1389: DivRoundByLastModulusManyNTT(level, nb, ctIn.Value[i], opOut.Value[i]) for every component) as one call: p0[i] ->
 * p1[i], i < n <= 16.  The same words as n calls of he_div_round_by_last_modulus_many_ntt; on a context whose submission queue is on
 * the n polynomials are filed together and ride in one batch with the other callers' (one round of the queue per Rescale). */
int he_rescale_polys(he_handle ring, int level, int nb, int n, const he_handle *p0, const he_handle *p1);

/* ---- automorphism: ring/automorphism.go ------------------------------------------ */
/* AutomorphismNTTIndex (:12): builds and keeps the index table on the device */
int he_automorphism_index_create(he_handle ring, uint64_t gal_el, he_handle *index);
int he_automorphism_index_destroy(he_handle index);
int he_automorphism_index_download(he_handle index, uint64_t *dst);
int he_automorphism_ntt_with_index(he_handle ring, int level, he_handle pin, he_handle index, he_handle pout);               /* :50 */
int he_automorphism_ntt_with_index_then_add_lazy(he_handle ring, int level, he_handle pin, he_handle index, he_handle pout); /* :82 */
int he_automorphism(he_handle ring, int level, he_handle pin, uint64_t gal_el, he_handle pout);                             /* :113 */

/* ---- basis extension: ring.BasisExtender (ring/basis_extension.go:14) -------------- */
int he_basis_extender_create(he_handle ringQ, he_handle ringP, he_handle *be);
int he_basis_extender_destroy(he_handle be);
int he_modup_q_to_p(he_handle be, int levelQ, int levelP, he_handle polQ, he_handle polP);                          /* :177 */
int he_modup_p_to_q(he_handle be, int levelP, int levelQ, he_handle polP, he_handle polQ);                          /* :195 */
int he_moddown_qp_to_q(he_handle be, int levelQ, int levelP, he_handle p1Q, he_handle p1P, he_handle p2Q);          /* :215 */
int he_moddown_qp_to_q_ntt(he_handle be, int levelQ, int levelP, he_handle p1Q, he_handle p1P, he_handle p2Q);      /* :235 */
int he_moddown_qp_to_p(he_handle be, int levelQ, int levelP, he_handle p1Q, he_handle p1P, he_handle p2P);          /* :262 */

/* ---- rlwe.Evaluator hot path (core/rlwe/evaluator*.go), the EvaluatorProvider
 *      operator interface of core/rlwe/rlwe.go:10-18.  Ciphertext components are
 *      separate poly handles, as rlwe.Ciphertext.Value []ring.Poly; a QP element
 *      is a (Q handle, P handle) pair, as ringqp.Poly{Q,P} (ring/ringqp/poly.go:17). */
/* ringP = 0: parameters without special primes (rlwe.ParametersLiteral.P = nil, core/rlwe/params.go:485): levelP = -1 in
 * every call below.  Such an evaluator serves base-2 gadget keys without a P part -- the reference's own P-less
 * configuration (core/rlwe/test_params.go:36-46) -- through gadgetProductSinglePAndBitDecompLazy's `ringP == nil`
 * branches (evaluator_gadget_product.go:284,302,330) and ModDown's `levelP == -1` copy (:74-96); the P handles of a QP
 * element are then passed as 0.  (With BaseTwoDecomposition = 0 the reference's P-less path is not functional --
 * DecomposeSingleNTT dereferences the nil ringP, :488 -- and is rejected here.) */
int he_evaluator_create(he_handle ringQ, he_handle ringP, he_handle *eval);
int he_evaluator_destroy(he_handle eval);
/* Coalescing of concurrent single-ciphertext calls.  The reference's operator API takes ONE ciphertext per call
 * (schemes/schemes.go:14-28) and scales by running many such calls at once -- one goroutine per ciphertext over evaluators that
 * share their tables and keys (Evaluator.ShallowCopy / WithKey, core/rlwe/evaluator.go:200-227; b.RunParallel in
 * schemes/ckks/ckks_benchmarks_test.go:116-207).  A GPU wants those independent callers in one launch: with max_batch > 1,
 * the single-ciphertext forms (batch-1 polynomials) of EVERY operator entry point of this header -- the ring-level methods
 * (he_ntt / he_intt, he_binop / he_unop / the scalar forms, he_shift, the automorphisms, the eight rescale variants), the basis
 * extender's ModUp / ModDown, all seven rlwe.EvaluatorProvider methods (core/rlwe/rlwe.go:10-18: he_decompose_ntt,
 * he_gadget_product[_lazy / _hoisted / _hoisted_lazy], he_moddown, he_eval_moddown_qp_to_q_ntt, he_relinearize,
 * he_automorphism_ct / _hoisted / _hoisted_lazy), he_ckks_mul_relin / he_bgv_mul_relin with or without a key, he_centered_lift,
 * he_decomp_fill, he_lintrans_mul_sum and he_lintrans_giant_step -- on the context of `eval` are filed in that context's submission queue; requests of the
 * same (operation, object, scalar arguments, key, aliasing pattern of the operands) waiting at the same time -- from any number of
 * OS threads -- are executed as ONE batched launch that addresses each caller's own polynomials and hoisting buffers through a
 * device table of entry offsets (no staging copies), and every call returns once its batch is enqueued on the context's stream
 * (the usual contract: results are visible after he_ctx_sync or a download).  A thread's own calls keep their order.  window_us
 * bounds how long a request waits for companions while the device is idle (a lone caller does not wait at all); while two earlier
 * batches are still in flight, gathering continues for free.  Results are bit-identical to the uncoalesced calls.  max_batch <= 1
 * switches it off.  Handles of a few entries (batch < max_batch: the drivers stack independent ciphertexts -- the two halves of a
 * bootstrap's EvalMod -- into one handle) are queued the same way, a request of nb entries taking nb rows of the entry table; calls on
 * larger handles and calls made while the context records a graph are launched directly as before; shapes
 * whose launches take no entry tables (base-2 gadgets, conjugate-invariant rings, evaluators without special primes, key switches
 * that write onto their own operand) are queued but served one by one.  The queue belongs to the CONTEXT (all objects of a context
 * share one stream): he_ctx_set_coalescing is the same switch addressed through the context handle. */
int he_evaluator_set_coalescing(he_handle eval, int max_batch, int window_us);
int he_ctx_set_coalescing(he_handle ctx, int max_batch, int window_us);
/* Deferred submission (an option of the queue above; depth > 0 switches it on, 0 off).  In the default mode a queued call returns
 * when its batch has been launched: two thread hand-overs per call (caller -> launching thread -> caller) that the device waits out
 * when the launches are short.  Deferred, a queued call checks its arguments, files its request and returns HE_OK at once; a
 * dispatcher thread owned by the context gathers the FIRST pending call of every calling thread, batches the ones that match and
 * launches them, so callers run ahead of the device by up to `depth` requests per thread (a goroutine that issues the calls of a
 * circuit never waits for a launch, as a goroutine of the reference never waits for anything but its own arithmetic).  Contract:
 *  - a thread's calls are launched in the order it made them; anything the same thread launches directly afterwards (handles of
 *    batch > 1, uploads / downloads, key creation, graph capture) first waits for that thread's pending requests;
 *  - he_ctx_sync launches everything filed before it, drains the stream and returns the first failure of a deferred launch, if any
 *    (a failure AFTER the arguments were accepted -- a device error -- can no longer be returned by the call that caused it);
 *  - polynomials may be freed right after the call that uses them (the request holds them);
 *  - data handed from one thread to another needs a he_ctx_sync in between (in the default mode a returned call is already in
 *    stream order; deferred, it may still be waiting behind the other thread's earlier calls).
 * Results are bit-identical to the direct calls.  Requires the queue to be on; he_ctx_set_coalescing(ctx, 0, 0) and he_ctx_destroy
 * end it (pending requests are launched first). */
int he_ctx_set_deferred(he_handle ctx, int depth);

/* GadgetCiphertext (core/rlwe/gadgetciphertext.go:19-42), BaseTwoDecomposition = 0.
 * Host image: q[beta][2][nQk][N], p[beta][2][nPk][N], NTT + Montgomery form.     */
int he_evk_create(he_handle eval, int beta, int nQk, int nPk, const uint64_t *q, const uint64_t *p, he_handle *evk);
/* Base-2 gadget (GadgetCiphertext.BaseTwoDecomposition = pw2 != 0, at most one special prime): one RNS digit per
 * Q-limb i with nj[i] = ceil(bits(q_i)/pw2) bit windows (core/rlwe/params.go:523-540); block (i, j) is stored at
 * index sum_{i'<i} nj[i'] + j.  Host image q[sum nj][2][nQk][N], p[sum nj][2][nPk][N]; nPk = 0 (p = NULL): a key without P part, for
 * an evaluator created without special primes.  Such keys are accepted by
 * he_gadget_product[_lazy], he_relinearize, he_automorphism_ct and the mul_relin entries
 * (gadgetProductSinglePAndBitDecompLazy, core/rlwe/evaluator_gadget_product.go:203); the hoisted entries reject
 * them, as the reference does (:381-383). */
int he_evk_create_base2(he_handle eval, int pw2, const int *nj, int n_rns_digits, int nQk, int nPk, const uint64_t *q,
                        const uint64_t *p, he_handle *evk);
int he_evk_destroy(he_handle evk);

/* Key replication across GPUs (evaluation keys are replicated on every GPU; the reference has no
 * counterpart -- its keys live in one address space, core/rlwe/keys.go:380-460).  he_evk_create / he_evk_create_base2 with
 * q == p == NULL allocate a zeroed key of the given shape; he_evk_device_buffer returns the key's device storage
 * ([beta][2][nQk + nPk][N] words: the q rows then the p rows of each (digit, component)) after draining the context's
 * stream, so that the host can fill it GPU-to-GPU (an RCCL broadcast over xGMI) on a stream of its own; he_evk_commit must
 * follow any external write (it refreshes the derived double-precision copy) once that write has completed. */
int he_evk_device_buffer(he_handle evk, void **ptr, size_t *bytes);
int he_evk_commit(he_handle evk);
/* The same replication with RCCL driven by the library itself, on the context's stream -- no framework in the data path, no
 * import order to respect (librccl.so.1 is loaded at run time and shares this library's HIP runtime).  One process per GPU:
 * rank 0 draws an id (he_rccl_unique_id, HE_RCCL_ID_BYTES bytes) and hands it to the other ranks over any control plane; every
 * rank then calls he_rccl_comm_create(ctx, id, rank, world) -- collective, returns when all ranks have joined -- and
 * he_evk_broadcast(comm, evk, root) on a key of the same shape (an empty one on the peers): the key words go GPU-to-GPU over
 * xGMI into the key's device storage and the derived double-precision copy is refreshed.  he_rccl_comm_ranks all-reduces a one
 * over the communicator: the number of ranks RCCL itself reaches (a self-test for multi-GPU runs).  he_poly_all_reduce_sum adds
 * a polynomial's words across the ranks in place (mod 2^64; the partial accumulators of a key switch split by digit,
 * he_gadget_product_hoisted_lazy_digits -- the caller reduces afterwards). */
#define HE_RCCL_ID_BYTES 128
int he_rccl_available(int *yes);  /* could librccl be loaded in this process?  (ask every rank BEFORE the first collective call) */
int he_rccl_unique_id(uint8_t *id);
int he_rccl_comm_create(he_handle ctx, const uint8_t *id, int rank, int world, he_handle *comm);
int he_rccl_comm_destroy(he_handle comm);
int he_rccl_comm_ranks(he_handle comm, int *ranks);
int he_evk_broadcast(he_handle comm, he_handle evk, int root);
int he_poly_all_reduce_sum(he_handle comm, he_handle poly);
/* the key words back on the host, [beta][2][nQk + nPk][N] (GadgetCiphertext.MarshalBinary's payload order per digit is q then p,
 * core/rlwe/gadgetciphertext.go:110-132) */
int he_evk_download(he_handle evk, uint64_t *dst, size_t n_words);

/* Decomposer.DecomposeAndSplit (ring/basis_extension.go:381): coefficient-domain
 * p0Q -> digit `digit` extended to p1Q (limbs 0..levelQ except, for multi-limb
 * digits, the digit's own) and p1P (limbs 0..levelP). */
int he_decompose_and_split(he_handle eval, int levelQ, int levelP, int nbPi, int digit,
                           he_handle p0Q, he_handle p1Q, he_handle p1P);

/* BuffDecompQP []ringqp.Poly of DecomposeNTT: an opaque device buffer holding, per
 * batch entry, beta digits of (Q limbs, P limbs). */
int he_decomp_create(he_handle eval, int batch, he_handle *decomp);
int he_decomp_destroy(he_handle decomp);
int he_decomp_download_limb(he_handle decomp, int b, int digit, int is_p, int limb, uint64_t *dst);
/* Evaluator.DecomposeNTT (core/rlwe/evaluator_gadget_product.go:459) */
int he_decompose_ntt(he_handle eval, int levelQ, int levelP, int nbPi, he_handle c2, int c2_is_ntt, he_handle decomp);

/* GadgetProductLazy (:108) -> (c0Q,c0P), (c1Q,c1P): NTT domain, canonical */
int he_gadget_product_lazy(he_handle eval, int levelQ, he_handle cx, he_handle evk,
                           he_handle c0Q, he_handle c0P, he_handle c1Q, he_handle c1P);
/* GadgetProductHoistedLazy (:379) */
int he_gadget_product_hoisted_lazy(he_handle eval, int levelQ, he_handle decomp, he_handle evk,
                                   he_handle c0Q, he_handle c0P, he_handle c1Q, he_handle c1P);
/* The same inner product over the digits [digit_begin, digit_end) only: sum_d decomp_d (.) evk_d, canonical.  The sum
 * over a partition of the digits, reduced once more, equals he_gadget_product_hoisted_lazy's output -- the primitive for
 * splitting ONE key switch over several GPUs by digit (each rank holds its digits of the key; the partial accumulators
 * are summed across ranks: lattigo_amd/dist.py SplitGadgetProductHoisted).  An empty range zeroes
 * the outputs.  Same restrictions as he_gadget_product_hoisted_lazy (:379-456). */
int he_gadget_product_hoisted_lazy_digits(he_handle eval, int levelQ, he_handle decomp, he_handle evk, int digit_begin, int digit_end,
                                          he_handle c0Q, he_handle c0P, he_handle c1Q, he_handle c1P);
/* Evaluator.ModDown (:39), NTT in / NTT out; levelP = -1 (no special primes, c0P = c1P = 0): the copy of :76-81 */
int he_moddown(he_handle eval, int levelQ, int levelP, he_handle c0Q, he_handle c0P, he_handle c1Q, he_handle c1P,
               he_handle out0, he_handle out1);
/* BasisExtender.ModDownQPtoQNTT (ring/basis_extension.go:235-256) through the evaluator's fused pipeline (three launches
 * instead of six: the strided NTT stages ride inside the basis extension and the final (x - p1Q) * P^-1 is the epilogue of
 * the forward row pass); p1Q in [0, 2q) as the reference's callers provide it, p2Q may alias p1Q; same canonical result
 * as he_moddown_qp_to_q_ntt. */
int he_eval_moddown_qp_to_q_ntt(he_handle eval, int levelQ, int levelP, he_handle p1Q, he_handle p1P, he_handle p2Q);
/* GadgetProduct (:16) and GadgetProductHoisted (:348) */
int he_gadget_product(he_handle eval, int levelQ, he_handle cx, he_handle evk, he_handle out0, he_handle out1);
int he_gadget_product_hoisted(he_handle eval, int levelQ, he_handle decomp, he_handle evk, he_handle out0, he_handle out1);
/* Evaluator.Relinearize (core/rlwe/evaluator_evaluationkey.go:117) */
int he_relinearize(he_handle eval, int level, he_handle in0, he_handle in1, he_handle in2, he_handle rlk,
                   he_handle out0, he_handle out1);
/* Evaluator.Automorphism (core/rlwe/evaluator_automorphism.go:13), NTT domain */
int he_automorphism_ct(he_handle eval, int level, he_handle in0, he_handle in1, uint64_t gal_el, he_handle gk,
                       he_handle out0, he_handle out1);
/* Evaluator.AutomorphismHoisted (:60): decomp = DecomposeNTT(in1) */
int he_automorphism_hoisted(he_handle eval, int level, he_handle in0, he_handle decomp, uint64_t gal_el, he_handle gk,
                            he_handle out0, he_handle out1);

/* EvaluatorProvider.AutomorphismHoistedLazy (core/rlwe/evaluator_automorphism.go:104): in0 = ctIn.Value[0],
 * decomp = DecomposeNTT(ctIn.Value[1]); output in QP, NTT domain, not divided by P */
int he_automorphism_hoisted_lazy(he_handle eval, int levelQ, he_handle in0, he_handle decomp, uint64_t gal_el, he_handle gk,
                                 he_handle c0Q, he_handle c0P, he_handle c1Q, he_handle c1P);

/* ---- scheme call sites ------------------------------------------------------------ */
/* CKKS Evaluator.mulRelin (schemes/ckks/evaluator.go:764), degree 1 x degree 1.
 * rlk = 0 -> no relinearisation: (out0,out1,out2); else (out0,out1), out2 ignored.  */
int he_ckks_mul_relin(he_handle eval, int level, he_handle a0, he_handle a1, he_handle b0, he_handle b1,
                      he_handle rlk, he_handle out0, he_handle out1, he_handle out2);
/* BGV Evaluator.tensorStandard (schemes/bgv/evaluator.go:592), plaintext modulus t */
int he_bgv_mul_relin(he_handle eval, int level, uint64_t t, he_handle a0, he_handle a1, he_handle b0, he_handle b1,
                     he_handle rlk, he_handle out0, he_handle out1, he_handle out2);
/* CKKS / BGV Evaluator.Rescale (ckks :477, bgv :1363) is, per ciphertext component,
 * he_div_round_by_last_modulus_many_ntt above. */

/* ---- pieces of circuits/ckks/bootstrapping Evaluator.ModUp ------------------------------------ */
/* The centred lifts of bootstrapping.Evaluator.ModUp (circuits/ckks/bootstrapping/evaluator.go:654-667, 677-696,
 * 742-755): c = src[limb 0][j] (coefficient domain, modulus q = Q[0]); neg = strict ? c > q/2 : c >= q/2; c = neg ?
 * q - c : c; every destination limb i gets t = BRedAdd(c, m_i), neg ? m_i - t : t -- Q limbs first_q..levelQ of dstQ
 * and, when levelP >= 0, P limbs 0..levelP of dstP.  dstQ may be src itself (limbs >= 1 are written).
 * strict = 3: the small-norm form of ringqp.Ring.ExtendBasisSmallNormAndCenter (ring/ringqp/operations.go:325): sign test
 * c > q/2 and |c| written without reduction (first_q > levelQ writes no Q limb). */
int he_centered_lift(he_handle eval, int strict, he_handle src, int first_q, int levelQ, he_handle dstQ, int levelP,
                     he_handle dstP);
/* Every digit of the hoisting buffer := (srcQ limbs 0..levelQ, srcP limbs 0..levelP), the way bootstrapping.ModUp fills
 * BuffDecompQP for the sparse-to-dense key switch (evaluator.go:699-718). */
int he_decomp_fill(he_handle decomp, int levelQ, int levelP, he_handle srcQ, he_handle srcP);

/* ---- fused driver step for circuits/common/lintrans --------------------------------------- */
/* Inner accumulation of lintrans.Evaluator.MultiplyByDiagMatrixBSGS / MultiplyByDiagMatrix
 * (circuits/common/lintrans/lintrans_evaluator.go:346-394 and :216-241): for k = 0,1
 *   out_k = Reduce( [out_k +] sum_{i<n} MulCoeffsMontgomeryLazy(pt_i, phi_i(ct_i[k])) )   over Q limbs 0..levelQ and
 * P limbs 0..levelP, i.e. the canonical value of the reference's ringqp MulCoeffsMontgomeryLazy[ThenAddLazy] / Reduce
 * chain (resp. MulCoeffsMontgomery[ThenAdd]).  Arrays have n entries (n <= 64).  ptQ/ptP: encoded diagonals (batch 1
 * broadcasts).  ctkP[i] == 0: term i has no P part (the P*ct term of the zero rotation, :349-352).  index[i] != 0: the
 * ciphertext of term i is read through that automorphism index (fuses AutomorphismNTTWithIndex, :224-225);
 * index == NULL: no automorphisms.  accumulate != 0 adds to the current (canonical) content of out. */
int he_lintrans_mul_sum(he_handle eval, int levelQ, int levelP, int n, const he_handle *ptQ, const he_handle *ptP,
                        const he_handle *ct0Q, const he_handle *ct0P, const he_handle *ct1Q, const he_handle *ct1P,
                        const he_handle *index, int accumulate, he_handle out0Q, he_handle out0P, he_handle out1Q,
                        he_handle out1P);
/* The giant step of lintrans.Evaluator.MultiplyByDiagMatrixBSGS (circuits/common/lintrans/lintrans_evaluator.go:397-441) as ONE
 * call -- the calls the reference makes after the inner loop of giant step j != 0, in its order:
 *   GadgetProductLazy(levelQ, cx, key, cQP)                                      (:412, core/rlwe/evaluator_gadget_product.go:45)
 *   ringQP.Add(cQP[0], (addQ, addP), cQP[0])                                     (:413)
 *   ringQP.AutomorphismNTTWithIndex(cQP[k], index_gal, out_k)                    (:419-420; accumulate == 0)
 *   ringQP.AutomorphismNTTWithIndexThenAddLazy(cQP[k], index_gal, out_k)         (:422-423; accumulate != 0: out_k += ..., no reduction)
 * with cx = the (ModDown'ed) inner sum of component 1, (addQ, addP) the inner sum of component 0 on Q and P, gal the Galois element
 * of the giant step's rotation, key its Galois key, out_k = (ckQ, ckP) the outer accumulators.  Word for word what the separate
 * calls produce; the intermediate ciphertext cQP is never materialised where the key inner products can store through the
 * automorphism themselves (standard rings, RNS gadgets).  The outputs must not alias the inputs. */
int he_lintrans_giant_step(he_handle eval, int levelQ, he_handle cx, he_handle key, uint64_t gal, he_handle addQ, he_handle addP,
                           he_handle c0Q, he_handle c0P, he_handle c1Q, he_handle c1P, int accumulate);

#ifdef __cplusplus
}
#endif
#endif /* HERING_H */
