/*
 * hering_debug.h -- diagnostics of libhering.so that are NOT part of the drop-in boundary
 * (include/hering.h): per-kernel timing used by bench.py's roofline leg and an instruction-rate probe.
 * Nothing in the reference corresponds to these; a Go binding does not need them.
 */
#ifndef HERING_DEBUG_H
#define HERING_DEBUG_H

#include "hering.h"

#ifdef __cplusplus
extern "C" {
#endif

/* per-kernel HIP-event timing on the context's stream: begin, run work, end -> per kernel id
 * launch counts and summed durations (adds two events per launch).  The state is per context and
 * guarded: other contexts may keep launching from other threads meanwhile and are not recorded. */
int he_prof_begin(he_handle ctx);
int he_prof_end(he_handle ctx, int max_kernels, int *counts, float *total_ms, int *n_kernels);
/* the same, with the summed ALGORITHMIC bytes of the recorded launches per kernel id (every polynomial stream a launch must
 * read or write, once; key rows shared by a batch once; twiddles / constants excluded) -- the numerator of the per-kernel
 * roofline figures; computed by the launchers themselves (csrc/kernels.hip, ProfScope) */
int he_prof_end_bytes(he_handle ctx, int max_kernels, int *counts, float *total_ms, double *total_bytes, int *n_kernels);
const char *he_prof_kernel_name(int id);
/* algorithmic bytes of the primitives called on the context since the last reset, by the per-primitive formulas of
 * SURVEY.md section 8(d) (NTT 2L, binary 3L, GadgetProduct 3L + 2 beta (L + alpha), ... limbs of N * 8 bytes, times the batch):
 * out[0] charges an evaluation key to every batch entry (the section's convention), out[1] reads it once per call */
int he_alg_bytes(he_handle ctx, int reset, double out[2]);
/* modular-multiply work of the primitives called on the context since the last reset, in closed form per primitive (SURVEY.md
 * section 8(d): NTT (N/2) logN + N per limb, basis extension L_src x L_dst x N, key inner product 2 beta (L + alpha) N, tensor 6 L N,
 * ...) and by the arithmetic class of the limb: out[0] multiply-equivalents on integer-class limbs (64-bit Montgomery products),
 * out[1] on limbs below 2^47 (exact double-precision products), out[2] / out[3] how many of these are NTT butterflies */
int he_alg_valu(he_handle ctx, int reset, double out[4]);
/* PCI bus id ("0000:05:00.0") of a HIP device of this process: bench.py finds the device's clock tables in sysfs through it
 * (/sys/bus/pci/devices/<id>/pp_dpm_sclk) -- the node's other GPUs are listed there too, whatever this process may see */
int he_debug_device_pci_bus_id(int device, char *out, int len);
/* dependent-MRedLazy throughput probe: returns modular multiplies per second */
int he_probe_modmul(he_handle ctx, int iters, double *mults_per_s);
/* the same for the exact double-precision product the limbs below 2^47 are computed with (error-free product + rounded quotient,
 * csrc/kernels.hip modmul_f64): the second ceiling of bench.py's `roofline.valu` */
int he_probe_modmul_f64(he_handle ctx, int iters, double *mults_per_s);
/* submission queue of an evaluator (he_evaluator_set_coalescing) since its creation: out[0] = single-ciphertext calls served,
 * out[1] = batched launches made for them, out[2] = largest batch, out[3] = calls that ran one by one because their pipeline
 * has launches without entry tables */
int he_evaluator_coalescing_stats(he_handle eval, uint64_t out[4]);
int he_ctx_coalescing_stats(he_handle ctx, uint64_t out[4]);  /* the same counters through the context handle */
/* diagnosis of the queue's gathering rule since the context was created: out[0..2] = batches launched because every recently
 * active caller was waiting / because the oldest request had waited 8 windows / because max_batch requests were pending; out[3],
 * out[4] = microseconds the leaders spent gathering and launching; out[5] / out[6] = sums of callers present / expected at launch;
 * out[7] = allocations the context's buffer cache could not serve (hipMalloc calls); deferred submission (he_ctx_set_deferred):
 * out[8] = microseconds the dispatcher paused because it was four batches ahead of the device, out[9] / out[10] = microseconds it
 * waited for callers while the device had at least two batches queued (free) / while the device was running dry, out[11] = batches;
 * out[12] (HERING_QUEUE_TIMING=1 only) = device microseconds spent inside the dispatcher's batches; out[13] / out[14] = entry
 * tables filled by a launch / reused from one of the context's table slots (same offsets as a recent batch); out[15] reserved */
int he_debug_queue_counters(he_handle ctx, uint64_t out[16]);
/* per operation of the queue (the CoOp numbering of csrc/api.cpp): out[2 i] = batches launched, out[2 i + 1] = requests served */
int he_debug_queue_op_stats(he_handle ctx, uint64_t out[64]);
/* files a request whose launch fails (HE_EDEVICE) after the call was accepted: in the default mode of the queue the call itself returns
 * the failure; under he_ctx_set_deferred the call returns HE_OK and the next he_ctx_sync reports it (tests) */
int he_debug_queue_inject_failure(he_handle ctx);
/* Concurrent single-ciphertext callers, the shape of the reference's parallel benchmarks (b.RunParallel,
 * schemes/ckks/ckks_benchmarks_test.go:116-207): n_threads OS threads (pthreads inside the library: no interpreter in the timed
 * region); thread i makes `iters` calls on its own batch-1 handles -- op 0: he_ckks_mul_relin(eval[i], level, a0[i], a1[i], b0[i],
 * b1[i], rlk[i], o0[i], o1[i], 0); op 1: he_bgv_mul_relin (t = plaintext modulus); op 2: he_automorphism_ct(eval[i], level, a0[i],
 * a1[i], t = Galois element, rlk[i] = Galois key, o0[i], o1[i]); op 3: he_relinearize(eval[i], level, a0[i], a1[i], b0[i], rlk[i],
 * o0[i], o1[i]); op 4: he_gadget_product(eval[i], level, a0[i], rlk[i], o0[i], o1[i]) (unused handle arrays may repeat another).
 * sync_each != 0: every call is followed by he_ctx_sync(ctx[i]) (a caller that needs each result before its next call),
 * otherwise one sync after the last call.  All threads start together; *wall_s = first start to last finish.  Returns the
 * first non-zero status of any call. */
int he_debug_concurrent_mul_relin(int n_threads, int iters, int sync_each, int op, int level, uint64_t t, const he_handle *ctx,
                                  const he_handle *eval, const he_handle *a0, const he_handle *a1, const he_handle *b0,
                                  const he_handle *b1, const he_handle *rlk, const he_handle *o0, const he_handle *o1,
                                  double *wall_s);

/* A whole circuit under the reference's concurrency shape (BenchmarkConcurrentBootstrap: b.RunParallel over bootstrappers,
 * circuits/ckks/bootstrapping/evaluator_benchmarks_test.go:14-42): n_threads OS threads (pthreads inside the library) each replay
 * `rounds` times a RECORDED sequence of calls of the public entry points of hering.h (lattigo_amd/_lib.py trace_begin / trace_end
 * records one run of a driver; encoding: csrc/replay.cpp).  Handles the recorded run created are replaced by the replaying
 * thread's own; the n_subst handles subst_from[] (the run's inputs) are replaced per thread by subst_to[thread * n_subst + i];
 * every other handle (rings, evaluator, keys, plaintexts) is shared.  watch[]: recorded handles whose per-thread counterparts of
 * the LAST round are returned in watch_out[thread * n_watch + i] and left alive (the caller downloads and frees them).  *wall_s:
 * common start to the last thread's final he_ctx_sync.  Returns the first non-zero status; err (optional) receives its message. */
int he_debug_replay(he_handle ctx, const uint64_t *program, size_t n_words, int n_threads, int rounds, const uint64_t *subst_from,
                    int n_subst, const uint64_t *subst_to, const uint64_t *watch, int n_watch, uint64_t *watch_out, double *wall_s,
                    char *err, size_t err_len);

/* where the replaying threads spent their time, per function number of the program encoding (csrc/replay.cpp): out[3 f] =
 * microseconds inside the calls of function f summed over the threads, out[3 f + 1] = calls, out[3 f + 2] = longest call */
int he_debug_replay_profile(uint64_t *out, int n_fn, int reset);

#ifdef __cplusplus
}
#endif
#endif /* HERING_DEBUG_H */
