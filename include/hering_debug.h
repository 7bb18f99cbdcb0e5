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
/* dependent-MRedLazy throughput probe: returns modular multiplies per second */
int he_probe_modmul(he_handle ctx, int iters, double *mults_per_s);

#ifdef __cplusplus
}
#endif
#endif /* HERING_DEBUG_H */
