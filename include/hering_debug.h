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
const char *he_prof_kernel_name(int id);
/* dependent-MRedLazy throughput probe: returns modular multiplies per second */
int he_probe_modmul(he_handle ctx, int iters, double *mults_per_s);

#ifdef __cplusplus
}
#endif
#endif /* HERING_DEBUG_H */
