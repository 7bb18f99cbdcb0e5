"""TEST SCAFFOLDING (tests/drivers, outside the shipped package lattigo_amd/ since round 5): host-side DRIVERS above the
operator boundary -- not MI355X-native work.

These modules restate, function by function, the control flow of the reference's own Go drivers that sit on
`rlwe.EvaluatorProvider` / `schemes.Evaluator` (circuits/common/lintrans, circuits/common/polynomial, circuits/ckks/mod1,
circuits/ckks/bootstrapping, circuits/ckks/dft, utils/cosine, the scheme evaluators' bookkeeping).  In a Go build the
reference's drivers themselves run unchanged over the cgo shim (go/hering); they exist here only because this image has no
Go toolchain and the parity tests need callers that exercise the device-resident operators (lattigo_amd.ring / .rlwe and the
C ABI in include/hering.h) the way the reference's circuits do.  The native pieces they use -- he_lintrans_mul_sum,
he_centered_lift, he_decomp_fill -- are part of the ABI.
"""
