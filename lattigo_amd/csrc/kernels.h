// kernels.h -- host-callable launchers of the gfx950 kernels (implemented in kernels.hip).
// All launchers enqueue on the given stream and return the hipError_t of the launch.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#include "modarith.h"

namespace he {

// run-time switch HERING_* (DESIGN.md section 9): set and non-zero
inline bool env_flag(const char *name) {
    const char *v = std::getenv(name);
    return v != nullptr && std::atoi(v) != 0;
}
constexpr int kMaxLimbs = 64;  // limbs addressed by one launch
constexpr int kMaxLogN = 20;   // the reference's MaxLogN (core/rlwe/params.go:21); the fused key-switch pipelines cover logN <= 17

// Which limbs a launch touches: entry y of the grid's y-dimension reads limb in_limb[y]
// of the input view(s), writes limb out_limb[y] and uses modulus record mod[y].
struct LimbTab {
    int n;
    uint8_t in_limb[kMaxLimbs];
    uint8_t out_limb[kMaxLimbs];
    uint8_t mod[kMaxLimbs];
};

// A device-resident polynomial batch: limb stride N words, batch stride bstride words.
// tab (optional): an entry table -- batch entry z lives at p + tab[z] words (a device array of word offsets, any allocation;
// differences wrap modulo 2^64) instead of p + z * bstride.  That is how concurrent single-ciphertext calls are coalesced into
// one batched launch over the callers' own, unrelated polynomials (he_evaluator_set_coalescing).  Only the launchers that
// say so accept it -- the caller-facing operands of the fused MulRelin / key-switch pipelines: launch_tensor; the input, the
// product prologue and the epilogues of launch_ntt_rows; the own-digit operand of launch_ks_inner / launch_ntt_mac_f64 and the
// latter's epilogue; the output of launch_gather -- every other launcher refuses a view that carries one.
struct View {
    uint64_t *p;
    size_t bstride;
    const size_t *tab = nullptr;
};
// fills a device entry table from host values (kernel arguments: no host buffer has to outlive the call)
hipError_t launch_tab_fill(size_t *dst, const size_t *vals, int n, hipStream_t s);

struct RingDev {
    int logN;
    int N;
    const ModConst *mc;      // [n_mod]
    const uint64_t *tw_fwd;  // [n_mod][N] RootsForward  (Montgomery form, bit-reversed order)
    const uint64_t *tw_inv;  // [n_mod][N] RootsBackward
    // HOST array [n_mod], kernel class per modulus: 2 = below 2^47 (double-precision row kernel),
    // 1 = below 2^58 (correction-free integer butterflies), 0 = generic
    const uint8_t *host_small;
    const double *twd_fwd;      // [n_mod][N] plain twiddles as doubles (entries of class-2 moduli only), or null
    const double *twd_inv;
    // [n_mod][16][2]: {w, floor(w 2^64 / q)} for the plain (non-Montgomery) forward twiddles RootsForward[0..15] -- the column
    // stages fused into the basis extension use Shoup products on the integer path; null when not built
    const uint64_t *tws_fwd = nullptr;
};

// ---- NTT ---------------------------------------------------------------------------
enum NttFlags {
    NTT_REDUCE_INPUT = 1,  // inputs are arbitrary 64-bit words: bring them to [0,2q) first
    NTT_LAZY_OUT = 2,      // forward: leave the output in [0,2q) instead of [0,q)
    NTT_ADD_SCALAR = 4,    // (set by launch_ntt when io_scalar is given)
    NTT_INPUT_F64 = 8,     // launch_ntt_rows, forward: the limbs of the double-precision class hold doubles (launch_modup_fused f64_raw)
};
struct NttEpilogue;
// io_scalar (optional, one word per launch limb): forward -- added to the input words before the transform (lazy, then reduced when
// NTT_REDUCE_INPUT is set); inverse -- out = CRed(INTT(in) + s).  epi (optional, forward only): the epilogue of
// launch_ntt_rows on the last pass.  Together they make DivRoundByLastModulusNTT two transforms (ring/scaling.go:101-122).
hipError_t launch_ntt(const RingDev &r, const LimbTab &tab, View in, View out, int batch, bool inverse, int flags,
                      hipStream_t s, const uint64_t *io_scalar = nullptr, const NttEpilogue *epi = nullptr);
// only the contiguous-row pass (the last min(logN,12) forward stages / the first ones of the inverse);
// the strided column stages are then done by the producer / consumer kernel (launch_modup_fused).
// For logN <= 12 this is the whole transform (inverse: N^-1 included).
// Optional epilogue of the forward row pass = the last op of ModDownQPtoQNTT (ring/basis_extension.go:252-255),
// optionally fused with the Ring.Add every key-switch caller applies next:
//   out = [w +] MRed(NTT(in) + 2q - y, s[limb])      (y, w, out indexed by tab.out_limb)
struct NttEpilogue {
    View y, w;
    bool has_w;
    uint64_t s[kMaxLimbs];
    // the limbs handled by the double-precision kernel hold y as IEEE doubles (integers, |y| < q) -- the accumulators
    // ntt_mac_f64 wrote with q_out_f64
    bool y_small_f64 = false;
    // y is caller-supplied (arbitrary 64-bit words): the double-precision kernel reduces it before converting
    bool y_reduce = false;
    // optional second output set: batch entries >= zsplit use (out2, y2, w2) with index z - zsplit
    // (both components of a ciphertext in one launch)
    int zsplit = 0;
    View out2, y2, w2;
    bool has_w2 = false;
    // tensor mode (fused MulRelin; launch_ntt_rows with zsplit = B): the addend of a component is formed from the four
    // inputs of the ciphertext product instead of being read back -- w0 = T(a0, b0), w1 = CRed(T(a0, b1) + T(a1, b0)),
    // T(x, y) = MRed(MRed(x, ts[limb]), y) (schemes/bgv/evaluator.go:634-647) -- and the two components of an entry are given
    // workgroups eight apart in launch order (the same XCD, back to back), so that the second one finds a0 / b0 in that L2
    bool tensor = false;
    View ta0, ta1, tb0, tb1;
    uint64_t ts[kMaxLimbs];
    // launch_ntt only: the epilogue pass writes here instead of `out` (which then only carries the column pass's
    // intermediate), so that dst may alias y
    bool has_dst = false;
    View dst;
    // launch_ntt_rows, not with `tensor`: the result is stored through the NTT-domain automorphism of Galois element g (ring/
    // automorphism.go:50-77: coefficient e of the transform lands where out[j] = in[index_g[j]] reads it); scatter_ginv = g^-1 mod 2N
    // (standard ring), 0 = plain stores.  Outputs must then not alias the addends (a thread reads w at e and writes elsewhere).
    uint32_t scatter_ginv = 0;
};
// Optional prologue of the INVERSE row pass over the limbs of the double-precision class (production row sizes): the input is
// formed in the kernel as the degree-2 term of a ciphertext product, c = T(a, b) = MRed(MRed(a, ts[limb]), b), canonical, and
// also written to `c` (limb-indexed like `in`) -- the tensor pass over those limbs and the read-back of its output disappear.
// The integer-class limbs of the launch read `in` as usual (the caller's tensor pass covers them).  ts by tab position.
struct NttProdIn {
    View a, b, c;
    uint64_t ts[kMaxLimbs];
};
int ntt_row_bits(int logN);  // row stages of the two-pass transform (the rest are column stages)
bool ntt_prod_in_supported(int logN);
bool epilogue_scatter_supported(int logN);  // NttEpilogue::scatter_ginv / NttMacEpilogue::scatter_ginv: the production row sizes
hipError_t launch_ntt_rows(const RingDev &r, const LimbTab &tab, View in, View out, int batch, bool inverse, int flags,
                           hipStream_t s, const NttEpilogue *epi = nullptr, const NttProdIn *prod = nullptr);

// conjugate-invariant fold (ring/ntt.go:764-769 forward, :1146-1151 backward), pairs (j, N-j) per thread:
//   forward : out[j] = in[j] + 2q - MRedLazy(in[N-j], F), out[0] = in[0]          (F = ModConst.pad0)
//   backward: out[j] = CRed(in[j] + q - MRed(in[N-j], F)), out[0] = 2*in[0] mod q  (F = ModConst.pad1, in canonical)
hipError_t launch_ci_fold(const RingDev &r, const LimbTab &tab, View in, View out, int batch, bool inverse, bool reduce_input,
                          hipStream_t s);

// INTTConjugateInvariantLazy of ONE limb with the reference's exact lazy words (see kernels.hip); `in` / `out` address that limb
// (limb stride folded into the pointer), mc_host is the host copy of the modulus record
hipError_t launch_ci_intt_lazy_ref(const RingDev &r, const ModConst &mc_host, int mod, View in, View out, int batch, hipStream_t s);

// ---- coefficient-wise -----------------------------------------------------------------
// op codes: 0..16 = he_binop, 100.. = he_unop, 200.. = scalar forms (scalar per limb in sc[])
enum EwOp {
    EW_ADD = 0, EW_ADD_LAZY, EW_SUB, EW_SUB_LAZY,
    EW_MUL_BARRETT, EW_MUL_BARRETT_LAZY, EW_MUL_BARRETT_THEN_ADD, EW_MUL_BARRETT_THEN_ADD_LAZY,
    EW_MUL_MONT, EW_MUL_MONT_LAZY, EW_MUL_MONT_LAZY_THEN_NEG,
    EW_MUL_MONT_THEN_ADD, EW_MUL_MONT_THEN_ADD_LAZY, EW_MUL_MONT_LAZY_THEN_ADD_LAZY,
    EW_MUL_MONT_THEN_SUB, EW_MUL_MONT_THEN_SUB_LAZY, EW_MUL_MONT_LAZY_THEN_SUB_LAZY,
    EW_NEG = 100, EW_REDUCE, EW_REDUCE_LAZY, EW_MFORM, EW_MFORM_LAZY, EW_IMFORM, EW_COPY,
    EW_ZERO,                       // 0 (no input read: the zero fill of a batch of fresh polynomials through an entry table)
    EW_ADD_SCALAR = 200,           // CRed(x + s)
    EW_SUB_SCALAR,                 // CRed(x + q - s)
    EW_MUL_SCALAR_MONT,            // MRed(x, s)
    EW_MUL_SCALAR_MONT_THEN_ADD,   // CRed(z + MRed(x, s))
    EW_ADD_SCALAR_LAZY,            // x + s
    // rescale / moddown fused forms (two-limb inputs)
    EW_SUB_THEN_MUL_SCALAR_MONT_2Q = 300,  // z = MRed(2q - y + x, s)          vec_ops.go:766
    EW_DIVROUND_COEFF,                     // z = MRed(x + (s0 + 2q - y), s)    scaling.go:138-142 (x = top limb + pHalf)
    EW_SUBMUL2Q_THEN_ADD,                  // z = CRed(z + MRed(2q - y + x, s)): ModDown epilogue fused with Ring.Add
};
struct ScalarTab {
    uint64_t s[kMaxLimbs];   // per launch-limb scalar
    uint64_t s2[kMaxLimbs];  // second scalar where needed
};
// z[out_limb] = op(x[in_limb], y[in_limb or y_limb], z)   (y uses tab.in_limb unless y_tab given)
hipError_t launch_ew(const RingDev &r, const LimbTab &tab, int op, View x, View y, View z, int batch,
                     const ScalarTab *sc, const uint8_t *x_limb_override, hipStream_t s);
// "double scalar" forms (Ring.{Add,Sub,Mul}DoubleRNSScalar, ring/operations.go:166-184,249-268): sc->s for the
// coefficients [0, N/2), sc->s2 for [N/2, N); op is one of the scalar EW ops
hipError_t launch_ew_double(const RingDev &r, const LimbTab &tab, int op, View x, View z, int batch, const ScalarTab *sc,
                            hipStream_t s);
// Ring.Shift (operations.go:279: out[j] = in[(j + k) mod N], k already reduced to [0, N)) and Ring.MultByMonomial
// (:307: p2 = p1 * X^k in Z[X]/(X^N+1), word-exact incl. the q - 0 = q representative), shift in [0, 2N); not in place
hipError_t launch_shift(const RingDev &r, const LimbTab &tab, View in, int k, View out, int batch, hipStream_t s);
hipError_t launch_mult_by_monomial(const RingDev &r, const LimbTab &tab, View in, int shift, View out, int batch, hipStream_t s);
// same with a separate addend view w for the *_THEN_ADD forms (z = f(x, y) + w)
hipError_t launch_ew_w(const RingDev &r, const LimbTab &tab, int op, View x, View y, View w, View z, int batch,
                       const ScalarTab *sc, hipStream_t s);

// ---- automorphism ------------------------------------------------------------------------
hipError_t launch_gather(const RingDev &r, const LimbTab &tab, View in, const uint32_t *index, View out, int batch,
                         bool then_add, hipStream_t s);
// coefficient-domain automorphism X^i -> X^(i*gal) (ring/automorphism.go:153-174)
hipError_t launch_automorphism_coeff(const RingDev &r, const LimbTab &tab, View in, uint64_t gal, View out, int batch,
                                     hipStream_t s, bool conjugate_invariant = false);
// lognth = log2(NthRoot) - 1: logN for the standard ring, logN + 1 for the conjugate-invariant one
hipError_t launch_build_automorphism_index(int logN, int lognth, uint64_t gal, uint32_t *index, hipStream_t s);

// ---- basis extension -----------------------------------------------------------------------
// One constant set of GenModUpConstants (ring/basis_extension.go:101) resident on the device.
struct ModUpDev {
    int nsrc, ndst;
    const uint64_t *a;         // [nsrc]           qoverqiinvqi
    const uint64_t *T;         // [ndst][nsrc]     qoverqimodp
    const uint64_t *vt;        // [ndst][nsrc+1]   vtimesqmodp
};
struct ModUpArgs {
    int nsrc, ndst;
    uint8_t src_limb[32], src_mod[32];
    uint64_t src_half[32];               // added (CRed) before reconstruction; 0 = none
    uint8_t dst_limb[kMaxLimbs], dst_mod[kMaxLimbs], dst_row[kMaxLimbs];
    uint64_t dst_half[kMaxLimbs];        // subtracted (CRed(x + p - half)) after
    uint8_t dst_view[kMaxLimbs];         // 0 -> dstA, 1 -> dstB
};
hipError_t launch_modup(const RingDev &r, const ModUpDev &c, const ModUpArgs &a, View src, View dstA, View dstB,
                        int batch, hipStream_t s);
// single-limb digit: centred copy into every listed limb (ring/basis_extension.go:402-436)
// strict bit 0: the sign test is c > q/2 instead of c >= q/2 (circuits/ckks/bootstrapping/evaluator.go:681 vs :657);
// bit 1: |c| is written unreduced (ringqp.Ring.ExtendBasisSmallNormAndCenter, ring/ringqp/operations.go:325)
hipError_t launch_center_copy(const RingDev &r, const ModUpArgs &a, View src, View dstA, View dstB, int batch,
                              hipStream_t s, int strict = 0);

// Fused basis extension for the key-switch pipelines: [last `a` inverse-NTT stages + N^-1 on the
// sources] -> ModUpExact (or the centred copy of a one-limb digit) -> [first `a` forward-NTT stages on
// every destination limb], a = logN - 12.  Sources are the output of the inverse ROWS pass, destinations
// feed the forward ROWS pass, so the strided "column" passes never touch HBM on their own.
// One descriptor per digit, resident in device memory (built once per (levelQ, levelP) plan).
#ifndef HE_MODUP_MAGIC
#define HE_MODUP_MAGIC 1  // split residues: exact running sum of the products, one reduction per coefficient (0: one modmul_f64 each)
#endif
// a source residue of 2^51 and above is split y = yh 2^kYSplitBits + yl for the double-precision destinations (both halves fit
// 32 bits for moduli below 2^61)
constexpr int kYSplitBits = HE_MODUP_MAGIC ? 29 : 26;
#ifndef HE_MODUP_R60
#define HE_MODUP_R60 1  // 0: the lean integer destinations always assemble their 128-bit sum (dst_fast 1)
#endif
struct ModUpDesc {
    int nsrc, ndst, single, reduce_out;
    const uint64_t *a, *T, *vt;
    // double-precision copies for destination moduli below 2^47: Td[row][i] = {T, T*2^kYSplitBits mod p} (plain integers),
    // vtd[row][v] = vt; a source residue y >= 2^51 is split as y = yh*2^kYSplitBits + yl (src_split[i])
    const double *Td, *vtd;
    // lean integer path (destination moduli below 2^58, see modup_fused_kernel): fc[row] = {vt[row][1] 2^64 mod p,
    // (p - dst_half) 2^64 mod p}; dst_fast[j] marks the destinations that take it
    const uint64_t *fc;
    // dst_fast = 3: the lean path with the sum reduced at radix 2^30; t60[row] = {T[i] 2^60 mod p (nsrc of them), vt[1] 2^60 mod p,
    // (p - dst_half) 2^60 mod p}
    const uint64_t *t60;
    uint8_t dst_fast[kMaxLimbs];
    uint8_t src_split[8];
    uint64_t src_half[8];
    uint8_t src_limb[8], src_mod[8];
    size_t dst_off;  // words added to the destination bases (digit block)
    uint8_t dst_limb[kMaxLimbs], dst_mod[kMaxLimbs], dst_row[kMaxLimbs], dst_view[kMaxLimbs];
    uint64_t dst_half[kMaxLimbs];
};
// all descriptors must share nsrc; returns hipErrorInvalidValue when (nsrc, logN) has no fused kernel
bool modup_fused_supported(int logN, int nsrc);
// dst_classes: bit 0 = some destination modulus is >= 2^47 (integer path), bit 1 = some is below (double path)
// total_limbs: source + destination limbs over the descriptors (the launch's algorithmic traffic, for the profiling leg)
hipError_t launch_modup_fused(const RingDev &r, const ModUpDesc *descs_dev, int ndesc, int nsrc, int dst_classes, View src,
                              View dstA, View dstB, int batch, hipStream_t s, bool f64_raw = false, int total_limbs = 0);
// f64_raw is exact only while the unreduced doubles stay below 2^53 through the remaining forward stages
bool modup_f64_raw_ok(int logN, int nsrc, uint64_t max_small_modulus);

// base-2 gadget decomposition (ring.MaskVec, ring/vec_ops.go:870, as used by
// core/rlwe/evaluator_gadget_product.go:256-258): block b = (RNS digit i, window j) gets
// (src[limb_i] >> shift_b) & mask replicated into every listed destination limb
struct MaskSpreadArgs {
    int nblk, ndst;
    uint64_t mask;
    uint8_t blk_limb[256], blk_shift[256];
    uint8_t dst_limb[kMaxLimbs];
};
hipError_t launch_mask_spread(const RingDev &r, const MaskSpreadArgs &a, View src, uint64_t *dec, size_t dec_bs, size_t dec_ds,
                              int batch, hipStream_t s);

// ---- key-switch inner product ---------------------------------------------------------------
// acc[k][l] = sum_d evk[d][k][l] * dec[d][l] * 2^-64 mod q_l, canonical
// (core/rlwe/evaluator_gadget_product.go:160-200 after its final Reduce).
struct KsArgs {
    int beta;
    int nlimbs;                       // launch limbs
    uint8_t dec_limb[kMaxLimbs];      // limb inside one digit block of dec
    uint8_t key_limb[kMaxLimbs];      // limb inside one (d,k) block of the key
    uint8_t out_limb[kMaxLimbs];
    uint8_t out_view[kMaxLimbs];      // 0 -> Q outputs, 1 -> P outputs
    uint8_t mod[kMaxLimbs];
    size_t dec_dstride;               // words between digits in dec
    size_t key_kstride;               // words between k=0 and k=1 blocks
    size_t key_dstride;               // words between digits in the key
    // optional: the digit's own limbs are read straight from the NTT-domain input instead of dec
    // (core/rlwe/evaluator_gadget_product.go:498-503 copies them); alpha = 0 disables
    int own_alpha;                    // digit d owns Q limbs [d*alpha, (d+1)*alpha)
    int own_nq;                       // launch limbs < own_nq are Q limbs with limb index == launch index
};
// Optional (AutomorphismHoistedLazy as ONE launch, core/rlwe/evaluator_automorphism.go:104-165): the accumulators are stored
// through the NTT-domain automorphism whose inverse Galois element is ginv (see NttEpilogue::scatter_ginv), and component 0 of
// the Q limbs is first increased by MRed(add0, add_s[launch limb]) -- the ctIn[0] * P term -- read at the source position.
// Round 6 -- the giant step of a baby-step giant-step linear transformation (circuits/common/lintrans/lintrans_evaluator.go:
// 397-441: GadgetProductLazy, ringQP.Add of the inner loop's component-0 sum, AutomorphismNTTWithIndex[ThenAddLazy] of both
// components into the outer accumulators) as the producers' own stores: `plain` -- the addend (add0 on the Q limbs, add0P on
// the P limbs) is added as it is, CRed(acc + add); `accumulate` -- the destination word is increased (no reduction, as
// ...ThenAddLazy) instead of overwritten.  launch_ntt_mac_f64 takes the same description for the double-precision limbs.
struct KsScatter {
    uint32_t ginv = 0;
    View add0{nullptr, 0};
    uint64_t add_s[kMaxLimbs];
    int plain = 0;
    View add0P{nullptr, 0};
    int accumulate = 0;
};
hipError_t launch_ks_inner(const RingDev &r, const KsArgs &a, View dec, View own, const uint64_t *key, View out0Q,
                           View out0P, View out1Q, View out1P, int batch, hipStream_t s, const KsScatter *sc = nullptr);

// Plaintext-diagonal x ciphertext multiply-accumulate (inner loop of lintrans, circuits/common/lintrans/
// lintrans_evaluator.go:346-394): out_k = Reduce(prev_k + sum_i MulCoeffsMontgomeryLazy(pt_i, ct_i[k])), k = 0,1, for the
// limbs [0, nlimbs) of one ring; exact 128-bit accumulation and one Montgomery reduction, i.e. the canonical value of the
// reference's lazy accumulation after its final Reduce.  pt_i is shared by the batch (bstride 0) or per entry; a term may
// be read through an automorphism index (gather), which fuses AutomorphismNTTWithIndex into the product.
constexpr int kMaxDiag = 64;
struct DiagMacArgs {
    int n, nlimbs, accumulate;
    int mod0;                                                    // modulus record of limb 0 in the ring's table
    const uint64_t *pt[kMaxDiag], *c0[kMaxDiag], *c1[kMaxDiag];  // c0/c1 null: the term is skipped on this ring
    const uint32_t *index[kMaxDiag];                             // null: identity
    size_t pt_bs[kMaxDiag], c0_bs[kMaxDiag], c1_bs[kMaxDiag];
    // optional entry tables of the terms' operands (View::tab for a term list): row term_rows * i + {0: pt, 1: c0, 2: c1} of
    // term_tab holds, per batch entry, the word offsets from the term's base pointers; replaces the batch strides
    const size_t *term_tab = nullptr;
    int term_rows = 3;
};
hipError_t launch_diag_mac(const RingDev &r, const DiagMacArgs &a, View out0, View out1, int batch, hipStream_t s);

// Fused "forward row NTT + key multiply-accumulate" for limbs below 2^47 (double-precision path): one workgroup
// owns (row, limb, batch entry) and loops over the beta digits; every non-own digit's row is transformed in LDS and
// multiplied by the two key rows straight out of LDS, the own digit's row is read from the NTT-domain input, the two
// accumulators stay in registers and are written once.  The transformed decomposition never goes to HBM.
struct NttMacArgs {
    int beta, nlimbs;
    uint8_t dec_limb[kMaxLimbs], key_limb[kMaxLimbs], out_limb[kMaxLimbs], out_view[kMaxLimbs], mod[kMaxLimbs];
    size_t dec_dstride, key_kstride, key_dstride;
    int own_alpha, own_nq;
    int dec_f64;     // the decomposed (non-own) words are doubles (launch_modup_fused f64_raw)
    int own_reduce;  // the own-digit words are caller-supplied (any uint64): reduce them before the conversion to double
    int q_out_f64;   // Q-limb accumulators are written as IEEE doubles (exact integers, |x| < q) for the f64 ModDown epilogue
};
// ModDown epilogue fused into the NTT + MAC kernel (production row sizes; every limb of the launch a Q limb of the
// double-precision class): after the key inner product of a (limb, row) the kernel transforms that row of the basis-extended P
// part of BOTH accumulators -- `ext`, [2 * batch] entries as launch_modup_fused left them, component c of entry b at c * batch + b
// -- and writes
//     out_c = [w_c +] MRed(NTT(ext_c) + 2q - acc_c, s[limb])        (the last op of ModDownQPtoQNTT, as NttEpilogue)
// against the accumulator still in registers: the Q accumulators are never written or read back, and the forward-row + epilogue
// launch over these limbs disappears.  tensor: w_c is formed from the product's inputs (NttEpilogue::tensor).
// sp[i] = IMForm(s[i]) and tsp[i] = IMForm(IMForm(ts[i])) as doubles, by launch limb.  out0Q / out1Q of the launch are unused.
struct NttMacEpilogue {
    View ext;
    bool ext_f64 = false;  // the extension's words are doubles (launch_modup_fused f64_raw)
    View out0, out1;
    bool has_w0 = false, has_w1 = false;
    View w0, w1;
    bool tensor = false;
    View ta0, ta1, tb0, tb1;
    double sp[kMaxLimbs], tsp[kMaxLimbs];
    uint32_t scatter_ginv = 0;  // as NttEpilogue::scatter_ginv
};
bool ntt_mac_epilogue_supported(int logN);
struct KsScatter;
// giant (optional, without an epilogue, production row sizes): the accumulators leave through KsScatter's giant-step stores
hipError_t launch_ntt_mac_f64(const RingDev &r, const NttMacArgs &a, View dec, View own, const double *keyd, View out0Q,
                              View out0P, View out1Q, View out1P, int batch, hipStream_t s, const NttMacEpilogue *epi = nullptr,
                              const KsScatter *giant = nullptr);
bool ntt_mac_giant_supported(int logN);
// keyd[i] = (double)IMForm(key[i]) for the limbs of class 2 (plain residues < 2^47), 0 elsewhere
hipError_t launch_key_to_f64(const RingDev &r, const uint64_t *key, double *keyd, int nblocks, const uint8_t *limb_mod_host,
                             int nlimbs, hipStream_t s);

// ---- ciphertext tensor product (schemes/ckks/evaluator.go:807-820, schemes/bgv/evaluator.go:634-647)
// c0 = MRed(MRed(a0,s),b0), c2 = MRed(MRed(a1,s),b1), c1 = CRed(MRed(MRed(a0,s),b1) + MRed(MRed(a1,s),b0))
// with the per-limb scalar s = 2^128 mod q (CKKS: MForm) or t*2^128 mod q (BGV: tMontgomery).
hipError_t launch_tensor(const RingDev &r, const LimbTab &tab, const uint64_t *scalar, View a0, View a1, View b0, View b1,
                         View c0, View c1, View c2, int batch, hipStream_t s);

// ---- per-kernel HIP-event profiling (diagnostics: bench.py's roofline leg) --------------------
enum KernelId {
    K_NTT_COLS_FWD = 0, K_NTT_ROWS_FWD, K_NTT_ROWS_INV, K_NTT_COLS_INV, K_EW, K_GATHER, K_AUTO_COEFF, K_INDEX,
    K_MODUP, K_CENTER, K_KS_INNER, K_TENSOR, K_PROBE, K_CI_FOLD, K_MASK_SPREAD, K_NTT_ROWS_FWD_F64, K_NTT_ROWS_INV_F64,
    K_NTT_MAC_F64, K_DIAG_MAC, K_COUNT
};
const char *kernel_name(int id);
void prof_begin(hipStream_t s);                                // start recording the launches enqueued on stream s
bool prof_active(hipStream_t s);                               // is stream s being recorded?
int prof_end(hipStream_t s, int *counts, float *total_ms, double *total_bytes = nullptr);  // stop, sync events, fill [K_COUNT] arrays

// throughput probe used by bench.py --microbench (not on the product path)
hipError_t launch_modmul_probe(uint64_t *buf, size_t n, int iters, uint64_t q, uint64_t qinv, hipStream_t s);
hipError_t launch_modmul_f64_probe(double *buf, size_t n, int iters, double q, hipStream_t s);

}  // namespace he
