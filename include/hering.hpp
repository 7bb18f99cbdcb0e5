// hering.hpp -- header-only C++17 host side of libhering.so (include/hering.h).
//
// The reference's host code is compiled (Go); this image has no Go toolchain, so beside the cgo package under go/hering (shipped
// as source) this header is the COMPILED mirror of the reference's operator interfaces for the hot path: the same type and method
// names, argument order (outputs last, caller-allocated) and error behaviour as
//     ring.Ring                    ring/ring.go, ring/ntt.go:127-152, ring/operations.go, ring/scaling.go, ring/automorphism.go
//     ring.BasisExtender           ring/basis_extension.go:14-280
//     rlwe.Evaluator               core/rlwe/evaluator*.go   (the seven EvaluatorProvider methods of core/rlwe/rlwe.go:10-18 included)
//     ckks / bgv Evaluator         MulRelin (schemes/ckks/evaluator.go:764, schemes/bgv/evaluator.go:592), Rescale (:477 / :1363)
// over device-resident polynomials.  A Go error is a C++ exception (hering::Error, carrying the library's status code and
// message); nothing here computes: every method is one call of the C ABI.  Objects are cheap shared references to library
// handles (a copy is another reference, as a Go pointer would be); the handle is released with the last reference.
//
// tests/cpp/parity.cpp is written against this header and reads like the reference's own ring_test.go / rlwe_test.go cases.
#ifndef HERING_HPP
#define HERING_HPP

#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hering.h"

namespace hering {

struct Error : std::runtime_error {
    int code;
    explicit Error(int rc) : std::runtime_error(std::string("hering: ") + he_last_error()), code(rc) {}
};
inline void check(int rc) {
    if (rc != HE_OK) throw Error(rc);
}

namespace detail {
struct Box {
    he_handle h = 0;
    int (*destroy)(he_handle) = nullptr;
    Box(he_handle h_, int (*d)(he_handle)) : h(h_), destroy(d) {}
    Box(const Box &) = delete;
    Box &operator=(const Box &) = delete;
    ~Box() {
        if (h && destroy) destroy(h);
    }
};
using Ref = std::shared_ptr<Box>;
inline Ref own(he_handle h, int (*d)(he_handle)) { return std::make_shared<Box>(h, d); }
}  // namespace detail

// One HIP device + stream.  There is no CPU fallback: without a device the constructor throws (HE_EDEVICE).
class Context {
    detail::Ref r_;

public:
    explicit Context(int device = 0) {
        he_handle h = 0;
        check(he_ctx_create(device, &h));
        r_ = detail::own(h, he_ctx_destroy);
    }
    he_handle h() const { return r_->h; }
    void Sync() const { check(he_ctx_sync(h())); }
    // the submission queue of the context (hering.h, he_ctx_set_coalescing): concurrent single-ciphertext calls of any operator --
    // one thread per ciphertext, the reference's b.RunParallel shape -- become batched launches over the callers' own polynomials
    void SetCoalescing(int maxBatch = 64, int windowMicros = 30) const { check(he_ctx_set_coalescing(h(), maxBatch, windowMicros)); }
    // deferred submission (hering.h, he_ctx_set_deferred): queued calls return once filed, the context's dispatcher thread launches
    // them; a failed launch is reported by the next Sync().  depth = 0: back to calls that return once launched
    void SetDeferred(int depth = 8) const { check(he_ctx_set_deferred(h(), depth)); }
    static int DeviceCount() {
        int n = 0;
        check(he_device_count(&n));
        return n;
    }
    static std::string Version() { return he_version(); }
    // CU count, LDS bytes per CU, clock kHz, HBM bytes
    std::array<uint64_t, 4> DeviceInfo() const {
        std::array<uint64_t, 4> v{};
        check(he_device_info(h(), v.data()));
        return v;
    }
    // HIP-event stopwatch on the context's stream
    void TimerStart() const { check(he_timer_start(h())); }
    float TimerStop() const {
        float ms = 0;
        check(he_timer_stop(h(), &ms));
        return ms;
    }
};

// A captured sequence of calls on a Context (he_graph_*): Capture(f) records what f enqueues; Launch replays it as one enqueue.
// f must have run once before (plans, scratch) and must not upload, download or Sync; the polynomials it touches must outlive the graph.
class Graph {
    detail::Ref r_;

public:
    template <class F>
    Graph(const Context &ctx, F &&f) {
        check(he_graph_begin(ctx.h()));
        he_handle g = 0;
        try {
            f();
        } catch (...) {
            if (he_graph_end(ctx.h(), &g) == HE_OK) he_graph_destroy(g);  // leave the context usable
            throw;
        }
        check(he_graph_end(ctx.h(), &g));
        r_ = detail::own(g, he_graph_destroy);
    }
    void Launch() const { check(he_graph_launch(r_->h)); }
    int Nodes() const {
        int n = 0;
        check(he_graph_nodes(r_->h, &n));
        return n;
    }
};

class Ring;

// A batch of ring.Poly in HBM: [batch][limbs][N] uint64 (ring/poly.go:13-16 with a leading batch axis).
class Poly {
    detail::Ref r_;
    int limbs_ = 0, batch_ = 0, n_ = 0;
    friend class Ring;
    Poly(he_handle h, int limbs, int batch, int n) : r_(detail::own(h, he_poly_free)), limbs_(limbs), batch_(batch), n_(n) {}

public:
    Poly() = default;
    he_handle h() const { return r_ ? r_->h : 0; }
    int Level() const { return limbs_ - 1; }
    int N() const { return n_; }
    int Batch() const { return batch_; }
    size_t Words() const { return (size_t)batch_ * limbs_ * n_; }
    // host image [batch][limbs][N]
    void Upload(const uint64_t *src, size_t words) { check(he_poly_upload(h(), src, words)); }
    void Upload(const std::vector<uint64_t> &src) { Upload(src.data(), src.size()); }
    void Download(uint64_t *dst, size_t words) const { check(he_poly_download(h(), dst, words)); }
    std::vector<uint64_t> Download() const {
        std::vector<uint64_t> v(Words());
        Download(v.data(), v.size());
        return v;
    }
    void CopyLvl(int level, const Poly &src) { check(he_poly_copy(h(), src.h(), level)); }  // ring.Poly.CopyLvl
    void Zero() { check(he_poly_zero(h())); }
    // one row of Coeffs [][]uint64: limb `limb` of batch entry b (N words)
    void UploadLimb(int b, int limb, const uint64_t *src) { check(he_poly_upload_limb(h(), b, limb, src)); }
    void DownloadLimb(int b, int limb, uint64_t *dst) const { check(he_poly_download_limb(h(), b, limb, dst)); }
    // limbs 0..level of entries [srcB0, srcB0 + nb) of src -> entries [dstB0, dstB0 + nb) of this batch
    void CopyBatch(int dstB0, const Poly &src, int srcB0, int nb, int level) { check(he_poly_copy_batch(h(), dstB0, src.h(), srcB0, nb, level)); }
    // what the library itself says about the handle (limbs, batch, N)
    std::array<int, 3> Shape() const {
        std::array<int, 3> v{};
        check(he_poly_shape(h(), &v[0], &v[1], &v[2]));
        return v;
    }
    // device storage for transports that move device memory themselves (drains the context's stream first)
    std::pair<void *, size_t> DeviceBuffer() const {
        void *p = nullptr;
        size_t n = 0;
        check(he_poly_device_buffer(h(), &p, &n));
        return {p, n};
    }
};

// the table ring.AutomorphismNTTIndex returns (ring/automorphism.go:12-34), on the device
class AutomorphismIndex {
    detail::Ref r_;
    int n_ = 0;
    friend class Ring;

public:
    AutomorphismIndex() = default;
    he_handle h() const { return r_ ? r_->h : 0; }
    std::vector<uint64_t> Download() const {
        std::vector<uint64_t> v((size_t)n_);
        check(he_automorphism_index_download(h(), v.data()));
        return v;
    }
};

enum class RingType { Standard = 0, ConjugateInvariant = 1 };  // ring/ring.go:24-27

// ring.Ring: the moduli chain with its NTT tables, at a level (AtLevel returns a shallow copy, ring/ring.go:186).
class Ring {
    detail::Ref r_;
    Context ctx_;
    std::vector<uint64_t> moduli_;
    int logN_ = 0, level_ = 0;
    RingType type_ = RingType::Standard;

public:
    // ring.NewRingFromType (ring/ring.go:267): throws HE_EPARAM for a modulus that is not an NTT-friendly prime, repeated moduli ...
    Ring(const Context &ctx, int logN, std::vector<uint64_t> moduli, RingType type = RingType::Standard)
        : ctx_(ctx), moduli_(std::move(moduli)), logN_(logN), level_((int)moduli_.size() - 1), type_(type) {
        he_handle h = 0;
        if (type == RingType::Standard) check(he_ring_create(ctx.h(), logN, moduli_.data(), (int)moduli_.size(), &h));  // ring.NewRing
        else check(he_ring_create_type(ctx.h(), logN, (int)type, moduli_.data(), (int)moduli_.size(), &h));
        r_ = detail::own(h, he_ring_destroy);
    }
    he_handle h() const { return r_->h; }
    const Context &Ctx() const { return ctx_; }
    int N() const { return 1 << logN_; }
    int LogN() const { return logN_; }
    int Level() const { return level_; }
    int MaxLevel() const { return (int)moduli_.size() - 1; }
    RingType Type() const { return type_; }
    const std::vector<uint64_t> &ModuliChain() const { return moduli_; }
    Ring AtLevel(int level) const {
        Ring c = *this;
        c.level_ = level;
        return c;
    }
    Poly NewPoly(int batch = 1) const {  // ring.Ring.NewPoly: zeroed, Level()+1 limbs
        he_handle h = 0;
        check(he_poly_alloc(this->h(), level_ + 1, batch, &h));
        return Poly(h, level_ + 1, batch, N());
    }
    Poly NewScratch(int batch = 1) const {  // not cleared: rlwe.BufferPool semantics (core/rlwe/pool.go)
        he_handle h = 0;
        check(he_poly_alloc_scratch(this->h(), level_ + 1, batch, &h));
        return Poly(h, level_ + 1, batch, N());
    }
    // SubRing constants (ring/subring.go:16-56): 0 Modulus, 1 MRedConstant, 2/3 BRedConstant, 4 NInv, 5 PrimitiveRoot
    uint64_t Constant(int limb, int which) const {
        uint64_t v = 0;
        check(he_ring_constant(h(), limb, which, &v));
        return v;
    }

    // ring/ntt.go:127-152
    void NTT(const Poly &p1, Poly &p2) const { check(he_ntt(h(), level_, p1.h(), p2.h())); }
    void NTTLazy(const Poly &p1, Poly &p2) const { check(he_ntt_lazy(h(), level_, p1.h(), p2.h())); }
    void INTT(const Poly &p1, Poly &p2) const { check(he_intt(h(), level_, p1.h(), p2.h())); }
    void INTTLazy(const Poly &p1, Poly &p2) const { check(he_intt_lazy(h(), level_, p1.h(), p2.h())); }

    // ring/operations.go
    void Add(const Poly &p1, const Poly &p2, Poly &p3) const { check(he_add(h(), level_, p1.h(), p2.h(), p3.h())); }
    void AddLazy(const Poly &p1, const Poly &p2, Poly &p3) const { bin(HE_ADD_LAZY, p1, p2, p3); }
    void Sub(const Poly &p1, const Poly &p2, Poly &p3) const { check(he_sub(h(), level_, p1.h(), p2.h(), p3.h())); }
    void SubLazy(const Poly &p1, const Poly &p2, Poly &p3) const { bin(HE_SUB_LAZY, p1, p2, p3); }
    void Neg(const Poly &p1, Poly &p2) const { check(he_neg(h(), level_, p1.h(), p2.h())); }
    void Reduce(const Poly &p1, Poly &p2) const { check(he_reduce(h(), level_, p1.h(), p2.h())); }
    void ReduceLazy(const Poly &p1, Poly &p2) const { un(HE_REDUCE_LAZY, p1, p2); }
    void MForm(const Poly &p1, Poly &p2) const { check(he_mform(h(), level_, p1.h(), p2.h())); }
    void MFormLazy(const Poly &p1, Poly &p2) const { un(HE_MFORM_LAZY, p1, p2); }
    void IMForm(const Poly &p1, Poly &p2) const { check(he_imform(h(), level_, p1.h(), p2.h())); }
    void MulCoeffsBarrett(const Poly &p1, const Poly &p2, Poly &p3) const { bin(HE_MUL_COEFFS_BARRETT, p1, p2, p3); }
    void MulCoeffsBarrettLazy(const Poly &p1, const Poly &p2, Poly &p3) const { bin(HE_MUL_COEFFS_BARRETT_LAZY, p1, p2, p3); }
    void MulCoeffsBarrettThenAdd(const Poly &p1, const Poly &p2, Poly &p3) const { bin(HE_MUL_COEFFS_BARRETT_THEN_ADD, p1, p2, p3); }
    void MulCoeffsBarrettThenAddLazy(const Poly &p1, const Poly &p2, Poly &p3) const { bin(HE_MUL_COEFFS_BARRETT_THEN_ADD_LAZY, p1, p2, p3); }
    void MulCoeffsMontgomery(const Poly &p1, const Poly &p2, Poly &p3) const { check(he_mul_coeffs_montgomery(h(), level_, p1.h(), p2.h(), p3.h())); }
    void MulCoeffsMontgomeryLazy(const Poly &p1, const Poly &p2, Poly &p3) const { check(he_mul_coeffs_montgomery_lazy(h(), level_, p1.h(), p2.h(), p3.h())); }
    void MulCoeffsMontgomeryLazyThenNeg(const Poly &p1, const Poly &p2, Poly &p3) const { bin(HE_MUL_COEFFS_MONTGOMERY_LAZY_THEN_NEG, p1, p2, p3); }
    void MulCoeffsMontgomeryThenAdd(const Poly &p1, const Poly &p2, Poly &p3) const { check(he_mul_coeffs_montgomery_then_add(h(), level_, p1.h(), p2.h(), p3.h())); }
    void MulCoeffsMontgomeryThenAddLazy(const Poly &p1, const Poly &p2, Poly &p3) const { bin(HE_MUL_COEFFS_MONTGOMERY_THEN_ADD_LAZY, p1, p2, p3); }
    void MulCoeffsMontgomeryLazyThenAddLazy(const Poly &p1, const Poly &p2, Poly &p3) const {
        check(he_mul_coeffs_montgomery_lazy_then_add_lazy(h(), level_, p1.h(), p2.h(), p3.h()));
    }
    void MulCoeffsMontgomeryThenSub(const Poly &p1, const Poly &p2, Poly &p3) const { bin(HE_MUL_COEFFS_MONTGOMERY_THEN_SUB, p1, p2, p3); }
    void MulCoeffsMontgomeryThenSubLazy(const Poly &p1, const Poly &p2, Poly &p3) const { bin(HE_MUL_COEFFS_MONTGOMERY_THEN_SUB_LAZY, p1, p2, p3); }
    void MulCoeffsMontgomeryLazyThenSubLazy(const Poly &p1, const Poly &p2, Poly &p3) const { bin(HE_MUL_COEFFS_MONTGOMERY_LAZY_THEN_SUB_LAZY, p1, p2, p3); }
    void AddScalar(const Poly &p1, uint64_t scalar, Poly &p2) const { sc(HE_ADD_SCALAR, p1, scalar, p2); }
    void SubScalar(const Poly &p1, uint64_t scalar, Poly &p2) const { sc(HE_SUB_SCALAR, p1, scalar, p2); }
    void MulScalar(const Poly &p1, uint64_t scalar, Poly &p2) const { sc(HE_MUL_SCALAR, p1, scalar, p2); }
    void MulScalarThenAdd(const Poly &p1, uint64_t scalar, Poly &p2) const { sc(HE_MUL_SCALAR_THEN_ADD, p1, scalar, p2); }
    void MulScalarThenSub(const Poly &p1, uint64_t scalar, Poly &p2) const { sc(HE_MUL_SCALAR_THEN_SUB, p1, scalar, p2); }
    void MulRNSScalarMontgomery(const Poly &p1, const std::vector<uint64_t> &scalar, Poly &p2) const {
        if ((int)scalar.size() <= level_) throw std::invalid_argument("MulRNSScalarMontgomery: one scalar per limb");
        check(he_mul_rns_scalar_montgomery(h(), level_, p1.h(), scalar.data(), p2.h()));
    }
    void Shift(const Poly &p1, int k, Poly &p2) const { check(he_shift(h(), level_, p1.h(), k, p2.h())); }
    void MultByMonomial(const Poly &p1, int k, Poly &p2) const { check(he_mult_by_monomial(h(), level_, p1.h(), k, p2.h())); }

    // ring/scaling.go
    void DivRoundByLastModulusNTT(const Poly &p0, Poly &p1) const { check(he_div_round_by_last_modulus_ntt(h(), level_, p0.h(), p1.h())); }
    void DivRoundByLastModulus(const Poly &p0, Poly &p1) const { check(he_div_round_by_last_modulus(h(), level_, p0.h(), p1.h())); }
    void DivFloorByLastModulusNTT(const Poly &p0, Poly &p1) const { check(he_div_floor_by_last_modulus_ntt(h(), level_, p0.h(), p1.h())); }
    void DivFloorByLastModulus(const Poly &p0, Poly &p1) const { check(he_div_floor_by_last_modulus(h(), level_, p0.h(), p1.h())); }
    void DivRoundByLastModulusManyNTT(int nbRescales, const Poly &p0, Poly &p1) const {
        check(he_div_round_by_last_modulus_many_ntt(h(), level_, nbRescales, p0.h(), p1.h()));
    }
    void DivRoundByLastModulusMany(int nbRescales, const Poly &p0, Poly &p1) const {
        check(he_div_round_by_last_modulus_many(h(), level_, nbRescales, p0.h(), p1.h()));
    }
    void DivFloorByLastModulusManyNTT(int nbRescales, const Poly &p0, Poly &p1) const {
        check(he_div_floor_by_last_modulus_many_ntt(h(), level_, nbRescales, p0.h(), p1.h()));
    }
    void DivFloorByLastModulusMany(int nbRescales, const Poly &p0, Poly &p1) const {
        check(he_div_floor_by_last_modulus_many(h(), level_, nbRescales, p0.h(), p1.h()));
    }

    // ring/automorphism.go: AutomorphismNTT (:36, not in place), Automorphism (:113, coefficient domain)
    void AutomorphismNTT(const Poly &pIn, uint64_t galEl, Poly &pOut) const {
        he_handle ix = 0;
        check(he_automorphism_index_create(h(), galEl, &ix));
        detail::Ref keep = detail::own(ix, he_automorphism_index_destroy);
        check(he_automorphism_ntt_with_index(h(), level_, pIn.h(), ix, pOut.h()));
    }
    void Automorphism(const Poly &pIn, uint64_t galEl, Poly &pOut) const { check(he_automorphism(h(), level_, pIn.h(), galEl, pOut.h())); }
    AutomorphismIndex AutomorphismNTTIndex(uint64_t galEl) const {  // :12
        he_handle ix = 0;
        check(he_automorphism_index_create(h(), galEl, &ix));
        AutomorphismIndex a;
        a.r_ = detail::own(ix, he_automorphism_index_destroy);
        a.n_ = N();
        return a;
    }
    void AutomorphismNTTWithIndex(const Poly &pIn, const AutomorphismIndex &index, Poly &pOut) const {  // :50
        check(he_automorphism_ntt_with_index(h(), level_, pIn.h(), index.h(), pOut.h()));
    }
    void AutomorphismNTTWithIndexThenAddLazy(const Poly &pIn, const AutomorphismIndex &index, Poly &pOut) const {  // :82
        check(he_automorphism_ntt_with_index_then_add_lazy(h(), level_, pIn.h(), index.h(), pOut.h()));
    }
    // {Add,Sub,Mul}ScalarBigint, MulScalarBigintThenAdd (operations.go:158,193,231,240): little-endian 64-bit words of |scalar|
    void AddScalarBigint(const Poly &p1, const std::vector<uint64_t> &words, Poly &p2) const {
        check(he_add_scalar_bigint(h(), level_, p1.h(), words.data(), (int)words.size(), p2.h()));
    }
    void SubScalarBigint(const Poly &p1, const std::vector<uint64_t> &words, Poly &p2) const {
        check(he_sub_scalar_bigint(h(), level_, p1.h(), words.data(), (int)words.size(), p2.h()));
    }
    void MulScalarBigint(const Poly &p1, const std::vector<uint64_t> &words, Poly &p2) const {
        check(he_mul_scalar_bigint(h(), level_, p1.h(), words.data(), (int)words.size(), p2.h()));
    }
    void MulScalarBigintThenAdd(const Poly &p1, const std::vector<uint64_t> &words, Poly &p2) const {
        check(he_mul_scalar_bigint_then_add(h(), level_, p1.h(), words.data(), (int)words.size(), p2.h()));
    }
    // {Add,Sub,Mul}DoubleRNSScalar, MulDoubleRNSScalarThenAdd (operations.go:166,176,249,260): scalar0 on coefficients [0, N/2), scalar1 on [N/2, N)
    void AddDoubleRNSScalar(const Poly &p1, const std::vector<uint64_t> &s0, const std::vector<uint64_t> &s1, Poly &p2) const { drs(0, p1, s0, s1, p2); }
    void SubDoubleRNSScalar(const Poly &p1, const std::vector<uint64_t> &s0, const std::vector<uint64_t> &s1, Poly &p2) const { drs(1, p1, s0, s1, p2); }
    void MulDoubleRNSScalar(const Poly &p1, const std::vector<uint64_t> &s0, const std::vector<uint64_t> &s1, Poly &p2) const { drs(2, p1, s0, s1, p2); }
    void MulDoubleRNSScalarThenAdd(const Poly &p1, const std::vector<uint64_t> &s0, const std::vector<uint64_t> &s1, Poly &p2) const { drs(3, p1, s0, s1, p2); }
    // MulByVectorMontgomery / ...ThenAddLazy (operations.go:363,370): every limb times limb 0 of the batch-1 polynomial `vector`
    void MulByVectorMontgomery(const Poly &p1, const Poly &vector, Poly &p2) const { check(he_mul_by_vector_montgomery(h(), level_, p1.h(), vector.h(), 0, p2.h())); }
    void MulByVectorMontgomeryThenAddLazy(const Poly &p1, const Poly &vector, Poly &p2) const {
        check(he_mul_by_vector_montgomery(h(), level_, p1.h(), vector.h(), 1, p2.h()));
    }
    // SubRing.RootsForward / RootsBackward (ring/subring.go:44-47)
    std::vector<uint64_t> RootsForward(int limb) const { return roots(limb, 0); }
    std::vector<uint64_t> RootsBackward(int limb) const { return roots(limb, 1); }
    // ring.NumberTheoreticTransformer on host slices (ring/ntt.go:17-22): the plug point of ring.NewRingWithCustomNTT
    void Forward(int limb, const uint64_t *p1, uint64_t *p2) const { check(he_subring_ntt_host(h(), limb, 0, 0, p1, p2)); }
    void ForwardLazy(int limb, const uint64_t *p1, uint64_t *p2) const { check(he_subring_ntt_host(h(), limb, 0, 1, p1, p2)); }
    void Backward(int limb, const uint64_t *p1, uint64_t *p2) const { check(he_subring_ntt_host(h(), limb, 1, 0, p1, p2)); }
    void BackwardLazy(int limb, const uint64_t *p1, uint64_t *p2) const { check(he_subring_ntt_host(h(), limb, 1, 1, p1, p2)); }

    // the selector forms (enum he_binop / he_unop / he_scalarop of hering.h), for table-driven callers
    void BinOp(int op, const Poly &p1, const Poly &p2, Poly &p3) const { bin(op, p1, p2, p3); }
    void UnOp(int op, const Poly &p1, Poly &p2) const { un(op, p1, p2); }
    void ScalarOp(int op, const Poly &p1, uint64_t scalar, Poly &p2) const { sc(op, p1, scalar, p2); }

private:
    void drs(int op, const Poly &p1, const std::vector<uint64_t> &s0, const std::vector<uint64_t> &s1, Poly &p2) const {
        if ((int)s0.size() <= level_ || (int)s1.size() <= level_) throw std::invalid_argument("DoubleRNSScalar: one scalar pair per limb");
        check(he_double_rns_scalarop(h(), level_, op, p1.h(), s0.data(), s1.data(), p2.h()));
    }
    std::vector<uint64_t> roots(int limb, int dir) const {
        std::vector<uint64_t> v((size_t)N());
        check(he_ring_roots(h(), limb, dir, v.data()));
        return v;
    }
    void bin(int op, const Poly &p1, const Poly &p2, Poly &p3) const { check(he_binop(h(), level_, op, p1.h(), p2.h(), p3.h())); }
    void un(int op, const Poly &p1, Poly &p2) const { check(he_unop(h(), level_, op, p1.h(), p2.h())); }
    void sc(int op, const Poly &p1, uint64_t s, Poly &p2) const { check(he_scalarop(h(), level_, op, p1.h(), s, p2.h())); }
};

// ring.BasisExtender (ring/basis_extension.go:14)
class BasisExtender {
    detail::Ref r_;

public:
    BasisExtender(const Ring &ringQ, const Ring &ringP) {
        he_handle h = 0;
        check(he_basis_extender_create(ringQ.h(), ringP.h(), &h));
        r_ = detail::own(h, he_basis_extender_destroy);
    }
    he_handle h() const { return r_->h; }
    void ModUpQtoP(int levelQ, int levelP, const Poly &polQ, Poly &polP) const { check(he_modup_q_to_p(h(), levelQ, levelP, polQ.h(), polP.h())); }
    void ModUpPtoQ(int levelP, int levelQ, const Poly &polP, Poly &polQ) const { check(he_modup_p_to_q(h(), levelP, levelQ, polP.h(), polQ.h())); }
    void ModDownQPtoQ(int levelQ, int levelP, const Poly &p1Q, const Poly &p1P, Poly &p2Q) const {
        check(he_moddown_qp_to_q(h(), levelQ, levelP, p1Q.h(), p1P.h(), p2Q.h()));
    }
    void ModDownQPtoQNTT(int levelQ, int levelP, const Poly &p1Q, const Poly &p1P, Poly &p2Q) const {
        check(he_moddown_qp_to_q_ntt(h(), levelQ, levelP, p1Q.h(), p1P.h(), p2Q.h()));
    }
    void ModDownQPtoP(int levelQ, int levelP, const Poly &p1Q, const Poly &p1P, Poly &p2P) const {
        check(he_moddown_qp_to_p(h(), levelQ, levelP, p1Q.h(), p1P.h(), p2P.h()));
    }
};

// rlwe.Ciphertext: Value []ring.Poly (core/rlwe/ciphertext.go:13, element.go:27)
struct Ciphertext {
    std::vector<Poly> Value;
    int Degree() const { return (int)Value.size() - 1; }
    int Level() const { return Value.empty() ? -1 : Value[0].Level(); }
};
// ringqp.Poly{Q, P} (ring/ringqp/poly.go:17); a lazy key-switch result is two of them
struct PolyQP {
    Poly Q, P;
};

class Evaluator;

// rlwe.GadgetCiphertext in HBM (core/rlwe/gadgetciphertext.go:19-42): the payload of a RelinearizationKey / GaloisKey
class EvaluationKey {
    detail::Ref r_;
    int nQk_ = 0, nPk_ = 0;
    friend class Evaluator;

public:
    EvaluationKey() = default;
    he_handle h() const { return r_ ? r_->h : 0; }
    int LevelQ() const { return nQk_ - 1; }
    int LevelP() const { return nPk_ - 1; }
    // the key words back on the host: [beta][2][nQk + nPk][N]
    void Download(uint64_t *dst, size_t words) const { check(he_evk_download(h(), dst, words)); }
    // device storage for an external (GPU-to-GPU) fill; Commit() after the write has completed
    std::pair<void *, size_t> DeviceBuffer() const {
        void *p = nullptr;
        size_t n = 0;
        check(he_evk_device_buffer(h(), &p, &n));
        return {p, n};
    }
    void Commit() const { check(he_evk_commit(h())); }
};

// BuffDecompQP of Evaluator.DecomposeNTT: an opaque device buffer (beta digits of (Q limbs, P limbs) per batch entry)
class Decomposition {
    detail::Ref r_;
    friend class Evaluator;

public:
    Decomposition() = default;
    he_handle h() const { return r_ ? r_->h : 0; }
};

// rlwe.Evaluator's key-switch path + the two scheme call sites on top of it
class Evaluator {
    detail::Ref r_;
    Ring ringQ_, ringP_;

public:
    Evaluator(const Ring &ringQ, const Ring &ringP) : ringQ_(ringQ), ringP_(ringP) {
        he_handle h = 0;
        check(he_evaluator_create(ringQ.h(), ringP.h(), &h));
        r_ = detail::own(h, he_evaluator_destroy);
    }
    he_handle h() const { return r_->h; }
    const Ring &RingQ() const { return ringQ_; }
    const Ring &RingP() const { return ringP_; }

    // q: [beta][2][nQk][N], p: [beta][2][nPk][N], NTT + Montgomery form (what GenRelinearizationKeyNew / GenGaloisKeyNew produce)
    EvaluationKey NewEvaluationKey(int beta, int nQk, int nPk, const std::vector<uint64_t> &q, const std::vector<uint64_t> &p) const {
        const size_t n = (size_t)ringQ_.N();
        if (q.size() != (size_t)beta * 2 * nQk * n || p.size() != (size_t)beta * 2 * nPk * n)
            throw std::invalid_argument("NewEvaluationKey: host image size");
        he_handle h = 0;
        check(he_evk_create(this->h(), beta, nQk, nPk, q.data(), p.data(), &h));
        EvaluationKey k;
        k.r_ = detail::own(h, he_evk_destroy);
        k.nQk_ = nQk;
        k.nPk_ = nPk;
        return k;
    }
    // BaseTwoDecomposition = pw2 != 0 (at most one special prime; nPk = 0 and p empty: a key without P part): nj[i] bit windows of Q-limb i
    EvaluationKey NewEvaluationKeyBase2(int pw2, const std::vector<int> &nj, int nQk, int nPk, const std::vector<uint64_t> &q,
                                        const std::vector<uint64_t> &p) const {
        he_handle h = 0;
        check(he_evk_create_base2(this->h(), pw2, nj.data(), (int)nj.size(), nQk, nPk, q.data(), nPk > 0 ? p.data() : nullptr, &h));
        EvaluationKey k;
        k.r_ = detail::own(h, he_evk_destroy);
        k.nQk_ = nQk;
        k.nPk_ = nPk;
        return k;
    }
    Decomposition NewDecomposition(int batch = 1) const {
        he_handle h = 0;
        check(he_decomp_create(this->h(), batch, &h));
        Decomposition d;
        d.r_ = detail::own(h, he_decomp_destroy);
        return d;
    }

    // ---- rlwe.EvaluatorProvider (core/rlwe/rlwe.go:10-18) ----
    void DecomposeNTT(int levelQ, int levelP, int nbPi, const Poly &c2, bool c2IsNTT, Decomposition &decompQP) const {
        check(he_decompose_ntt(h(), levelQ, levelP, nbPi, c2.h(), c2IsNTT ? 1 : 0, decompQP.h()));
    }
    void GadgetProductLazy(int levelQ, const Poly &cx, const EvaluationKey &gadgetCt, std::array<PolyQP, 2> &ct) const {
        check(he_gadget_product_lazy(h(), levelQ, cx.h(), gadgetCt.h(), ct[0].Q.h(), ct[0].P.h(), ct[1].Q.h(), ct[1].P.h()));
    }
    void GadgetProductHoistedLazy(int levelQ, const Decomposition &decompQP, const EvaluationKey &gadgetCt, std::array<PolyQP, 2> &ct) const {
        check(he_gadget_product_hoisted_lazy(h(), levelQ, decompQP.h(), gadgetCt.h(), ct[0].Q.h(), ct[0].P.h(), ct[1].Q.h(), ct[1].P.h()));
    }
    void AutomorphismHoistedLazy(int levelQ, const Ciphertext &ctIn, const Decomposition &c1DecompQP, uint64_t galEl, const EvaluationKey &gk,
                                 std::array<PolyQP, 2> &ctQP) const {
        check(he_automorphism_hoisted_lazy(h(), levelQ, ctIn.Value.at(0).h(), c1DecompQP.h(), galEl, gk.h(), ctQP[0].Q.h(), ctQP[0].P.h(),
                                           ctQP[1].Q.h(), ctQP[1].P.h()));
    }
    void ModDownQPtoQNTT(int levelQ, int levelP, const Poly &p1Q, const Poly &p1P, Poly &p2Q) const {
        check(he_eval_moddown_qp_to_q_ntt(h(), levelQ, levelP, p1Q.h(), p1P.h(), p2Q.h()));
    }
    // (CheckAndGetGaloisKey / AutomorphismIndex are key-set and table look-ups of the Go side: go/hering/evaluator.go)

    // Decomposer.DecomposeAndSplit (ring/basis_extension.go:381): coefficient-domain p0Q -> digit `digit` extended to (p1Q, p1P)
    void DecomposeAndSplit(int levelQ, int levelP, int nbPi, int digit, const Poly &p0Q, Poly &p1Q, Poly &p1P) const {
        check(he_decompose_and_split(h(), levelQ, levelP, nbPi, digit, p0Q.h(), p1Q.h(), p1P.h()));
    }
    // the inner product over the digits [digitBegin, digitEnd) only (a key switch split over GPUs by digit)
    void GadgetProductHoistedLazyDigits(int levelQ, const Decomposition &decompQP, const EvaluationKey &gadgetCt, int digitBegin, int digitEnd,
                                        std::array<PolyQP, 2> &ct) const {
        check(he_gadget_product_hoisted_lazy_digits(h(), levelQ, decompQP.h(), gadgetCt.h(), digitBegin, digitEnd, ct[0].Q.h(), ct[0].P.h(),
                                                    ct[1].Q.h(), ct[1].P.h()));
    }
    // one limb of the hoisting buffer back on the host (tests)
    void DecompositionLimb(const Decomposition &d, int b, int digit, bool isP, int limb, uint64_t *dst) const {
        check(he_decomp_download_limb(d.h(), b, digit, isP ? 1 : 0, limb, dst));
    }
    // pieces of bootstrapping.Evaluator.ModUp (circuits/ckks/bootstrapping/evaluator.go:654-755)
    void CenteredLift(int strict, const Poly &src, int firstQ, int levelQ, Poly &dstQ, int levelP, Poly &dstP) const {
        check(he_centered_lift(h(), strict, src.h(), firstQ, levelQ, dstQ.h(), levelP, dstP.h()));
    }
    void DecompositionFill(Decomposition &d, int levelQ, int levelP, const Poly &srcQ, const Poly &srcP) const {
        check(he_decomp_fill(d.h(), levelQ, levelP, srcQ.h(), srcP.h()));
    }
    // inner accumulation of lintrans.Evaluator.MultiplyByDiagMatrix[BSGS] (lintrans_evaluator.go:216-241, :346-394): term i =
    // (plaintext diagonal, ciphertext, optional automorphism index); out_k = Reduce([out_k +] sum_i pt_i * phi_i(ct_i[k])) on Q and P
    struct DiagTerm {
        PolyQP pt, c0, c1;             // c0.P / c1.P empty: the term has no P part
        const AutomorphismIndex *index;  // nullptr: no automorphism
    };
    void LinTransMulSum(int levelQ, int levelP, const std::vector<DiagTerm> &terms, bool accumulate, std::array<PolyQP, 2> &out) const {
        const int n = (int)terms.size();
        std::vector<he_handle> ptQ(n), ptP(n), c0Q(n), c0P(n), c1Q(n), c1P(n), ix(n);
        bool any_index = false;
        for (int i = 0; i < n; i++) {
            ptQ[i] = terms[i].pt.Q.h(); ptP[i] = terms[i].pt.P.h();
            c0Q[i] = terms[i].c0.Q.h(); c0P[i] = terms[i].c0.P.h();
            c1Q[i] = terms[i].c1.Q.h(); c1P[i] = terms[i].c1.P.h();
            ix[i] = terms[i].index ? terms[i].index->h() : 0;
            any_index = any_index || ix[i] != 0;
        }
        check(he_lintrans_mul_sum(h(), levelQ, levelP, n, ptQ.data(), ptP.data(), c0Q.data(), c0P.data(), c1Q.data(), c1P.data(),
                                  any_index ? ix.data() : nullptr, accumulate ? 1 : 0, out[0].Q.h(), out[0].P.h(), out[1].Q.h(), out[1].P.h()));
    }

    // the giant step of lintrans.Evaluator.MultiplyByDiagMatrixBSGS (lintrans_evaluator.go:397-441) as one call: GadgetProductLazy(cx)
    // -> cQP; cQP[0] += add; out[k] (+)= AutomorphismNTTWithIndex[ThenAddLazy](cQP[k]) -- the key inner products store through
    // the automorphism themselves (he_lintrans_giant_step)
    void LinTransGiantStep(int levelQ, const Poly &cx, const EvaluationKey &galoisKey, uint64_t galEl, const PolyQP &add, bool accumulate,
                           std::array<PolyQP, 2> &out) const {
        check(he_lintrans_giant_step(h(), levelQ, cx.h(), galoisKey.h(), galEl, add.Q.h(), add.P.h(), out[0].Q.h(), out[0].P.h(), out[1].Q.h(),
                                     out[1].P.h(), accumulate ? 1 : 0));
    }

    // ---- rlwe.Evaluator ----
    void ModDown(int levelQ, int levelP, const std::array<PolyQP, 2> &ctQP, Ciphertext &ct) const {
        check(he_moddown(h(), levelQ, levelP, ctQP[0].Q.h(), ctQP[0].P.h(), ctQP[1].Q.h(), ctQP[1].P.h(), ct.Value.at(0).h(), ct.Value.at(1).h()));
    }
    void GadgetProduct(int levelQ, const Poly &cx, const EvaluationKey &gadgetCt, Ciphertext &ct) const {
        check(he_gadget_product(h(), levelQ, cx.h(), gadgetCt.h(), ct.Value.at(0).h(), ct.Value.at(1).h()));
    }
    void GadgetProductHoisted(int levelQ, const Decomposition &decompQP, const EvaluationKey &gadgetCt, Ciphertext &ct) const {
        check(he_gadget_product_hoisted(h(), levelQ, decompQP.h(), gadgetCt.h(), ct.Value.at(0).h(), ct.Value.at(1).h()));
    }
    void Relinearize(const Ciphertext &ctIn, const EvaluationKey &rlk, Ciphertext &opOut) const {  // evaluator_evaluationkey.go:117
        if (ctIn.Degree() != 2) throw std::invalid_argument("cannot relinearize: ctIn.Degree() should be 2 but is " + std::to_string(ctIn.Degree()));
        check(he_relinearize(h(), ctIn.Level(), ctIn.Value[0].h(), ctIn.Value[1].h(), ctIn.Value[2].h(), rlk.h(), opOut.Value.at(0).h(),
                             opOut.Value.at(1).h()));
    }
    void Automorphism(const Ciphertext &ctIn, uint64_t galEl, const EvaluationKey &gk, Ciphertext &opOut) const {  // evaluator_automorphism.go:13
        check(he_automorphism_ct(h(), ctIn.Level(), ctIn.Value.at(0).h(), ctIn.Value.at(1).h(), galEl, gk.h(), opOut.Value.at(0).h(),
                                 opOut.Value.at(1).h()));
    }
    void AutomorphismHoisted(int level, const Ciphertext &ctIn, const Decomposition &c1DecompQP, uint64_t galEl, const EvaluationKey &gk,
                             Ciphertext &opOut) const {  // :60
        check(he_automorphism_hoisted(h(), level, ctIn.Value.at(0).h(), c1DecompQP.h(), galEl, gk.h(), opOut.Value.at(0).h(), opOut.Value.at(1).h()));
    }

    // ---- schemes: degree 1 x degree 1.  relin = false: the degree-2 result (opOut needs three components) ----
    void MulRelinCKKS(const Ciphertext &op0, const Ciphertext &op1, const EvaluationKey *rlk, Ciphertext &opOut) const {
        mul(false, 0, op0, op1, rlk, opOut);
    }
    void MulRelinBGV(uint64_t t, const Ciphertext &op0, const Ciphertext &op1, const EvaluationKey *rlk, Ciphertext &opOut) const {
        mul(true, t, op0, op1, rlk, opOut);
    }
    // Evaluator.Rescale: DivRoundByLastModulusManyNTT per component (schemes/ckks/evaluator.go:477, schemes/bgv/evaluator.go:1363)
    void Rescale(int nbRescales, const Ciphertext &op0, Ciphertext &opOut) const {
        std::vector<he_handle> in, out;
        for (size_t i = 0; i < op0.Value.size(); i++) { in.push_back(op0.Value[i].h()); out.push_back(opOut.Value.at(i).h()); }
        check(he_rescale_polys(ringQ_.h(), op0.Level(), nbRescales, (int)in.size(), in.data(), out.data()));  // the loop as one call
    }
    // the reference's parallel mode (many goroutines, one ciphertext per call) gathered into batched launches: hering.h
    // (the queue belongs to the evaluator's context and serves every operator of that context: Context::SetCoalescing is the same switch)
    void SetCoalescing(int maxBatch = 64, int windowMicros = 30) const { check(he_evaluator_set_coalescing(h(), maxBatch, windowMicros)); }

private:
    void mul(bool bgv, uint64_t t, const Ciphertext &op0, const Ciphertext &op1, const EvaluationKey *rlk, Ciphertext &opOut) const {
        if (op0.Degree() != 1 || op1.Degree() != 1) throw std::invalid_argument("MulRelin: operands of degree 1");
        const he_handle k = rlk ? rlk->h() : 0, o2 = rlk ? 0 : opOut.Value.at(2).h();
        const int level = op0.Level() < op1.Level() ? op0.Level() : op1.Level();
        if (bgv)
            check(he_bgv_mul_relin(h(), level, t, op0.Value[0].h(), op0.Value[1].h(), op1.Value[0].h(), op1.Value[1].h(), k, opOut.Value.at(0).h(),
                                   opOut.Value.at(1).h(), o2));
        else
            check(he_ckks_mul_relin(h(), level, op0.Value[0].h(), op0.Value[1].h(), op1.Value[0].h(), op1.Value[1].h(), k, opOut.Value.at(0).h(),
                                    opOut.Value.at(1).h(), o2));
    }
};

// One process per GPU: the RCCL communicator of a context, driven by the library on the context's stream (key replication over xGMI;
// the all-reduce of a key switch split by digit).  Rank 0 draws the id and hands it to the others over any control plane.
class Communicator {
    detail::Ref r_;

public:
    static bool Available() {
        int yes = 0;
        check(he_rccl_available(&yes));
        return yes != 0;
    }
    static std::array<uint8_t, HE_RCCL_ID_BYTES> UniqueId() {
        std::array<uint8_t, HE_RCCL_ID_BYTES> id{};
        check(he_rccl_unique_id(id.data()));
        return id;
    }
    Communicator(const Context &ctx, const std::array<uint8_t, HE_RCCL_ID_BYTES> &id, int rank, int world) {  // collective
        he_handle h = 0;
        check(he_rccl_comm_create(ctx.h(), id.data(), rank, world, &h));
        r_ = detail::own(h, he_rccl_comm_destroy);
    }
    int Ranks() const {  // the ranks RCCL itself reaches (an all-reduce of ones)
        int n = 0;
        check(he_rccl_comm_ranks(r_->h, &n));
        return n;
    }
    void Broadcast(const EvaluationKey &evk, int root) const { check(he_evk_broadcast(r_->h, evk.h(), root)); }
    void AllReduceSum(Poly &p) const { check(he_poly_all_reduce_sum(r_->h, p.h())); }
};

}  // namespace hering
#endif  // HERING_HPP
