package hering

/*
#include "hering.h"
*/
import "C"

import "unsafe"

// ---- ring.Ring hot-path methods on device twins (ring/ntt.go:127-152, ring/operations.go, ring/scaling.go,
// ring/automorphism.go).  Outputs are caller-allocated and last, in-place aliasing as the reference allows it.

func (r *Ring) NTT(p1, p2 *Poly) error {
	return lockedCall(func() C.int { return C.he_ntt(r.h, C.int(r.level), p1.h, p2.h) }, r, p1, p2)
}
func (r *Ring) NTTLazy(p1, p2 *Poly) error {
	return lockedCall(func() C.int { return C.he_ntt_lazy(r.h, C.int(r.level), p1.h, p2.h) }, r, p1, p2)
}
func (r *Ring) INTT(p1, p2 *Poly) error {
	return lockedCall(func() C.int { return C.he_intt(r.h, C.int(r.level), p1.h, p2.h) }, r, p1, p2)
}
func (r *Ring) INTTLazy(p1, p2 *Poly) error {
	return lockedCall(func() C.int { return C.he_intt_lazy(r.h, C.int(r.level), p1.h, p2.h) }, r, p1, p2)
}

func (r *Ring) Add(p1, p2, p3 *Poly) error {
	return lockedCall(func() C.int { return C.he_add(r.h, C.int(r.level), p1.h, p2.h, p3.h) }, r, p1, p2, p3)
}
func (r *Ring) Sub(p1, p2, p3 *Poly) error {
	return lockedCall(func() C.int { return C.he_sub(r.h, C.int(r.level), p1.h, p2.h, p3.h) }, r, p1, p2, p3)
}
func (r *Ring) Neg(p1, p2 *Poly) error {
	return lockedCall(func() C.int { return C.he_neg(r.h, C.int(r.level), p1.h, p2.h) }, r, p1, p2)
}
func (r *Ring) Reduce(p1, p2 *Poly) error {
	return lockedCall(func() C.int { return C.he_reduce(r.h, C.int(r.level), p1.h, p2.h) }, r, p1, p2)
}
func (r *Ring) MForm(p1, p2 *Poly) error {
	return lockedCall(func() C.int { return C.he_mform(r.h, C.int(r.level), p1.h, p2.h) }, r, p1, p2)
}
func (r *Ring) IMForm(p1, p2 *Poly) error {
	return lockedCall(func() C.int { return C.he_imform(r.h, C.int(r.level), p1.h, p2.h) }, r, p1, p2)
}
func (r *Ring) MulCoeffsMontgomery(p1, p2, p3 *Poly) error {
	return lockedCall(func() C.int { return C.he_mul_coeffs_montgomery(r.h, C.int(r.level), p1.h, p2.h, p3.h) }, r, p1, p2, p3)
}
func (r *Ring) MulCoeffsMontgomeryThenAdd(p1, p2, p3 *Poly) error {
	return lockedCall(func() C.int { return C.he_mul_coeffs_montgomery_then_add(r.h, C.int(r.level), p1.h, p2.h, p3.h) }, r, p1, p2, p3)
}
func (r *Ring) MulCoeffsMontgomeryLazy(p1, p2, p3 *Poly) error {
	return lockedCall(func() C.int { return C.he_mul_coeffs_montgomery_lazy(r.h, C.int(r.level), p1.h, p2.h, p3.h) }, r, p1, p2, p3)
}
func (r *Ring) MulCoeffsMontgomeryLazyThenAddLazy(p1, p2, p3 *Poly) error {
	return lockedCall(func() C.int {
		return C.he_mul_coeffs_montgomery_lazy_then_add_lazy(r.h, C.int(r.level), p1.h, p2.h, p3.h)
	}, r, p1, p2, p3)
}

// MulScalar / AddScalar: ring/operations.go:201,151 (selector values of enum he_scalar_op in hering.h).
func (r *Ring) MulScalar(p1 *Poly, scalar uint64, p2 *Poly) error {
	return lockedCall(func() C.int { return C.he_scalarop(r.h, C.int(r.level), C.HE_MUL_SCALAR, p1.h, C.uint64_t(scalar), p2.h) })
}
func (r *Ring) AddScalar(p1 *Poly, scalar uint64, p2 *Poly) error {
	return lockedCall(func() C.int { return C.he_scalarop(r.h, C.int(r.level), C.HE_ADD_SCALAR, p1.h, C.uint64_t(scalar), p2.h) })
}

// MulRNSScalarMontgomery: ring/operations.go:216.
func (r *Ring) MulRNSScalarMontgomery(p1 *Poly, scalar []uint64, p2 *Poly) error {
	return lockedCall(func() C.int {
		return C.he_mul_rns_scalar_montgomery(r.h, C.int(r.level), p1.h, (*C.uint64_t)(unsafe.Pointer(&scalar[0])), p2.h)
	})
}

// DivRoundByLastModulusManyNTT: ring/scaling.go:148 (the rescale of CKKS / BGV).
func (r *Ring) DivRoundByLastModulusManyNTT(nbRescales int, p0, p1 *Poly) error {
	return lockedCall(func() C.int {
		return C.he_div_round_by_last_modulus_many_ntt(r.h, C.int(r.level), C.int(nbRescales), p0.h, p1.h)
	})
}

// AutomorphismIndex is the device twin of the table ring.AutomorphismNTTIndex returns.
type AutomorphismIndex struct{ h Handle }

func (r *Ring) AutomorphismNTTIndex(galEl uint64) (*AutomorphismIndex, error) {
	ix := &AutomorphismIndex{}
	return ix, lockedCall(func() C.int { return C.he_automorphism_index_create(r.h, C.uint64_t(galEl), &ix.h) })
}

// Download returns the table as the reference's []uint64.
func (ix *AutomorphismIndex) Download(n int) ([]uint64, error) {
	out := make([]uint64, n)
	return out, lockedCall(func() C.int {
		return C.he_automorphism_index_download(ix.h, (*C.uint64_t)(unsafe.Pointer(&out[0])))
	})
}

// AutomorphismNTTWithIndex: ring/automorphism.go:50 (not in place).
func (r *Ring) AutomorphismNTTWithIndex(pIn *Poly, ix *AutomorphismIndex, pOut *Poly) error {
	return lockedCall(func() C.int { return C.he_automorphism_ntt_with_index(r.h, C.int(r.level), pIn.h, ix.h, pOut.h) })
}

// BasisExtender: ring.BasisExtender (ring/basis_extension.go:14).
type BasisExtender struct{ h Handle }

func NewBasisExtender(ringQ, ringP *Ring) (*BasisExtender, error) {
	be := &BasisExtender{}
	return be, lockedCall(func() C.int { return C.he_basis_extender_create(ringQ.h, ringP.h, &be.h) })
}
func (be *BasisExtender) ModUpQtoP(levelQ, levelP int, polQ, polP *Poly) error {
	return lockedCall(func() C.int { return C.he_modup_q_to_p(be.h, C.int(levelQ), C.int(levelP), polQ.h, polP.h) })
}
func (be *BasisExtender) ModUpPtoQ(levelP, levelQ int, polP, polQ *Poly) error {
	return lockedCall(func() C.int { return C.he_modup_p_to_q(be.h, C.int(levelP), C.int(levelQ), polP.h, polQ.h) })
}
func (be *BasisExtender) ModDownQPtoQ(levelQ, levelP int, p1Q, p1P, p2Q *Poly) error {
	return lockedCall(func() C.int { return C.he_moddown_qp_to_q(be.h, C.int(levelQ), C.int(levelP), p1Q.h, p1P.h, p2Q.h) })
}
func (be *BasisExtender) ModDownQPtoQNTT(levelQ, levelP int, p1Q, p1P, p2Q *Poly) error {
	return lockedCall(func() C.int {
		return C.he_moddown_qp_to_q_ntt(be.h, C.int(levelQ), C.int(levelP), p1Q.h, p1P.h, p2Q.h)
	})
}
func (be *BasisExtender) ModDownQPtoP(levelQ, levelP int, p1Q, p1P, p2P *Poly) error {
	return lockedCall(func() C.int { return C.he_moddown_qp_to_p(be.h, C.int(levelQ), C.int(levelP), p1Q.h, p1P.h, p2P.h) })
}
