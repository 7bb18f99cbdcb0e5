package hering

/*
#include "hering.h"
*/
import "C"

import (
	"fmt"

	"github.com/tuneinsight/lattigo/v6/core/rlwe"
	"github.com/tuneinsight/lattigo/v6/schemes"
)

// SchemeEvaluator implements schemes.Evaluator (schemes/schemes.go:14-28) for ciphertext x ciphertext operands on device
// twins: the interface the circuits layer embeds (circuits/common/lintrans/lintrans_evaluator.go:13, circuits/common/
// polynomial/polynomial_evaluator.go:24, power_basis.go:57).  Plaintext and scalar operands are host-side encodings in the
// reference (they go through the scheme's encoder); they are delegated to the embedded reference evaluator `Host`, after
// which the result's twin is refreshed -- the hot operations (ct x ct Mul / MulRelin / MulThenAdd, Relinearize, Rescale,
// and everything the embedded EvaluatorProvider serves) never leave the device.
//
// Scale and level bookkeeping is the scheme's (MetaData stays on the Go side, core/rlwe/metadata.go): the two callbacks
// below are the scheme-specific parts -- CKKS multiplies scales and MForms an operand (schemes/ckks/evaluator.go:764-872), BGV
// multiplies by T * 2^64 first and tracks the scale in Z_T (schemes/bgv/evaluator.go:592-685).
type SchemeEvaluator struct {
	*Evaluator                   // rlwe.ParameterProvider + rlwe.EvaluatorProvider
	Host       schemes.Evaluator // the reference evaluator of the same scheme, for host-side operands
	BGVPlainT  uint64            // 0: CKKS tensor (MForm), otherwise BGV's plaintext modulus T
	MulScale   func(op0, op1 *rlwe.Ciphertext, out *rlwe.Ciphertext)
	RescaleTo  func(op0, out *rlwe.Ciphertext) (nbRescales int, err error)
}

var _ schemes.Evaluator = (*SchemeEvaluator)(nil)

func (s *SchemeEvaluator) ct(op rlwe.Operand) (*rlwe.Ciphertext, bool) {
	c, ok := op.(*rlwe.Ciphertext)
	return c, ok
}

// hostBinary runs op on the reference evaluator for operands that are not ciphertexts and refreshes the twin of the result.
func (s *SchemeEvaluator) hostBinary(f func() error, op0, out *rlwe.Ciphertext) error {
	if err := s.DownloadCiphertext(op0); err != nil {
		return err
	}
	if err := f(); err != nil {
		return err
	}
	for _, v := range out.Value {
		if err := s.Upload(s.RingQ, v); err != nil {
			return err
		}
	}
	return nil
}

func minLevel(a, b, c *rlwe.Ciphertext) int {
	l := a.Level()
	if b.Level() < l {
		l = b.Level()
	}
	if c.Level() < l {
		l = c.Level()
	}
	return l
}

func (s *SchemeEvaluator) addSub(op0 *rlwe.Ciphertext, op1 rlwe.Operand, opOut *rlwe.Ciphertext, sub bool) error {
	c1, ok := s.ct(op1)
	if !ok || op0.Degree() != c1.Degree() || op0.Scale.Cmp(c1.Scale) != 0 {
		// plaintext / scalar operands, unequal degrees or scales that need matching: the scheme's host logic decides
		if sub {
			return s.hostBinary(func() error { return s.Host.Sub(op0, op1, opOut) }, op0, opOut)
		}
		return s.hostBinary(func() error { return s.Host.Add(op0, op1, opOut) }, op0, opOut)
	}
	level := minLevel(op0, c1, opOut)
	r := s.RingQ.AtLevel(level)
	for i := range op0.Value {
		a, err := s.twin(s.RingQ, op0.Value[i], true)
		if err != nil {
			return err
		}
		b, err := s.twin(s.RingQ, c1.Value[i], true)
		if err != nil {
			return err
		}
		o, err := s.twin(s.RingQ, opOut.Value[i], false)
		if err != nil {
			return err
		}
		if sub {
			err = r.Sub(a, b, o)
		} else {
			err = r.Add(a, b, o)
		}
		if err != nil {
			return err
		}
	}
	opOut.Resize(op0.Degree(), level)
	*opOut.MetaData = *op0.MetaData
	return nil
}

// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (s *SchemeEvaluator) Add(op0 *rlwe.Ciphertext, op1 rlwe.Operand, opOut *rlwe.Ciphertext) error {
	return s.addSub(op0, op1, opOut, false)
}
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (s *SchemeEvaluator) Sub(op0 *rlwe.Ciphertext, op1 rlwe.Operand, opOut *rlwe.Ciphertext) error {
	return s.addSub(op0, op1, opOut, true)
}
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (s *SchemeEvaluator) AddNew(op0 *rlwe.Ciphertext, op1 rlwe.Operand) (*rlwe.Ciphertext, error) {
	out := rlwe.NewCiphertext(s, op0.Degree(), op0.Level())
	return out, s.Add(op0, op1, out)
}
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (s *SchemeEvaluator) SubNew(op0 *rlwe.Ciphertext, op1 rlwe.Operand) (*rlwe.Ciphertext, error) {
	out := rlwe.NewCiphertext(s, op0.Degree(), op0.Level())
	return out, s.Sub(op0, op1, out)
}

// mul is the degree-1 x degree-1 tensor, with (relin) or without the key switch of the degree-2 term -- one fused device call
// (he_ckks_mul_relin / he_bgv_mul_relin: tensor kernel + gadget product + ModDown with the Add folded in).
func (s *SchemeEvaluator) mul(op0 *rlwe.Ciphertext, op1 rlwe.Operand, opOut *rlwe.Ciphertext, relin bool) error {
	c1, ok := s.ct(op1)
	if !ok || op0.Degree() != 1 || c1.Degree() != 1 {
		if relin {
			return s.hostBinary(func() error { return s.Host.MulRelin(op0, op1, opOut) }, op0, opOut)
		}
		return s.hostBinary(func() error { return s.Host.Mul(op0, op1, opOut) }, op0, opOut)
	}
	level := minLevel(op0, c1, opOut)
	var in [4]*Poly
	var err error
	if in[0], err = s.twin(s.RingQ, op0.Value[0], true); err != nil {
		return err
	}
	if in[1], err = s.twin(s.RingQ, op0.Value[1], true); err != nil {
		return err
	}
	if in[2], err = s.twin(s.RingQ, c1.Value[0], true); err != nil {
		return err
	}
	if in[3], err = s.twin(s.RingQ, c1.Value[1], true); err != nil {
		return err
	}
	degree := 2
	var rlk Handle
	if relin {
		key, err := s.keys.GetRelinearizationKey()
		if err != nil {
			return fmt.Errorf("cannot MulRelin: %w", err)
		}
		k, err := s.evk(&key.GadgetCiphertext)
		if err != nil {
			return err
		}
		rlk, degree = k.h, 1
	}
	opOut.Resize(degree, level)
	var out [3]Handle
	for i := 0; i <= degree; i++ {
		o, err := s.twin(s.RingQ, opOut.Value[i], false)
		if err != nil {
			return err
		}
		out[i] = o.h
	}
	if s.BGVPlainT == 0 {
		err = lockedCall(func() C.int {
			return C.he_ckks_mul_relin(s.h, C.int(level), in[0].h, in[1].h, in[2].h, in[3].h, rlk, out[0], out[1], out[2])
		})
	} else {
		err = lockedCall(func() C.int {
			return C.he_bgv_mul_relin(s.h, C.int(level), C.uint64_t(s.BGVPlainT), in[0].h, in[1].h, in[2].h, in[3].h, rlk, out[0], out[1], out[2])
		})
	}
	if err != nil {
		return err
	}
	*opOut.MetaData = *op0.MetaData
	s.MulScale(op0, c1, opOut)
	return nil
}

// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (s *SchemeEvaluator) Mul(op0 *rlwe.Ciphertext, op1 rlwe.Operand, opOut *rlwe.Ciphertext) error {
	return s.mul(op0, op1, opOut, false)
}
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (s *SchemeEvaluator) MulRelin(op0 *rlwe.Ciphertext, op1 rlwe.Operand, opOut *rlwe.Ciphertext) error {
	return s.mul(op0, op1, opOut, true)
}
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (s *SchemeEvaluator) MulNew(op0 *rlwe.Ciphertext, op1 rlwe.Operand) (*rlwe.Ciphertext, error) {
	out := rlwe.NewCiphertext(s, 2, op0.Level())
	return out, s.Mul(op0, op1, out)
}
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (s *SchemeEvaluator) MulRelinNew(op0 *rlwe.Ciphertext, op1 rlwe.Operand) (*rlwe.Ciphertext, error) {
	out := rlwe.NewCiphertext(s, 1, op0.Level())
	return out, s.MulRelin(op0, op1, out)
}

// MulThenAdd: opOut += op0 * op1 (schemes/ckks/evaluator.go:1081, schemes/bgv/evaluator.go:1230); the ct x ct product goes
// through the device tensor into a scratch ciphertext, the accumulation is a device Add.
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (s *SchemeEvaluator) MulThenAdd(op0 *rlwe.Ciphertext, op1 rlwe.Operand, opOut *rlwe.Ciphertext) error {
	c1, ok := s.ct(op1)
	if !ok || op0.Degree() != 1 || c1.Degree() != 1 {
		return s.hostBinary(func() error { return s.Host.MulThenAdd(op0, op1, opOut) }, op0, opOut)
	}
	tmp := rlwe.NewCiphertext(s, 2, minLevel(op0, c1, opOut))
	if err := s.mul(op0, c1, tmp, false); err != nil {
		return err
	}
	if opOut.Degree() < 2 {
		opOut.Resize(2, opOut.Level())
	}
	return s.addSub(opOut, tmp, opOut, false)
}

// Relinearize: schemes.Evaluator.Relinearize(op0, op1) -> rlwe.Evaluator.Relinearize.
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (s *SchemeEvaluator) Relinearize(op0, op1 *rlwe.Ciphertext) error { return s.Evaluator.Relinearize(op0, op1) }

// Rescale: per component DivRoundByLastModulusManyNTT (schemes/ckks/evaluator.go:477-515, schemes/bgv/evaluator.go:1363); how
// many levels to drop and the new scale are the scheme's decision (RescaleTo).
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (s *SchemeEvaluator) Rescale(op0, op1 *rlwe.Ciphertext) error {
	nb, err := s.RescaleTo(op0, op1)
	if err != nil || nb == 0 {
		return err
	}
	r := s.RingQ.AtLevel(op0.Level())
	op1.Resize(op0.Degree(), op0.Level()-nb)
	// the loop over the components as ONE call (he_rescale_polys): with the context's submission queue on, the polynomials of this
	// ciphertext share a batch with the other goroutines'
	in := make([]Handle, len(op0.Value))
	out := make([]Handle, len(op0.Value))
	keep := make([]any, 0, 2*len(op0.Value))
	for i := range op0.Value {
		a, err := s.twin(s.RingQ, op0.Value[i], true)
		if err != nil {
			return err
		}
		o, err := s.twin(s.RingQ, op1.Value[i], false)
		if err != nil {
			return err
		}
		in[i], out[i] = a.h, o.h
		keep = append(keep, a, o)
	}
	return lockedCall(func() C.int {
		return C.he_rescale_polys(r.h, C.int(r.level), C.int(nb), C.int(len(in)), &in[0], &out[0])
	}, keep...)
}
