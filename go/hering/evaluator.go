package hering

/*
#include "hering.h"
*/
import "C"

import (
	"fmt"
	"sync"
	"sync/atomic"
	"unsafe"

	"github.com/tuneinsight/lattigo/v6/core/rlwe"
	"github.com/tuneinsight/lattigo/v6/ring"
	"github.com/tuneinsight/lattigo/v6/ring/ringqp"
)

// Evaluator is the device twin of rlwe.Evaluator's key-switch path.  It implements rlwe.EvaluatorProvider
// (core/rlwe/rlwe.go:10-18): the seven methods below take the reference's own host types, find (or create) their device
// twins, run the HIP path and mark the results device-fresh.  Host memory is only touched by Upload / Download: the
// circuit drivers above call the provider back-to-back on the same objects, so operands stay in HBM between calls, and a
// caller that wants to read a result on the host (decrypt, serialise) calls Evaluator.Download on it first.
//
// STALENESS CONTRACT (applies to every method below that takes a ring.Poly / rlwe.Ciphertext / ringqp.Poly; each repeats it):
//   - a twin is keyed on the address of the host polynomial's first coefficient.  The FIRST time a host polynomial is seen as an
//     input its words are uploaded; afterwards the DEVICE copy is the authoritative one and the host words are NOT read again;
//   - a caller that modifies a host polynomial on the host after its first use (encoder, sampler, CopyLvl, a CPU ring call)
//     MUST call Upload(ring, p) before passing it in again, or the device computes on the old words -- silently;
//   - results exist on the device only: the host words of an output are stale until Download / DownloadCiphertext / Sync;
//   - Sync() downloads every output twin that has not been downloaded since it was last written (a full barrier: after it the
//     host view equals the device view for everything this evaluator produced);
//   - Forget(p) drops a twin; the cache is bounded by MaxTwins: when it would grow beyond, the twins that have not been handed
//     out for the longest time are evicted one by one (downloaded first when they are newer than the host copy) -- never a
//     twin handed out recently, so the operands of an operation in progress (this goroutine's or another's) stay put.
type Evaluator struct {
	params rlwe.Parameters
	ctx    *Context
	h      Handle
	RingQ  *Ring
	RingP  *Ring
	keys   rlwe.EvaluationKeySet

	// MaxTwins bounds the twin cache (0: DefaultMaxTwins): the key pointers keep every host polynomial alive and every twin
	// resident in HBM, so a long circuit without Forget would otherwise grow without limit.  It is a SOFT bound: twins handed out
	// during the last MaxTwins / 2 twin() calls are never evicted (they may be operands of a call in progress), and nothing is
	// evicted while a Graph captured on the context is alive (its nodes address twins by device pointer) -- Close graphs
	// explicitly (Graph.Close) in long-running services instead of leaving them to the finalizer, and watch TwinCount().
	MaxTwins int

	mu     sync.Mutex
	polys  map[*uint64]*Poly                      // twin of a ring.Poly, keyed by the address of its first coefficient
	hosts  map[*uint64]ring.Poly                  // the host polynomial of each twin (for Sync)
	dirty  map[*uint64]bool                       // twins written on the device since their last Download
	used   map[*uint64]uint64                     // tick of the last time a twin was handed out (eviction order)
	tick   uint64
	evks   map[*rlwe.GadgetCiphertext]*EvaluationKey // keys are immutable once generated: uploaded once
	decs   map[*uint64]*Decomposition             // twin of a BuffDecompQP slice, keyed like polys on its first Q row
	index  map[uint64]*AutomorphismIndex
	batch  int
	parent *Evaluator // ShallowCopy: the evaluator that owns the shared maps (evks, index) and whose lock guards them
}

// NewEvaluator mirrors rlwe.NewEvaluator(params, evk) on GPU `ctx`.
func NewEvaluator(ctx *Context, params rlwe.ParameterProvider, evk rlwe.EvaluationKeySet) (*Evaluator, error) {
	p := *params.GetRLWEParameters()
	e := &Evaluator{params: p, ctx: ctx, keys: evk, batch: 1,
		polys: map[*uint64]*Poly{}, hosts: map[*uint64]ring.Poly{}, dirty: map[*uint64]bool{}, used: map[*uint64]uint64{},
		evks: map[*rlwe.GadgetCiphertext]*EvaluationKey{}, decs: map[*uint64]*Decomposition{},
		index: map[uint64]*AutomorphismIndex{}}
	var err error
	if e.RingQ, err = NewRing(ctx, p.RingQ()); err != nil {
		return nil, err
	}
	var hp Handle // 0: parameters without special primes (levelP = -1)
	if p.RingP() != nil {
		if e.RingP, err = NewRing(ctx, p.RingP()); err != nil {
			return nil, err
		}
		hp = e.RingP.h
	}
	if err = lockedCall(func() C.int { return C.he_evaluator_create(e.RingQ.h, hp, &e.h) }); err != nil {
		return nil, err
	}
	// The reference's callers scale by running many single-ciphertext calls at once (b.RunParallel over ShallowCopy'd
	// evaluators, schemes/ckks/ckks_benchmarks_test.go:116-207): on by default, such calls are gathered into batched launches.
	if err = e.SetCoalescing(DefaultCoalesceBatch, DefaultCoalesceWindowMicros); err != nil {
		return nil, err
	}
	return e, nil
}

// Defaults of the submission queue (see SetCoalescing).
const (
	DefaultCoalesceBatch        = 64
	DefaultCoalesceWindowMicros = 30
)

// SetCoalescing configures the evaluator's submission queue (he_evaluator_set_coalescing, include/hering.h): MulRelin calls made
// at the same time from different goroutines -- on this Evaluator or on its ShallowCopy's, which share the device evaluator --
// are executed as one batched launch over the callers' own device twins; every call returns once its batch is enqueued.
// maxBatch <= 1 switches it off.  A lone caller is not delayed: the gathering window only applies while calls overlap.
func (e *Evaluator) SetCoalescing(maxBatch, windowMicros int) error {
	return lockedCall(func() C.int { return C.he_evaluator_set_coalescing(e.h, C.int(maxBatch), C.int(windowMicros)) }, e)
}

// ShallowCopy mirrors rlwe.Evaluator.ShallowCopy (core/rlwe/evaluator.go:200-214): a copy for another goroutine that shares
// everything read-only -- here the device evaluator with its tables, plans, uploaded keys and submission queue -- and has its
// own twin cache (each goroutine works on its own ciphertexts).
func (e *Evaluator) ShallowCopy() *Evaluator {
	root := e
	if e.parent != nil {
		root = e.parent
	}
	c := &Evaluator{params: e.params, ctx: e.ctx, h: e.h, RingQ: e.RingQ, RingP: e.RingP, keys: e.keys, MaxTwins: e.MaxTwins, batch: e.batch,
		polys: map[*uint64]*Poly{}, hosts: map[*uint64]ring.Poly{}, dirty: map[*uint64]bool{}, used: map[*uint64]uint64{},
		evks: root.evks, decs: map[*uint64]*Decomposition{}, index: root.index, parent: root}
	return c
}

// sharedMu guards the maps a ShallowCopy shares with the evaluator it was copied from (uploaded keys, automorphism indices).
func (e *Evaluator) sharedMu() *sync.Mutex {
	if e.parent != nil {
		return &e.parent.mu
	}
	return &e.mu
}

// GetRLWEParameters: rlwe.ParameterProvider.
func (e *Evaluator) GetRLWEParameters() *rlwe.Parameters { return &e.params }

// ---- device twins ------------------------------------------------------------------------------------------------------

func key(p ring.Poly) *uint64 { return &p.Coeffs[0][0] }

// TwinCount returns how many device twins the evaluator holds at the moment (MaxTwins is a soft bound, see there).
func (e *Evaluator) TwinCount() int {
	e.mu.Lock()
	defer e.mu.Unlock()
	return len(e.polys)
}

// DefaultMaxTwins is the twin-cache bound when Evaluator.MaxTwins is 0.
const DefaultMaxTwins = 4096

// twin returns the device polynomial of a host polynomial of ring r; upload says whether the host content is the current
// one (an input seen for the first time) or about to be overwritten (an output).
func (e *Evaluator) twin(r *Ring, p ring.Poly, upload bool) (*Poly, error) {
	e.mu.Lock()
	e.tick++
	d, ok := e.polys[key(p)]
	if ok {
		e.used[key(p)] = e.tick
		if !upload {
			e.dirty[key(p)] = true // an output: the device copy is about to become newer than the host's
		}
	}
	e.mu.Unlock()
	if ok && d.limbs >= len(p.Coeffs) {
		return d, nil
	}
	if err := e.evict(); err != nil {
		return nil, err
	}
	d, err := r.AtLevel(len(p.Coeffs) - 1).NewScratch(e.batch)
	if err != nil {
		return nil, err
	}
	if upload {
		if err = d.Upload(0, p); err != nil {
			return nil, err
		}
	}
	e.mu.Lock()
	e.polys[key(p)] = d
	e.hosts[key(p)] = p
	e.used[key(p)] = e.tick
	if !upload {
		e.dirty[key(p)] = true
	}
	e.mu.Unlock()
	return d, nil
}

// evict bounds the twin cache.  An operation fetches its twins one after another and then makes ONE library call with all of
// them, so a twin handed out a moment ago may be an operand of a call that has not been made yet (on this goroutine or another
// one): only twins that have not been handed out during the last MaxTwins / 2 twin() calls are candidates.  A candidate that
// is newer than its host polynomial is downloaded before it is dropped, so dropping never loses a result and a later Download
// of that polynomial has nothing left to do.  When nothing is old enough the cache grows past the bound for the moment.
func (e *Evaluator) evict() error {
	max := e.MaxTwins
	if max == 0 {
		max = DefaultMaxTwins
	}
	e.mu.Lock()
	if len(e.polys) < max || atomic.LoadInt32(&e.ctx.liveGraphs) > 0 { // (a live graph addresses twins by device pointer)
		e.mu.Unlock()
		return nil
	}
	horizon := uint64(max / 2)
	victims := make([]*uint64, 0, len(e.polys)/4+1)
	for k, t := range e.used {
		if e.tick-t > horizon {
			victims = append(victims, k)
			if len(victims) >= len(e.polys)/4+1 {
				break
			}
		}
	}
	e.mu.Unlock()
	for _, k := range victims {
		e.mu.Lock()
		d, host, dirty, t := e.polys[k], e.hosts[k], e.dirty[k], e.used[k]
		stillOld := d != nil && e.tick-t > horizon
		e.mu.Unlock()
		if !stillOld { // handed out again meanwhile
			continue
		}
		if dirty {
			if err := d.Download(0, host); err != nil {
				return err
			}
		}
		e.mu.Lock()
		if e.polys[k] == d && e.tick-e.used[k] > horizon { // (not re-used while it was being downloaded)
			delete(e.polys, k)
			delete(e.hosts, k)
			delete(e.dirty, k)
			delete(e.used, k)
		}
		e.mu.Unlock()
	}
	return nil
}

// Sync downloads every twin that was written on the device since its last Download, so that the host view of everything this
// evaluator produced is current (see the staleness contract at the type).  It waits for the context's stream.
func (e *Evaluator) Sync() error {
	e.mu.Lock()
	todo := make([]*uint64, 0, len(e.dirty))
	for k, d := range e.dirty {
		if d {
			todo = append(todo, k)
		}
	}
	e.mu.Unlock()
	for _, k := range todo {
		e.mu.Lock()
		d, host := e.polys[k], e.hosts[k]
		e.mu.Unlock()
		if d == nil {
			continue
		}
		if err := d.Download(0, host); err != nil {
			return err
		}
		e.mu.Lock()
		e.dirty[k] = false
		e.mu.Unlock()
	}
	return e.ctx.Sync()
}

// Upload refreshes the device twin of a host polynomial that was modified on the host.
func (e *Evaluator) Upload(r *Ring, p ring.Poly) error {
	d, err := e.twin(r, p, false)
	if err != nil {
		return err
	}
	e.mu.Lock()
	e.dirty[key(p)] = false // host and device agree after the upload
	e.mu.Unlock()
	return d.Upload(0, p)
}

// Download copies the device twin of p (if there is one) back to the host polynomial.
func (e *Evaluator) Download(p ring.Poly) error {
	e.mu.Lock()
	d, ok := e.polys[key(p)]
	e.mu.Unlock()
	if !ok {
		return nil
	}
	if err := d.Download(0, p); err != nil {
		return err
	}
	e.mu.Lock()
	e.dirty[key(p)] = false
	e.mu.Unlock()
	return nil
}

// DownloadCiphertext brings every component of ct back to the host.
func (e *Evaluator) DownloadCiphertext(ct *rlwe.Ciphertext) error {
	for _, v := range ct.Value {
		if err := e.Download(v); err != nil {
			return err
		}
	}
	return nil
}

// Forget drops the device twin of p (its HBM returns to the context's pool when the finalizer runs).
func (e *Evaluator) Forget(p ring.Poly) {
	e.mu.Lock()
	delete(e.polys, key(p))
	delete(e.hosts, key(p))
	delete(e.dirty, key(p))
	delete(e.used, key(p))
	e.mu.Unlock()
}

// EvaluationKey is a rlwe.GadgetCiphertext in HBM: [beta][2][Q limbs | P limbs][N].
type EvaluationKey struct {
	h        Handle
	nQk, nPk int
	pw2      int
}

// evk uploads a gadget ciphertext once (always NTT + Montgomery, core/rlwe/keygenerator.go:314).
func (e *Evaluator) evk(g *rlwe.GadgetCiphertext) (*EvaluationKey, error) {
	sm := e.sharedMu()
	sm.Lock()
	k, ok := e.evks[g]
	sm.Unlock()
	if ok {
		return k, nil
	}
	nQk, nPk, n := g.LevelQ()+1, g.LevelP()+1, e.params.N()
	// flatten Value[i][j][k].{Q,P} into the host images the ABI takes: q[block][2][nQk][N], p[block][2][nPk][N]
	var q, p []uint64
	nj := make([]C.int, 0, len(g.Value))
	for i := range g.Value {
		nj = append(nj, C.int(len(g.Value[i])))
		for j := range g.Value[i] {
			for c := 0; c < 2; c++ {
				for _, row := range g.Value[i][j][c].Q.Coeffs[:nQk] {
					q = append(q, row[:n]...)
				}
				if nPk > 0 {
					for _, row := range g.Value[i][j][c].P.Coeffs[:nPk] {
						p = append(p, row[:n]...)
					}
				}
			}
		}
	}
	k = &EvaluationKey{nQk: nQk, nPk: nPk, pw2: g.BaseTwoDecomposition}
	var pp *C.uint64_t
	if nPk > 0 {
		pp = (*C.uint64_t)(unsafe.Pointer(&p[0]))
	}
	var err error
	if g.BaseTwoDecomposition == 0 {
		err = lockedCall(func() C.int {
			return C.he_evk_create(e.h, C.int(len(g.Value)), C.int(nQk), C.int(nPk), (*C.uint64_t)(unsafe.Pointer(&q[0])), pp, &k.h)
		})
	} else {
		err = lockedCall(func() C.int {
			return C.he_evk_create_base2(e.h, C.int(g.BaseTwoDecomposition), &nj[0], C.int(len(nj)), C.int(nQk), C.int(nPk),
				(*C.uint64_t)(unsafe.Pointer(&q[0])), pp, &k.h)
		})
	}
	if err != nil {
		return nil, err
	}
	sm.Lock()
	if prev, dup := e.evks[g]; dup { // another goroutine uploaded the same key meanwhile: keep one
		sm.Unlock()
		return prev, nil
	}
	e.evks[g] = k
	sm.Unlock()
	return k, nil
}

// Decomposition is the device twin of BuffDecompQP []ringqp.Poly (the hoisting buffer of DecomposeNTT).
type Decomposition struct{ h Handle }

func (e *Evaluator) decomp(buff []ringqp.Poly) (*Decomposition, error) {
	k := key(buff[0].Q)
	e.mu.Lock()
	d, ok := e.decs[k]
	e.mu.Unlock()
	if ok {
		return d, nil
	}
	d = &Decomposition{}
	if err := lockedCall(func() C.int { return C.he_decomp_create(e.h, C.int(e.batch), &d.h) }); err != nil {
		return nil, err
	}
	e.mu.Lock()
	e.decs[k] = d
	e.mu.Unlock()
	return d, nil
}

// qp returns the device twins of a ringqp.Poly; P is nil when the parameters have no special primes.
func (e *Evaluator) qp(p ringqp.Poly, levelP int, upload bool) (q, pp *Poly, err error) {
	if q, err = e.twin(e.RingQ, p.Q, upload); err != nil {
		return
	}
	if levelP >= 0 {
		pp, err = e.twin(e.RingP, p.P, upload)
	}
	return
}

func h(p *Poly) Handle {
	if p == nil {
		return 0
	}
	return p.h
}

// ---- rlwe.EvaluatorProvider (core/rlwe/rlwe.go:10-18) -----------------------------------------------------------------------

// DecomposeNTT: core/rlwe/evaluator_gadget_product.go:459.
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (e *Evaluator) DecomposeNTT(level, levelP, pCount int, c1 ring.Poly, isNTT bool, BuffDecompQP []ringqp.Poly) {
	c, err := e.twin(e.RingQ, c1, true)
	if err != nil {
		panic(err) // the reference's method has no error return; invariant violations panic there too
	}
	d, err := e.decomp(BuffDecompQP)
	if err != nil {
		panic(err)
	}
	ntt := 0
	if isNTT {
		ntt = 1
	}
	if err = lockedCall(func() C.int {
		return C.he_decompose_ntt(e.h, C.int(level), C.int(levelP), C.int(pCount), c.h, C.int(ntt), d.h)
	}); err != nil {
		panic(err)
	}
}

// CheckAndGetGaloisKey: core/rlwe/evaluator.go:123 (host bookkeeping, unchanged).
func (e *Evaluator) CheckAndGetGaloisKey(galEl uint64) (evk *rlwe.GaloisKey, err error) {
	if e.keys == nil {
		return nil, fmt.Errorf("evaluation key interface is nil")
	}
	if evk, err = e.keys.GetGaloisKey(galEl); err != nil {
		return nil, fmt.Errorf("%w: key for galEl %d = 5^{%d} key is missing", err, galEl, e.params.SolveDiscreteLogGaloisElement(galEl))
	}
	return
}

// GadgetProductLazy: core/rlwe/evaluator_gadget_product.go:108 (ct.IsNTT domain handling as there).
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (e *Evaluator) GadgetProductLazy(levelQ int, cx ring.Poly, gadgetCt *rlwe.GadgetCiphertext, ct *rlwe.Element[ringqp.Poly]) (err error) {
	if ct.LevelP() < gadgetCt.LevelP() {
		return fmt.Errorf("ctQP.LevelP()=%d < gadgetCt.LevelP()=%d", ct.LevelP(), gadgetCt.LevelP())
	}
	k, err := e.evk(gadgetCt)
	if err != nil {
		return err
	}
	rQ := e.RingQ.AtLevel(levelQ)
	c, err := e.twin(e.RingQ, cx, true)
	if err != nil {
		return err
	}
	if !ct.IsNTT { // cx in the coefficient domain: transform a scratch copy (:142-152)
		t, err := rQ.NewScratch(e.batch)
		if err != nil {
			return err
		}
		if err = rQ.NTT(c, t); err != nil {
			return err
		}
		c = t
	}
	q0, p0, err := e.qp(ct.Value[0], gadgetCt.LevelP(), false)
	if err != nil {
		return err
	}
	q1, p1, err := e.qp(ct.Value[1], gadgetCt.LevelP(), false)
	if err != nil {
		return err
	}
	if err = lockedCall(func() C.int {
		return C.he_gadget_product_lazy(e.h, C.int(levelQ), c.h, k.h, q0.h, h(p0), q1.h, h(p1))
	}); err != nil {
		return err
	}
	if !ct.IsNTT { // ringQP.INTT of the result (:121-125)
		for _, pr := range [][2]*Poly{{q0, p0}, {q1, p1}} {
			if err = rQ.INTT(pr[0], pr[0]); err != nil {
				return err
			}
			if pr[1] != nil {
				if err = e.RingP.AtLevel(gadgetCt.LevelP()).INTT(pr[1], pr[1]); err != nil {
					return err
				}
			}
		}
	}
	return nil
}

// GadgetProductHoistedLazyDigits is the inner product of GadgetProductHoistedLazy over the RNS digits [digitBegin, digitEnd)
// only (NTT domain, canonical): the per-GPU share when one key switch is split over several devices by digit -- each device
// holds its digits of the key, the partial (Q, P) accumulators are summed across devices (RCCL all-reduce on
// he_poly_device_buffer storage) and reduced before ModDown.  No counterpart in the reference (one address space).
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (e *Evaluator) GadgetProductHoistedLazyDigits(levelQ int, BuffQPDecompQP []ringqp.Poly, gadgetCt *rlwe.GadgetCiphertext, digitBegin, digitEnd int, ct *rlwe.Element[ringqp.Poly]) (err error) {
	if gadgetCt.BaseTwoDecomposition != 0 {
		return fmt.Errorf("method is unsupported for BaseTwoDecomposition != 0")
	}
	k, err := e.evk(gadgetCt)
	if err != nil {
		return err
	}
	d, err := e.decomp(BuffQPDecompQP)
	if err != nil {
		return err
	}
	q0, p0, err := e.qp(ct.Value[0], gadgetCt.LevelP(), false)
	if err != nil {
		return err
	}
	q1, p1, err := e.qp(ct.Value[1], gadgetCt.LevelP(), false)
	if err != nil {
		return err
	}
	return lockedCall(func() C.int {
		return C.he_gadget_product_hoisted_lazy_digits(e.h, C.int(levelQ), d.h, k.h, C.int(digitBegin), C.int(digitEnd), q0.h, h(p0), q1.h, h(p1))
	})
}

// GadgetProductHoistedLazy: core/rlwe/evaluator_gadget_product.go:379.
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (e *Evaluator) GadgetProductHoistedLazy(levelQ int, BuffQPDecompQP []ringqp.Poly, gadgetCt *rlwe.GadgetCiphertext, ct *rlwe.Element[ringqp.Poly]) (err error) {
	if gadgetCt.BaseTwoDecomposition != 0 {
		return fmt.Errorf("method is unsupported for BaseTwoDecomposition != 0")
	}
	if ct.LevelP() < gadgetCt.LevelP() {
		return fmt.Errorf("ctQP.LevelP()=%d < gadgetCt.LevelP()=%d", ct.Level(), gadgetCt.LevelP())
	}
	k, err := e.evk(gadgetCt)
	if err != nil {
		return err
	}
	d, err := e.decomp(BuffQPDecompQP)
	if err != nil {
		return err
	}
	q0, p0, err := e.qp(ct.Value[0], gadgetCt.LevelP(), false)
	if err != nil {
		return err
	}
	q1, p1, err := e.qp(ct.Value[1], gadgetCt.LevelP(), false)
	if err != nil {
		return err
	}
	if err = lockedCall(func() C.int {
		return C.he_gadget_product_hoisted_lazy(e.h, C.int(levelQ), d.h, k.h, q0.h, h(p0), q1.h, h(p1))
	}); err != nil {
		return err
	}
	if !ct.IsNTT {
		for _, pr := range [][2]*Poly{{q0, p0}, {q1, p1}} {
			if err = e.RingQ.AtLevel(levelQ).INTT(pr[0], pr[0]); err != nil {
				return err
			}
			if err = e.RingP.AtLevel(gadgetCt.LevelP()).INTT(pr[1], pr[1]); err != nil {
				return err
			}
		}
	}
	return nil
}

// AutomorphismHoistedLazy: core/rlwe/evaluator_automorphism.go:104 (NTT-domain ciphertexts, as every caller in circuits/).
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (e *Evaluator) AutomorphismHoistedLazy(levelQ int, ctIn *rlwe.Ciphertext, c1DecompQP []ringqp.Poly, galEl uint64, ctQP *rlwe.Element[ringqp.Poly]) (err error) {
	gk, err := e.CheckAndGetGaloisKey(galEl)
	if err != nil {
		return fmt.Errorf("cannot apply AutomorphismHoistedLazy: %w", err)
	}
	if ctQP.LevelP() < gk.LevelP() {
		return fmt.Errorf("ctQP.LevelP()=%d < GaloisKey[%d].LevelP()=%d", ctQP.LevelP(), galEl, gk.LevelP())
	}
	if !ctQP.IsNTT {
		return fmt.Errorf("hering: AutomorphismHoistedLazy is device-resident for NTT-domain ciphertexts only")
	}
	k, err := e.evk(&gk.GadgetCiphertext)
	if err != nil {
		return err
	}
	d, err := e.decomp(c1DecompQP)
	if err != nil {
		return err
	}
	in0, err := e.twin(e.RingQ, ctIn.Value[0], true)
	if err != nil {
		return err
	}
	q0, p0, err := e.qp(ctQP.Value[0], gk.LevelP(), false)
	if err != nil {
		return err
	}
	q1, p1, err := e.qp(ctQP.Value[1], gk.LevelP(), false)
	if err != nil {
		return err
	}
	return lockedCall(func() C.int {
		return C.he_automorphism_hoisted_lazy(e.h, C.int(levelQ), in0.h, d.h, C.uint64_t(galEl), k.h, q0.h, h(p0), q1.h, h(p1))
	})
}

// LinTransGiantStep is NOT a method of rlwe.EvaluatorProvider: it replaces lines 412-423 of
// lintrans.Evaluator.MultiplyByDiagMatrixBSGS (circuits/common/lintrans/lintrans_evaluator.go) -- GadgetProductLazy(levelQ, cx, gk, cQP);
// ringQP.Add(cQP.Value[0], add, cQP.Value[0]); ringQP.AutomorphismNTTWithIndex[ThenAddLazy](cQP.Value[k], index, outQP[k]) -- by one
// native call whose key inner products store through the automorphism into the outer accumulators (he_lintrans_giant_step; word for
// word what the separate calls produce).  A maintainer who wants it patches that loop body to
//
//	if dev, ok := eval.EvaluatorProvider.(*hering.Evaluator); ok {
//		err = dev.LinTransGiantStep(levelQ, tmp1QP.Q, galEl, tmp0QP, c0OutQP, c1OutQP, cnt0 != 0)
//	} else { ... the reference's own calls ... }
//
// Staleness: as AutomorphismHoistedLazy (inputs from their device twins, outputs on the device only).
func (e *Evaluator) LinTransGiantStep(levelQ int, cx ring.Poly, galEl uint64, add, out0, out1 ringqp.Poly, accumulate bool) (err error) {
	gk, err := e.CheckAndGetGaloisKey(galEl)
	if err != nil {
		return fmt.Errorf("cannot apply LinTransGiantStep: %w", err)
	}
	k, err := e.evk(&gk.GadgetCiphertext)
	if err != nil {
		return err
	}
	in, err := e.twin(e.RingQ, cx, true)
	if err != nil {
		return err
	}
	aq, ap, err := e.qp(add, gk.LevelP(), true)
	if err != nil {
		return err
	}
	// (accumulating: the accumulators' current words are inputs too)
	q0, p0, err := e.qp(out0, gk.LevelP(), accumulate)
	if err != nil {
		return err
	}
	q1, p1, err := e.qp(out1, gk.LevelP(), accumulate)
	if err != nil {
		return err
	}
	acc := C.int(0)
	if accumulate {
		acc = 1
	}
	return lockedCall(func() C.int {
		return C.he_lintrans_giant_step(e.h, C.int(levelQ), in.h, k.h, C.uint64_t(galEl), aq.h, h(ap), q0.h, h(p0), q1.h, h(p1), acc)
	})
}

// ModDownQPtoQNTT: ring/basis_extension.go:235 through the evaluator's fused three-launch pipeline.
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (e *Evaluator) ModDownQPtoQNTT(levelQ, levelP int, p1Q, p1P, p2Q ring.Poly) {
	a, err := e.twin(e.RingQ, p1Q, true)
	if err != nil {
		panic(err)
	}
	b, err := e.twin(e.RingP, p1P, true)
	if err != nil {
		panic(err)
	}
	c, err := e.twin(e.RingQ, p2Q, false)
	if err != nil {
		panic(err)
	}
	if err = lockedCall(func() C.int {
		return C.he_eval_moddown_qp_to_q_ntt(e.h, C.int(levelQ), C.int(levelP), a.h, b.h, c.h)
	}); err != nil {
		panic(err)
	}
}

// AutomorphismIndex: core/rlwe/evaluator.go:147 -- the table itself (host slice, for callers that index with it) comes from
// the device-built one so that both sides agree.
func (e *Evaluator) AutomorphismIndex(galEl uint64) []uint64 {
	ix, err := e.autoIndex(galEl)
	if err != nil {
		panic(err)
	}
	out, err := ix.Download(e.params.N())
	if err != nil {
		panic(err)
	}
	return out
}

func (e *Evaluator) autoIndex(galEl uint64) (*AutomorphismIndex, error) {
	sm := e.sharedMu()
	sm.Lock()
	ix, ok := e.index[galEl]
	sm.Unlock()
	if ok {
		return ix, nil
	}
	ix, err := e.RingQ.AutomorphismNTTIndex(galEl)
	if err != nil {
		return nil, err
	}
	sm.Lock()
	e.index[galEl] = ix
	sm.Unlock()
	return ix, nil
}

// ---- the full (non-lazy) operators of rlwe.Evaluator the scheme layer calls -----------------------------------------------------

// GadgetProduct: core/rlwe/evaluator_gadget_product.go:16.
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (e *Evaluator) GadgetProduct(levelQ int, cx ring.Poly, gadgetCt *rlwe.GadgetCiphertext, ct *rlwe.Ciphertext) error {
	k, err := e.evk(gadgetCt)
	if err != nil {
		return err
	}
	c, err := e.twin(e.RingQ, cx, true)
	if err != nil {
		return err
	}
	o0, err := e.twin(e.RingQ, ct.Value[0], false)
	if err != nil {
		return err
	}
	o1, err := e.twin(e.RingQ, ct.Value[1], false)
	if err != nil {
		return err
	}
	return lockedCall(func() C.int { return C.he_gadget_product(e.h, C.int(levelQ), c.h, k.h, o0.h, o1.h) })
}

// Relinearize: core/rlwe/evaluator_evaluationkey.go:117 (degree 2 -> degree 1).
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (e *Evaluator) Relinearize(ctIn, opOut *rlwe.Ciphertext) error {
	if ctIn.Degree() != 2 {
		return fmt.Errorf("cannot relinearize: ctIn.Degree() should be 2 but is %d", ctIn.Degree())
	}
	rlk, err := e.keys.GetRelinearizationKey()
	if err != nil {
		return fmt.Errorf("cannot relinearize: %w", err)
	}
	k, err := e.evk(&rlk.GadgetCiphertext)
	if err != nil {
		return err
	}
	level := ctIn.Level()
	if opOut.Level() < level {
		level = opOut.Level()
	}
	var in [3]*Poly
	for i := range in {
		if in[i], err = e.twin(e.RingQ, ctIn.Value[i], true); err != nil {
			return err
		}
	}
	o0, err := e.twin(e.RingQ, opOut.Value[0], false)
	if err != nil {
		return err
	}
	o1, err := e.twin(e.RingQ, opOut.Value[1], false)
	if err != nil {
		return err
	}
	if err = lockedCall(func() C.int {
		return C.he_relinearize(e.h, C.int(level), in[0].h, in[1].h, in[2].h, k.h, o0.h, o1.h)
	}); err != nil {
		return err
	}
	opOut.Resize(1, level)
	*opOut.MetaData = *ctIn.MetaData
	return nil
}

// Automorphism: core/rlwe/evaluator_automorphism.go:13 (NTT domain).
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (e *Evaluator) Automorphism(ctIn *rlwe.Ciphertext, galEl uint64, opOut *rlwe.Ciphertext) error {
	gk, err := e.CheckAndGetGaloisKey(galEl)
	if err != nil {
		return fmt.Errorf("cannot apply Automorphism: %w", err)
	}
	k, err := e.evk(&gk.GadgetCiphertext)
	if err != nil {
		return err
	}
	level := ctIn.Level()
	if opOut.Level() < level {
		level = opOut.Level()
	}
	i0, err := e.twin(e.RingQ, ctIn.Value[0], true)
	if err != nil {
		return err
	}
	i1, err := e.twin(e.RingQ, ctIn.Value[1], true)
	if err != nil {
		return err
	}
	o0, err := e.twin(e.RingQ, opOut.Value[0], false)
	if err != nil {
		return err
	}
	o1, err := e.twin(e.RingQ, opOut.Value[1], false)
	if err != nil {
		return err
	}
	if err = lockedCall(func() C.int {
		return C.he_automorphism_ct(e.h, C.int(level), i0.h, i1.h, C.uint64_t(galEl), k.h, o0.h, o1.h)
	}); err != nil {
		return err
	}
	opOut.Resize(1, level)
	*opOut.MetaData = *ctIn.MetaData
	return nil
}

// AutomorphismHoisted: core/rlwe/evaluator_automorphism.go:60.
// Staleness: inputs are read from their device twins (uploaded on first sight only -- call Upload after a host-side change);
// outputs are written on the device only (Download / Sync before reading them on the host).
func (e *Evaluator) AutomorphismHoisted(level int, ctIn *rlwe.Ciphertext, c1DecompQP []ringqp.Poly, galEl uint64, opOut *rlwe.Ciphertext) error {
	gk, err := e.CheckAndGetGaloisKey(galEl)
	if err != nil {
		return fmt.Errorf("cannot apply AutomorphismHoisted: %w", err)
	}
	k, err := e.evk(&gk.GadgetCiphertext)
	if err != nil {
		return err
	}
	d, err := e.decomp(c1DecompQP)
	if err != nil {
		return err
	}
	i0, err := e.twin(e.RingQ, ctIn.Value[0], true)
	if err != nil {
		return err
	}
	o0, err := e.twin(e.RingQ, opOut.Value[0], false)
	if err != nil {
		return err
	}
	o1, err := e.twin(e.RingQ, opOut.Value[1], false)
	if err != nil {
		return err
	}
	return lockedCall(func() C.int {
		return C.he_automorphism_hoisted(e.h, C.int(level), i0.h, d.h, C.uint64_t(galEl), k.h, o0.h, o1.h)
	})
}
