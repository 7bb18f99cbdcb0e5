// Package hering binds libhering.so (include/hering.h), the MI355X ring-arithmetic backend, to Lattigo v6.
//
// It is the Go side of the drop-in boundary described in INTEGRATION.md: device-resident twins of ring.Poly /
// rlwe.Ciphertext / rlwe.GadgetCiphertext (handles into HBM) and, over them, implementations of the reference's two operator
// interfaces -- rlwe.EvaluatorProvider (core/rlwe/rlwe.go:10-18) and schemes.Evaluator (schemes/schemes.go:14-28) -- so that
// the reference's circuit drivers (lintrans, polynomial evaluation, mod1, bootstrapping) run on it unchanged.
//
// NOTE: the build image of this repository has no Go toolchain; this package is shipped as source and is checked against the
// C header by tools/check_go_abi.py (every C.he_* call: the symbol exists and the argument count matches).
package hering

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../lattigo_amd -lhering -Wl,-rpath,${SRCDIR}/../../lattigo_amd
#include <stdlib.h>
#include "hering.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"sync/atomic"
	"unsafe"

	"github.com/tuneinsight/lattigo/v6/ring"
)

// Handle is an opaque libhering object (context, ring, polynomial batch, key ...).
type Handle = C.he_handle

// check turns a libhering status into an error.  he_last_error() is thread-local: the calling goroutine is pinned to its OS
// thread for the duration of a failing call chain by the callers below (lockedCall).
func check(rc C.int) error {
	if rc == 0 {
		return nil
	}
	return errors.New("hering: " + C.GoString(C.he_last_error()))
}

// lockedCall runs f on a pinned OS thread so that a non-zero status and its message are read on the same thread.  `keep` lists
// the Go objects whose handles f passes to C: handles are plain integers, so without it the collector could run a finalizer
// (he_*_destroy / he_poly_free) on an object whose last Go reference was the read of its handle, while the call is in flight.
func lockedCall(f func() C.int, keep ...any) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	err := check(f())
	runtime.KeepAlive(keep)
	return err
}

// Context owns one HIP stream on one GPU; all work of the objects created from it is enqueued there.
type Context struct {
	h Handle
	// graphs captured on this context that are still alive: their nodes address device polynomials by pointer, so the
	// evaluators of the context do not evict twins meanwhile (Evaluator.evict)
	liveGraphs int32
}

// NewContext opens GPU `device`.  There is no CPU fallback: without a HIP device this fails.
func NewContext(device int) (*Context, error) {
	c := &Context{}
	if err := lockedCall(func() C.int { return C.he_ctx_create(C.int(device), &c.h) }); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(c, func(c *Context) { C.he_ctx_destroy(c.h) })
	return c, nil
}

// Sync waits for everything enqueued on the context.
func (c *Context) Sync() error { return lockedCall(func() C.int { return C.he_ctx_sync(c.h) }) }

// SetCoalescing configures the context's submission queue (he_ctx_set_coalescing, include/hering.h): single-ciphertext calls of
// ANY operator made at the same time from different goroutines -- ring methods, Rescale, the rlwe.EvaluatorProvider methods,
// MulRelin -- are executed as batched launches over the callers' own device twins.  maxBatch <= 1 switches it off.
func (c *Context) SetCoalescing(maxBatch, windowMicros int) error {
	return lockedCall(func() C.int { return C.he_ctx_set_coalescing(c.h, C.int(maxBatch), C.int(windowMicros)) })
}

// SetDeferred switches the queue to deferred submission (he_ctx_set_deferred, include/hering.h): a queued call returns once it is
// filed and the context's dispatcher thread launches it; a failed launch is reported by the next Sync.  The library orders a
// caller's requests by OS thread, so a goroutine that issues calls in this mode must stay on its thread: call
// runtime.LockOSThread() at its start (a goroutine that migrated between two calls could see them launched out of order).
// depth = 0 switches back to calls that return once launched.
func (c *Context) SetDeferred(depth int) error {
	return lockedCall(func() C.int { return C.he_ctx_set_deferred(c.h, C.int(depth)) })
}

// Graph is a captured sequence of calls on a Context (he_graph_*, include/hering.h): one enqueue replays them all.
type Graph struct {
	ctx *Context
	h   Handle
}

// Capture records the device work f enqueues on c instead of executing it.  f must have run once before (plans and scratch are
// built on first use) and must not upload, download or Sync.  The replay reads and writes the same device polynomials: while a
// graph of the context is alive, its evaluators keep every device twin (no eviction); a caller that Forgets or frees a
// polynomial a graph addresses must Close the graph first.  A panic inside f still ends the capture.
func (c *Context) Capture(f func() error) (g *Graph, err error) {
	if err = lockedCall(func() C.int { return C.he_graph_begin(c.h) }); err != nil {
		return nil, err
	}
	g = &Graph{ctx: c}
	ended := false
	end := func() error {
		ended = true
		return lockedCall(func() C.int { return C.he_graph_end(c.h, &g.h) })
	}
	defer func() {
		if r := recover(); r != nil {
			if !ended && end() == nil { // leave the context usable, drop what was recorded
				C.he_graph_destroy(g.h)
			}
			panic(r)
		}
	}()
	ferr := f()
	err = end()
	if ferr != nil {
		if err == nil {
			C.he_graph_destroy(g.h)
		}
		return nil, ferr
	}
	if err != nil {
		return nil, err
	}
	atomic.AddInt32(&c.liveGraphs, 1)
	runtime.SetFinalizer(g, func(g *Graph) { g.Close() })
	return g, nil
}

// Close destroys the graph (idempotent); the context's evaluators may evict twins again once no graph is left.
func (g *Graph) Close() {
	if g.h != 0 {
		C.he_graph_destroy(g.h)
		g.h = 0
		atomic.AddInt32(&g.ctx.liveGraphs, -1)
	}
}

// Launch enqueues the captured sequence on the context's stream.
func (g *Graph) Launch() error {
	err := lockedCall(func() C.int { return C.he_graph_launch(g.h) })
	runtime.KeepAlive(g)
	return err
}

// Version of the loaded library.
func Version() string { return C.GoString(C.he_version()) }

// Ring is the device twin of a ring.Ring: same moduli chain, tables built by the library (ring/subring.go:99-159 restated in
// csrc/host_math.cpp).  Like ring.Ring.AtLevel, AtLevel returns a shallow copy carrying the level.
type Ring struct {
	ctx   *Context
	h     Handle
	n     int
	level int
	host  *ring.Ring
	own   *ringOwner // shared by every AtLevel copy: the handle lives as long as any of them
}

// ringOwner carries the finalizer of a ring handle.  AtLevel copies hold a pointer to it, so he_ring_destroy runs only when the
// original AND every copy are unreachable (a finalizer on the original alone would pull the handle from under the copies).
type ringOwner struct{ h Handle }

// NewRing mirrors ring.NewRingFromType for an existing reference ring.
func NewRing(ctx *Context, r *ring.Ring) (*Ring, error) {
	moduli := r.ModuliChain()
	d := &Ring{ctx: ctx, n: r.N(), level: r.MaxLevel(), host: r}
	typ := 0
	if r.Type() == ring.ConjugateInvariant {
		typ = 1
	}
	err := lockedCall(func() C.int {
		return C.he_ring_create_type(ctx.h, C.int(r.LogN()), C.int(typ), (*C.uint64_t)(unsafe.Pointer(&moduli[0])), C.int(len(moduli)), &d.h)
	})
	if err != nil {
		return nil, err
	}
	d.own = &ringOwner{h: d.h}
	runtime.SetFinalizer(d.own, func(o *ringOwner) { C.he_ring_destroy(o.h) })
	return d, nil
}

// AtLevel: ring.Ring.AtLevel (ring/ring.go:186).  The copy shares the handle and its owner.
func (r *Ring) AtLevel(level int) *Ring { c := *r; c.level = level; return &c }

// Level, N as the reference.
func (r *Ring) Level() int { return r.level }
func (r *Ring) N() int     { return r.n }

// Poly is a device-resident batch of ring.Poly: [batch][limbs][N] uint64 in HBM.
type Poly struct {
	h      Handle
	limbs  int
	batch  int
	shared bool // a view owned by somebody else: no finalizer
}

// NewPoly allocates a zeroed batch at the ring's level (ring.Ring.NewPoly).
func (r *Ring) NewPoly(batch int) (*Poly, error) { return r.newPoly(batch, true) }

// NewScratch allocates without clearing: for results the next operation overwrites (rlwe.BufferPool semantics,
// core/rlwe/pool.go:12-60).
func (r *Ring) NewScratch(batch int) (*Poly, error) { return r.newPoly(batch, false) }

func (r *Ring) newPoly(batch int, zero bool) (*Poly, error) {
	p := &Poly{limbs: r.level + 1, batch: batch}
	err := lockedCall(func() C.int {
		if zero {
			return C.he_poly_alloc(r.h, C.int(r.level+1), C.int(batch), &p.h)
		}
		return C.he_poly_alloc_scratch(r.h, C.int(r.level+1), C.int(batch), &p.h)
	})
	if err != nil {
		return nil, err
	}
	runtime.SetFinalizer(p, func(p *Poly) { C.he_poly_free(p.h) })
	return p, nil
}

// Upload copies batch entry b from a reference polynomial, one row per call (a [][]uint64 cannot cross cgo; each row is
// borrowed for the duration of its call only).
func (p *Poly) Upload(b int, src ring.Poly) error {
	if len(src.Coeffs) > p.limbs {
		return fmt.Errorf("hering: polynomial has %d limbs, the device twin %d", len(src.Coeffs), p.limbs)
	}
	for i, row := range src.Coeffs {
		row := row
		if err := lockedCall(func() C.int {
			return C.he_poly_upload_limb(p.h, C.int(b), C.int(i), (*C.uint64_t)(unsafe.Pointer(&row[0])))
		}); err != nil {
			return err
		}
	}
	return nil
}

// Download copies batch entry b back into a reference polynomial.
func (p *Poly) Download(b int, dst ring.Poly) error {
	for i, row := range dst.Coeffs {
		if i >= p.limbs {
			break
		}
		row := row
		if err := lockedCall(func() C.int {
			return C.he_poly_download_limb(p.h, C.int(b), C.int(i), (*C.uint64_t)(unsafe.Pointer(&row[0])))
		}); err != nil {
			return err
		}
	}
	return nil
}

// CopyLvl: ring.Poly.CopyLvl.
func (p *Poly) CopyLvl(level int, src *Poly) error {
	return lockedCall(func() C.int { return C.he_poly_copy(p.h, src.h, C.int(level)) })
}
