// gen_wire_fixture.go -- produces the byte fixtures tests/test_wire.py::test_reference_generated_fixture consumes.
//
// NEEDS A GO TOOLCHAIN (there is none in this repository's build image, which is why the fixtures are not committed yet):
//
//	cd <a checkout of github.com/tuneinsight/lattigo at v6.2.0>
//	cp <this repo>/tests/golden/wire/gen_wire_fixture.go ./cmd_gen_wire_fixture/main.go
//	go run ./cmd_gen_wire_fixture <this repo>/tests/golden/wire
//
// It serialises, with the reference's own WriteTo / MarshalBinary (ring/poly.go:132-176, ring/ringqp/poly.go:105-176,
// core/rlwe/gadgetciphertext.go:101-165, core/rlwe/keys.go:628-707, core/rlwe/element.go:335-432), objects whose words
// follow a closed formula instead of the keyed PRNG, so that the Python side can rebuild them without Lattigo's sampler:
//
//	word(tag, i, j, q) = (tag*1000003 + i*7919 + j*104729 + 1) mod q        (limb i, coefficient j, modulus q)
//
// Files written: poly.bin (ring.Poly, tag 1), polyqp.bin (ringqp.Poly, tags 2 / 3), gadget.bin (rlwe.GadgetCiphertext: digit d,
// component c -> tags 100 + 10 d + c for Q and 200 + 10 d + c for P), galoiskey.bin (the same gadget as a rlwe.GaloisKey with
// GaloisElement 5), ciphertext.bin (rlwe.Ciphertext, degree 1, tags 7 / 8, IsNTT = IsMontgomery = true, Scale 2^40) and
// manifest.json (the parameters the Python test reads back).
package main

import (
	"encoding/json"
	"fmt"
	"os"
	"path/filepath"

	"github.com/tuneinsight/lattigo/v6/core/rlwe"
	"github.com/tuneinsight/lattigo/v6/ring"
	"github.com/tuneinsight/lattigo/v6/ring/ringqp"
)

func word(tag, i, j int, q uint64) uint64 {
	return (uint64(tag)*1000003 + uint64(i)*7919 + uint64(j)*104729 + 1) % q
}

func fill(p ring.Poly, tag int, moduli []uint64) {
	for i := range p.Coeffs {
		for j := range p.Coeffs[i] {
			p.Coeffs[i][j] = word(tag, i, j, moduli[i])
		}
	}
}

func must(err error) {
	if err != nil {
		panic(err)
	}
}

func write(dir, name string, data []byte, err error) {
	must(err)
	must(os.WriteFile(filepath.Join(dir, name), data, 0o644))
}

func main() {
	if len(os.Args) != 2 {
		fmt.Println("usage: gen_wire_fixture OUTPUT_DIR")
		os.Exit(2)
	}
	dir := os.Args[1]
	params, err := rlwe.NewParametersFromLiteral(rlwe.ParametersLiteral{
		LogN:    8,
		LogQ:    []int{55, 45, 45},
		LogP:    []int{55, 46},
		NTTFlag: true,
	})
	must(err)
	Q, P := params.RingQ().ModuliChain(), params.RingP().ModuliChain()

	poly := params.RingQ().NewPoly()
	fill(poly, 1, Q)
	data, err := poly.MarshalBinary()
	write(dir, "poly.bin", data, err)

	qp := ringqp.Poly{Q: params.RingQ().NewPoly(), P: params.RingP().NewPoly()}
	fill(qp.Q, 2, Q)
	fill(qp.P, 3, P)
	data, err = qp.MarshalBinary()
	write(dir, "polyqp.bin", data, err)

	gct := rlwe.NewGadgetCiphertext(params, 1, params.MaxLevelQ(), params.MaxLevelP(), 0)
	for d := range gct.Value {
		for b := range gct.Value[d] {
			for c := range gct.Value[d][b] {
				fill(gct.Value[d][b][c].Q, 100+10*d+c, Q)
				fill(gct.Value[d][b][c].P, 200+10*d+c, P)
			}
		}
	}
	data, err = gct.MarshalBinary()
	write(dir, "gadget.bin", data, err)

	gk := rlwe.NewGaloisKey(params)
	gk.GaloisElement = 5
	gk.GadgetCiphertext = *gct
	data, err = gk.MarshalBinary()
	write(dir, "galoiskey.bin", data, err)

	ct := rlwe.NewCiphertext(params, 1, params.MaxLevel())
	fill(ct.Value[0], 7, Q)
	fill(ct.Value[1], 8, Q)
	ct.IsNTT, ct.IsMontgomery = true, true
	ct.Scale = rlwe.NewScale(1 << 40)
	data, err = ct.MarshalBinary()
	write(dir, "ciphertext.bin", data, err)

	manifest, err := json.MarshalIndent(map[string]any{
		"lattigo": "v6.2.0", "LogN": 8, "Q": Q, "P": P, "NthRoot": params.RingQ().NthRoot(),
		"beta": len(gct.Value), "GaloisElement": 5, "LogScale": 40,
	}, "", " ")
	write(dir, "manifest.json", manifest, err)
	fmt.Println("wrote poly.bin polyqp.bin gadget.bin galoiskey.bin ciphertext.bin manifest.json to", dir)
}
