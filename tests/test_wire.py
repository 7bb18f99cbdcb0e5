"""Host-side wire format of ring.Poly (utils/structs/matrix.go:82-150): layout and round trip."""
import struct

import numpy as np
import pytest

from lattigo_amd import wire


def test_poly_layout_and_roundtrip():
    a = np.arange(3 * 8, dtype=np.uint64).reshape(3, 8) * np.uint64(0x0101010101010101)
    b = wire.poly_marshal(a)
    assert len(b) == wire.poly_binary_size(3, 8)
    assert struct.unpack_from("<Q", b, 0)[0] == 3 and struct.unpack_from("<Q", b, 8)[0] == 8
    assert struct.unpack_from("<Q", b, 16)[0] == 0 and struct.unpack_from("<Q", b, 24)[0] == 0x0101010101010101
    assert np.array_equal(wire.poly_unmarshal(b), a)


def test_poly_unmarshal_rejects_short_and_ragged():
    a = np.ones((2, 4), dtype=np.uint64)
    b = wire.poly_marshal(a)
    with pytest.raises(ValueError):
        wire.poly_unmarshal(b[:-1])
    ragged = struct.pack("<Q", 2) + struct.pack("<Q", 1) + struct.pack("<Q", 7) + struct.pack("<Q", 2) + struct.pack("<QQ", 1, 2)
    with pytest.raises(ValueError):
        wire.poly_unmarshal(ragged)
    assert wire.poly_unmarshal(struct.pack("<Q", 0)).size == 0


def test_gadget_ciphertext_layout_sizes_and_roundtrip():
    rng = np.random.default_rng(7)
    N, LQ, LP = 16, 3, 2
    for base_two, nj in ((0, [1, 1]), (12, [3, 2, 3])):
        D = sum(nj)
        kq = rng.integers(0, 1 << 62, size=(D, 2, LQ, N), dtype=np.uint64)
        kp = rng.integers(0, 1 << 62, size=(D, 2, LP, N), dtype=np.uint64)
        b = wire.gadget_ciphertext_marshal(kq, kp, base_two, nj)
        # BinarySize identities of the reference (gadgetciphertext.go:86, matrix.go, vector.go, ringqp/poly.go)
        assert len(b) == wire.gadget_ciphertext_binary_size(nj, LQ, LP, N)
        assert struct.unpack_from("<QQQ", b, 0) == (base_two, len(nj), nj[0])
        assert struct.unpack_from("<QQQ", b, 24) == (2, LQ, N)  # Vector[ringqp.Poly] count, then Q as a matrix
        assert struct.unpack_from("<Q", b, 48)[0] == int(kq[0, 0, 0, 0])
        q2, p2, bt, nj2 = wire.gadget_ciphertext_unmarshal(b)
        assert bt == base_two and nj2 == nj and np.array_equal(q2, kq) and np.array_equal(p2, kp)
        g = wire.galois_key_marshal(5, 2 * N, kq, kp, base_two, nj)
        assert len(g) == 16 + len(b) and g[16:] == b
        ge, nth, q3, p3, bt3, nj3 = wire.galois_key_unmarshal(g)
        assert (ge, nth, bt3, nj3) == (5, 2 * N, base_two, nj) and np.array_equal(q3, kq) and np.array_equal(p3, kp)
        with pytest.raises(ValueError):
            wire.gadget_ciphertext_unmarshal(b[:-3])
    with pytest.raises(ValueError):
        wire.gadget_ciphertext_marshal(kq, kp, 12, [1, 1])


def test_metadata_json_has_the_reference_fixed_size():
    from fractions import Fraction
    for scale in (1, 2 ** 45, 2.0 ** 40 + 0.5, Fraction(2 ** 90 + 1, 3), 65537, 1e-3):
        m = wire.metadata_marshal(scale=scale, scale_mod=65537, log_cols=13, is_ntt=True, is_montgomery=False)
        assert len(m) == wire.METADATA_BINARY_SIZE == 277  # metadata.go:30: 44 + (84 + 111) + 38
        d = wire.metadata_unmarshal(m)
        assert abs(d["scale"] / Fraction(scale) - 1) < Fraction(1, 10 ** 38)
        assert d["scale_mod"] == 65537 and d["log_cols"] == 13 and d["is_ntt"] and not d["is_montgomery"]
    m = wire.metadata_marshal(scale=2 ** 45)
    assert b'"Value":"3.518437208883200000000000000000000000000e+13"' in m
    assert b'"Mod":"0.000000000000000000000000000000000000000e+00"' in m
    assert m.endswith(b'"CiphertextMetaData":{"IsNTT":"0x01","IsMontgomery":"0x00"}}')
    assert wire._bigfloat_text(Fraction(99999999999999999999999999999999999999999, 10 ** 40)).startswith("1.0000")


def test_ciphertext_roundtrip():
    rng = np.random.default_rng(9)
    v = rng.integers(0, 1 << 60, size=(2, 3, 8), dtype=np.uint64)
    meta = wire.metadata_marshal(scale=2 ** 30, log_cols=2)
    b = wire.ciphertext_marshal(v, meta)
    assert len(b) == 1 + wire.METADATA_BINARY_SIZE + 8 + 2 * wire.poly_binary_size(3, 8)  # element.go:318-331
    v2, d = wire.ciphertext_unmarshal(b)
    assert np.array_equal(v2, v) and d["scale"] == 2 ** 30 and d["log_cols"] == 2
    v3, d3 = wire.ciphertext_unmarshal(wire.ciphertext_marshal(v, None))
    assert d3 is None and np.array_equal(v3, v)


def test_literal_bytes_assembled_from_the_reference_layout_rules():
    """Byte strings written out by hand from the reference's layout rules -- Matrix.WriteTo = u64 row count, then every row as a
    Vector (utils/structs/matrix.go:82-106); Vector.WriteTo = u64 length, then the elements, uint64 elements little-endian
    (utils/structs/vector.go:82-100, utils/buffer/writer.go:311-325); ringqp.Poly = Q then P (ring/ringqp/poly.go:105-125);
    GadgetCiphertext = u64 BaseTwoDecomposition, then Value as Matrix[VectorQP] (core/rlwe/gadgetciphertext.go:101-118) -- and
    compared with lattigo_amd.wire in both directions.  This pins wire.py to the documented layout; it is NOT a comparison with
    bytes produced by Go (no toolchain here, and the reference holds no regenerable serialised fixture): parity with real Go
    output remains unpinned."""
    # ring.Poly{Coeffs: [[1, 2], [0x0102030405060708, 3]]}
    poly_hex = ("0200000000000000"    # Matrix: 2 rows (limbs)
                "0200000000000000" "0100000000000000" "0200000000000000"   # Vector len 2: 1, 2
                "0200000000000000" "0807060504030201" "0300000000000000")  # Vector len 2: 0x0102030405060708, 3
    coeffs = np.array([[1, 2], [0x0102030405060708, 3]], dtype=np.uint64)
    assert wire.poly_marshal(coeffs) == bytes.fromhex(poly_hex)
    assert np.array_equal(wire.poly_unmarshal(bytes.fromhex(poly_hex)), coeffs)

    # GadgetCiphertext{BaseTwoDecomposition: 0, Value: [[ [ {Q:[[10,11]], P:[[20,21]]}, {Q:[[12,13]], P:[[22,23]]} ] ]]}
    # (one RNS digit, one power-of-two block, the two components of the key; one Q limb, one P limb, N = 2)
    def u(x):
        return struct.pack("<Q", x).hex()
    gct_hex = (u(0) +                      # BaseTwoDecomposition
               u(1) +                      # Matrix[VectorQP]: 1 row (RNS digits)
               u(1) +                      # Vector[VectorQP]: 1 block (bit windows of that digit)
               u(2) +                      # VectorQP = Vector[ringqp.Poly]: 2 components
               u(1) + u(2) + u(10) + u(11) +   # component 0, Q: Matrix 1 row; Vector len 2: 10, 11
               u(1) + u(2) + u(20) + u(21) +   # component 0, P
               u(1) + u(2) + u(12) + u(13) +   # component 1, Q
               u(1) + u(2) + u(22) + u(23))    # component 1, P
    kq = np.array([[[[10, 11]], [[12, 13]]]], dtype=np.uint64)   # [beta][2][LQ][N]
    kp = np.array([[[[20, 21]], [[22, 23]]]], dtype=np.uint64)
    assert wire.gadget_ciphertext_marshal(kq, kp, 0, [1]) == bytes.fromhex(gct_hex)
    q2, p2, bt, nj = wire.gadget_ciphertext_unmarshal(bytes.fromhex(gct_hex))
    assert bt == 0 and nj == [1] and np.array_equal(q2, kq) and np.array_equal(p2, kp)
    # base-2 gadget: BaseTwoDecomposition = 3, one digit with two bit windows
    gct2_hex = (u(3) + u(1) + u(2) +
                u(2) + u(1) + u(2) + u(10) + u(11) + u(1) + u(2) + u(20) + u(21) + u(1) + u(2) + u(12) + u(13) + u(1) + u(2) + u(22) + u(23) +
                u(2) + u(1) + u(2) + u(30) + u(31) + u(1) + u(2) + u(40) + u(41) + u(1) + u(2) + u(32) + u(33) + u(1) + u(2) + u(42) + u(43))
    kq2 = np.array([[[[10, 11]], [[12, 13]]], [[[30, 31]], [[32, 33]]]], dtype=np.uint64)
    kp2 = np.array([[[[20, 21]], [[22, 23]]], [[[40, 41]], [[42, 43]]]], dtype=np.uint64)
    assert wire.gadget_ciphertext_marshal(kq2, kp2, 3, [2]) == bytes.fromhex(gct2_hex)
    # GaloisKey = u64 GaloisElement, u64 NthRoot, then the EvaluationKey (core/rlwe/keys.go:628-657)
    assert wire.galois_key_marshal(5, 8, kq, kp, 0, [1]) == bytes.fromhex(u(5) + u(8) + gct_hex)


def test_reference_generated_fixture():
    """Bytes produced by the reference's OWN MarshalBinary (tests/golden/wire/*.bin, written by tests/golden/wire/gen_wire_fixture.go
    -- needs a Go toolchain, which this repository's image does not have: until a maintainer runs it, this test skips and the
    wire format stays "parity unpinned", see README.md).  Every object's words follow the closed formula of the generator, so the
    expected arrays are rebuilt here and compared with lattigo_amd.wire in both directions: parse(Go bytes) == arrays and
    marshal(arrays) == Go bytes."""
    import json
    import os
    import pytest
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wire")
    if not os.path.exists(os.path.join(d, "manifest.json")):
        pytest.skip("no reference-generated fixture (run tests/golden/wire/gen_wire_fixture.go with a Go toolchain)")
    m = json.load(open(os.path.join(d, "manifest.json")))
    N, Q, P = 1 << m["LogN"], [int(x) for x in m["Q"]], [int(x) for x in m["P"]]
    word = lambda tag, mods: np.array([[(tag * 1000003 + i * 7919 + j * 104729 + 1) % q for j in range(N)] for i, q in enumerate(mods)],
                                      dtype=np.uint64)
    rd = lambda name: open(os.path.join(d, name), "rb").read()
    # ring.Poly
    assert np.array_equal(wire.poly_unmarshal(rd("poly.bin")), word(1, Q))
    assert wire.poly_marshal(word(1, Q)) == rd("poly.bin")
    # ringqp.Poly
    assert wire.polyqp_marshal(word(2, Q), word(3, P)) == rd("polyqp.bin")
    # rlwe.GadgetCiphertext / rlwe.GaloisKey
    beta = m["beta"]
    kq = np.stack([np.stack([word(100 + 10 * dg + c, Q) for c in range(2)]) for dg in range(beta)])
    kp = np.stack([np.stack([word(200 + 10 * dg + c, P) for c in range(2)]) for dg in range(beta)])
    q2, p2, base_two, nj = wire.gadget_ciphertext_unmarshal(rd("gadget.bin"))
    assert base_two == 0 and nj == [1] * beta and np.array_equal(q2, kq) and np.array_equal(p2, kp)
    assert wire.gadget_ciphertext_marshal(kq, kp, 0, [1] * beta) == rd("gadget.bin")
    g, nth, q3, p3, bt3, nj3 = wire.galois_key_unmarshal(rd("galoiskey.bin"))
    assert (g, nth, bt3, nj3) == (m["GaloisElement"], m["NthRoot"], 0, [1] * beta) and np.array_equal(q3, kq) and np.array_equal(p3, kp)
    assert wire.galois_key_marshal(m["GaloisElement"], m["NthRoot"], kq, kp, 0, [1] * beta) == rd("galoiskey.bin")
    # rlwe.Ciphertext incl. its JSON MetaData
    value, meta = wire.ciphertext_unmarshal(rd("ciphertext.bin"))
    assert np.array_equal(value, np.stack([word(7, Q), word(8, Q)]))
    assert meta["scale"] == 1 << m["LogScale"] and meta["is_ntt"] and meta["is_montgomery"]

