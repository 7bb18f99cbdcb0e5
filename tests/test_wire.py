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
