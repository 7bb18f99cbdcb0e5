"""Wire format of ring.Poly, host-side only.

ring.Poly.WriteTo (ring/poly.go:132) writes Coeffs as a structs.Matrix[uint64]
(utils/structs/matrix.go:82-106): a little-endian uint64 row count, then per row a
little-endian uint64 length followed by the row's words (utils/structs/vector.go:82-100,
utils/buffer/writer.go).  This lets Go-produced polynomials (ciphertext components, key
limbs) be loaded straight into device handles.

Parity note: the reference holds no serialized fixture for this path that can be regenerated
without its blake2b-keyed sampler, so this module is restated from the writer/reader code
only ("parity unpinned" for N4)."""
from __future__ import annotations

import struct

import numpy as np


def poly_marshal(coeffs: np.ndarray) -> bytes:
    """[limbs, N] uint64 -> bytes of ring.Poly.MarshalBinary."""
    a = np.ascontiguousarray(coeffs, dtype="<u8")
    if a.ndim != 2:
        raise ValueError("expected a [limbs, N] array")
    out = [struct.pack("<Q", a.shape[0])]
    for row in a:
        out.append(struct.pack("<Q", row.shape[0]))
        out.append(row.tobytes())
    return b"".join(out)


def poly_unmarshal(buf: bytes) -> np.ndarray:
    """bytes of ring.Poly.MarshalBinary -> [limbs, N] uint64 (rows must share a length)."""
    if len(buf) < 8:
        raise ValueError("short buffer")
    (rows,) = struct.unpack_from("<Q", buf, 0)
    off = 8
    out = []
    for _ in range(rows):
        if off + 8 > len(buf):
            raise ValueError("short buffer")
        (n,) = struct.unpack_from("<Q", buf, off)
        off += 8
        if off + 8 * n > len(buf):
            raise ValueError("short buffer")
        out.append(np.frombuffer(buf, dtype="<u8", count=n, offset=off).astype(np.uint64))
        off += 8 * n
    if len({r.shape[0] for r in out}) > 1:
        raise ValueError("ragged polynomial")
    return np.stack(out) if out else np.zeros((0, 0), dtype=np.uint64)


def poly_binary_size(limbs: int, N: int) -> int:
    """ring.Poly.BinarySize"""
    return 8 + limbs * (8 + 8 * N)


# ----------------------------------------------------------------------------------------------------
# ringqp.Poly, rlwe.GadgetCiphertext / EvaluationKey / GaloisKey, rlwe.Ciphertext (Element[ring.Poly])
# ----------------------------------------------------------------------------------------------------
class _Reader:
    def __init__(self, buf: bytes, off: int = 0):
        self.buf, self.off = buf, off

    def u64(self) -> int:
        if self.off + 8 > len(self.buf):
            raise ValueError("short buffer")
        (v,) = struct.unpack_from("<Q", self.buf, self.off)
        self.off += 8
        return v

    def u8(self) -> int:
        if self.off + 1 > len(self.buf):
            raise ValueError("short buffer")
        v = self.buf[self.off]
        self.off += 1
        return v

    def take(self, n: int) -> bytes:
        if self.off + n > len(self.buf):
            raise ValueError("short buffer")
        v = self.buf[self.off: self.off + n]
        self.off += n
        return v

    def poly(self) -> np.ndarray:
        rows = self.u64()
        out = []
        for _ in range(rows):
            n = self.u64()
            out.append(np.frombuffer(self.take(8 * n), dtype="<u8").astype(np.uint64))
        if len({r.shape[0] for r in out}) > 1:
            raise ValueError("ragged polynomial")
        return np.stack(out) if out else np.zeros((0, 0), dtype=np.uint64)


def polyqp_marshal(q: np.ndarray, p: np.ndarray) -> bytes:
    """ringqp.Poly.WriteTo (ring/ringqp/poly.go:105): Q then P, each a ring.Poly."""
    return poly_marshal(q) + poly_marshal(p)


def gadget_ciphertext_marshal(kq: np.ndarray, kp: np.ndarray, base_two: int = 0, nj=None) -> bytes:
    """rlwe.GadgetCiphertext.WriteTo (core/rlwe/gadgetciphertext.go:101): uint64 BaseTwoDecomposition, then Value, a
    structs.Matrix[VectorQP] (utils/structs/matrix.go:82, vector.go:82, core/rlwe/keys.go:168): uint64 rows; per row
    uint64 length; per entry a Vector[ringqp.Poly] of the 2 key components.  kq/kp: [blocks][2][limbs][N] with the
    blocks of row i (its nj[i] base-2 digits; one per row when base_two == 0) stored consecutively."""
    kq, kp = np.asarray(kq, dtype=np.uint64), np.asarray(kp, dtype=np.uint64)
    D = kq.shape[0]
    nj = [1] * D if nj is None else [int(x) for x in nj]
    if sum(nj) != D or kp.shape[0] != D:
        raise ValueError("block count does not match the decomposition shape")
    out = [struct.pack("<Q", base_two), struct.pack("<Q", len(nj))]
    blk = 0
    for n in nj:
        out.append(struct.pack("<Q", n))
        for _ in range(n):
            out.append(struct.pack("<Q", kq.shape[1]))
            for k in range(kq.shape[1]):
                out.append(polyqp_marshal(kq[blk, k], kp[blk, k]))
            blk += 1
    return b"".join(out)


def _gadget_ciphertext_read(r: _Reader):
    base_two = r.u64()
    rows = r.u64()
    nj, bq, bp = [], [], []
    for _ in range(rows):
        n = r.u64()
        nj.append(n)
        for _ in range(n):
            comps = r.u64()
            cq, cp = [], []
            for _ in range(comps):
                cq.append(r.poly())
                cp.append(r.poly())
            bq.append(np.stack(cq))
            bp.append(np.stack(cp))
    return np.stack(bq), np.stack(bp), base_two, nj


def gadget_ciphertext_unmarshal(buf: bytes):
    """-> (kq [blocks][2][limbsQ][N], kp [blocks][2][limbsP][N], BaseTwoDecomposition, nj)"""
    return _gadget_ciphertext_read(_Reader(buf))


def gadget_ciphertext_binary_size(nj, limbsQ: int, limbsP: int, N: int, comps: int = 2) -> int:
    """rlwe.GadgetCiphertext.BinarySize (core/rlwe/gadgetciphertext.go:86)"""
    per = 8 + comps * (poly_binary_size(limbsQ, N) + poly_binary_size(limbsP, N))
    return 8 + 8 + sum(8 + n * per for n in nj)


def galois_key_marshal(galois_element: int, nth_root: int, kq, kp, base_two: int = 0, nj=None) -> bytes:
    """rlwe.GaloisKey.WriteTo (core/rlwe/keys.go:628): GaloisElement, NthRoot, then the EvaluationKey (= its
    GadgetCiphertext, keys.go:443)."""
    return struct.pack("<QQ", galois_element, nth_root) + gadget_ciphertext_marshal(kq, kp, base_two, nj)


def galois_key_unmarshal(buf: bytes):
    r = _Reader(buf)
    g, nth = r.u64(), r.u64()
    return (g, nth) + _gadget_ciphertext_read(r)


ScalePrecisionLog10 = 39  # ceil(128 / log2(10)), core/rlwe/scale.go:17


def _bigfloat_text(x) -> str:
    """big.Float.Text('e', 39) of an exactly representable value (int, float or Fraction)"""
    import decimal
    from fractions import Fraction
    fr = Fraction(x)
    with decimal.localcontext() as c:
        c.prec = 200
        c.rounding = decimal.ROUND_HALF_EVEN
        d = decimal.Decimal(fr.numerator) / decimal.Decimal(fr.denominator)
        if d == 0:
            return "0." + "0" * ScalePrecisionLog10 + "e+00"
        exp = d.adjusted()
        mant = (d.scaleb(-exp)).quantize(decimal.Decimal(1).scaleb(-ScalePrecisionLog10))
        if abs(mant) >= 10:  # rounding carried into a new digit
            mant, exp = (mant / 10).quantize(decimal.Decimal(1).scaleb(-ScalePrecisionLog10)), exp + 1
        return f"{mant:f}e{'+' if exp >= 0 else '-'}{abs(exp):02d}"


def metadata_marshal(scale=1, scale_mod=None, is_batched=True, is_bit_reversed=False, log_rows=0, log_cols=0,
                     is_ntt=True, is_montgomery=False) -> bytes:
    """rlwe.MetaData.MarshalBinary = MarshalJSON (core/rlwe/metadata.go:68-82,198-226,350-370; scale.go:192-218)"""
    mod = _bigfloat_text(scale_mod) if scale_mod is not None else "0." + "0" * ScalePrecisionLog10 + "e+00"
    s = ('{"PlaintextMetaData":{"Scale":{"Value":"%s","Mod":"%s"},"IsBatched":"0x%02x","IsBitReversed":"0x%02x",'
         '"LogDimensions":["0x%02x","0x%02x"]},"CiphertextMetaData":{"IsNTT":"0x%02x","IsMontgomery":"0x%02x"}}'
         % (_bigfloat_text(scale), mod, int(is_batched), int(is_bit_reversed), log_rows & 0xFF, log_cols & 0xFF,
            int(is_ntt), int(is_montgomery)))
    return s.encode()


METADATA_BINARY_SIZE = 44 + (84 + 21 + 2 * (ScalePrecisionLog10 + 6)) + 38  # metadata.go:30,151,303; scale.go:175


def metadata_unmarshal(buf: bytes) -> dict:
    import json
    from fractions import Fraction
    j = json.loads(buf.decode())
    pm, cm = j["PlaintextMetaData"], j["CiphertextMetaData"]
    hx = lambda s: int(s, 16)
    mod = Fraction(pm["Scale"]["Mod"])
    return {"scale": Fraction(pm["Scale"]["Value"]), "scale_mod": None if mod == 0 else mod,
            "is_batched": hx(pm["IsBatched"]) == 1, "is_bit_reversed": hx(pm["IsBitReversed"]) == 1,
            "log_rows": hx(pm["LogDimensions"][0]), "log_cols": hx(pm["LogDimensions"][1]),
            "is_ntt": hx(cm["IsNTT"]) == 1, "is_montgomery": hx(cm["IsMontgomery"]) == 1}


def ciphertext_marshal(value: np.ndarray, metadata: bytes | None) -> bytes:
    """rlwe.Element[ring.Poly].WriteTo (core/rlwe/element.go:335): a flag byte, the MetaData when present, then Value,
    a structs.Vector[ring.Poly] (uint64 count + the polynomials).  value: [degree+1][limbs][N]."""
    value = np.asarray(value, dtype=np.uint64)
    out = [b"\x01" + metadata if metadata is not None else b"\x00", struct.pack("<Q", value.shape[0])]
    for p in value:
        out.append(poly_marshal(p))
    return b"".join(out)


def ciphertext_unmarshal(buf: bytes):
    """-> (value [degree+1][limbs][N], metadata dict or None)"""
    r = _Reader(buf)
    meta = metadata_unmarshal(r.take(METADATA_BINARY_SIZE)) if r.u8() == 1 else None
    n = r.u64()
    return np.stack([r.poly() for _ in range(n)]), meta
