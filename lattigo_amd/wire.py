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
