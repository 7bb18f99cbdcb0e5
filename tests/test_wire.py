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
