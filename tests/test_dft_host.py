"""drivers.dft host logic (numpy only): the special FFT, its factorisation into sparse diagonal-form factors, and the
double-precision slot encoder's rounding (circuits/ckks/dft/dft.go:368-470; schemes/ckks/encoder.go:160-330)."""
from fractions import Fraction

import numpy as np
import pytest

from drivers import dft as DFT
from oracle import oracle as O


def apply(diags, v):
    return sum(d * np.roll(v, -k) for k, d in diags.items())


@pytest.mark.parametrize("logN,groups", [(5, [(0, 4)]), (9, [(0, 4), (4, 8)]), (10, [(0, 3), (3, 6), (6, 9)])])
def test_factors_multiply_back_to_the_special_fft(logN, groups):
    N, n = 1 << logN, 1 << (logN - 1)
    rng = np.random.default_rng(logN)
    w = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    br = DFT.bitrev_indices(n)
    v = w[br]
    for f in DFT.factor_diagonals(N, groups, False):
        v = apply(f, v)
    assert np.max(np.abs(v - DFT.special_fft(w, N))) < 1e-10
    # the slot roots: z_j = sum_k w_k zeta_j^k with zeta_j = exp(i pi 5^j / N)
    j = 3
    zeta = np.exp(1j * np.pi * pow(5, j, 2 * N) / N)
    assert abs(v[j] - np.sum(w * zeta ** np.arange(n))) < 1e-9
    for f in DFT.factor_diagonals(N, [g for g in reversed(groups)], True):
        v = apply(f, v)
    assert np.max(np.abs(v[br] - w)) < 1e-10
    assert np.max(np.abs(DFT.special_ifft(DFT.special_fft(w, N), N) - w)) < 1e-12


def test_encode_rounds_the_scaled_coefficients_exactly():
    logN = 6
    N, n = 1 << logN, 1 << (logN - 1)
    q, _ = O.GenModuli(logN + 1, [50, 40], [])
    rng = np.random.default_rng(3)
    z = rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)
    scale = Fraction(1 << 35)
    res = DFT.fast_encode_rns(z, N, scale, q)
    w = DFT.special_ifft(z, N)
    c = np.concatenate([w.real, w.imag]) * float(scale)
    want = np.rint(c).astype(np.int64)
    for i, qi in enumerate(q):
        assert np.array_equal(res[i].astype(object), np.array([int(x) % int(qi) for x in want], dtype=object))
