"""Fixtures shared by the `-m gpu` parity tests: one context, rings on both sides."""
import numpy as np
import pytest

import lattigo_amd as la
from oracle import oracle as O
from tests.conftest import Pi60, Qi60


@pytest.fixture(scope="session")
def ctx():
    c = la.Context(0)
    yield c
    c.sync()


class Pair:
    """The same ring on the GPU (la.*) and in the oracle (O.*)."""

    def __init__(self, ctx, logN, nq, np_=0, qmods=None, pmods=None, ci=False):
        """ci: conjugate-invariant rings Z[X + X^-1]/(X^2N + 1) (moduli = 1 mod 4N)"""
        self.N = 1 << logN
        self.q = list(qmods) if qmods else Qi60[:nq]
        self.p = list(pmods) if pmods else Pi60[:np_]
        self.gQ, self.oQ = la.Ring(ctx, self.N, self.q, conjugate_invariant=ci), O.Ring(self.N, self.q, ci)
        if self.p:
            self.gP, self.oP = la.Ring(ctx, self.N, self.p, conjugate_invariant=ci), O.Ring(self.N, self.p, ci)

    def up(self, ring, arr, batch=1):
        arr = np.asarray(arr, dtype=np.uint64)
        nl = arr.shape[-2]
        return la.Poly(ring, nl, batch).upload(arr)
