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

    def __init__(self, ctx, logN, nq, np_=0, qmods=None, pmods=None):
        self.N = 1 << logN
        self.q = list(qmods) if qmods else Qi60[:nq]
        self.p = list(pmods) if pmods else Pi60[:np_]
        self.gQ, self.oQ = la.Ring(ctx, self.N, self.q), O.Ring(self.N, self.q)
        if self.p:
            self.gP, self.oP = la.Ring(ctx, self.N, self.p), O.Ring(self.N, self.p)

    def up(self, ring, arr, batch=1):
        arr = np.asarray(arr, dtype=np.uint64)
        nl = arr.shape[-2]
        return la.Poly(ring, nl, batch).upload(arr)
