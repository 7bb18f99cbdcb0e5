"""Host-side mirror of utils/cosine (cosine_approx.go): the Han-Ki approximation of cos(2 pi (x - 1/4) / 2^r) that is only
accurate where the bootstrapping needs it -- in windows of half-width 1/dev around the integers of [-K + 1, K - 1] -- obtained by
interpolation at Chebyshev-type nodes of those windows and expressed in the Chebyshev basis of [-K / 2^r, K / 2^r].
100-digit decimal arithmetic (the reference: 256-bit floats)."""
from __future__ import annotations

import decimal
import math

D = decimal.Decimal
PREC = 100


def _pi():
    from .mod1 import _pi as p
    return p(PREC)


def _cos(x):
    """cos in the current decimal context (argument reduction + Taylor series)"""
    pi = _pi()
    two_pi = 2 * pi
    x = abs(x) % two_pi
    if x > pi:
        x = two_pi - x
    sign = 1
    if x > pi / 2:
        x, sign = pi - x, -1
    x2, term, total, k = x * x, D(1), D(0), 0
    eps = D(10) ** -(PREC + 5)
    while abs(term) > eps:
        total += term
        k += 2
        term = -term * x2 / (k * (k - 1))
    return sign * total


def _gen_degrees(degree: int, K: int, dev: float):
    """genDegrees (cosine_approx.go:63-150): how many interpolation nodes each integer window gets"""
    log2, log2TwoPi = math.log2, math.log2(2 * math.pi)
    degbdd, totdeg, err = degree + 1, 2 * K - 1, 1.0 / dev
    deg = [1] * K
    temp = -sum(log2(float(i)) for i in range(1, 2 * K)) + (2 * float(K) - 1) * log2TwoPi + log2(err)
    bdd = []
    for i in range(K):
        b = temp
        for j in range(1, K - 1 - i + 1):
            b += log2(float(j) + err)
        for j in range(1, K - 1 + i + 1):
            b += log2(float(j) + err)
        bdd.append(b)
    for _ in range(200):
        if totdeg >= degbdd:
            break
        maxi = max(range(K), key=lambda i: (bdd[i], -i))  # first index of the maximum
        if maxi != 0:
            if totdeg + 2 > degbdd:
                break
            for i in range(K):
                bdd[i] -= log2(float(totdeg + 1)) + log2(float(totdeg + 2))
                bdd[i] += 2.0 * log2TwoPi
                if i != maxi:
                    bdd[i] += log2(abs(float(i - maxi)) + err) + log2(float(i + maxi) + err)
                else:
                    bdd[i] += log2(err) - 1.0 + log2(2.0 * float(i) + err)
            totdeg += 2
        else:
            bdd[0] -= log2(float(totdeg + 1))
            bdd[0] += log2(err) - 1.0 + log2TwoPi
            for i in range(1, K):
                bdd[i] -= log2(float(totdeg + 1))
                bdd[i] += log2TwoPi + log2(float(i) + err)
            totdeg += 1
        deg[maxi] += 1
    return deg, totdeg


def ApproximateCos(K: int, degree: int, dev: float, scnum: int):
    """ApproximateCos (cosine_approx.go:22-30): Chebyshev coefficients (decimals) on [-K / 2^scnum, K / 2^scnum]"""
    with decimal.localcontext() as ctx:
        ctx.prec = PREC
        pi = _pi()
        deg, totdeg = _gen_degrees(degree, K, dev)
        scfac, intersize = D(1 << scnum), D(1) / D(dev)
        # genNodes (:152-215)
        nodes = [D(0)] * totdeg
        cnt = 1 if deg[0] % 2 != 0 else 0
        for i in range(K - 1, 0, -1):
            for j in range(deg[i]):
                t = _cos(pi * (2 * j) / (2 * deg[i])) * intersize
                nodes[cnt] = D(i) + t
                nodes[cnt + 1] = -nodes[cnt]
                cnt += 2
        for j in range(deg[0] // 2):
            t = _cos(pi * (2 * j) / (2 * deg[0])) * intersize
            nodes[cnt] = t
            nodes[cnt + 1] = -t
            cnt += 2
        # cos2PiXMinusQuarterOverR (:32-43) shifts and scales its argument IN PLACE: from here on the nodes are the
        # polynomial's own variable t = (x - 1/4) / 2^scnum
        nodes = [(x - D("0.25")) / scfac for x in nodes]
        y = [_cos(2 * pi * t) for t in nodes]
        # solve (:217-330): Newton divided differences, resampling at Chebyshev points, Chebyshev coefficients
        for j in range(1, totdeg):
            for i in range(totdeg - j):
                y[i] = (y[i + 1] - y[i]) / (nodes[i + j] - nodes[i])
        n = totdeg + 1
        half = D(K) / scfac
        xs = [half * _cos(D(i) * pi / (n - 1)) for i in range(n)]
        p = []
        for i in range(n):
            v = y[0]
            for j in range(1, n - 1):
                v = v * (xs[i] - nodes[j]) + y[j]
            p.append(v)
        T = []
        for i in range(n):
            row = [D(1), xs[i] / half]
            for j in range(2, n):
                row.append(2 * (xs[i] / half) * row[j - 1] - row[j - 2])
            T.append(row)
        for i in range(n - 1):  # Gaussian elimination with partial pivoting
            mx = max(range(i, n), key=lambda r: (abs(T[r][i]), -r))
            if mx != i:
                T[mx], T[i] = T[i], T[mx]
                p[mx], p[i] = p[i], p[mx]
            piv = T[i][i]
            for j in range(i + 1, n):
                T[i][j] /= piv
            p[i] /= piv
            T[i][i] = D(1)
            for j in range(i + 1, n):
                f = T[j][i]
                p[j] -= f * p[i]
                for l in range(i + 1, n):
                    T[j][l] -= f * T[i][l]
                T[j][i] = D(0)
        c = [D(0)] * n
        c[n - 1] = p[n - 1] / T[n - 1][n - 1]
        for i in range(n - 2, -1, -1):
            c[i] = p[i] - sum(T[i][j] * c[j] for j in range(i + 1, n))
        return [+v for v in c[:totdeg]]
