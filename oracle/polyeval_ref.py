"""TEST INFRASTRUCTURE (never imported by lattigo_amd/).  A second, independently organised restatement of the reference's
polynomial evaluation on ciphertexts, written from the Go sources and not from tests/drivers/polyeval.py, so that the
GPU tests stop comparing that driver with itself:

  circuits/common/polynomial/polynomial_evaluator.go:33-359   Evaluate, baby / giant steps, monomial combination
  circuits/common/polynomial/polynomial.go:32-141             Factorize metadata, PatersonStockmeyerPolynomial, recursePS
  circuits/common/polynomial/power_basis.go:31-160            SplitDegree, GenPower / genPower (lazy, Chebyshev)
  circuits/common/polynomial/polynomial_evaluator_sim.go      SimPowerBasis
  circuits/ckks/polynomial/polynomial_evaluator_sim.go:21-89  level / scale planning, CKKS (LevelsConsumedPerRescaling = 1)
  circuits/bgv/polynomial/polynomial_evaluator_sim.go:19-104  level / scale planning, BGV (standard tensoring)
  utils/bignum/polynomial.go:14-24, 258-314                   OptimalSplit, Factorize (monomial / Chebyshev)
  circuits/ckks/mod1/mod1_evaluator.go:28-144                 mod1 Evaluate around the polynomial (evaluate_mod1)

Organisation (deliberately unlike the driver's class-per-Go-type mirror): a polynomial is a plain dict, the level / scale
planner is one object holding two closures per scheme, planning returns a flat list of leaves, and the evaluator is a set of
module functions over any ``schemes.Evaluator``-shaped backend.  ``Trace`` wraps a backend and records, for every primitive
call, (name, level, scale, degree) of the destination -- the level / scale schedule itself -- so a test can require two
implementations to issue the same primitive sequence, not only to land on the same final words.
Scales: exact rationals (CKKS) or residues mod t (BGV); coefficients: (re, im) Fractions (CKKS) or ints mod t (BGV)."""
from __future__ import annotations

from fractions import Fraction


# ------------------------------------------------------------------------------------------------------------------
# recording proxy
# ------------------------------------------------------------------------------------------------------------------
class Trace:
    _DST_LAST = ("Add", "Sub", "Mul", "MulRelin", "MulThenAdd", "Relinearize", "Rescale")
    _DST_RET = ("MulNew", "MulRelinNew", "NewCiphertext", "CopyNew")

    def __init__(self, backend):
        self._b, self.log = backend, []

    def __getattr__(self, name):
        target = getattr(self._b, name)
        if name not in self._DST_LAST and name not in self._DST_RET:
            return target

        def call(*a, **k):
            out = target(*a, **k)
            dst = out if name in self._DST_RET else a[-1]
            self.log.append((name, dst.Level(), dst.Scale, dst.Degree()))
            return out

        return call


# ------------------------------------------------------------------------------------------------------------------
# integer helpers
# ------------------------------------------------------------------------------------------------------------------
def _blen(n: int) -> int:
    return int(n).bit_length()


def optimal_split(log_degree: int) -> int:
    """bignum.OptimalSplit: the baby-step width that minimises non-scalar multiplications"""
    s = log_degree // 2
    rest = log_degree - s
    cost_here = 2 ** s + 2 ** rest + rest - 3
    cost_next = 2 ** (s + 1) + 2 ** (rest - 1) + rest - 4
    return s + 1 if cost_here > cost_next else s


def split_degree(n: int):
    """power_basis.go:31-49: X^n = X^a * X^b with the depth-optimal (a, b)"""
    assert n > 0
    if n & (n - 1) == 0:
        return n >> 1, n >> 1
    top = 1 << (_blen(n - 1) - 1)
    return top - 1, n + 1 - top


def _wanted(i: int, even: bool, odd: bool) -> bool:
    """the parity filter the reference applies everywhere: both flags (or neither) set = keep every index"""
    if not even and not odd:
        return True
    return even if i % 2 == 0 else odd


# ------------------------------------------------------------------------------------------------------------------
# coefficients and polynomials (dicts)
# ------------------------------------------------------------------------------------------------------------------
def _as_pair(c):
    if isinstance(c, tuple):
        return Fraction(c[0]), Fraction(c[1])
    if isinstance(c, complex):
        return Fraction(c.real), Fraction(c.imag)
    return Fraction(c), Fraction(0)


def make_poly(coeffs, basis="Monomial", even=True, odd=True, lazy=False):
    c = list(coeffs)
    return dict(c=c, basis=basis, even=even, odd=odd, lazy=lazy, lead=True, maxdeg=len(c) - 1, level=None, scale=None)


def _deg(p) -> int:
    return len(p["c"]) - 1


def _split_poly(p, n: int, arith):
    """p = q * X^n + r (monomial) or q * T_n + r (Chebyshev, T_i = 2 T_n T_{i-n} - T_{2n-i}); metadata as in
    polynomial.go:32-52: the quotient inherits MaxDeg and Lead, the remainder gets n-1 or the shortened MaxDeg"""
    add, neg = arith
    d = _deg(p)
    assert n >= d >> 1
    src = p["c"]
    r = list(src[:n])
    q = [None] * (d - n + 1)
    q[0] = src[n]
    for i in range(n + 1, d + 1):
        if src[i] is None or not _wanted(i, p["even"], p["odd"]):
            continue
        if p["basis"] == "Chebyshev":
            q[i - n] = add(src[i], src[i])
            mirror = 2 * n - i
            r[mirror] = add(r[mirror], neg(src[i])) if r[mirror] is not None else neg(src[i])
        else:
            q[i - n] = src[i]
    common = dict(basis=p["basis"], even=p["even"], odd=p["odd"], lazy=False, level=None, scale=None)
    pq = dict(common, c=q, lead=p["lead"], maxdeg=p["maxdeg"])
    pr = dict(common, c=r, lead=False, maxdeg=(n - 1) if p["maxdeg"] == d else p["maxdeg"] - (d - n + 1))
    return pq, pr


# ------------------------------------------------------------------------------------------------------------------
# level / scale planner
# ------------------------------------------------------------------------------------------------------------------
class Planner:
    """The simulated evaluator of either scheme.  An operand is a (level, scale) tuple."""

    def __init__(self, Q, t=None):
        self.Q = [int(q) for q in Q]
        self.t = None if t is None else int(t)
        if self.t is None:
            self.smul = lambda a, b: Fraction(a) * Fraction(b)
            self.sdiv = lambda a, b: Fraction(a) / Fraction(b)
            self.same = lambda a, b: Fraction(a) == Fraction(b)
        else:
            t_ = self.t
            self.smul = lambda a, b: int(a) * int(b) % t_
            self.sdiv = lambda a, b: int(a) * pow(int(b), -1, t_) % t_
            self.same = lambda a, b: int(a) % t_ == int(b) % t_

    def depth(self, degree: int) -> int:
        assert degree > 0
        return _blen(degree) - 1

    def product(self, x, y):
        return min(x[0], y[0]), self.smul(x[1], y[1])

    def rescaled(self, x):
        return x[0] - 1, self.sdiv(x[1], self.Q[x[0]])

    def leaf(self, lead: bool, level: int, scale):
        """UpdateLevelAndScaleBabyStep: a leading leaf is evaluated one modulus larger, to be rescaled at the very end"""
        return level, (self.smul(scale, self.Q[level]) if lead else scale)

    def quotient(self, lead: bool, level: int, scale, xpow_scale):
        """UpdateLevelAndScaleGiantStep: the quotient of a split lives one level higher; its scale is chosen so that
        rescale(q) * X^n lands exactly on the scale of the remainder"""
        q = self.Q[level] if lead else self.Q[level + 1]
        return level + 1, self.sdiv(self.smul(scale, q), xpow_scale)

    def powers(self, level, scale, upto_pow2: int, log_split: int):
        table = {1: (level, scale)}

        def need(n):
            if n < 2:
                return
            a, b = split_degree(n)
            need(a)
            need(b)
            table[n] = self.rescaled(self.product(table[a], table[b]))  # recomputed on every visit, like the reference

        need(upto_pow2)
        for i in range((1 << log_split) - 1, 2, -1):
            need(i)
        return table


def plan_leaves(planner: Planner, poly, level: int, scale, target_scale, arith):
    """PatersonStockmeyerPolynomial + recursePS: the flat list of baby-step polynomials, each with its level and scale, in
    the reference's order (quotients before remainders)"""
    d = _deg(poly)
    log_degree = _blen(d)
    log_split = optimal_split(log_degree)
    xp = planner.powers(level, scale, 1 << log_degree, log_split)
    leaves = []

    def descend(p, ls, lvl, out_scale):
        dp = _deg(p)
        if dp < (1 << ls):
            md = p["maxdeg"]
            if p["lead"] and ls > 1 and md > (1 << _blen(md)) - (1 << (ls - 1)):
                return descend(p, optimal_split(_blen(dp)), lvl, out_scale)
            p["level"], p["scale"] = planner.leaf(p["lead"], lvl, out_scale)
            leaves.append(p)
            return p["level"], p["scale"]
        step = 1 << ls
        while step < (dp >> 1) + 1:
            step <<= 1
        pq, pr = _split_poly(p, step, arith)
        q_level, q_scale = planner.quotient(p["lead"], lvl, out_scale, xp[step][1])
        top = descend(pq, ls, q_level, q_scale)
        top = planner.product(planner.rescaled(top), xp[step])
        low = descend(pr, ls, lvl, top[1])
        if not planner.same(low[1], top[1]):
            raise RuntimeError(f"plan: remainder scale {low[1]} != quotient * X^{step} scale {top[1]}")
        return top

    descend(poly, log_split, level - planner.depth(d), target_scale)
    return leaves


# ------------------------------------------------------------------------------------------------------------------
# evaluation over a schemes.Evaluator-shaped backend
# ------------------------------------------------------------------------------------------------------------------
def _batch_of(ct):
    v = ct.Value[0]
    return getattr(v, "batch", 1)


def _new_ct(ev, degree, level, like):
    try:
        return ev.NewCiphertext(degree, level, _batch_of(like))
    except TypeError:
        return ev.NewCiphertext(degree, level)


def grow_powers(ev, X: dict, basis: str, n: int, lazy: bool):
    """PowerBasis.GenPower: make X[n] (and what it needs), rescaled"""
    if n in X:
        return

    def build(m, lz) -> bool:
        """genPower: returns whether X[m] was produced now (its rescale is then owed by the caller)"""
        if m in X:
            return False
        a, b = split_degree(m)
        pow2 = m & (m - 1) == 0
        owe_a = build(a, lz and not pow2)
        owe_b = build(b, lz and not pow2)
        if lz:
            for k in (a, b):
                if X[k].Degree() == 2:
                    ev.Relinearize(X[k], X[k])
        if owe_a:
            ev.Rescale(X[a], X[a])
        if owe_b:
            ev.Rescale(X[b], X[b])
        X[m] = ev.MulNew(X[a], X[b]) if lz else ev.MulRelinNew(X[a], X[b])
        if basis == "Chebyshev":  # T_m = 2 T_a T_b - T_|a-b|
            gap = abs(a - b)
            ev.Add(X[m], X[m], X[m])
            if gap == 0:
                ev.Add(X[m], -1, X[m])
            else:
                grow_powers(ev, X, basis, gap, lz)
                ev.Sub(X[m], X[gap], X[m])
        return True

    if build(n, lazy):
        ev.Rescale(X[n], X[n])


def eval_leaf(ev, X: dict, leaf):
    """EvaluatePolynomialVectorFromPowerBasis, mapping == nil: sum_k c_k X[k] at the leaf's level and scale"""
    c, even, odd = leaf["c"], leaf["even"], leaf["odd"]
    d = len(c) - 1
    low = d - 1 if (even and not odd) else d
    widest = max([X[i].Degree() for i in range(d, 0, -1) if i in X], default=0)
    out = _new_ct(ev, 1 if low == 0 else widest, leaf["level"], X[1])
    out.Scale = leaf["scale"]
    if even:
        ev.Add(out, c[0], out)
    if low == 0:
        return out
    for k in range(d, 0, -1):
        if _wanted(k, even, odd):
            ev.MulThenAdd(X[k], c[k], out)
    return out


def combine(ev, planner: Planner, X: dict, parts):
    """EvaluatePatersonStockmeyerPolynomialVector after the baby steps: parts = [[degree, ct], ...] in ascending order of
    the reference's reversed list; neighbours of equal degree merge as low + rescale(high) * X^(2^k)"""
    while len(parts) > 1:
        n = len(parts)
        action = [0] * n
        i = 0
        while i < n:
            if i == n - 1:
                action[i] = 2
            elif parts[i][0] == parts[i + 1][0]:
                action[i] = 1
                i += 1
            i += 1
        for i in range(n):
            if action[i] == 2:
                parts[i][0] = parts[i - 1][0]
            elif action[i] == 1:
                low, high = parts[i], parts[i + 1]
                width = 1 << _blen(low[0])
                hv, xv = high[1], X[width]
                if hv.Degree() == 2:
                    ev.Relinearize(hv, hv)
                ev.Rescale(hv, hv)
                ev.Mul(hv, xv, hv)
                if not planner.same(low[1].Scale, hv.Scale):
                    raise RuntimeError(f"combine: scale discrepancy {hv.Scale} != {low[1].Scale}")
                ev.Add(hv, low[1], hv)
                high[0] = 2 * width - 1
                parts[i] = None
        parts = [p for p in parts if p is not None]
    res = parts[0][1]
    if res.Degree() == 2:
        ev.Relinearize(res, res)
    ev.Rescale(res, res)
    return res


def evaluate_polynomial(ev, ct, coeffs, target_scale, basis="Monomial", even=True, odd=True, lazy=False):
    """polynomial.Evaluator.Evaluate for one polynomial on one ciphertext.  ``ev``: a bgv.Evaluator mirror (has .t) or a
    ckks.Evaluator mirror; ``coeffs[i]`` may be None where the parity flags exclude index i."""
    t = getattr(ev, "t", None)
    if t is not None:
        t = int(t)
        cs = [None if c is None else int(c) % t for c in coeffs]
        arith = (lambda a, b: (a + b) % t, lambda a: (-a) % t)
        target_scale = int(target_scale) % t
    else:
        cs = [None if c is None else _as_pair(c) for c in coeffs]
        arith = (lambda a, b: (a[0] + b[0], a[1] + b[1]), lambda a: (-a[0], -a[1]))
        target_scale = Fraction(target_scale)
    poly = make_poly(cs, basis, even, odd, lazy)
    d = _deg(poly)
    X = {1: ev.CopyNew(ct)}
    depth = (d - 1).bit_length() if d > 1 else 0  # ceil(log2 d)
    if X[1].Level() < depth:
        raise ValueError(f"{X[1].Level()} levels < {depth} log(d) -> cannot evaluate poly")
    log_degree = _blen(d)
    log_split = optimal_split(log_degree)
    grow_powers(ev, X, basis, 1 << (log_degree - 1), False)
    for i in range((1 << log_split) - 1, 2, -1):
        if _wanted(i, even, odd):
            grow_powers(ev, X, basis, i, lazy)
    planner = Planner(ev.Q, t)
    leaves = plan_leaves(planner, poly, X[1].Level(), X[1].Scale, target_scale, arith)
    parts = [None] * len(leaves)
    for i, leaf in enumerate(leaves):
        parts[len(leaves) - 1 - i] = [len(leaf["c"]) - 1, eval_leaf(ev, X, leaf)]
    return combine(ev, planner, X, parts)


# ------------------------------------------------------------------------------------------------------------------
# mod1 (circuits/ckks/mod1/mod1_evaluator.go:28-144) around the polynomial; the approximation's coefficients are inputs
# ------------------------------------------------------------------------------------------------------------------
def _keep_bits(x, bits=128):
    """nearest value with a `bits`-bit significand, ties to even (a big.Float result of that precision)"""
    x = Fraction(x)
    if x == 0:
        return x
    lo, hi = -4096, 4096  # find s with 2^(bits-1) <= x 2^s < 2^bits by bisection on the exponent
    while lo < hi:
        mid = (lo + hi) // 2
        if x * Fraction(2) ** mid >= 1 << (bits - 1):
            hi = mid
        else:
            lo = mid + 1
    scaled = x * Fraction(2) ** lo
    down = scaled.numerator // scaled.denominator
    frac = scaled - down
    up = frac > Fraction(1, 2) or (frac == Fraction(1, 2) and down % 2 == 1)
    return Fraction(down + (1 if up else 0)) / Fraction(2) ** lo


def _sqrt_bits(x, bits=128):
    """square root to `bits` bits, correctly rounded: bracket it between consecutive (bits+8)-bit integers, then keep `bits`"""
    import math
    x = Fraction(x)
    shift = 0
    while x * Fraction(4) ** shift < 1 << (2 * (bits + 8)):
        shift += 1
    while shift > 0 and x * Fraction(4) ** (shift - 1) >= 1 << (2 * (bits + 8)):
        shift -= 1
    big = x * Fraction(4) ** shift
    root = math.isqrt(big.numerator // big.denominator)
    # root <= sqrt(big) < root + 1; the true root is irrational or equal to `root`: a half-unit nudge cannot cross a rounding boundary
    est = Fraction(root) if Fraction(root * root) == big else Fraction(2 * root + 1, 2)
    return _keep_bits(est / Fraction(2) ** shift, bits)


def evaluate_mod1(ev, ct, *, level_q, log_scale, cosine: bool, K: float, double_angle: int, sqrt2pi, poly_coeffs, poly_even,
                  poly_odd, inv_coeffs=None):
    import math
    if ct.Level() < level_q:
        raise ValueError("cannot Evaluate: ct.Level() < Mod1Parameters.LevelQ")
    x = ev.CopyNew(ct)
    if x.Level() > level_q:
        if hasattr(ev, "_resize"):
            ev._resize(x, x.Degree(), level_q)
        else:
            ev._set(x, x.Value, level_q)
    x.Scale = Fraction(1 << log_scale)
    Q = [int(q) for q in ev.Q]
    d = len(poly_coeffs) - 1
    poly_depth = (d - 1).bit_length() if d > 1 else 0
    goal = Fraction(x.Scale)
    for i in range(double_angle):  # each squaring doubles the scale exponent: undo it ahead of time, in 128-bit floats like the reference
        goal = _sqrt_bits(_keep_bits(goal * Q[x.Level() - poly_depth - double_angle + i + 1]))
    if cosine:  # cos(2 pi (y - 1/4)) = sin(2 pi y)
        shrink = 2.0 ** double_angle
        shift = Fraction(-0.5) / (Fraction(2 * (K / shrink)) * Fraction(shrink))
        ev.Add(x, (shift, 0), x)
    y = evaluate_polynomial(ev, x, poly_coeffs, goal, "Chebyshev", poly_even, poly_odd)
    c = Fraction(sqrt2pi)
    for _ in range(double_angle):
        c = c * c
        ev.MulRelin(y, y, y)
        ev.Add(y, y, y)
        ev.Add(y, (-c, 0), y)
        ev.Rescale(y, y)
    if inv_coeffs is not None:
        y = evaluate_polynomial(ev, y, inv_coeffs, y.Scale, "Monomial", False, True)
    y.Scale = ct.Scale
    return y


class Mod1Ref:
    """``EvaluateNew`` over evaluate_mod1 for callers that hold a mod1 evaluator object (the bootstrapping tests); ``params`` is
    anything carrying the reference's Mod1Parameters fields (LevelQ, LogDefaultScale, Mod1Type, K, DoubleAngle, Sqrt2Pi,
    Mod1Poly, Mod1InvPoly -- mod1_parameters.go:68-92); Mod1Type 1 is SinContinuous."""

    def __init__(self, ev, params):
        self.ev, self.Parameters = ev, params

    def EvaluateNew(self, ct):
        pm = self.Parameters
        return evaluate_mod1(self.ev, ct, level_q=pm.LevelQ, log_scale=pm.LogDefaultScale, cosine=pm.Mod1Type != 1, K=pm.K,
                             double_angle=pm.DoubleAngle, sqrt2pi=pm.Sqrt2Pi, poly_coeffs=pm.Mod1Poly.Coeffs,
                             poly_even=pm.Mod1Poly.IsEven, poly_odd=pm.Mod1Poly.IsOdd,
                             inv_coeffs=None if pm.Mod1InvPoly is None else pm.Mod1InvPoly.Coeffs)
