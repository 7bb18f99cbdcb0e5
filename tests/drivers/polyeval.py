"""Host-side mirror of circuits/common/polynomial + circuits/bgv/polynomial: Paterson-Stockmeyer evaluation of a polynomial
with integer coefficients on a BGV ciphertext (power basis, baby steps from the power basis, giant steps by monomials,
level / scale planning by the simulated evaluator).  Pure control flow over a ``schemes.Evaluator``-shaped backend
(``schemes.BGVCiphertextEvaluator`` / ``schemes.CKKSCiphertextEvaluator`` on the device); every ring operation it triggers
runs in the HIP kernels.  Monomial and Chebyshev bases, single polynomial (no slot mapping); the CKKS instantiation
(circuits/ckks/polynomial) keeps scales as exact rationals."""
from __future__ import annotations


def OptimalSplit(logDegree: int) -> int:
    """utils/bignum/polynomial.go:14"""
    logSplit = logDegree >> 1
    a = (1 << logSplit) + (1 << (logDegree - logSplit)) + logDegree - logSplit - 3
    b = (1 << (logSplit + 1)) + (1 << (logDegree - logSplit - 1)) + logDegree - logSplit - 4
    return logSplit + 1 if a > b else logSplit


def SplitDegree(n: int):
    """circuits/common/polynomial/power_basis.go:31"""
    if n <= 0:
        raise ValueError(f"invalid n: n={n} should be greater than zero")
    if n & (n - 1) == 0:
        return n // 2, n // 2
    k = (n - 1).bit_length() - 1
    return (1 << k) - 1, n + 1 - (1 << k)


class Polynomial:
    """polynomial.Polynomial over bignum.Polynomial in the monomial basis (circuits/common/polynomial/polynomial.go:12-30,
    utils/bignum/polynomial.go:44-151); NewPolynomial marks it both even and odd, i.e. no parity filtering."""

    def __init__(self, coeffs, MaxDeg=None, Lead=True, Lazy=False, Basis="Monomial"):
        self.Coeffs = list(coeffs)
        self.Basis = Basis
        self.IsOdd = self.IsEven = True  # bignum.NewPolynomial: both set = no parity filtering; clear one for odd / even polys
        self.MaxDeg = len(self.Coeffs) - 1 if MaxDeg is None else MaxDeg
        self.Lead, self.Lazy = Lead, Lazy
        self.Level, self.Scale = 0, 1

    def Degree(self):
        return len(self.Coeffs) - 1

    def Depth(self):
        d = self.Degree()
        return (d - 1).bit_length() if d > 1 else 0  # ceil(log2(degree))

    def Factorize(self, n: int):
        """p = q * X^n + r, resp. q * T_n + r (polynomial.go:32-52 over utils/bignum/polynomial.go:258-314)"""
        if n < self.Degree() >> 1:
            raise ValueError("cannot Factorize: n < p.Degree()/2")
        even, odd = self.IsEven, self.IsOdd
        keep = lambda i: self.Coeffs[i] is not None and (not (even or odd) or (i & 1 == 0 and even) or (i & 1 == 1 and odd))
        r = list(self.Coeffs[:n])
        q = [None] * (self.Degree() - n + 1)
        q[0] = self.Coeffs[n]
        for i in range(n + 1, self.Degree() + 1):
            if not keep(i):
                continue
            if self.Basis == "Chebyshev":  # T_i = 2 T_n T_{i-n} - T_{2n-i}
                j = i - n
                q[i - n] = _cadd(self.Coeffs[i], self.Coeffs[i])
                r[n - j] = _csub(r[n - j], self.Coeffs[i]) if r[n - j] is not None else _csub(0, self.Coeffs[i])
            else:
                q[i - n] = self.Coeffs[i]
        pr = Polynomial(r, Lead=False, Basis=self.Basis)
        pq = Polynomial(q, Lead=False, Basis=self.Basis)
        pq.MaxDeg = self.MaxDeg
        pr.MaxDeg = n - 1 if self.MaxDeg == self.Degree() else self.MaxDeg - (self.Degree() - n + 1)
        pq.Lead = self.Lead
        pq.Lazy = pr.Lazy = False
        pq.IsOdd = pr.IsOdd = self.IsOdd
        pq.IsEven = pr.IsEven = self.IsEven
        return pq, pr


def _cpair(c):
    from fractions import Fraction
    if isinstance(c, tuple):
        return Fraction(c[0]), Fraction(c[1])
    if isinstance(c, complex):
        return Fraction(c.real), Fraction(c.imag)
    return Fraction(c), Fraction(0)


def _cadd(a, b):
    (ar, ai), (br, bi) = _cpair(a), _cpair(b)
    return (ar + br, ai + bi)


def _csub(a, b):
    (ar, ai), (br, bi) = _cpair(a), _cpair(b)
    return (ar - br, ai - bi)


class SimOperand:
    def __init__(self, Level, Scale):
        self.Level, self.Scale = Level, Scale


class BGVSimEvaluator:
    """circuits/bgv/polynomial/polynomial_evaluator_sim.go (standard tensoring)"""

    def __init__(self, Q, t):
        self.Q, self.t = [int(q) for q in Q], int(t)

    def _div(self, a, b):
        return a * pow(b, -1, self.t) % self.t

    def PolynomialDepth(self, degree: int) -> int:
        if degree <= 0:
            raise ValueError(f"invalid degree: degree={degree} should be greater than zero")
        return degree.bit_length() - 1

    def Rescale(self, op0: SimOperand):
        op0.Scale = self._div(op0.Scale, self.Q[op0.Level] % self.t)
        op0.Level -= 1

    def MulNew(self, op0: SimOperand, op1: SimOperand) -> SimOperand:
        return SimOperand(min(op0.Level, op1.Level), op0.Scale * op1.Scale % self.t)

    def UpdateLevelAndScaleBabyStep(self, lead, tLevelOld, tScaleOld):
        return tLevelOld, (tScaleOld * (self.Q[tLevelOld] % self.t) % self.t if lead else tScaleOld)

    def UpdateLevelAndScaleGiantStep(self, lead, tLevelOld, tScaleOld, xPowScale):
        tLevelNew = tLevelOld
        tScaleNew = self._div(tScaleOld, xPowScale)
        currentQi = self.Q[tLevelNew] if lead else self.Q[tLevelNew + 1]
        return tLevelNew + 1, tScaleNew * (currentQi % self.t) % self.t


class CKKSSimEvaluator:
    """circuits/ckks/polynomial/polynomial_evaluator_sim.go with LevelsConsumedPerRescaling = 1; exact rational scales"""

    def __init__(self, Q):
        from fractions import Fraction
        self.Q = [Fraction(int(q)) for q in Q]

    def PolynomialDepth(self, degree: int) -> int:
        if degree <= 0:
            raise ValueError(f"invalid degree: degree={degree} should be greater than zero")
        return degree.bit_length() - 1

    def Rescale(self, op0: SimOperand):
        op0.Scale = op0.Scale / self.Q[op0.Level]
        op0.Level -= 1

    def MulNew(self, op0: SimOperand, op1: SimOperand) -> SimOperand:
        return SimOperand(min(op0.Level, op1.Level), op0.Scale * op1.Scale)

    def UpdateLevelAndScaleBabyStep(self, lead, tLevelOld, tScaleOld):
        return tLevelOld, (tScaleOld * self.Q[tLevelOld] if lead else tScaleOld)

    def UpdateLevelAndScaleGiantStep(self, lead, tLevelOld, tScaleOld, xPowScale):
        qi = self.Q[tLevelOld] if lead else self.Q[tLevelOld + 1]
        return tLevelOld + 1, tScaleOld * qi / xPowScale


def _sim_gen_power(d: dict, n: int, sim):
    """SimPowerBasis.GenPower (polynomial_evaluator_sim.go:25-38)"""
    if n < 2:
        return
    a, b = SplitDegree(n)
    _sim_gen_power(d, a, sim)
    _sim_gen_power(d, b, sim)
    d[n] = sim.MulNew(d[a], d[b])
    sim.Rescale(d[n])


def _recurse_ps(logSplit, targetLevel, p: Polynomial, pb: dict, outputScale, sim):
    """recursePS (polynomial.go:92-141)"""
    if p.Degree() < (1 << logSplit):
        if p.Lead and logSplit > 1 and p.MaxDeg > (1 << p.MaxDeg.bit_length()) - (1 << (logSplit - 1)):
            logDegree = p.Degree().bit_length()
            return _recurse_ps(OptimalSplit(logDegree), targetLevel, p, pb, outputScale, sim)
        p.Level, p.Scale = sim.UpdateLevelAndScaleBabyStep(p.Lead, targetLevel, outputScale)
        return [p], SimOperand(p.Level, p.Scale)
    nextPower = 1 << logSplit
    while nextPower < (p.Degree() >> 1) + 1:
        nextPower <<= 1
    XPow = pb[nextPower]
    coeffsq, coeffsr = p.Factorize(nextPower)
    tLevelNew, tScaleNew = sim.UpdateLevelAndScaleGiantStep(p.Lead, targetLevel, outputScale, XPow.Scale)
    bsgsQ, res = _recurse_ps(logSplit, tLevelNew, coeffsq, pb, tScaleNew, sim)
    sim.Rescale(res)
    res = sim.MulNew(res, XPow)
    bsgsR, tmp = _recurse_ps(logSplit, targetLevel, coeffsr, pb, res.Scale, sim)
    if tmp.Scale != res.Scale:
        raise RuntimeError(f"recursePS: res.Scale != tmp.Scale: {res.Scale} != {tmp.Scale}")
    return bsgsQ + bsgsR, res


def PatersonStockmeyerPolynomial(p: Polynomial, inputLevel, inputScale, outputScale, sim):
    """Polynomial.PatersonStockmeyerPolynomial (polynomial.go:62-90) -> the list of baby-step polynomials"""
    logDegree = p.Degree().bit_length()
    logSplit = OptimalSplit(logDegree)
    pb = {1: SimOperand(inputLevel, inputScale)}
    _sim_gen_power(pb, 1 << logDegree, sim)
    for i in range((1 << logSplit) - 1, 2, -1):
        _sim_gen_power(pb, i, sim)
    ps, _ = _recurse_ps(logSplit, inputLevel - sim.PolynomialDepth(p.Degree()), p, pb, outputScale, sim)
    return ps


class PowerBasis:
    """polynomial.PowerBasis (power_basis.go:17-160), monomial basis"""

    def __init__(self, ct, evaluator, Basis="Monomial"):
        self.Value = {1: evaluator.CopyNew(ct)}
        self.Basis = Basis

    def GenPower(self, n: int, lazy: bool, ev):
        if n not in self.Value:
            if self._gen(n, lazy, True, ev):
                ev.Rescale(self.Value[n], self.Value[n])

    def _gen(self, n, lazy, rescale, ev) -> bool:
        if n in self.Value:
            return False
        a, b = SplitDegree(n)
        isPow2 = n & (n - 1) == 0
        rescaleA = self._gen(a, lazy and not isPow2, rescale, ev)
        rescaleB = self._gen(b, lazy and not isPow2, rescale, ev)
        if lazy:
            for k in (a, b):
                if self.Value[k].Degree() == 2:
                    ev.Relinearize(self.Value[k], self.Value[k])
        if rescaleA:
            ev.Rescale(self.Value[a], self.Value[a])
        if rescaleB:
            ev.Rescale(self.Value[b], self.Value[b])
        self.Value[n] = ev.MulNew(self.Value[a], self.Value[b]) if lazy else ev.MulRelinNew(self.Value[a], self.Value[b])
        if self.Basis == "Chebyshev":  # T_n = 2 T_a T_b - T_|a-b| (power_basis.go:135-157)
            c = abs(a - b)
            ev.Add(self.Value[n], self.Value[n], self.Value[n])
            if c == 0:
                ev.Add(self.Value[n], -1, self.Value[n])
            else:
                self.GenPower(c, lazy, ev)
                ev.Sub(self.Value[n], self.Value[c], self.Value[n])
        return True


class _BabyStep:
    def __init__(self, Degree, Value):
        self.Degree, self.Value = Degree, Value


class PolynomialEvaluator:
    """polynomial.Evaluator[uint64] (circuits/common/polynomial/polynomial_evaluator.go) with the BGV coefficient getter
    and simulated evaluator (circuits/bgv/polynomial/polynomial_evaluator.go); `evaluator` is a bgv.Evaluator mirror
    (schemes.BGVCiphertextEvaluator)."""

    def __init__(self, evaluator):
        self.eval = evaluator
        self.bgv = getattr(evaluator, "t", None) is not None
        self.sim = BGVSimEvaluator(evaluator.Q, evaluator.t) if self.bgv else CKKSSimEvaluator(evaluator.Q)

    def Evaluate(self, ct, coeffs, targetScale: int):
        """Evaluator.Evaluate (:33-92): ct -> p(ct), p = sum coeffs[i] X^i over Z_t"""
        ev = self.eval
        if isinstance(coeffs, Polynomial):
            p = coeffs
        elif self.bgv:
            p = Polynomial([int(c) % ev.t for c in coeffs])
        else:
            p = Polynomial([_cpair(c) for c in coeffs])
        pb = PowerBasis(ct, ev, p.Basis)
        level, depth = pb.Value[1].Level(), p.Depth()
        if level < depth:
            raise ValueError(f"{level} levels < {depth} log(d) -> cannot evaluate poly")
        logDegree = p.Degree().bit_length()
        logSplit = OptimalSplit(logDegree)
        pb.GenPower(1 << (logDegree - 1), False, ev)
        even, odd = p.IsEven, p.IsOdd
        for i in range((1 << logSplit) - 1, 2, -1):
            if not (even or odd) or (i & 1 == 0 and even) or (i & 1 == 1 and odd):
                pb.GenPower(i, p.Lazy, ev)
        if self.bgv:
            targetScale = int(targetScale) % ev.t
        else:
            from fractions import Fraction
            targetScale = Fraction(targetScale)
        PS = PatersonStockmeyerPolynomial(p, pb.Value[1].Level(), pb.Value[1].Scale, targetScale, self.sim)
        return self.EvaluatePatersonStockmeyerPolynomial(PS, pb)

    def EvaluatePatersonStockmeyerPolynomial(self, polys, pb: PowerBasis):
        """EvaluatePatersonStockmeyerPolynomialVector (:100-160)"""
        ev = self.eval
        split = len(polys)
        babySteps = [None] * split
        for i in range(split):
            babySteps[split - i - 1] = self.EvaluateBabyStep(polys[i], pb)
        while len(babySteps) != 1:
            n = len(babySteps)
            giantsteps = [0] * n
            i = 0
            while i < n:
                if i == n - 1:
                    giantsteps[i] = 2
                elif babySteps[i].Degree == babySteps[i + 1].Degree:
                    giantsteps[i] = 1
                    i += 1
                i += 1
            for i in range(n):
                self.EvaluateGiantStep(i, giantsteps, babySteps, pb)
            babySteps = [b for b in babySteps if b is not None]
        res = babySteps[0].Value
        if res.Degree() == 2:
            ev.Relinearize(res, res)
        ev.Rescale(res, res)
        return res

    def EvaluateBabyStep(self, poly: Polynomial, pb: PowerBasis):
        return _BabyStep(poly.Degree(), self.EvaluatePolynomialFromPowerBasis(poly.Level, poly, pb, poly.Scale))

    def EvaluateGiantStep(self, i, giantSteps, babySteps, pb: PowerBasis):
        """:185-208 (the reference's local `i++` has no effect on the caller's loop)"""
        if giantSteps[i] == 2:
            babySteps[i].Degree = babySteps[i - 1].Degree
        elif giantSteps[i] == 1:
            even, odd = babySteps[i], babySteps[i + 1]
            deg = 1 << babySteps[i].Degree.bit_length()
            self.EvaluateMonomial(even.Value, odd.Value, pb.Value[deg])
            odd.Degree = 2 * deg - 1
            babySteps[i] = None

    def EvaluateMonomial(self, a, b, xpow):
        """b = rescale(b) * X^n + a (:211-236)"""
        ev = self.eval
        if b.Degree() == 2:
            ev.Relinearize(b, b)
        ev.Rescale(b, b)
        ev.Mul(b, xpow, b)
        if a.Scale != b.Scale:
            raise RuntimeError(f"evalMonomial: scale discrepency: (rescale(b) * X^n).Scale = {b.Scale} != a.Scale = {a.Scale}")
        ev.Add(b, a, b)

    def EvaluatePolynomialFromPowerBasis(self, targetLevel, pol: Polynomial, pb: PowerBasis, targetScale):
        """EvaluatePolynomialVectorFromPowerBasis, single polynomial (:239-359, mapping == nil branch)"""
        ev, X = self.eval, pb.Value
        B = X[1].Value[0].batch if hasattr(X[1].Value[0], "batch") else 1
        even, odd = pol.IsEven, pol.IsOdd
        minimumDegreeNonZeroCoefficient = len(pol.Coeffs) - 1
        if even and not odd:
            minimumDegreeNonZeroCoefficient -= 1
        maximumCiphertextDegree = 0
        for i in range(pol.Degree(), 0, -1):
            if i in X:
                maximumCiphertextDegree = max(maximumCiphertextDegree, X[i].Degree())
        if minimumDegreeNonZeroCoefficient == 0:
            res = ev.NewCiphertext(1, targetLevel, B)
            res.Scale = targetScale
            if even:
                ev.Add(res, pol.Coeffs[0], res)
            return res
        res = ev.NewCiphertext(maximumCiphertextDegree, targetLevel, B)
        res.Scale = targetScale
        if even:
            ev.Add(res, pol.Coeffs[0], res)
        for key in range(pol.Degree(), 0, -1):
            if not (even or odd) or (key & 1 == 0 and even) or (key & 1 == 1 and odd):
                ev.MulThenAdd(X[key], pol.Coeffs[key], res)
        return res
