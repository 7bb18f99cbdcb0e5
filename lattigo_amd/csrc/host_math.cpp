// host_math.cpp -- see host_math.h.
#include "host_math.h"

#include <algorithm>

namespace he {

uint64_t mulmod(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)(((u128)a * b) % m); }
uint64_t powmod(uint64_t a, uint64_t e, uint64_t m) {
    uint64_t r = 1 % m;
    a %= m;
    while (e) {
        if (e & 1) r = mulmod(r, a, m);
        a = mulmod(a, a, m);
        e >>= 1;
    }
    return r;
}
uint64_t invmod(uint64_t a, uint64_t p) { return powmod(a % p, p - 2, p); }
uint64_t to_mont(uint64_t a, uint64_t q) { return (uint64_t)((((u128)(a % q)) << 64) % q); }

bool is_prime_u64(uint64_t n) {
    static const uint64_t bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return false;
    for (uint64_t b : bases)
        if (n % b == 0) return n == b;
    uint64_t d = n - 1;
    int s = 0;
    while (!(d & 1)) { d >>= 1; s++; }
    for (uint64_t b : bases) {
        uint64_t x = powmod(b, d, n);
        if (x == 1 || x == n - 1) continue;
        bool composite = true;
        for (int r = 1; r < s; r++) {
            x = mulmod(x, x, n);
            if (x == n - 1) { composite = false; break; }
        }
        if (composite) return false;
    }
    return true;
}

static uint64_t gcd_u64(uint64_t a, uint64_t b) {
    while (b) { uint64_t t = a % b; a = b; b = t; }
    return a;
}
static uint64_t rho_factor(uint64_t n) {
    if (!(n & 1)) return 2;
    for (uint64_t c = 1;; c++) {
        uint64_t x = 2, y = 2, d = 1;
        auto f = [&](uint64_t v) { return (mulmod(v, v, n) + c) % n; };
        while (d == 1) {
            x = f(x);
            y = f(f(y));
            d = gcd_u64(x > y ? x - y : y - x, n);
        }
        if (d != n) return d;
    }
}
static void factor_into(uint64_t n, std::vector<uint64_t> &out) {
    if (n == 1) return;
    if (is_prime_u64(n)) {
        if (std::find(out.begin(), out.end(), n) == out.end()) out.push_back(n);
        return;
    }
    uint64_t d = rho_factor(n);
    factor_into(d, out);
    factor_into(n / d, out);
}
std::vector<uint64_t> unique_prime_factors(uint64_t n) {
    std::vector<uint64_t> out;
    for (uint64_t p = 2; p < 4096 && p * p <= n; p++) {
        if (n % p == 0) {
            out.push_back(p);
            while (n % p == 0) n /= p;
        }
    }
    factor_into(n, out);
    return out;
}

static uint64_t bitrev(uint64_t x, int bits) {
    uint64_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

static bool build_subring_nth(int logN, uint64_t q, uint64_t nth, SubRingHost &out, std::string &err);
bool build_subring(int logN, uint64_t q, SubRingHost &out, std::string &err) {
    return build_subring_nth(logN, q, 2ull << logN, out, err);
}
bool build_subring_ci(int logN, uint64_t q, SubRingHost &out, std::string &err) {
    SubRingHost big;
    if (!build_subring_nth(logN, q, 4ull << logN, big, err)) return false;
    const uint64_t N = 1ull << logN;
    out.mc = big.mc;
    out.primroot = big.primroot;
    out.mc.pad0 = big.roots_fwd[1];
    out.mc.pad1 = big.roots_bwd[1];
    out.roots_fwd.assign(N, 0);
    out.roots_bwd.assign(N, 0);
    for (uint64_t h = 1; h < N; h <<= 1)
        for (uint64_t i = 0; i < h; i++) {
            out.roots_fwd[h + i] = big.roots_fwd[2 * h + i];
            out.roots_bwd[h + i] = big.roots_bwd[2 * h + i];
        }
    return true;
}
static bool build_subring_nth(int logN, uint64_t q, uint64_t nth, SubRingHost &out, std::string &err) {
    const uint64_t N = 1ull << logN;
    // as the reference: its inverse butterfly forms U + 4q - V in 64 bits ("not possible ... if Q > 61 bits", ring/ntt.go:169),
    // and the word-serial Montgomery products of the integer butterflies here need 5q < 2^64
    if (q >> 61) { err = "modulus must be below 2^61 (the reference's own limit, ring/ntt.go:169)"; return false; }
    if (!is_prime_u64(q)) { err = "invalid modulus: " + std::to_string(q) + " is not prime"; return false; }
    if ((q & (nth - 1)) != 1) { err = "invalid modulus: " + std::to_string(q) + " != 1 mod NthRoot"; return false; }
    ModConst &mc = out.mc;
    mc.q = q;
    // q^-1 mod 2^64 by Newton iteration
    uint64_t inv = q;  // correct to 3 bits
    for (int i = 0; i < 6; i++) inv *= 2 - q * inv;
    mc.qinv = inv;
    // floor(2^128 / q)
    u128 rem = ((u128)1 << 64) % q;
    mc.brc0 = (uint64_t)(((u128)1 << 64) / q);
    mc.brc1 = (uint64_t)((rem << 64) / q);
    const uint64_t half = nth >> 1;  // table length: N (standard) or 2N (conjugate invariant)
    int loghalf = 0;
    while ((1ull << loghalf) < half) loghalf++;
    mc.ninv = to_mont(invmod(half % q, q), q);
    mc.r2 = to_mont(to_mont(1, q), q);
    mc.pad0 = mc.pad1 = 0;
    mc.rq = 1.0 / (double)q;
    mc.pad2 = 0;
    // smallest generator g >= 3 of Z_q^* (ring/subring.go:181-193)
    std::vector<uint64_t> fac = unique_prime_factors(q - 1);
    uint64_t g = 3;
    for (;; g++) {
        bool ok = true;
        for (uint64_t f : fac)
            if (powmod(g, (q - 1) / f, q) == 1) { ok = false; break; }
        if (ok) break;
    }
    out.primroot = g;
    const uint64_t psi = powmod(g, (q - 1) / nth, q), psiinv = invmod(psi, q);
    (void)N;
    out.roots_fwd.assign(half, 0);
    out.roots_bwd.assign(half, 0);
    uint64_t pf = to_mont(1, q), pb = pf;
    const uint64_t psim = to_mont(psi, q), psiinvm = to_mont(psiinv, q);
    for (uint64_t j = 0; j < half; j++) {
        const uint64_t idx = bitrev(j, loghalf);
        out.roots_fwd[idx] = pf;
        out.roots_bwd[idx] = pb;
        pf = mred(pf, psim, q, mc.qinv);
        pb = mred(pb, psiinvm, q, mc.qinv);
    }
    return true;
}

std::vector<std::vector<uint64_t>> build_rescale_constants(const std::vector<uint64_t> &moduli) {
    const int n = (int)moduli.size();
    std::vector<std::vector<uint64_t>> rc(n > 1 ? n - 1 : 0);
    for (int j = n - 1; j > 0; j--) {
        rc[j - 1].resize(j);
        for (int i = 0; i < j; i++) {
            const uint64_t qi = moduli[i];
            rc[j - 1][i] = to_mont(qi - invmod(moduli[j] % qi, qi), qi);
        }
    }
    return rc;
}

ModUpHost build_modup_constants(const std::vector<uint64_t> &S, const std::vector<uint64_t> &D) {
    ModUpHost c;
    c.nsrc = (int)S.size();
    c.ndst = (int)D.size();
    c.a.resize(c.nsrc);
    c.T.assign((size_t)c.ndst * c.nsrc, 0);
    c.vt.assign((size_t)c.ndst * (c.nsrc + 1), 0);
    for (int i = 0; i < c.nsrc; i++) {
        const uint64_t si = S[i];
        uint64_t star = 1;
        for (int k = 0; k < c.nsrc; k++)
            if (k != i) star = mulmod(star, S[k] % si, si);
        c.a[i] = to_mont(invmod(star, si), si);
        for (int j = 0; j < c.ndst; j++) {
            const uint64_t dj = D[j];
            uint64_t t = 1;
            for (int k = 0; k < c.nsrc; k++)
                if (k != i) t = mulmod(t, S[k] % dj, dj);
            c.T[(size_t)j * c.nsrc + i] = to_mont(t, dj);
        }
    }
    for (int j = 0; j < c.ndst; j++) {
        const uint64_t dj = D[j];
        uint64_t smod = 1;
        for (int k = 0; k < c.nsrc; k++) smod = mulmod(smod, S[k] % dj, dj);
        const uint64_t v = dj - smod;  // -S mod dj (S mod dj != 0 for coprime bases; == dj otherwise, as the reference)
        uint64_t acc = 0;
        for (int i = 1; i <= c.nsrc; i++) {
            acc = acc + v;
            if (acc >= dj) acc -= dj;
            c.vt[(size_t)j * (c.nsrc + 1) + i] = acc;
        }
    }
    return c;
}

uint64_t half_product_mod(const std::vector<uint64_t> &S, uint64_t m) {
    // floor(P/2) = (P-1)/2 for odd P:  (P mod m - 1) * 2^-1 mod m, m odd
    uint64_t p = 1;
    for (uint64_t s : S) p = mulmod(p, s % m, m);
    const uint64_t pm1 = (p + m - 1) % m;
    return mulmod(pm1, (m + 1) / 2, m);
}

uint64_t inv_product_mont(const std::vector<uint64_t> &S, uint64_t q) {
    uint64_t p = 1;
    for (uint64_t s : S) p = mulmod(p, s % q, q);
    return to_mont(invmod(p, q), q);
}

uint64_t words_mod(const uint64_t *w, int n, uint64_t q) {
    u128 rem = 0;
    for (int i = n - 1; i >= 0; i--) rem = ((rem << 64) | w[i]) % q;
    return (uint64_t)rem;
}

}  // namespace he
