// parity.cpp -- the C++ host mirror (include/hering.hpp) against the CPU oracle (oracle/lattigo_oracle.h, test infrastructure),
// written the way the reference's own tests read: ring_test.go's testNTT / testMForm / testMulPoly cases (ring/ring_test.go:433-
// 512, ring/ntt_test.go:95-121), rlwe_test.go's testGadgetProduct / testAutomorphism / testRelinearize (core/rlwe/rlwe_test.go:
// 666-1010) and the schemes' MulRelin + Rescale (schemes/bgv/bgv_test.go:560-640), on the same seeded inputs, word for word.
//
//   make -C tests/cpp && tests/cpp/parity          (needs an MI355X; without a GPU it must fail loudly -- there is no CPU path)
//   tests/cpp/parity --link-only                   exercises only what needs no device (symbol resolution, error strings)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "hering.hpp"
extern "C" {
#include "lattigo_oracle.h"
}

#include "ntt_kat.inc"  // the reference's own known-answer vectors (ring/ntt_test.go:10-89)

using hering::Ciphertext;
using hering::Poly;
using u64v = std::vector<uint64_t>;

static int g_checks = 0;
#define REQUIRE(cond)                                                              \
    do {                                                                           \
        if (!(cond)) {                                                             \
            std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
            std::exit(1);                                                          \
        }                                                                          \
        g_checks++;                                                                \
    } while (0)

// coefficients i.i.d. uniform in [0, q_i) per limb, [entries][limbs][N]
static u64v uniform(std::mt19937_64 &rng, const u64v &moduli, int N, int entries = 1) {
    u64v out((size_t)entries * moduli.size() * N);
    size_t k = 0;
    for (int e = 0; e < entries; e++)
        for (uint64_t q : moduli) {
            std::uniform_int_distribution<uint64_t> d(0, q - 1);
            for (int j = 0; j < N; j++) out[k++] = d(rng);
        }
    return out;
}
static void gen_moduli(int logN, const std::vector<int> &logq, const std::vector<int> &logp, u64v &q, u64v &p) {
    q.resize(logq.size());
    p.resize(logp.size());
    REQUIRE(lo_gen_moduli(logN + 1, logq.data(), (int)logq.size(), logp.data(), (int)logp.size(), q.data(), p.data()) == 0);
}
static Poly upload(const hering::Ring &r, const u64v &host, int batch = 1) {
    Poly p = r.NewScratch(batch);
    p.Upload(host);
    return p;
}
static Ciphertext new_ct(const hering::Ring &r, int degree, int batch = 1) {
    Ciphertext ct;
    for (int i = 0; i <= degree; i++) ct.Value.push_back(r.NewPoly(batch));
    return ct;
}
// component k of entry b of a [entries][degree+1][limbs][N] host ciphertext batch -> a [entries][limbs][N] image
static u64v component(const u64v &cts, int entries, int comps, int k, size_t words) {
    u64v out((size_t)entries * words);
    for (int b = 0; b < entries; b++) std::memcpy(&out[(size_t)b * words], &cts[((size_t)b * comps + k) * words], words * 8);
    return out;
}

static void test_ring(const hering::Context &ctx, int logN) {
    const int N = 1 << logN;
    u64v q, p;
    gen_moduli(logN, {60, 45, 55, 40}, {}, q, p);
    hering::Ring ringQ(ctx, logN, q);
    lo_ring *oQ = lo_ring_new(N, q.data(), (int)q.size());
    REQUIRE(oQ != nullptr);
    const int level = ringQ.MaxLevel();
    std::mt19937_64 rng(1000 + logN);
    const u64v x = uniform(rng, q, N), y = uniform(rng, q, N);
    u64v want(x.size()), want2(x.size());
    Poly px = upload(ringQ, x), py = upload(ringQ, y), pz = ringQ.NewPoly();
    // testNTT: NTT / INTT against the reference algorithm, and the round trip in place
    ringQ.NTT(px, pz);
    lo_ntt(oQ, level, x.data(), want.data());
    REQUIRE(pz.Download() == want);
    ringQ.INTT(pz, pz);
    REQUIRE(pz.Download() == x);
    // testMForm
    ringQ.MForm(px, pz);
    lo_unop(oQ, level, LO_MFORM, x.data(), want.data());
    REQUIRE(pz.Download() == want);
    ringQ.IMForm(pz, pz);
    REQUIRE(pz.Download() == x);
    // testMulPoly: Montgomery and Barrett products agree with the oracle's, Add / Sub / Neg too
    ringQ.MulCoeffsMontgomery(px, py, pz);
    lo_binop(oQ, level, LO_MUL_MONT, x.data(), y.data(), want.data());
    REQUIRE(pz.Download() == want);
    ringQ.MulCoeffsBarrett(px, py, pz);
    lo_binop(oQ, level, LO_MUL_BARRETT, x.data(), y.data(), want.data());
    REQUIRE(pz.Download() == want);
    ringQ.Add(px, py, pz);
    lo_binop(oQ, level, LO_ADD, x.data(), y.data(), want.data());
    REQUIRE(pz.Download() == want);
    ringQ.Sub(px, py, pz);
    lo_binop(oQ, level, LO_SUB, x.data(), y.data(), want.data());
    REQUIRE(pz.Download() == want);
    // rescale: DivRoundByLastModulusNTT at the top level
    ringQ.NTT(px, pz);
    lo_ntt(oQ, level, x.data(), want.data());
    Poly pr = ringQ.AtLevel(level - 1).NewPoly();
    ringQ.DivRoundByLastModulusNTT(pz, pr);
    lo_div_round_by_last_modulus_ntt(oQ, level, want.data(), want2.data());
    want2.resize((size_t)level * N);
    REQUIRE(pr.Download() == want2);
    // automorphism in the NTT domain
    const uint64_t galEl = 5 * 5 * 5 % (2 * (uint64_t)N);
    Poly pa = ringQ.NewPoly();
    ringQ.AutomorphismNTT(pz, galEl, pa);
    u64v index(N), wa(x.size());
    lo_automorphism_ntt_index(N, 2 * (uint64_t)N, galEl, index.data());
    lo_automorphism_ntt_with_index(oQ, level, want.data(), index.data(), wa.data());
    REQUIRE(pa.Download() == wa);
    // AutomorphismNTTIndex: the table itself and the WithIndex forms
    hering::AutomorphismIndex ix = ringQ.AutomorphismNTTIndex(galEl);
    REQUIRE(ix.Download() == index);
    Poly pa2 = ringQ.NewPoly();
    ringQ.AutomorphismNTTWithIndex(pz, ix, pa2);
    REQUIRE(pa2.Download() == wa);
    ringQ.AutomorphismNTTWithIndexThenAddLazy(pz, ix, pa2);  // pa2 += phi(pz), lazily
    lo_automorphism_ntt_with_index_then_add_lazy(oQ, level, want.data(), index.data(), wa.data());
    REQUIRE(pa2.Download() == wa);
    // the per-limb tables and the host-slice transformer (ring.NumberTheoreticTransformer, BASELINE config 1's plumbing)
    for (int i = 0; i <= level; i++) {
        const u64v rf = ringQ.RootsForward(i);
        REQUIRE(std::equal(rf.begin(), rf.end(), lo_ring_roots_fwd(oQ, i)));
        REQUIRE(ringQ.Constant(i, 0) == q[i]);
    }
    {
        u64v f(N), bk(N), wf(x.size());
        ringQ.Forward(1, &x[(size_t)N], f.data());
        lo_ntt(oQ, level, x.data(), wf.data());
        REQUIRE(std::equal(f.begin(), f.end(), wf.begin() + N));
        ringQ.Backward(1, f.data(), bk.data());
        REQUIRE(std::equal(bk.begin(), bk.end(), x.begin() + N));
    }
    // rows of Coeffs one by one, the handle's own shape, AddScalarBigint with a multi-word scalar
    {
        Poly pl2 = ringQ.NewPoly();
        for (int i = 0; i <= level; i++) pl2.UploadLimb(0, i, &x[(size_t)i * N]);
        REQUIRE(pl2.Download() == x);
        u64v row(N);
        pl2.DownloadLimb(0, level, row.data());
        REQUIRE(std::equal(row.begin(), row.end(), x.begin() + (size_t)level * N));
        const auto shape = pl2.Shape();
        REQUIRE(shape[0] == level + 1 && shape[1] == 1 && shape[2] == N);
        const u64v big = {0x123456789abcdef0ull, 0x0fedcba987654321ull, 0x1ull};
        ringQ.AddScalarBigint(px, big, pl2);
        lo_add_scalar_bigint(oQ, level, x.data(), big.data(), (int)big.size(), want2.data());
        u64v w2(want2.begin(), want2.begin() + x.size());
        REQUIRE(pl2.Download() == w2);
    }
    // a captured sequence replays to the same words
    {
        Poly g1 = ringQ.NewPoly(), g2 = ringQ.NewPoly();
        auto seq = [&] { ringQ.NTT(px, g1); ringQ.MulCoeffsMontgomery(g1, g1, g2); ringQ.INTT(g2, g2); };
        seq();
        const u64v first = g2.Download();
        g2.Zero();
        hering::Graph graph(ctx, seq);
        REQUIRE(graph.Nodes() >= 3);
        graph.Launch();
        REQUIRE(g2.Download() == first);
    }
    lo_ntt(oQ, level, x.data(), want.data());
    // AtLevel: only the first limbs are touched
    Poly pl = ringQ.NewPoly();
    ringQ.AtLevel(1).NTT(px, pl);
    u64v got = pl.Download();
    lo_ntt(oQ, level, x.data(), want.data());
    REQUIRE(std::equal(got.begin(), got.begin() + 2 * N, want.begin()));
    for (size_t i = 2 * (size_t)N; i < got.size(); i++) REQUIRE(got[i] == 0);
    lo_ring_free(oQ);
}

static void test_rlwe(const hering::Context &ctx, int logN) {
    const int N = 1 << logN, B = 2;
    u64v q, p;
    gen_moduli(logN, {55, 45, 45, 45, 45}, {55, 55}, q, p);
    const int nq = (int)q.size(), np = (int)p.size(), level = nq - 1, levelP = np - 1;
    hering::Ring ringQ(ctx, logN, q), ringP(ctx, logN, p);
    hering::Evaluator eval(ringQ, ringP);
    lo_ring *oQ = lo_ring_new(N, q.data(), nq), *oP = lo_ring_new(N, p.data(), np);
    lo_evaluator *oev = lo_evaluator_new(oQ, oP);
    REQUIRE(oev != nullptr);
    std::mt19937_64 rng(2000 + logN);
    const int beta = lo_base_rns_decomposition_vector_size(level, levelP);
    // a uniformly random key: the key switch is bit-exact whatever the key holds (decrypt-and-check lives in the Python suite)
    const u64v kq = uniform(rng, q, N, beta * 2), kp = uniform(rng, p, N, beta * 2);
    hering::EvaluationKey evk = eval.NewEvaluationKey(beta, nq, np, kq, kp);
    lo_evk oevk{};
    oevk.beta = beta; oevk.nQk = nq; oevk.nPk = np; oevk.q = kq.data(); oevk.p = kp.data(); oevk.pw2 = 0;
    for (int &v : oevk.nj) v = 1;
    const size_t words = (size_t)nq * N;

    // testGadgetProduct: Evaluator.GadgetProduct, the lazy form + ModDown, and the hoisted form agree with the oracle and each other
    const u64v cx = uniform(rng, q, N, B);
    Poly pcx = upload(ringQ, cx, B);
    Ciphertext ct = new_ct(ringQ, 1, B);
    eval.GadgetProduct(level, pcx, evk, ct);
    const u64v g0 = ct.Value[0].Download(), g1 = ct.Value[1].Download();
    for (int b = 0; b < B; b++) {
        u64v want(2 * words);
        lo_gadget_product(oev, level, &cx[(size_t)b * words], &oevk, want.data());
        REQUIRE(std::equal(want.begin(), want.begin() + words, g0.begin() + (size_t)b * words));
        REQUIRE(std::equal(want.begin() + words, want.end(), g1.begin() + (size_t)b * words));
    }
    std::array<hering::PolyQP, 2> qp{{{ringQ.NewPoly(B), ringP.NewPoly(B)}, {ringQ.NewPoly(B), ringP.NewPoly(B)}}};
    eval.GadgetProductLazy(level, pcx, evk, qp);
    Ciphertext ct2 = new_ct(ringQ, 1, B);
    eval.ModDown(level, levelP, qp, ct2);
    REQUIRE(ct2.Value[0].Download() == g0 && ct2.Value[1].Download() == g1);
    hering::Decomposition dec = eval.NewDecomposition(B);
    eval.DecomposeNTT(level, levelP, levelP + 1, pcx, true, dec);
    eval.GadgetProductHoisted(level, dec, evk, ct2);
    REQUIRE(ct2.Value[0].Download() == g0 && ct2.Value[1].Download() == g1);

    // testAutomorphism: Evaluator.Automorphism and AutomorphismHoisted
    const u64v cts = uniform(rng, q, N, B * 2);  // [B][2][limbs][N]
    Ciphertext cin;
    cin.Value = {upload(ringQ, component(cts, B, 2, 0, words), B), upload(ringQ, component(cts, B, 2, 1, words), B)};
    const uint64_t galEl = 2 * (uint64_t)N - 1;  // the conjugation
    Ciphertext rot = new_ct(ringQ, 1, B);
    eval.Automorphism(cin, galEl, evk, rot);
    for (int b = 0; b < B; b++) {
        u64v want(2 * words);
        lo_automorphism_ct(oev, level, &cts[(size_t)b * 2 * words], galEl, &oevk, want.data());
        const u64v r0 = rot.Value[0].Download(), r1 = rot.Value[1].Download();
        REQUIRE(std::equal(want.begin(), want.begin() + words, r0.begin() + (size_t)b * words));
        REQUIRE(std::equal(want.begin() + words, want.end(), r1.begin() + (size_t)b * words));
    }
    eval.DecomposeNTT(level, levelP, levelP + 1, cin.Value[1], true, dec);
    Ciphertext rot2 = new_ct(ringQ, 1, B);
    eval.AutomorphismHoisted(level, cin, dec, galEl, evk, rot2);
    REQUIRE(rot2.Value[0].Download() == rot.Value[0].Download() && rot2.Value[1].Download() == rot.Value[1].Download());

    // the giant step of lintrans.Evaluator.MultiplyByDiagMatrixBSGS (lintrans_evaluator.go:397-441) as one call against the reference's
    // own sequence of calls through this header: GadgetProductLazy, ringQP.Add, AutomorphismNTTWithIndex, then ...ThenAddLazy
    {
        const uint64_t gs = 5;  // the generator: a rotation by one slot
        const u64v addq = uniform(rng, q, N, B), addp = uniform(rng, p, N, B);
        hering::PolyQP add{upload(ringQ, addq, B), upload(ringP, addp, B)};
        std::array<hering::PolyQP, 2> fused{{{ringQ.NewPoly(B), ringP.NewPoly(B)}, {ringQ.NewPoly(B), ringP.NewPoly(B)}}};
        std::array<hering::PolyQP, 2> sep{{{ringQ.NewPoly(B), ringP.NewPoly(B)}, {ringQ.NewPoly(B), ringP.NewPoly(B)}}};
        const hering::AutomorphismIndex ixq = ringQ.AutomorphismNTTIndex(gs);
        for (int round = 0; round < 2; round++) {  // overwrite, then accumulate onto the result
            eval.LinTransGiantStep(level, pcx, evk, gs, add, round != 0, fused);
            eval.GadgetProductLazy(level, pcx, evk, qp);
            ringQ.Add(qp[0].Q, add.Q, qp[0].Q);
            ringP.Add(qp[0].P, add.P, qp[0].P);
            for (int k = 0; k < 2; k++) {
                if (round == 0) {
                    ringQ.AutomorphismNTTWithIndex(qp[k].Q, ixq, sep[k].Q);
                    ringP.AutomorphismNTTWithIndex(qp[k].P, ixq, sep[k].P);
                } else {
                    ringQ.AutomorphismNTTWithIndexThenAddLazy(qp[k].Q, ixq, sep[k].Q);
                    ringP.AutomorphismNTTWithIndexThenAddLazy(qp[k].P, ixq, sep[k].P);
                }
                REQUIRE(fused[k].Q.Download() == sep[k].Q.Download() && fused[k].P.Download() == sep[k].P.Download());
            }
        }
    }

    // BGV MulRelin (tensorStandard + Relinearize) and the degree-2 product followed by Relinearize; then Rescale
    const uint64_t T = 65537;
    const u64v cts1 = uniform(rng, q, N, 2);  // one ciphertext: [2][limbs][N]
    const u64v cts0(cts.begin(), cts.begin() + 2 * words);
    Ciphertext a, b2;
    a.Value = {upload(ringQ, component(cts0, 1, 2, 0, words)), upload(ringQ, component(cts0, 1, 2, 1, words))};
    b2.Value = {upload(ringQ, component(cts1, 1, 2, 0, words)), upload(ringQ, component(cts1, 1, 2, 1, words))};
    Ciphertext prod = new_ct(ringQ, 1);
    eval.MulRelinBGV(T, a, b2, &evk, prod);
    u64v want(2 * words);
    lo_bgv_mul_relin(oev, level, T, cts0.data(), cts1.data(), &oevk, 1, want.data());
    u64v got = prod.Value[0].Download(), got1 = prod.Value[1].Download();
    REQUIRE(std::equal(want.begin(), want.begin() + words, got.begin()) && std::equal(want.begin() + words, want.end(), got1.begin()));
    Ciphertext deg2 = new_ct(ringQ, 2), relin = new_ct(ringQ, 1);
    eval.MulRelinBGV(T, a, b2, nullptr, deg2);
    eval.Relinearize(deg2, evk, relin);
    REQUIRE(relin.Value[0].Download() == got && relin.Value[1].Download() == got1);
    Ciphertext ck = new_ct(ringQ, 1);
    eval.MulRelinCKKS(a, b2, &evk, ck);
    lo_ckks_mul_relin(oev, level, cts0.data(), cts1.data(), &oevk, 1, want.data());
    got = ck.Value[0].Download(); got1 = ck.Value[1].Download();
    REQUIRE(std::equal(want.begin(), want.begin() + words, got.begin()) && std::equal(want.begin() + words, want.end(), got1.begin()));
    Ciphertext res;
    res.Value = {ringQ.AtLevel(level - 1).NewPoly(), ringQ.AtLevel(level - 1).NewPoly()};
    eval.Rescale(1, ck, res);
    u64v wres(2 * (size_t)level * N);
    lo_rescale(oQ, level, 1, 1, want.data(), wres.data());
    got = res.Value[0].Download(); got1 = res.Value[1].Download();
    REQUIRE(std::equal(got.begin(), got.end(), wres.begin()) && std::equal(got1.begin(), got1.end(), wres.begin() + (size_t)level * N));

    // error behaviour: a Go error is an exception carrying the library's status
    bool threw = false;
    try {
        eval.Relinearize(prod, evk, relin);  // degree 1
    } catch (const std::invalid_argument &e) {
        threw = std::string(e.what()).find("cannot relinearize: ctIn.Degree() should be 2 but is 1") != std::string::npos;
    }
    REQUIRE(threw);
    threw = false;
    try {
        eval.GadgetProduct(-1, pcx, evk, ct);
    } catch (const hering::Error &e) {
        threw = e.code == HE_EINVAL;
    }
    REQUIRE(threw);
    // a level above the key's is NOT an error: levelQ = utils.Min(levelQ, gadgetCt.LevelQ()) (evaluator_gadget_product.go:18)
    eval.GadgetProduct(level + 1, pcx, evk, ct2);
    REQUIRE(ct2.Value[0].Download() == g0 && ct2.Value[1].Download() == g1);
    lo_evaluator_free(oev);
    lo_ring_free(oQ);
    lo_ring_free(oP);
}

int main(int argc, char **argv) {
    const bool link_only = argc > 1 && std::string(argv[1]) == "--link-only";
    std::printf("%s, %d HIP device(s)\n", he_version(), hering::Context::DeviceCount());
    if (link_only || hering::Context::DeviceCount() == 0) {
        // no device: the constructor must refuse loudly (HE_EDEVICE) -- there is no CPU path to fall back to
        try {
            hering::Context ctx(0);
            if (!link_only) { std::fprintf(stderr, "FAIL: a context without a device\n"); return 1; }
        } catch (const hering::Error &e) {
            std::printf("no device: %s (code %d)\n", e.what(), e.code);
            return e.code == HE_EDEVICE ? (link_only ? 0 : 3) : 1;
        }
        if (link_only) return 0;
    }
    hering::Context ctx(0);
    // ring.NewRing's parameter errors
    bool threw = false;
    try {
        hering::Ring bad(ctx, 10, {0x1fffffffffe00001ull, 0x1fffffffffe00001ull});
    } catch (const hering::Error &e) {
        threw = e.code == HE_EPARAM && std::string(e.what()).find("moduli are not distinct") != std::string::npos;
    }
    REQUIRE(threw);
    // TestNTT's known answers (ring/ntt_test.go:95-121): NTT(poly) == polyNTT, then INTT in place gives poly back
    REQUIRE(kNttKat.size() == 6);
    for (const NttKat &k : kNttKat) {
        hering::Ring r(ctx, k.logN, k.Qis);
        Poly px = r.NewPoly(), pz = r.NewPoly();
        px.Upload(k.poly);
        r.NTT(px, pz);
        REQUIRE(pz.Download() == k.polyNTT);
        r.INTT(pz, pz);
        REQUIRE(pz.Download() == k.poly);
    }
    for (int logN : {10, 13, 15}) test_ring(ctx, logN);
    for (int logN : {11, 13}) test_rlwe(ctx, logN);
    ctx.Sync();
    // the same checks through the context's submission queue, in its default mode and with deferred submission (every call of
    // the mirror is then filed and launched by the context's dispatcher thread; downloads wait for the caller's pending requests)
    const int direct = g_checks;
    ctx.SetCoalescing(16, 30);
    test_ring(ctx, 13);
    test_rlwe(ctx, 13);
    ctx.SetDeferred(8);
    test_ring(ctx, 13);
    test_rlwe(ctx, 11);
    ctx.Sync();
    ctx.SetDeferred(0);
    ctx.SetCoalescing(0, 0);
    std::printf("(%d of the checks through the submission queue)\n", g_checks - direct);
    std::printf("PASS: %d checks\n", g_checks);
    return 0;
}
