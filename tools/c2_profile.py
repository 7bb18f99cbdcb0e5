import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import lattigo_amd as la
from bench import uniform
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from bench_configs import C2_Q, C2_P
ctx = la.Context(0); rng = np.random.default_rng(0)
N = 1 << 14
rq, rp = la.Ring(ctx, N, C2_Q), la.Ring(ctx, N, C2_P)
ev = la.Evaluator(rq, rp)
L, B = len(C2_Q), 128
rlk = ev.NewEvaluationKey(uniform(rng, C2_Q, N, (L, 2)), uniform(rng, C2_P, N, (L, 2)))
a = [la.Poly(rq, L, B).upload(uniform(rng, C2_Q, N, (B,))) for _ in range(2)]
b = [la.Poly(rq, L, B).upload(uniform(rng, C2_Q, N, (B,))) for _ in range(2)]
o3 = [la.Poly(rq, L, B) for _ in range(3)]
r3 = [la.Poly(rq, L - 1, B) for _ in range(3)]
for name, fn in (("mul", lambda: ev.CKKSMulRelin(L - 1, a, b, None, o3)), ("rescale3", lambda: ev.Rescale(L - 1, 1, o3, r3)),
                 ("mulrelin", lambda: ev.CKKSMulRelin(L - 1, a, b, rlk, o3[:2])), ("rescale2", lambda: ev.Rescale(L - 1, 1, o3[:2], r3[:2]))):
    for _ in range(2): fn()
    ctx.prof_begin()
    for _ in range(5): fn()
    pr = ctx.prof_end()
    print(name, round(sum(v[1] for v in pr.values()) / 5, 3), {k: (v[0] // 5, round(v[1] / 5, 3)) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][1])})
