import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, json
import lattigo_amd as la
from bench import uniform
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from bench_configs import C4_Q, C4_P
ctx = la.Context(0); rng = np.random.default_rng(0)
N = 1 << 16
rq, rp = la.Ring(ctx, N, C4_Q), la.Ring(ctx, N, C4_P)
ev = la.Evaluator(rq, rp)
L, B, beta = len(C4_Q), int(sys.argv[1]) if len(sys.argv) > 1 else 16, 5
gk = ev.NewEvaluationKey(uniform(rng, C4_Q, N, (beta, 2)), uniform(rng, C4_P, N, (beta, 2)))
ct = [la.Poly(rq, L, B).upload(uniform(rng, C4_Q, N, (B,))) for _ in range(2)]
o2 = [la.Poly(rq, L, B) for _ in range(2)]
gal = 5
for _ in range(2): ev.Automorphism(L - 1, ct, gal, gk, o2)
ctx.prof_begin()
for _ in range(5): ev.Automorphism(L - 1, ct, gal, gk, o2)
pr = ctx.prof_end()
print({k: (v[0] // 5, round(v[1] / 5, 3)) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][1])})
