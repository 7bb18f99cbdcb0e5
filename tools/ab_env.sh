# usage (GPU box): bash tools/ab_env.sh "ENV1=a ENV2=b" "ENV1=c" ... : bench.py under each environment setting, interleaved twice
# ("-" = no extra environment).  AB_ARGS: extra bench.py arguments.
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
for rep in 1 2; do for v in "$@"; do
  e="$v"; [ "$v" = "-" ] && e=""
  env $e python $R/bench.py --no-cpu-baseline --no-ntt --steps 20 $AB_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print('[$v]', round(d['value']), d['verified'], {a: round(b,3) for a,b in k.items()})"
done; done
