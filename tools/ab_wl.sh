# usage (GPU box): AB_WL="c4 c5" bash tools/ab_wl.sh variantA variantB ... : bench.py --workload W on each lattigo_amd/variants/libhering_<v>.so
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
for w in ${AB_WL:-c4}; do for rep in 1 2; do for v in "$@"; do
  HERING_LIB=$R/lattigo_amd/variants/libhering_$v.so python $R/bench.py --workload $w --no-cpu-baseline --no-ntt --no-b1 --steps 5 $AB_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print('$w $v', round(d['value'],1), d['verified'], {a: round(b,3) for a,b in list(k.items())[:6]})"
done; done; done
