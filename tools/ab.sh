# usage (GPU box): bash tools/ab.sh variantA variantB ... : bench.py on each lattigo_amd/variants/libhering_<v>.so, interleaved twice
# (AB_ARGS: extra bench.py arguments, e.g. --no-verify for timing experiments that compute wrong results by design)
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in "$@"; do
  HERING_LIB=$R/lattigo_amd/variants/libhering_$v.so python $R/bench.py --no-cpu-baseline --no-ntt --steps 20 $AB_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print('$v', round(d['value']), d['verified'], {a: round(b,3) for a,b in k.items()})"
done; done
