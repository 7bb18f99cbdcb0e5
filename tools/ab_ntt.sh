R=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in base6 rskip; do
  HERING_LIB=$R/lattigo_amd/variants/libhering_$v.so python $R/bench.py --no-cpu-baseline --no-b1 --no-concurrent --no-other-configs --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); n=d['ntt']; print('$v', round(d['value']), {k: (round(v['ntt']['limb_ntt_per_s']/1e6,3), round(v['intt']['limb_ntt_per_s']/1e6,3)) for k,v in n.items()})"
done; done
