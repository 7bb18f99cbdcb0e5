# final tree of round 5: GPU suite, smoke, bench lines of all four workloads (the kernels are those of profiles/r05_kernel_stats_*)
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()"
for B in 24 32; do timeout 600 python bench.py --workload c5 --batch $B --steps 3 --no-ntt --no-b1 --no-concurrent --no-cpu-baseline 2> gpurun_out/c5_b$B.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c5 batch $B', d['value'], d.get('verified'), d['ms_per_step'])"; done
