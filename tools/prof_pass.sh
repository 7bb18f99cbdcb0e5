# usage (on the GPU box): bash tools/prof_pass.sh TAG "COUNTER COUNTER ..." [bench args]   -> gpurun_out/pmc_TAG/
# one rocprofv3 counter pass of bench.py (kernel trace + the given PMC counters, csv)
R=$GRAFT_REPO_ROOT; TAG=$1; CTRS=$2; shift 2
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$TAG
timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-ntt --no-concurrent --no-other-configs "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
ls $R/gpurun_out/pmc_$TAG | head -3
