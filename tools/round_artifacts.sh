R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE $R/gpurun_out/prof_final
for C in FETCH_SIZE WRITE_SIZE; do timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$C.log 2>&1; done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_final_prof.log 2>&1
cd $R; timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_final.json
timeout 300 python tools/bootstrap_c5_shape.py 8 > gpurun_out/bootstrap_c5_shape.jsonl 2> gpurun_out/bootstrap_c5_shape.err; timeout 200 python tools/bootstrap_c5_shape.py 1 >> gpurun_out/bootstrap_c5_shape.jsonl 2>> gpurun_out/bootstrap_c5_shape.err
bash $R/tools/pmc_sq.sh > $R/gpurun_out/pmc_sq.log 2>&1
