# Round artefacts on the GPU box (gpurun_out/ is merged back; tools/collect_profiles.py copies the summaries into profiles/):
#   bash tools/round_artifacts.sh          -> bench line, rocprofv3 kernel stats of the same command, PMC traffic + SQ passes
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE $R/gpurun_out/prof_final $R/gpurun_out/pmc_sq1 $R/gpurun_out/pmc_sq2
BARGS="--no-cpu-baseline --no-verify --no-ntt"
for C in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -o bench -- python $R/bench.py --steps 5 --warmup 1 $BARGS > $R/gpurun_out/pmc_$C.log 2>&1; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o bench -- python $R/bench.py --steps 20 --warmup 3 $BARGS > $R/gpurun_out/bench_final_prof.log 2>&1
bash $R/tools/prof_pass.sh sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" > /dev/null
bash $R/tools/prof_pass.sh sq2 "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" > /dev/null
cd $R; timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 400 gpurun_out/bench_final.json
timeout 300 python tools/bench_configs.py > gpurun_out/bench_configs.jsonl 2> gpurun_out/bench_configs.err
for w in c4 c5; do timeout 600 python bench.py --workload $w --steps 5 --no-ntt >> gpurun_out/bench_configs.jsonl 2>> gpurun_out/bench_configs.err; done
timeout 200 python tools/ntt_prof.py >> gpurun_out/bench_configs.jsonl 2>> gpurun_out/bench_configs.err
