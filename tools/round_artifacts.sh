# Round artefacts on the GPU box (gpurun_out/ is merged back; tools/collect_profiles.py copies the summaries into profiles/):
#   bash tools/round_artifacts.sh          -> bench line, rocprofv3 kernel stats of the same command, PMC traffic (c2, c3, c4, c5) + SQ passes
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_* $R/gpurun_out/prof_final
BARGS="--no-cpu-baseline --no-verify --no-ntt --no-b1 --no-concurrent --no-other-configs"
# PMC traffic: separate FETCH_SIZE / WRITE_SIZE passes (kernel trace only, as the guide prescribes); steps of bench.py's step() per
# run = warmup + steps (no HIP-event leg under the profiler)
for W in c2 c3 c4 c5; do
  S=5; [ $W = c5 ] && S=2
  for C in FETCH_SIZE WRITE_SIZE; do timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${C}_$W -o bench -- python $R/bench.py --workload $W --steps $S --warmup 1 --no-kernel-timing $BARGS > $R/gpurun_out/pmc_${C}_$W.log 2>&1; done
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o bench -- python $R/bench.py --steps 20 --warmup 3 $BARGS > $R/gpurun_out/bench_final_prof.log 2>&1
bash $R/tools/prof_pass.sh sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --no-b1 > /dev/null
bash $R/tools/prof_pass.sh sq2 "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" --no-b1 > /dev/null
# the same first SQ pass and a kernel-trace summary for the secondary workloads (round 5)
for W in c2 c4 c5; do
  S=3; [ $W = c5 ] && S=1
  rm -rf $R/gpurun_out/pmc_sq1_$W $R/gpurun_out/stats_$W
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq1_$W -o b -- python $R/bench.py --workload $W --steps $S --warmup 1 $BARGS --no-kernel-timing > $R/gpurun_out/pmc_sq1_$W.log 2>&1
  find $R/gpurun_out/pmc_sq1_$W -name '*kernel_trace.csv' -delete
  S=10; [ $W = c5 ] && S=3
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$W -o bench -- python $R/bench.py --workload $W --steps $S --warmup 2 $BARGS --no-kernel-timing > $R/gpurun_out/stats_$W.log 2>&1
  find $R/gpurun_out/stats_$W -name '*kernel_trace.csv' -delete
done
cd $R; timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 400 gpurun_out/bench_final.json
rm -f gpurun_out/bench_configs.jsonl gpurun_out/bench_configs.err
for w in c2 c4 c5; do S=10; [ $w = c5 ] && S=5; timeout 900 python bench.py --workload $w --steps $S --no-ntt --no-other-configs >> gpurun_out/bench_configs.jsonl 2>> gpurun_out/bench_configs.err; done
timeout 300 python tools/bench_configs.py >> gpurun_out/bench_configs.jsonl 2>> gpurun_out/bench_configs.err
timeout 200 python tools/ntt_prof.py >> gpurun_out/bench_configs.jsonl 2>> gpurun_out/bench_configs.err
# machine probes behind DESIGN.md's ceilings: instruction issue rates and what HBM gives a streaming kernel by read : write mix
bash $R/tools/r05_ntt_pmc.sh > /dev/null 2>&1
for p in instr_probe hbm_probe; do hipcc --offload-arch=gfx950 -O3 $R/tools/$p.hip -o /tmp/$p 2>/dev/null && /tmp/$p > $R/gpurun_out/$p.txt 2>&1; done
