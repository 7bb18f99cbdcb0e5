# final tree of round 5: GPU suite, smoke, bench lines of all four workloads (the kernels are those of profiles/r05_kernel_stats_*)
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()"
T0=$(date +%s); timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default bench.py: rc=$? $(( $(date +%s) - T0 )) s"
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo rc=$?
rm -f gpurun_out/bench_configs.jsonl gpurun_out/bench_configs.err
for w in c2 c4 c5; do S=10; [ $w = c5 ] && S=5; timeout 900 python bench.py --workload $w --steps $S --no-ntt >> gpurun_out/bench_configs.jsonl 2>> gpurun_out/bench_configs.err; done
python - <<'P'
import json
d=json.load(open("gpurun_out/bench_final.json"))
print("c3", d["value"], d["verified"], d.get("accounting_problems"))
cb=d["concurrent_b1"]; print({k:cb[k] for k in cb if k in ("K","coalesced","coalesced_sync_each","deferred","deferred_sync_each","verified")})
for r in cb.get("compiled_host",[]): print({k:r.get(k) for k in ("K","sync_each","deferred_depth","mean_batch","ops_per_s","verified_callers","error")})
for l in open("gpurun_out/bench_configs.jsonl"):
    d=json.loads(l); print(d["config"]["workload"][:30], d["value"], d.get("verified"), d.get("accounting_problems"))
    cb=d.get("concurrent_b1") or {}
    for r in cb.get("compiled_host",[]): print("   ", {k:r.get(k) for k in ("K","sync_each","coalescing","max_batch","deferred_depth","mean_batch","ops_per_s","verified_callers","error")})
    if "deferred" in cb: print("   ", {k:cb[k] for k in cb if k in ("K","coalesced","deferred","deferred_sync_each","lone_caller","uncoalesced_K4","verified","error")})
P
