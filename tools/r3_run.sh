# round-3 GPU check: the whole GPU suite, then the driver's bench invocation (timed)
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3_gpu_tests.log
S0=$(date +%s.%N); timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "bench rc=$? wall=$(echo "$(date +%s.%N) - $S0" | bc)"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["batch_ms"])
for k,v in d["secondary"].items():
    if isinstance(v, dict):
        print(k, {x: v.get(x) for x in ("value","ms_per_step","wall_s","error","batch_ms")}, v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("traffic"), (v.get("cpu_baseline") or {}).get("value"))
    else: print(k, v)
print(len(json.dumps(d)), "bytes")
PY
# experiments riding along: 64-column tiles for the edge GEMM; which memory-side counters rocprofv3 offers
for B in 0 64; do GNX_GEMM_BN=$B timeout -k 10 200 python bench.py --dims core --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GNX_GEMM_BN=$B core', d['ms_per_step'], d['roofline']['all_kernels_us'])"; done
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "TCC_EA0?_(RD|WR)|DRAM|MALL|TCC_REQ|TCC_HIT|TCC_MISS|HBM|FETCH_SIZE|WRITE_SIZE|TCC_BUBBLE|TCC_TAG" | sort -u | head -60) > gpurun_out/r3_counters.txt 2>&1; wc -l gpurun_out/r3_counters.txt
