#!/bin/bash
# Round 5, sixth GPU call: suite; the default bench line (with its live traced child) timed; c3 / c5 lines
TAG=${1:-r05f}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
S=$(date +%s)
timeout -k 5 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
echo "default bench.py: $(( $(date +%s) - S )) s, rc $?"
tail -3 $OUT/bench.err
python - <<PY
import json
p=json.loads(open("$OUT/bench_line.json").read().strip().splitlines()[-1])
print("value %.3f G ops/s ms %.4f t_device %.4f frac %.4f live %s trace_s %s" % (p["value"]/1e9,p["ms_per_step"],p["t_device_ms"],p["roofline"]["frac"],p["roofline"].get("kernels_live"),p["roofline"].get("kernels_trace_seconds")))
print("traffic", p["roofline"].get("traffic"), "cpu_baseline", {k:(v if not isinstance(v,dict) else '...') for k,v in p.get("cpu_baseline",{}).items()})
for w in p.get("workloads",[]): print("  %-60s ms %.3f dev %.3f" % (w["workload"][:60], w["ms_per_step"], w["t_device_ms"]))
print("sharding_model", [ (r["n_gpus"], round(r["projected_speedup"],3)) for r in p.get("sharding_model",{}).get("projected",[])])
print("apply", [(r["batch_changes"], round(r["ms"],3)) for r in p.get("apply_changes",{}).get("batches",[])] if isinstance(p.get("apply_changes"),dict) else None)
for k in (p["roofline"].get("kernels") or [])[:8]: print("   ", k["kernel"], k["avg_us"], k.get("pmc_traffic_bytes"))
PY
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $OUT/kt3 -o run -- python bench.py --workload c3_map_lww --steps 10 --warmup 3 --no-sublines --no-cpu-baseline --no-live-trace > $OUT/c3_under_trace.json 2> $OUT/kt3.err
python tools/rocpd_timeline.py $OUT/kt3/run_results.db -2 > $OUT/c3_timeline.txt 2>&1
rm -rf $OUT/kt3
head -22 $OUT/c3_timeline.txt
