#!/bin/bash
# Round 5, Backend.load of the config-5 document: GPU suite, then the host side of the load with the chunked inflate on / off
# (AM355_TRACE laps of one load each, then bench lines)  -> gpurun_out/TAG/*
TAG=${1:-r05a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
nproc > $OUT/host.txt; lscpu | grep -i "model name\|^CPU(s)\|Thread\|Socket" >> $OUT/host.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
B="python bench.py --workload c5_doc_mixed --steps 6 --warmup 2 --prewarm 0.2 --no-sublines --no-cpu-baseline"
for mode in 1 0; do
  AM355_PINFLATE=$mode AM355_TRACE=1 timeout 300 $B > $OUT/c5_trace_pinflate$mode.json 2> $OUT/c5_trace_pinflate$mode.err
  AM355_PINFLATE=$mode timeout 300 $B > $OUT/c5_bench_pinflate$mode.json 2> $OUT/c5_bench_pinflate$mode.err
  python - <<PY
import json
p=json.loads(open("$OUT/c5_bench_pinflate$mode.json").read().strip().splitlines()[-1])
print("pinflate=$mode ms_per_step %.2f t_device_ms %s value %.0f M rows/s" % (p["ms_per_step"], p.get("t_device_ms"), p["value"]/1e6))
PY
done
grep "load_document" $OUT/c5_trace_pinflate1.err | tail -8
grep "load_document" $OUT/c5_trace_pinflate0.err | tail -8
