#!/bin/bash
# Round 5, second GPU call: suite, Backend.load (one call) with laps, full bench line, kernel trace of the headline replay -> gpurun_out/TAG
TAG=${1:-r05b}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
B5="python bench.py --workload c5_doc_mixed --steps 6 --warmup 2 --prewarm 0.2 --no-sublines --no-cpu-baseline"
AM355_TRACE=1 timeout 300 $B5 > $OUT/c5_trace.json 2> $OUT/c5_trace.err
timeout 300 $B5 > $OUT/c5_bench.json 2> $OUT/c5_bench.err
python - <<PY
import json
p=json.loads(open("$OUT/c5_bench.json").read().strip().splitlines()[-1])
print("c5 ms_per_step %.2f t_device_ms %s value %.0f M rows/s" % (p["ms_per_step"], p.get("t_device_ms"), p["value"]/1e6))
PY
grep "load_document\|replay_document" $OUT/c5_trace.err | tail -16
timeout -k 5 400 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
head -c 400 $OUT/bench_line.json; echo
B="python bench.py --steps 10 --warmup 3 --no-sublines --no-cpu-baseline"
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $OUT/kt -o run -- $B > $OUT/bench_under_trace.json 2> $OUT/kt.err
python tools/rocpd_summary.py $OUT/kt/run_results.db 26 > $OUT/kernel_stats.txt 2>&1
python tools/rocpd_timeline.py $OUT/kt/run_results.db -2 > $OUT/timeline.txt 2>&1
rm -rf $OUT/kt
head -45 $OUT/timeline.txt
