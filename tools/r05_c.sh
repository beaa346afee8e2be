#!/bin/bash
# Round 5, third GPU call: suite; headline with the speculative decode launch on / off (bench lines + timelines); Backend.load laps
TAG=${1:-r05c}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
B="python bench.py --steps 40 --warmup 10 --no-sublines --no-cpu-baseline"
for mode in 1 0 1 0; do
  AM355_SPEC_DECODE=$mode timeout 200 $B 2>/dev/null | python -c "import json,sys;p=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('spec=$mode value %.3f G ops/s, ms %.4f, t_device_ms %.4f'%(p['value']/1e9,p['ms_per_step'],p['t_device_ms']))"
done
B2="python bench.py --steps 10 --warmup 3 --no-sublines --no-cpu-baseline"
for mode in 1 0; do
  AM355_SPEC_DECODE=$mode timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $OUT/kt$mode -o run -- $B2 > $OUT/bench_under_trace_spec$mode.json 2> $OUT/kt$mode.err
  python tools/rocpd_timeline.py $OUT/kt$mode/run_results.db -2 > $OUT/timeline_spec$mode.txt 2>&1
  rm -rf $OUT/kt$mode
done
cat $OUT/timeline_spec1.txt | head -48
timeout 100 python tools/trace_run.py torch > $OUT/trace_run.txt 2>&1
B5="python bench.py --workload c5_doc_mixed --steps 8 --warmup 2 --prewarm 0.2 --no-sublines --no-cpu-baseline"
AM355_TRACE=1 timeout 300 $B5 > $OUT/c5_trace.json 2> $OUT/c5_trace.err
timeout 300 $B5 2> $OUT/c5_bench.err | python -c "import json,sys;p=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c5 ms_per_step %.2f t_device_ms %.3f'%(p['ms_per_step'],p['t_device_ms']))"
grep "load_document" $OUT/c5_trace.err | tail -9
