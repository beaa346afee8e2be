#!/bin/bash
# Round 5, fourth GPU call: suite; headline A/B (speculative decode x helper-thread enqueue); c3 line + timeline; headline timeline
TAG=${1:-r05d}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
B="python bench.py --steps 40 --warmup 10 --no-sublines --no-cpu-baseline --no-live-trace"
for rep in 1 2; do
for cfg in "1 thread" "0 thread" "1 main" "0 main"; do
  set -- $cfg
  AM355_SPEC_DECODE=$1 AM355_HASH_ENQUEUE=$2 timeout 200 $B 2>/dev/null | python -c "import json,sys;p=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('spec=$1 enqueue=$2 value %.3f G ops/s, ms %.4f, t_device_ms %.4f'%(p['value']/1e9,p['ms_per_step'],p['t_device_ms']))"
done; done
for wl in c3_map_lww c4_text_multi; do
  timeout 200 python bench.py --workload $wl --steps 30 --warmup 5 --no-sublines --no-cpu-baseline --no-live-trace 2>/dev/null | python -c "import json,sys;p=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$wl value %.3f G ops/s, ms %.4f, t_device_ms %.4f'%(p['value']/1e9,p['ms_per_step'],p['t_device_ms']))"
done
B2="python bench.py --steps 10 --warmup 3 --no-sublines --no-cpu-baseline --no-live-trace"
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $OUT/kt -o run -- $B2 > $OUT/bench_under_trace.json 2> $OUT/kt.err
python tools/rocpd_timeline.py $OUT/kt/run_results.db -2 > $OUT/timeline.txt 2>&1
rm -rf $OUT/kt
head -48 $OUT/timeline.txt
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $OUT/kt3 -o run -- python bench.py --workload c3_map_lww --steps 10 --warmup 3 --no-sublines --no-cpu-baseline --no-live-trace > $OUT/c3_under_trace.json 2> $OUT/kt3.err
python tools/rocpd_timeline.py $OUT/kt3/run_results.db -2 > $OUT/c3_timeline.txt 2>&1
rm -rf $OUT/kt3
head -70 $OUT/c3_timeline.txt
timeout 100 python tools/trace_run.py torch > $OUT/trace_run.txt 2>&1
tail -22 $OUT/trace_run.txt
