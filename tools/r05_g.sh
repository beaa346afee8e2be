#!/bin/bash
# Round 5: Backend.load device time with the single-pass scan on / off; kernel stats of the load
TAG=${1:-r05g}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
B5="python bench.py --workload c5_doc_mixed --steps 6 --warmup 2 --prewarm 0.2 --no-sublines --no-cpu-baseline --no-live-trace"
for rep in 1 2; do for mode in 1 0; do
  AM355_SCAN_LOOKBACK=$mode timeout 300 $B5 2>/dev/null | python -c "import json,sys;p=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('lookback=$mode c5 ms_per_step %.2f t_device_ms %.3f'%(p['ms_per_step'],p['t_device_ms']), p['phases_ms'])"
done; done
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/kt -o run -- $B5 > $OUT/c5_bench_under_trace.json 2> $OUT/c5_kt.err
python tools/rocpd_summary.py $OUT/kt/run_results.db 8 > $OUT/c5_kernel_stats.txt 2>&1
rm -rf $OUT/kt
head -30 $OUT/c5_kernel_stats.txt
