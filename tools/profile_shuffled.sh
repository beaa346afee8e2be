#!/bin/bash
# Timeline + kernel stats of the headline log in shuffled delivery order (general path): gpurun_out/TAG/shuffled_*
set -u
TAG=${1:-shuf}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/kts -o run -- python tools/trace_shuffled.py > $OUT/shuffled_trace_stdout.txt 2> $OUT/shuffled_trace.err
python tools/rocpd_timeline.py $OUT/kts/run_results.db -1 > $OUT/shuffled_timeline.txt 2>&1
python tools/rocpd_summary.py $OUT/kts/run_results.db 4 > $OUT/shuffled_kernel_stats.txt 2>&1
rm -rf $OUT/kts
AM355_HOST_SCHEDULE=1 timeout 120 python tools/trace_shuffled.py > $OUT/shuffled_host_schedule.txt 2>&1
timeout 120 python tools/trace_shuffled.py > $OUT/shuffled_device_schedule.txt 2>&1
tail -2 $OUT/shuffled_host_schedule.txt $OUT/shuffled_device_schedule.txt
