#!/bin/bash
# Kernel stats + timeline of one replay of a bench workload: gpurun_out/TAG/WORKLOAD_*   (tools/profile_workload.sh TAG WORKLOAD [ANCHOR KERNEL])
set -u
TAG=${1:-w}; W=${2:-c3_map_lww}; ANCHOR=${3:-k_parse_changes}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
B="python bench.py --workload $W --steps 20 --warmup 5 --prewarm 0.2 --no-sublines --no-cpu-baseline --no-live-trace"
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $OUT/kt -o run -- $B > $OUT/${W}_bench_under_trace.json 2> $OUT/${W}_kt.err
python tools/rocpd_summary.py $OUT/kt/run_results.db 8 > $OUT/${W}_kernel_stats.txt 2>&1
python tools/rocpd_timeline.py $OUT/kt/run_results.db -2 $ANCHOR > $OUT/${W}_timeline.txt 2>&1
rm -rf $OUT/kt
timeout -k 5 200 $B > $OUT/${W}_bench.json 2> $OUT/${W}_bench.err
head -100 $OUT/${W}_timeline.txt
