#!/bin/bash
# Kernel stats / timeline / PMC of Backend.load on the config-5 document (12 M rows): gpurun_out/TAG/c5_*
set -u
TAG=${1:-c5}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
B="python bench.py --workload c5_doc_mixed --steps 6 --warmup 2 --prewarm 0.2 --no-sublines --no-cpu-baseline --no-live-trace --detail $OUT/c5_bench_detail.json"
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/kt -o run -- $B > $OUT/c5_bench_under_trace.json 2> $OUT/c5_kt.err
python tools/rocpd_summary.py $OUT/kt/run_results.db 8 > $OUT/c5_kernel_stats.txt 2>&1
python tools/rocpd_timeline.py $OUT/kt/run_results.db -2 k_scan_term_sums > $OUT/c5_timeline.txt 2>&1
timeout -k 5 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o f --output-format csv -- $B > /dev/null 2> $OUT/c5_pf.err
timeout -k 5 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw -o w --output-format csv -- $B > /dev/null 2> $OUT/c5_pw.err
F=$(find $OUT/pf -name "*counter_collection.csv" | head -1); W=$(find $OUT/pw -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $F $W > $OUT/c5_pmc_hbm_traffic.txt 2>&1
rm -rf $OUT/kt $OUT/pf $OUT/pw
timeout -k 5 200 $B > $OUT/c5_bench.json 2> $OUT/c5_bench.err
head -40 $OUT/c5_kernel_stats.txt
