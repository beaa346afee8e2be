#!/bin/bash
# One GPU call that produces the round's measurement artefacts under gpurun_out/$1/ :
#   bench line (full), rocprofv3 kernel stats + timeline of one replay, PMC passes (FETCH_SIZE, WRITE_SIZE), kernel table, stage split,
#   JS end-to-end. Copy what is to be judged into profiles/ afterwards.
set -u
TAG=${1:-round}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
B="python bench.py --steps 10 --warmup 3 --no-sublines --no-cpu-baseline --no-live-trace --detail $OUT/bench_under_trace.json"
timeout -k 5 400 python bench.py --detail $OUT/bench_detail.json > $OUT/bench_line.json 2> $OUT/bench.err
# (every profiler pass under a hard limit: a pass that stalls must not eat the GPU budget; if the first one stalls, the passes are
# repeated with AM355_STAGE_SYNC=1 = am355_load_changes waits for its copies, and the note is written next to the results)
timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $OUT/kt -o run -- $B > /dev/null 2> $OUT/kt.err
if [ $? -ge 124 ]; then
  echo "kernel-trace pass stalled without AM355_STAGE_SYNC; profiler passes run with AM355_STAGE_SYNC=1" > $OUT/profiler_note.txt
  export AM355_STAGE_SYNC=1
  rm -rf $OUT/kt
  timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $OUT/kt -o run -- $B > /dev/null 2> $OUT/kt.err || { echo "kernel-trace pass stalled again" >> $OUT/profiler_note.txt; exit 1; }
fi
python tools/rocpd_summary.py $OUT/kt/run_results.db 26 > $OUT/kernel_stats.txt 2>&1
python tools/rocpd_timeline.py $OUT/kt/run_results.db -2 > $OUT/timeline.txt 2>&1
timeout -k 5 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o f --output-format csv -- $B > /dev/null 2> $OUT/pf.err
timeout -k 5 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw -o w --output-format csv -- $B > /dev/null 2> $OUT/pw.err
F=$(find $OUT/pf -name "*counter_collection.csv" | head -1); W=$(find $OUT/pw -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $F $W > $OUT/pmc_hbm_traffic.txt 2>&1
python tools/kernel_table.py --db $OUT/kt/run_results.db --fetch $F --write $W --line $OUT/bench_under_trace.json --replays 26 \
   --out $OUT/kernel_table.json --source "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $B" > $OUT/kernel_table.txt 2>&1
rm -rf $OUT/kt $OUT/pf $OUT/pw
unset AM355_STAGE_SYNC
timeout -k 5 150 python tools/time_stages.py > $OUT/stages.json 2> $OUT/stages.err
python -c "from automerge_classic_amd import loggen; loggen.config('c4_text_single', 1.0, False).save('/tmp/c4.bin')"
timeout -k 5 100 node automerge_classic_amd/js/bench_e2e.js /tmp/c4.bin 9 > $OUT/e2e.json 2> $OUT/e2e.err
head -c 300 $OUT/bench_line.json; echo; head -30 $OUT/kernel_table.txt
