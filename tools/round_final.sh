#!/bin/bash
# The measurement call of a round on its final code (one gpurun call): the GPU suite (every test, no early exit) + smoke, then the
# headline artefacts (tools/profile_round.sh: bench line + full record, kernel stats / timeline / PMC / kernel table of one replay,
# stage split, JS end-to-end), the applyChanges timings and the history timings. Copy what is to be judged into profiles/ afterwards.
#   tools/round_final.sh r06_final
TAG=${1:-round_final}
mkdir -p gpurun_out/$TAG
timeout 1000 python -m pytest tests -m gpu -q > gpurun_out/$TAG/pytest_gpu.log 2>&1
tail -4 gpurun_out/$TAG/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1
timeout 300 python tools/time_apply.py > gpurun_out/$TAG/apply_changes_timings.jsonl 2> gpurun_out/$TAG/apply_changes_timings.err
# applyChanges call after call onto a kept state (resident path): ms per call for several batch sizes, and the stream commands of one call
{
  for spec in "c4_text_single 1 40" "c4_text_single 4 20" "c4_text_single 16 8" "c4_text_single 40 8" "c4_text_single 64 8" "c4_text_single 96 6" "c4_text_single 144 5" "c4_text_single 200 5" "c2_text_typing 1 40" "c4_text_multi 2 40" "c3_map_lww 1 40"; do
    set -- $spec
    echo "== $1: $2 change(s) per call"; timeout 200 python tools/profile_apply_seq.py $1 1.0 $2 $3
    echo "   full replay per call:"; AM355_NO_RESIDENT=1 timeout 200 python tools/profile_apply_seq.py $1 1.0 $2 $3
  done
} > gpurun_out/$TAG/apply_seq_timings.txt 2>&1
( export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/prof_apply_seq -o run -- python tools/profile_apply_seq.py c4_text_single 1.0 1 40 > gpurun_out/$TAG/prof_apply_seq.log 2>&1
  python tools/rocpd_timeline.py $(find gpurun_out/$TAG/prof_apply_seq -name "*.db" | head -1) -3 k_decode > gpurun_out/$TAG/apply_seq_timeline.txt 2>&1
  rm -rf gpurun_out/$TAG/prof_apply_seq
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/prof_apply40 -o run -- python tools/profile_apply_seq.py c4_text_single 1.0 40 8 > gpurun_out/$TAG/prof_apply40.log 2>&1
  python tools/rocpd_timeline.py $(find gpurun_out/$TAG/prof_apply40 -name "*.db" | head -1) -3 k_decode > gpurun_out/$TAG/apply40_timeline.txt 2>&1
  rm -rf gpurun_out/$TAG/prof_apply40 )
{
echo "== 1 map change (8 keys) per call onto the 1 M-op text + map document (tools/profile_apply_mixed.py)"; timeout 200 python tools/profile_apply_mixed.py 1.0 1 40
echo "   the whole merge per call (AM355_NO_MAPS_ONLY=1):"; AM355_NO_MAPS_ONLY=1 timeout 200 python tools/profile_apply_mixed.py 1.0 1 40
echo "   full replay per call (AM355_NO_RESIDENT=1):"; AM355_NO_RESIDENT=1 timeout 200 python tools/profile_apply_mixed.py 1.0 1 40
echo "== 1 text change (250 ops) + 1 map change per call: list rows merged in place, then the map half"; timeout 200 python tools/profile_apply_mixed.py 1.0 1 40 both
echo "   the whole merge per call (AM355_NO_MAPS_ONLY=1):"; AM355_NO_MAPS_ONLY=1 timeout 200 python tools/profile_apply_mixed.py 1.0 1 40 both
echo "   full replay per call (AM355_NO_RESIDENT=1):"; AM355_NO_RESIDENT=1 timeout 200 python tools/profile_apply_mixed.py 1.0 1 40 both
echo "== 8 map changes per call"; timeout 200 python tools/profile_apply_mixed.py 1.0 8 20
echo "   the whole merge per call (AM355_NO_MAPS_ONLY=1):"; AM355_NO_MAPS_ONLY=1 timeout 200 python tools/profile_apply_mixed.py 1.0 8 20
} > gpurun_out/$TAG/apply_mixed_timings.txt 2>&1
timeout 200 python tools/time_history.py > gpurun_out/$TAG/history_trace.txt 2>&1
ls gpurun_out/$TAG
cat gpurun_out/$TAG/bench_line.json; echo
head -36 gpurun_out/$TAG/kernel_table.txt
cat gpurun_out/$TAG/apply_changes_timings.jsonl
cat gpurun_out/$TAG/apply_seq_timings.txt | cut -c1-40,200-
