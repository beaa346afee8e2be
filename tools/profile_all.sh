#!/bin/bash
# Everything the round's measurement section cites, in one GPU call: gpurun_out/TAG/*  (headline artefacts, config-5 load, shuffled
# delivery, c3 timeline, history after load, apply timings)
TAG=${1:-round}
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1
bash tools/profile_c5.sh $TAG > gpurun_out/${TAG}_c5.log 2>&1
bash tools/profile_shuffled.sh $TAG > gpurun_out/${TAG}_shuffled.log 2>&1
bash tools/profile_workload.sh $TAG c3_map_lww > gpurun_out/${TAG}_c3.log 2>&1
timeout 200 python tools/time_history.py > gpurun_out/$TAG/history_trace.txt 2>&1
timeout 300 python tools/time_apply.py > gpurun_out/$TAG/apply_changes_timings.jsonl 2> gpurun_out/$TAG/apply_changes_timings.err
ls gpurun_out/$TAG
