#!/bin/bash
# re-measure the headline part of the round's profile on the final code (tools/profile_round.sh) + the fat-parse test
TAG=${1:-r05_final}
timeout 300 python -m pytest tests -m gpu -x -q -k "fat_changes or golden_reference or full_size_headline" 2>&1 | tail -2
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1
head -c 400 gpurun_out/$TAG/bench_line.json; echo
head -16 gpurun_out/$TAG/kernel_table.txt
head -44 gpurun_out/$TAG/timeline.txt
