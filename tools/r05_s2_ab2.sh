#!/bin/bash
# same-box A/B: old = 96fa574 (the measured library of the second session), new = k_emit gated on Counts.n_list_inc + k_list_scan_edits
# (list counts / scan / edits without per-position arrays); new:AM355_LIST_UNFUSED=1 isolates the k_emit change.
TAG=${1:-r05_s2_ab2}
mkdir -p gpurun_out/$TAG
{
echo "# headline"; bash tools/ab_libs.sh 3 old new new:AM355_LIST_UNFUSED=1
echo "# c4_text_multi"; AB_ARGS="--workload c4_text_multi" bash tools/ab_libs.sh 2 old new
echo "# c2_text_typing"; AB_ARGS="--workload c2_text_typing" bash tools/ab_libs.sh 2 old new
} > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x 2>&1 | tail -2
