#!/bin/bash
# same-box A/B: k_map_group_rank with / without the quick test on the sorted first-eight-bytes keys; then the GPU suite on the final library
TAG=${1:-r05_s2_ab6}
mkdir -p gpurun_out/$TAG
{ echo "# c3_map_lww"; AB_ARGS="--workload c3_map_lww" bash tools/ab_libs.sh 3 prev new; } > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2
