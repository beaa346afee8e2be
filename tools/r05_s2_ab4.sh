#!/bin/bash
# same-box A/B: prev = key stats from sixteen byte loads per emission, new = two 8-byte loads; then the GPU suite on the final library
TAG=${1:-r05_s2_ab4}
mkdir -p gpurun_out/$TAG
{
echo "# c3_map_lww"; AB_ARGS="--workload c3_map_lww" bash tools/ab_libs.sh 3 prev new new:AM355_MAP_ALL_PASSES=1
} > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2
