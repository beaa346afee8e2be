#!/bin/bash
# same-box A/B of the map order: prev = constant bytes skipped, trigger passes; new = + the values of a key ranked by k_map_group_rank instead of
# three radix passes over the trigger ids; new:AM355_MAP_TRIGGER_PASSES=1 = the new library with the trigger passes; then the GPU suite
TAG=${1:-r05_s2_ab5}
mkdir -p gpurun_out/$TAG
{
echo "# c3_map_lww"; AB_ARGS="--workload c3_map_lww" bash tools/ab_libs.sh 3 prev new new:AM355_MAP_TRIGGER_PASSES=1
} > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2
