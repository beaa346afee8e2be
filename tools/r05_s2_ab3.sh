#!/bin/bash
# same-box A/B of the map order: constant key bytes / a constant key length not sorted by (MapKeyStats) against every pass (AM355_MAP_ALL_PASSES=1)
TAG=${1:-r05_s2_ab3}
mkdir -p gpurun_out/$TAG
{
echo "# c3_map_lww"; AB_ARGS="--workload c3_map_lww" bash tools/ab_libs.sh 3 new new:AM355_MAP_ALL_PASSES=1
echo "# headline"; bash tools/ab_libs.sh 2 new new:AM355_MAP_ALL_PASSES=1
} > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x 2>&1 | tail -2
