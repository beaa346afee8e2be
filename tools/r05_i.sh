#!/bin/bash
# same-box A/B of the replay: round-4 library | round-4 + the two-wavefront SHA kernel | current (tools/ab_libs.sh; libraries in _ab/)
TAG=${1:-r05i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
AB_ARGS="--no-live-trace" bash tools/ab_libs.sh 3 r04 new 2>&1 | tee $OUT/ab_headline.txt
AB_ARGS="--no-live-trace --workload c4_text_multi" bash tools/ab_libs.sh 2 r04 new 2>&1 | tee $OUT/ab_c4multi.txt
