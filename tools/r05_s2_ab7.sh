#!/bin/bash
# A/B of the decoder's column-group split on batches of thousands of changes (default: four wavefronts per change up to 512 changes)
mkdir -p gpurun_out/r05_s2_ab7
{ echo "# headline"; bash tools/ab_libs.sh 2 new new:AM355_DECODE_SPLIT_MAX=8192
  echo "# c4_text_multi"; AB_ARGS="--workload c4_text_multi" bash tools/ab_libs.sh 1 new new:AM355_DECODE_SPLIT_MAX=8192
  echo "# c2_text_typing"; AB_ARGS="--workload c2_text_typing" bash tools/ab_libs.sh 1 new new:AM355_DECODE_SPLIT_MAX=8192; } > gpurun_out/r05_s2_ab7/ab.txt 2>&1
cat gpurun_out/r05_s2_ab7/ab.txt
