#!/bin/bash
# the decoder's split bound at 1536 changes (default now) against 512 (rounds 3-5): c2_text_typing (1001 changes); then the GPU suite
mkdir -p gpurun_out/r05_s2_ab8
{ echo "# c2_text_typing"; AB_ARGS="--workload c2_text_typing" bash tools/ab_libs.sh 3 new new:AM355_DECODE_SPLIT_MAX=512; } > gpurun_out/r05_s2_ab8/ab.txt 2>&1
cat gpurun_out/r05_s2_ab8/ab.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2
