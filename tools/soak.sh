#!/bin/bash
# Soak loop of the full GPU suite: N plain runs and C runs with AM355_CANARY=1 (red zones behind every device
# carve-out, csrc/am355_canary.h), every run with its COMPLETE log kept; a failed run is repeated once with AMD_LOG_LEVEL=1.
#   tools/soak.sh TAG N_PLAIN N_CANARY      -> gpurun_out/TAG/soak_*.log, gpurun_out/TAG/soak_summary.txt
set -u
TAG=${1:-soak}; N=${2:-6}; C=${3:-2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/soak_summary.txt
run() {  # name, env...
  local name=$1; shift
  local t0=$(date +%s)
  env "$@" timeout -k 10 600 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/soak_$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc $(( $(date +%s) - t0 ))s  $(tail -1 $OUT/soak_$name.log)" >> $OUT/soak_summary.txt
  if [ $rc -ne 0 ]; then
    env "$@" AMD_LOG_LEVEL=1 timeout -k 10 600 python -X faulthandler -m pytest tests -m gpu -x -v -p no:cacheprovider > $OUT/soak_${name}_rerun_loglevel1.log 2>&1
    echo "$name rerun rc=$? (AMD_LOG_LEVEL=1)" >> $OUT/soak_summary.txt
  fi
}
for i in $(seq 1 $C); do run canary_$i AM355_CANARY=1; done
for i in $(seq 1 $N); do run plain_$i AM355_SOAK=1; done
echo "code: $(cat .git_head 2>/dev/null)  lib sha256: $(sha256sum automerge_classic_amd/csrc/libam355.so | cut -c1-16)" >> $OUT/soak_summary.txt
cat $OUT/soak_summary.txt
