#!/bin/bash
# same-box A/B of am355_apply_changes between library variants in _ab/ (built like tools/ab_libs.sh's): tools/ab_apply.sh <rounds> <variant>...
# (variant = lib[:ENV=VAL]); prints tools/time_apply.py's lines per variant
rounds=$1; shift
L=automerge_classic_amd/csrc/libam355.so
cp $L /tmp/lib_orig.so
for r in $(seq $rounds); do for v in "$@"; do
  lib=${v%%:*}; envs=""; [ "$v" != "$lib" ] && envs=${v#*:}
  cp _ab/lib_$lib.so $L
  echo "# $v (round $r)"
  env $envs timeout -k 5 200 python tools/time_apply.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  %-16s %5d onto %5d  %.3f ms' % (d['workload'], d['batch_changes'], d['doc_changes'], d['ms']))"
done; done
cp /tmp/lib_orig.so $L
