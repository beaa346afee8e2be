#!/bin/bash
# same-box A/B of library variants: [AB_ARGS="--workload c3_map_lww"] tools/ab_libs.sh <rounds> <variant>...   (variant = lib[:ENV=VAL])
# (build each variant to _ab/lib_<name>.so: the directory is git-ignored but travels with the gpurun snapshot)
# Examples (what rounds 5's one-off scripts did): tools/ab_libs.sh 3 old new          AB_ARGS="--workload c3_map_lww" tools/ab_libs.sh 3 prev new new:AM355_MAP_ALL_PASSES=1
rounds=$1; shift
L=automerge_classic_amd/csrc/libam355.so
cp $L /tmp/lib_orig.so
for r in $(seq $rounds); do for v in "$@"; do
  lib=${v%%:*}; envs=""; [ "$v" != "$lib" ] && envs=${v#*:}
  cp _ab/lib_$lib.so $L
  env $envs timeout -k 5 100 python bench.py --steps 60 --warmup 10 --no-sublines --no-cpu-baseline --detail /tmp/ab.json $AB_ARGS > /dev/null 2>&1
  python - "$v" <<PY
import json,sys
p=json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print("%-28s value %.0f ms %.4f t_device_ms %.4f"%(sys.argv[1],p["value"]/1e6,p["ms_per_step"],p["t_device_ms"]),{k[3:]:round(v,3) for k,v in p["phases_ms"].items()})
PY
done; done
cp /tmp/lib_orig.so $L
