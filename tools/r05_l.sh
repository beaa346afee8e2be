#!/bin/bash
TAG=${1:-r05l}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
for wl in c3_map_lww c3_map_lww; do
  timeout 200 python bench.py --workload $wl --steps 30 --warmup 5 --no-sublines --no-cpu-baseline --no-live-trace 2>/dev/null | python -c "import json,sys;p=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$wl value %.3f G ops/s, ms %.4f, t_device_ms %.4f'%(p['value']/1e9,p['ms_per_step'],p['t_device_ms']), p['phases_ms'])"
done
AM355_MAP_SORT_TILED=1 timeout 200 python bench.py --workload c3_map_lww --steps 30 --warmup 5 --no-sublines --no-cpu-baseline --no-live-trace 2>/dev/null | python -c "import json,sys;p=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c3 tiled sort: ms %.4f, t_device_ms %.4f'%(p['ms_per_step'],p['t_device_ms']))"
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $OUT/kt3 -o run -- python bench.py --workload c3_map_lww --steps 10 --warmup 3 --no-sublines --no-cpu-baseline --no-live-trace > $OUT/c3_under_trace.json 2> $OUT/kt3.err
python tools/rocpd_timeline.py $OUT/kt3/run_results.db -2 > $OUT/c3_timeline.txt 2>&1
rm -rf $OUT/kt3
cat $OUT/c3_timeline.txt | tail -22
