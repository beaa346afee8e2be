#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of the headline replay under environment variants:
#   tools/ab_kernels.sh '<kernel name regex>' 'VAR=val VAR2=val' '' ...       ('' = defaults)
export TMPDIR=/tmp
pat=$1; shift
for v in "$@"; do
  rm -rf /tmp/abk
  env $v timeout -k 5 120 rocprofv3 --kernel-trace --stats -d /tmp/abk -o run -- python bench.py --steps 10 --warmup 3 --no-sublines --no-cpu-baseline > /tmp/abk.json 2> /tmp/abk.err
  echo "== ${v:-(defaults)}: $(python -c "import json;p=json.loads(open('/tmp/abk.json').read().strip().splitlines()[-1]);print('t_device_ms %.4f'%p['t_device_ms'])" 2>/dev/null)"
  python tools/rocpd_summary.py /tmp/abk/run_results.db 26 2>&1 | grep -E "$pat"
done
