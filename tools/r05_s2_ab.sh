#!/bin/bash
# Round 5, second session: GPU suite on the new code (counters / rows without a value inside lists; wave-level object lookup in
# k_doc_resolve), then same-box A/B of the library before (_ab/lib_old.so = 21fc5aa) and after (_ab/lib_new.so): headline, c3, config 5.
TAG=${1:-r05_s2_ab}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.log 2>&1
tail -3 gpurun_out/$TAG/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
{
echo "# headline"; bash tools/ab_libs.sh 3 old new
echo "# c3_map_lww"; AB_ARGS="--workload c3_map_lww" bash tools/ab_libs.sh 3 old new
echo "# c5_doc_mixed"
L=automerge_classic_amd/csrc/libam355.so
cp $L /tmp/lib_orig.so
for r in 1 2; do for v in old new; do
  cp _ab/lib_$v.so $L
  timeout -k 5 200 python bench.py --workload c5_doc_mixed --steps 6 --warmup 2 --prewarm 0.2 --no-sublines --no-cpu-baseline 2>/dev/null | python -c "import json,sys;p=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v','value %.0f M rows/s, ms %.2f, t_device_ms %.3f'%(p['value']/1e6,p['ms_per_step'],p['t_device_ms']))"
done; done
cp /tmp/lib_orig.so $L
} > gpurun_out/$TAG/ab.txt 2>&1
cat gpurun_out/$TAG/ab.txt
