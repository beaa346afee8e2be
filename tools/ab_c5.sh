L=automerge_classic_amd/csrc/libam355.so
cp $L /tmp/lib_orig.so
for r in 1 2; do for v in head scan; do
  cp _ab/lib_$v.so $L
  timeout -k 5 200 python bench.py --workload c5_doc_mixed --steps 6 --warmup 2 --prewarm 0.2 --no-sublines --no-cpu-baseline 2>/dev/null | python -c "import json,sys;p=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v','value %.0f M rows/s, ms %.2f, t_device_ms %.3f'%(p['value']/1e6,p['ms_per_step'],p['t_device_ms']))"
done; done
cp /tmp/lib_orig.so $L
