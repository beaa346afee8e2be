mkdir -p gpurun_out/s3k
timeout 900 python -m pytest tests/test_apply_engine.py tests/test_js_host.py -m gpu -q -x 2>&1 | tail -3
{
echo "== 1 text change (250 ops) + 1 map change (8 keys) per call onto the 1 M-op text + map document"; timeout 200 python tools/profile_apply_mixed.py 1.0 1 40 both
echo "   the whole merge per call (AM355_NO_MAPS_ONLY=1):"; AM355_NO_MAPS_ONLY=1 timeout 200 python tools/profile_apply_mixed.py 1.0 1 40 both
echo "   full replay per call (AM355_NO_RESIDENT=1):"; AM355_NO_RESIDENT=1 timeout 200 python tools/profile_apply_mixed.py 1.0 1 40 both
echo "== 1 map change per call"; timeout 200 python tools/profile_apply_mixed.py 1.0 1 40
} > gpurun_out/s3k/apply_mixed_both.txt 2>&1
cat gpurun_out/s3k/apply_mixed_both.txt | cut -c1-70,255-
timeout 900 python tools/soak_resident.py 110000 250 2>&1 | tail -2
