mkdir -p gpurun_out/s3j
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
{
echo "== 1 map change (8 keys) per call onto the 1 M-op text + map document"; timeout 200 python tools/profile_apply_mixed.py 1.0 1 40
echo "   the whole merge per call (AM355_NO_MAPS_ONLY=1):"; AM355_NO_MAPS_ONLY=1 timeout 200 python tools/profile_apply_mixed.py 1.0 1 40
echo "   full replay per call (AM355_NO_RESIDENT=1):"; AM355_NO_RESIDENT=1 timeout 200 python tools/profile_apply_mixed.py 1.0 1 40
echo "== 8 map changes per call"; timeout 200 python tools/profile_apply_mixed.py 1.0 8 20
echo "   the whole merge per call (AM355_NO_MAPS_ONLY=1):"; AM355_NO_MAPS_ONLY=1 timeout 200 python tools/profile_apply_mixed.py 1.0 8 20
} > gpurun_out/s3j/apply_mixed.txt 2>&1
cat gpurun_out/s3j/apply_mixed.txt | cut -c1-60,250-
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s3j/prof -o run -- python tools/profile_apply_mixed.py 1.0 1 8 > gpurun_out/s3j/prof.log 2>&1
python tools/rocpd_timeline.py $(find gpurun_out/s3j/prof -name "*.db" | head -1) -3 k_decode > gpurun_out/s3j/mixed_timeline.txt 2>&1
rm -rf gpurun_out/s3j/prof
head -16 gpurun_out/s3j/mixed_timeline.txt
AM355_TRACE=1 timeout 200 python tools/profile_apply_mixed.py 1.0 1 4 2>&1 | tail -22
timeout 900 python tools/soak_resident.py 80000 200 2>&1 | tail -2
