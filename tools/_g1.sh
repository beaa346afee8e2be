mkdir -p gpurun_out/s3d
AM355_TRACE=1 timeout 200 python tools/profile_apply_seq.py c4_text_single 1.0 40 4 > gpurun_out/s3d/trace40.txt 2>&1
grep -n "resident:\|delta\|ms per call" gpurun_out/s3d/trace40.txt | tail -60
{
for spec in "c4_text_single 1 20" "c4_text_single 16 8" "c4_text_single 40 8" "c4_text_single 64 8" "c4_text_single 96 6"; do
  set -- $spec
  echo "== $1: $2 change(s) per call"; timeout 200 python tools/profile_apply_seq.py $1 1.0 $2 $3
done
} > gpurun_out/s3d/apply_seq.txt 2>&1
cat gpurun_out/s3d/apply_seq.txt | cut -c1-150
export TMPDIR=/tmp
for per in 40; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s3d/prof$per -o run -- python tools/profile_apply_seq.py c4_text_single 1.0 $per 6 > gpurun_out/s3d/prof$per.log 2>&1
python tools/rocpd_timeline.py $(find gpurun_out/s3d/prof$per -name "*.db" | head -1) -3 k_decode > gpurun_out/s3d/apply${per}_timeline.txt 2>&1
rm -rf gpurun_out/s3d/prof$per
cat gpurun_out/s3d/apply${per}_timeline.txt | head -12
done
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
