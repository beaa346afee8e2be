mkdir -p gpurun_out/s3c
{
for spec in "c4_text_single 1 20" "c4_text_single 4 12" "c4_text_single 16 8" "c4_text_single 40 8" "c4_text_single 64 8" "c4_text_single 96 6" "c4_text_multi 2 20" "c2_text_typing 1 20"; do
  set -- $spec
  echo "== $1: $2 change(s) per call"; timeout 200 python tools/profile_apply_seq.py $1 1.0 $2 $3
done
} > gpurun_out/s3c/apply_seq.txt 2>&1
cat gpurun_out/s3c/apply_seq.txt | cut -c1-150
export TMPDIR=/tmp
for per in 40; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s3c/prof$per -o run -- python tools/profile_apply_seq.py c4_text_single 1.0 $per 6 > gpurun_out/s3c/prof$per.log 2>&1
python tools/rocpd_timeline.py $(find gpurun_out/s3c/prof$per -name "*.db" | head -1) -3 k_decode > gpurun_out/s3c/apply${per}_timeline.txt 2>&1
rm -rf gpurun_out/s3c/prof$per
cat gpurun_out/s3c/apply${per}_timeline.txt | tail -45
done
timeout 600 python -m pytest tests/test_apply_engine.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python tools/soak_resident.py 12000 150 2>&1 | tail -2
