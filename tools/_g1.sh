mkdir -p gpurun_out/s3a
{
for spec in "c4_text_single 1 20" "c4_text_single 16 8" "c4_text_single 40 8" "c4_text_single 64 8" "c4_text_single 96 6"; do
  set -- $spec
  echo "== $1: $2 change(s) per call"; timeout 200 python tools/profile_apply_seq.py $1 1.0 $2 $3
done
} > gpurun_out/s3a/apply_seq.txt 2>&1
cat gpurun_out/s3a/apply_seq.txt | cut -c1-120
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s3a/prof40 -o run -- python tools/profile_apply_seq.py c4_text_single 1.0 40 8 > gpurun_out/s3a/prof40.log 2>&1
python tools/rocpd_timeline.py $(find gpurun_out/s3a/prof40 -name "*.db" | head -1) -3 k_decode > gpurun_out/s3a/apply40_timeline.txt 2>&1
rm -rf gpurun_out/s3a/prof40
cat gpurun_out/s3a/apply40_timeline.txt | tail -60
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
