mkdir -p gpurun_out/s3f
AM355_TRACE=1 timeout 200 python tools/profile_apply_seq.py c4_text_single 1.0 40 4 > gpurun_out/s3f/trace40.txt 2>&1
tail -16 gpurun_out/s3f/trace40.txt
{
for spec in "c4_text_single 1 20" "c4_text_single 4 12" "c4_text_single 16 8" "c4_text_single 40 8" "c4_text_single 64 8" "c4_text_single 96 6" "c4_text_single 144 5" "c4_text_multi 2 20" "c2_text_typing 1 20" "c3_map_lww 1 20"; do
  set -- $spec
  echo "== $1: $2 change(s) per call"; timeout 200 python tools/profile_apply_seq.py $1 1.0 $2 $3
done
} > gpurun_out/s3f/apply_seq.txt 2>&1
cat gpurun_out/s3f/apply_seq.txt | cut -c1-150
timeout 900 python -m pytest tests/test_apply_engine.py tests/test_js_host.py -m gpu -q 2>&1 | tail -3
timeout 900 python tools/soak_resident.py 30000 200 2>&1 | tail -2
