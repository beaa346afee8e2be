mkdir -p gpurun_out/s3h
AM355_TRACE=1 timeout 200 python tools/profile_apply_seq.py c3_map_lww 1.0 1 4 > gpurun_out/s3h/trace_c3.txt 2>&1
tail -40 gpurun_out/s3h/trace_c3.txt
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s3h/prof -o run -- python tools/profile_apply_seq.py c3_map_lww 1.0 1 6 > gpurun_out/s3h/prof.log 2>&1
python tools/rocpd_timeline.py $(find gpurun_out/s3h/prof -name "*.db" | head -1) -3 k_decode > gpurun_out/s3h/c3_timeline.txt 2>&1
rm -rf gpurun_out/s3h/prof
cat gpurun_out/s3h/c3_timeline.txt
