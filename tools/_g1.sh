mkdir -p gpurun_out/s3g
python -c "from automerge_classic_amd import loggen; loggen.config('c4_text_single', 1.0, False).save('/tmp/c4.bin')"
for spec in "1 20" "40 8"; do set -- $spec
AM355_JS_PROFILE=1 timeout 200 node automerge_classic_amd/js/bench_apply.js /tmp/c4.bin $1 $2
done > gpurun_out/s3g/js_apply.txt 2>&1
cat gpurun_out/s3g/js_apply.txt
