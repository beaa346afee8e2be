#!/bin/bash
# Round 5, second session, the measurement call on the final code: the GPU suite (every test, no early exit) + smoke, then the
# headline artefacts (tools/profile_round.sh: full bench line with its live traced child, kernel stats / timeline / PMC / kernel
# table, stage split, JS end-to-end), the apply timings (the delta stage changed this session) and the history timings. The
# config-5 / shuffled / c3 profiles of the first session (profiles/r05_c5_*, r05_sched_*, r05_c3_*) are of kernels this session
# did not touch; their numbers are re-measured by the bench line's sub-lines.
TAG=${1:-r05_s2_final}
mkdir -p gpurun_out/$TAG
timeout 1000 python -m pytest tests -m gpu -q > gpurun_out/$TAG/pytest_gpu.log 2>&1
tail -4 gpurun_out/$TAG/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1
timeout 300 python tools/time_apply.py > gpurun_out/$TAG/apply_changes_timings.jsonl 2> gpurun_out/$TAG/apply_changes_timings.err
timeout 200 python tools/time_history.py > gpurun_out/$TAG/history_trace.txt 2>&1
ls gpurun_out/$TAG
head -c 700 gpurun_out/$TAG/bench_line.json; echo
head -36 gpurun_out/$TAG/kernel_table.txt
cat gpurun_out/$TAG/apply_changes_timings.jsonl
