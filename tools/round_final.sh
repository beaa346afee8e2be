#!/bin/bash
# The measurement call of a round on its final code (one gpurun call): the GPU suite (every test, no early exit) + smoke, then the
# headline artefacts (tools/profile_round.sh: bench line + full record, kernel stats / timeline / PMC / kernel table of one replay,
# stage split, JS end-to-end), the applyChanges timings and the history timings. Copy what is to be judged into profiles/ afterwards.
#   tools/round_final.sh r06_final
TAG=${1:-round_final}
mkdir -p gpurun_out/$TAG
timeout 1000 python -m pytest tests -m gpu -q > gpurun_out/$TAG/pytest_gpu.log 2>&1
tail -4 gpurun_out/$TAG/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1
timeout 300 python tools/time_apply.py > gpurun_out/$TAG/apply_changes_timings.jsonl 2> gpurun_out/$TAG/apply_changes_timings.err
timeout 200 python tools/time_history.py > gpurun_out/$TAG/history_trace.txt 2>&1
ls gpurun_out/$TAG
cat gpurun_out/$TAG/bench_line.json; echo
head -36 gpurun_out/$TAG/kernel_table.txt
cat gpurun_out/$TAG/apply_changes_timings.jsonl
