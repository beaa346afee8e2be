#!/bin/bash
# Round 5, the measurement call on the final code: GPU suite + smoke, then tools/profile_all.sh (bench line with its live traced child,
# kernel stats / timeline / PMC of the headline replay, config-5 load, shuffled delivery, c3 timeline, history, apply timings).
TAG=${1:-r05_final}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.log 2>&1
tail -2 gpurun_out/$TAG/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_all.sh $TAG
head -c 600 gpurun_out/$TAG/bench_line.json; echo
head -40 gpurun_out/$TAG/kernel_table.txt
