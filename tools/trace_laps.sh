#!/bin/bash
# host-side laps of a few headline replays (AM355_TRACE=1): where the calling thread is while the device works
mkdir -p gpurun_out/laps
AM355_TRACE=1 timeout 120 python - > gpurun_out/laps/laps.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, '.')
from automerge_classic_amd import engine, loggen
log = loggen.config('c4_text_single', 1.0, False)
eng = engine.Engine(0)
for i in range(6):
    sys.stderr.write('--- replay %d\n' % i)
    eng.load_changes(log); eng.replay(); eng.fetch_ir()
PY
tail -60 gpurun_out/laps/laps.txt
