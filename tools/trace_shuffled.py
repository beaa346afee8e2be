import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from automerge_classic_amd import engine, loggen
log = loggen.config("c4_text_single", 1.0, False)
log = log.reordered(np.random.default_rng(4).permutation(log.n_changes))
eng = engine.Engine(0)
eng.load_changes(log)
for i in range(4):
    if i == 3: os.environ["AM355_DEBUG_TIMING"] = "1"; os.environ["AM355_TRACE"] = "1"
    eng.replay()
print(eng.stats().fast_path, eng.stats().ms_host_schedule)
