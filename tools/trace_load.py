import os, sys
sys.path.insert(0, '.')
import torch
from automerge_classic_amd import engine, loggen
log = loggen.config("c4_text_single", 1.0, "deflate" in sys.argv)
eng = engine.Engine(0)
for i in range(6):
    if i == 5: os.environ["AM355_TRACE"] = "1"
    eng.load_changes(log)
    eng.replay(); eng.fetch_ir()
