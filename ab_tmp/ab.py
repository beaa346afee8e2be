import sys, time, os
sys.path.insert(0, os.getcwd())
from automerge_classic_amd import engine, loggen
libs = {"old": "ab_tmp/libam355_old.so", "new": "automerge_classic_amd/csrc/libam355.so"}
for wl in ("c3_map_lww", "c4_text_single", "c2_text_typing"):
    log = loggen.config(wl, 1.0, False)
    for rnd in range(2):
        for name, path in libs.items():
            eng = engine.Engine(0, os.path.abspath(path))
            eng.load_changes(log)
            for _ in range(5): eng.replay()
            t0 = time.perf_counter(); dec = 0.0
            for _ in range(20):
                eng.replay(); dec += eng.stats().ms_decode
            dt = (time.perf_counter() - t0) / 20
            print(wl, name, "ms/step %.3f" % (dt * 1e3), "decode %.3f" % (dec / 20), flush=True)
            eng.close()
