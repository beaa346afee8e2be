"""A/B of engine knobs read at context creation: median am355_replay wall time of the headline log under each environment.
   python tools/ab_replay.py 'AM355_HASH_CUS=16' 'AM355_HASH_START=intern' ...   (each argument: space-separated VAR=value pairs, '' = defaults)"""
import os
import statistics
import subprocess
import sys
import time
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch  # noqa: F401
    from automerge_classic_amd import engine, loggen
    log = loggen.config(os.environ.get("AB_WORKLOAD", "c4_text_single"), 1.0, False)
    eng = engine.Engine(0)
    eng.load_changes(log)
    ts = []
    for i in range(40):
        t0 = time.perf_counter(); eng.replay(); ts.append(time.perf_counter() - t0)
    print("%.4f ms median, %.4f ms min" % (statistics.median(ts[8:]) * 1e3, min(ts[8:]) * 1e3))
else:
    for spec in sys.argv[1:] or [""]:
        env = dict(os.environ)
        for kv in spec.split():
            k, v = kv.split("=", 1)
            env[k] = v
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=120)
        print("%-44s %s" % (spec or "(defaults)", out.stdout.strip() or out.stderr.strip()[-300:]), flush=True)
