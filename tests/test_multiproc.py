"""N>1 path of bench.py on CPU: world_size 2 over gloo. Replicas only: each rank replays its own
document (here through the CPU emulation build of the kernels) and the ranks only exchange the timing contract."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
from automerge_classic_amd import dist_util, engine, loggen
import oracle_lib
dist.init_process_group("gloo")
rank, world, _ = dist_util.rank_world()
assert world == 2 and rank == dist.get_rank()
seed = dist_util.rank_seed(0x5EED0004, rank)
log = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=6, n_rounds=2, ins_per_change=25, del_per_change=5, n_objects=1, seed=seed)
eng = engine.Engine(0, os.path.join(ROOT, "tests", "emu", "libam355_emu.so"))
eng.load_changes(log); eng.replay()
assert eng.patch_json() == oracle_lib.OracleDoc(log).patch_json()
# ranks hold different documents
hs = [None, None]
dist.all_gather_object(hs, eng.hashes().tobytes())
assert hs[0] != hs[1]
elapsed, ops = dist_util.aggregate(1.0 + rank, float(eng.stats().n_ops), dist)
assert elapsed == 2.0 and ops == 2.0 * eng.stats().n_ops
dist.barrier()
dist.destroy_process_group()
sys.stdout.write("rank%dok\n" % rank); sys.stdout.flush()
'''


def test_two_rank_replicas_over_gloo(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\n" + WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "rank0ok" in out.stdout and "rank1ok" in out.stdout
