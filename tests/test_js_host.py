"""The JavaScript host side (north_star: host code stays JavaScript, calling HIP through an N-API addon)."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, "automerge_classic_amd", "js")
NODE = shutil.which("node")


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_js_backend_reproduces_reference_goldens_on_gpu():
    """node -> mi355x-backend (index.js) -> am355_napi.node -> libam355.so -> MI355X, checked against the golden patches
    of the unmodified reference (tests/golden)."""
    addon = os.path.join(JS, "am355_napi.node")
    if not os.path.exists(addon):
        import __graft_entry__ as g
        g.build_js_addon()
    out = subprocess.run([NODE, os.path.join(JS, "test_golden.js"), os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "golden fixtures reproduced" in out.stdout


@pytest.mark.skipif(NODE is None or not os.path.isdir("/root/reference"), reason="needs node and the reference tree (build container only)")
def test_js_backend_plumbing_runs_reference_suites():
    """BASELINE config 1 (plumbing, no GPU): the reference's own suites run against the wrapper with every call
    delegated to the JS backend (the mechanism of the reference's test/wasm.js)."""
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"), MI355X_BACKEND_JS_ONLY="1",
               AUTOMERGE_BACKEND_PATH="/root/reference/backend")
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "run_ref_tests.js"), "test.js", "text_test.js", "sync_test.js"],
                         capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " 0 failed" in out.stdout


EMU_LIB = os.path.join(ROOT, "tests", "emu", "libam355_emu.so")


def _emu_env(**extra):
    """node with the CPU emulation build of the engine preloaded: the addon's am355_* calls resolve to it (test infrastructure;
    the product addon links automerge_classic_amd/csrc/libam355.so, which needs the GPU)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    if not os.path.exists(os.path.join(JS, "am355_napi.node")):
        import __graft_entry__ as g
        g.build_js_addon()
    return dict(os.environ, LD_PRELOAD=EMU_LIB, **extra)


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_js_host_materialises_reference_patches_emulated():
    """node -> index.js -> addon -> (emulated) engine -> patch IR -> materialize.js: every golden fixture through the JS Backend surface
    and a slice of the vectors captured from the reference's suites, JSON.stringify-exact against the reference's patches."""
    env = _emu_env()
    out = subprocess.run([NODE, os.path.join(JS, "test_golden.js"), os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "golden fixtures reproduced" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    out = subprocess.run([NODE, os.path.join(JS, "test_vectors.js"), os.path.join(ROOT, "tests", "golden", "ref_suite_vectors.json.gz"), "12", "5"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and '"failed":0' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.skipif(NODE is None or not os.path.isdir("/root/reference"), reason="needs node and the reference tree (build container only)")
def test_reference_suites_and_frontend_flows_over_the_engine_enabled_wrapper_emulated():
    """The reference's own suites and frontend flows (load -> getAllChanges -> change, merge of loaded documents, clone, sync, history
    snapshots) against mi355x-backend WITH the engine serving Backend.load / loadChanges (emulated kernels): GpuState handles,
    freezing and lazy hydration behave like reference handles."""
    env = _emu_env(NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"), AUTOMERGE_BACKEND_PATH="/root/reference/backend")
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "wrapper_flows.js")], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "wrapper flows ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "run_ref_tests.js"), "test.js", "text_test.js", "sync_test.js", "backend_test.js"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and " 0 failed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    served = json.loads(out.stdout.split("served by: ")[1].splitlines()[0])
    # Backend.applyChanges goes to the engine too (am355_apply_changes); the few calls it refuses (assignments to list elements ...)
    # are the only ones the reference path serves
    assert served["gpuLoad"] >= 10 and served["gpuApplyChanges"] >= 100 and served["fallbackToJs"] <= served["gpuApplyChanges"] // 10


@pytest.mark.skipif(NODE is None or not os.path.isdir("/root/reference"), reason="needs node and the reference tree (build container only)")
def test_differential_campaigns_against_the_live_reference_emulated():
    """Engine (emulated kernels, addon called directly: no JS fallback in between) against the live reference: unusual values / keys
    (patch text, materialised patch, save, load) and the history of damaged documents. The engine may refuse; it must never differ."""
    env = _emu_env(NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"))
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "values_campaign.js")], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "DISAGREE 0" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "history_campaign.js"), "12", "3"], capture_output=True, text=True, env=env, timeout=1500)
    assert out.returncode == 0 and "DISAGREE 0" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    # tables, bulk list operations (multi-insert / multi-delete ops), lists of lists, containers replaced by scalars: 5 results per scenario
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "structure_campaign.js"), "12", "9"], capture_output=True, text=True, env=env, timeout=1500)
    assert out.returncode == 0 and "DISAGREE 0" in out.stdout and "84 results identical" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.skipif(NODE is None or not os.path.isdir("/root/reference"), reason="needs node and the reference tree (build container only)")
def test_sync_protocol_over_the_engine_against_the_live_reference_emulated():
    """two peers with diverged histories sync until they agree -- reference on both sides, then the engine-enabled
    wrapper on the receiving side, on the sending side and on both: every message byte-identical, every patch equal, nothing served
    by the JS fallback (the bulk receive is am355_apply_changes, the Bloom filters are built and probed on the device)."""
    env = _emu_env(NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"), AUTOMERGE_BACKEND_PATH="/root/reference/backend")
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "sync_campaign.js"), "6"], capture_output=True, text=True, env=env, timeout=1500)
    assert out.returncode == 0 and "DISAGREE 0" in out.stdout and "18 runs identical" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    served = json.loads(out.stdout.split("served by: ")[1].splitlines()[0])
    assert served["gpuApplyChanges"] >= 12 and served["fallbackToJs"] == 0 and served["hydrations"] == 0
    # both peers start from SAVED documents (Backend.load): the protocol's queries rebuild the hash graph (am355_doc_changes, told to the
    # engine by am355_hash_graph_known) and the first message's changes are applied onto the loaded document -- all on the engine
    env["SYNC_LOADED"] = "1"
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "sync_campaign.js"), "4"], capture_output=True, text=True, env=env, timeout=1500)
    assert out.returncode == 0 and "DISAGREE 0" in out.stdout and "12 runs identical" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    served = json.loads(out.stdout.split("served by: ")[1].splitlines()[0])
    assert served["gpuLoad"] >= 8 and served["gpuApplyChanges"] >= 8 and served["fallbackToJs"] == 0 and served["hydrations"] == 0


@pytest.mark.skipif(NODE is None or not os.path.isdir("/root/reference"), reason="needs node and the reference tree (build container only)")
def test_states_built_by_several_engine_calls_hydrate_to_the_reference_state_emulated():
    """every call the engine refuses, applyLocalChange and clone run on the HYDRATED reference handle, whose objectMeta depends
    on where the calls that built the state ended: hydrate() replays the recorded calls one by one. Sessions of the applyChanges
    campaigns through the engine, and after every second call the NEXT call through the reference on the clone of the engine's state:
    the patch must be the one the reference recorded. The sessions of apply_campaign_loaded.json.gz start with Backend.load on the engine
    and go onto the loaded document (am355_apply_changes on a document context); the wrapper's own patch of every call is compared too."""
    env = _emu_env(NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"), AUTOMERGE_BACKEND_PATH="/root/reference/backend",
                   MAX_SESSIONS="8")   # (8 sessions of each fixture; `node oracle/js/hydrate_check.js <fixtures>` runs them all)
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "hydrate_check.js"), "apply_campaign.json.gz", "apply_campaign_conflicts.json.gz", "apply_campaign_loaded.json.gz"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "DIFFERENT 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert int(out.stdout.split("hydrate check: ")[1].split()[0]) >= 60


@pytest.mark.skipif(NODE is None or not os.path.isdir("/root/reference"), reason="needs node and the reference tree (build container only)")
def test_loaded_lineages_survive_a_context_that_moved_on_emulated():
    """Backend.load + applyChanges sessions (tests/golden/apply_campaign_loaded.json.gz) through the wrapper with ONE engine context that
    another document takes after every call: every call replays the retained changes of its state first -- the rebuilt changes of the
    document + what the calls since applied, told to the engine as such (am355_forget_call_history(n), am355_hash_graph_known), the
    changes the reference left queued while it rebuilt its hash graph handed over behind the next batch. Every patch the reference's."""
    env = _emu_env(NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"), AUTOMERGE_BACKEND_PATH="/root/reference/backend",
                   STEAL="1", MI355X_CONTEXTS="1", HYDRATE_EVERY="1000", MAX_SESSIONS="12")
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "hydrate_check.js"), "apply_campaign_loaded.json.gz"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "DIFFERENT 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert int(out.stdout.split("reference calls on hydrated states, ")[1].split()[0]) >= 100   # (calls the engine served; 12 of the 48 sessions)


@pytest.mark.skipif(NODE is None or not os.path.isdir("/root/reference"), reason="needs node and the reference tree (build container only)")
def test_states_after_calls_onto_loaded_documents_save_like_the_reference_emulated():
    """Backend.load + applyChanges sessions through the wrapper and the reference side by side: Backend.save and Backend.getAllChanges of
    the two states byte-identical after every fourth call (oracle/js/loaded_state_check.js)."""
    env = _emu_env(NODE_PATH=os.path.join(ROOT, "oracle", "js_shims", "node_modules"), AUTOMERGE_BACKEND_PATH="/root/reference/backend", EVERY_SESSION="2")
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "loaded_state_check.js")], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "diff 0; getAllChanges" in out.stdout and out.stdout.strip().split("getAllChanges same ")[1].split()[2] == "0", out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_js_host_materialises_incremental_patches_emulated():
    """Backend.applyChanges calls of the reference's suites: node -> addon -> (emulated) am355_apply_changes -> record tables ->
    materialize.js, JSON.stringify-exact against the patch the reference returned (a slice; the GPU suite runs all of them)."""
    env = _emu_env()
    out = subprocess.run([NODE, os.path.join(JS, "test_apply_vectors.js"), os.path.join(ROOT, "tests", "golden", "ref_apply_vectors.json.gz"), "4", "120"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["failed"] == 0 and res["equal"] >= 100
    # lists whose elements are assigned to (update edits, re-insertions, the multi-insert that keeps one value): six sessions
    out = subprocess.run([NODE, os.path.join(JS, "test_apply_vectors.js"), os.path.join(ROOT, "tests", "golden", "apply_campaign_lists.json.gz"), "0", "6"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["failed"] == 0 and res["refused"] == 0 and res["equal"] >= 60
    # sessions onto loaded documents: Backend.load, then the calls (ten sessions; the GPU suite runs all 48)
    out = subprocess.run([NODE, os.path.join(JS, "test_apply_vectors.js"), os.path.join(ROOT, "tests", "golden", "apply_campaign_loaded.json.gz"), "0", "10"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["failed"] == 0 and res["refused"] == 0 and res["equal"] >= 80


def _apply_tables_logs(tmp_path):
    from automerge_classic_amd import loggen
    logs = [loggen.generate(loggen.KIND_TEXT_TYPING, n_ops=300, ops_per_change=7, seed=73),
            loggen.generate(loggen.KIND_TEXT_CONCURRENT, n_actors=5, n_rounds=5, ins_per_change=9, del_per_change=3, n_objects=2, seed=74),
            loggen.generate(loggen.KIND_MAP_LWW, n_actors=4, n_rounds=5, n_keys=30, seed=75)]
    paths = []
    for i, log in enumerate(logs):
        paths.append(str(tmp_path / ("log%d.bin" % i)))
        log.save(paths[-1])
    return paths


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_js_host_keeps_its_per_state_tables_across_applychanges_calls_emulated(tmp_path):
    """States made by Backend.applyChanges call after call (index.js: application order as a view of 0 .. n - 1, hashes appended to a
    shared store, the binding's arena mirror extended by the batch's bytes only -- am355_arena_epoch / am355_get_hashes_range /
    am355_applied_in_input_order): getAllChanges / getChangeByHash agree with the change bytes, every patch equals the one a state built
    in one go gets for the same call. Typing, concurrent text with deletions, a map with conflicts."""
    env = _emu_env()
    for path in _apply_tables_logs(tmp_path):
        out = subprocess.run([NODE, os.path.join(JS, "test_apply_tables.js"), path], capture_output=True, text=True, env=env, timeout=900)
        assert out.returncode == 0 and '"ok":true' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_js_host_keeps_its_per_state_tables_across_applychanges_calls_on_gpu(tmp_path):
    if not os.path.exists(os.path.join(JS, "am355_napi.node")):
        import __graft_entry__ as g
        g.build_js_addon()
    for path in _apply_tables_logs(tmp_path):
        out = subprocess.run([NODE, os.path.join(JS, "test_apply_tables.js"), path], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and '"ok":true' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_js_host_reproduces_the_incremental_patches_of_the_reference_suites_on_gpu():
    if not os.path.exists(os.path.join(JS, "am355_napi.node")):
        import __graft_entry__ as g
        g.build_js_addon()
    out = subprocess.run([NODE, os.path.join(JS, "test_apply_vectors.js")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["failed"] == 0 and res["equal"] == 1578 and res["refused"] == 0 and res["rejected"] == 4   # all 1582 captured calls
    out = subprocess.run([NODE, os.path.join(JS, "test_apply_vectors.js"), os.path.join(ROOT, "tests", "golden", "apply_campaign_loaded.json.gz")],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["failed"] == 0 and res["refused"] == 0 and res["equal"] == 681   # sessions onto loaded documents
    out = subprocess.run([NODE, os.path.join(JS, "test_apply_vectors.js"), os.path.join(ROOT, "tests", "golden", "apply_campaign_lists.json.gz")],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["failed"] == 0 and res["refused"] == 0 and res["equal"] == 412


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_js_host_reproduces_all_reference_suite_vectors_on_gpu():
    """All 1499 vectors captured from the reference's suites through node -> addon -> MI355X -> materialize.js."""
    if not os.path.exists(os.path.join(JS, "am355_napi.node")):
        import __graft_entry__ as g
        g.build_js_addon()
    out = subprocess.run([NODE, os.path.join(JS, "test_vectors.js")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["failed"] == 0 and res["vectors"] >= 1499 and res["equal"] >= 1490 and res["rejected"] == [14, 16, 25, 33] and res["unsupported"] == []   # (56, 911 -- a counter inside a list -- are served since round 5)


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_js_backend_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    if not os.path.exists(os.path.join(JS, "am355_napi.node")):
        import __graft_entry__ as g
        g.build_js_addon()
    out = subprocess.run([NODE, "-e", f"require({os.path.join(JS, 'index.js')!r})"], capture_output=True, text=True)
    assert out.returncode != 0 and "no CPU fallback" in out.stderr


@pytest.mark.skipif(NODE is None, reason="node not installed")
@pytest.mark.parametrize("gpus", [2, 3])
def test_js_sharded_pool_over_worker_processes_emulated(gpus, tmp_path):
    """js/sharded.js: one worker process per GPU, the collective inside the library (here the emulated engine and a stand-in for
    librccl between the processes): every golden change log and saved document through `gpus` ranks == the reference's patches; a
    batch one rank rejects is rejected as a whole and the pool goes on."""
    env = _emu_env(AM355_RCCL_LIB=os.path.join(ROOT, "tests", "emu", "libfake_rccl.so"), AM355_SHARD_ALL_ON_DEVICE0="1", TMPDIR=str(tmp_path))
    out = subprocess.run([NODE, os.path.join(JS, "test_sharded.js"), os.path.join(ROOT, "tests", "golden"), str(gpus)], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "sharded fixtures reproduced" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    info = json.loads(out.stdout.strip().splitlines()[-2])
    assert info["sharded_fixtures"] >= 15 and info["sharded_documents"] >= 15 and len(info["fragmentBytes"]) == gpus


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_js_sharded_pool_on_gpu_world_1():
    """The same through the real addon, the MI355X and librccl.so.1 (a communicator of one rank: the test box has one GPU)."""
    out = subprocess.run([NODE, os.path.join(JS, "test_sharded.js"), os.path.join(ROOT, "tests", "golden"), "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "sharded fixtures reproduced" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
