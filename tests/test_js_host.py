"""The JavaScript host side (north_star: host code stays JavaScript, calling HIP through an N-API addon)."""
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
