import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tc-viml_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Builds the HIP library (cross-compiles without a GPU) and the C oracle once per session."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tcv_build", os.path.join(ROOT, "tc-viml_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()
    import orc
    orc.build()
    return True


@pytest.fixture(scope="session")
def gpu(built, request):
    import tcv
    if any(os.path.basename(str(it.fspath)) in ("test_gpu_replay.py", "test_gpu_teacher.py") for it in request.session.items):
        import replay_cache
        replay_cache.prefetch()          # CPU worker processes for the oracle replays, started before this process touches the device
    if tcv.lib().tcv_device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box (the product has no CPU fallback)")
    return tcv
