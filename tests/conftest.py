import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"])) if "meta" in d else {}
    return d


@pytest.fixture(scope="session")
def built():
    """Builds (once) the product libraries and the oracle; the GPU box uses the prebuilt in-tree .so files."""
    import __graft_entry__ as g
    g.build()
    return True


_scene_cache = {}


def demo_scene(name, args=()):
    from raytracer_project_amd import capi
    key = (name, tuple(args))
    if key not in _scene_cache:
        _scene_cache[key] = capi.DemoScene(name, *args)
    return _scene_cache[key]


def rel_err(a, b, floor=1e-9):
    """per-channel relative error |a-b| / max(|b|, floor)"""
    return np.abs(a - b) / np.maximum(np.abs(b), floor)
