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


class FlatScene:
    """A flattened world from a fixture (tests/golden/<name>.npz: every array of a zr_scene_desc, structs as raw bytes, + camera, environment, seed): owns
    the arrays the SceneDesc points into.  `mats` / `ops` / `objects` are numpy views of the struct arrays and may be edited before the scene is committed."""

    def __init__(self, name):
        import ctypes as C
        from raytracer_project_amd import capi
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(str(z["meta"]))
        self.seed = int(self.meta["seed"])
        self.a = {k: np.ascontiguousarray(z[k]).copy() for k in z.files if k != "meta"}
        self.camera = capi.Camera.from_buffer_copy(self.a["camera"].tobytes())
        self.env = capi.Env.from_buffer_copy(self.a["env"].tobytes())
        self.capi, self.C = capi, C

    def records(self, key, ctype):
        """the struct array `key` as a list of ctypes records (copies)"""
        n = self.a[key].size // self.C.sizeof(ctype)
        return [ctype.from_buffer_copy(self.a[key][i * self.C.sizeof(ctype):(i + 1) * self.C.sizeof(ctype)].tobytes()) for i in range(n)]

    def set_records(self, key, recs):
        self.a[key] = np.frombuffer(b"".join(bytes(r) for r in recs), dtype=np.uint8).copy()

    @property
    def desc(self):
        capi, C, a = self.capi, self.C, self.a
        d = capi.SceneDesc()
        ptr = lambda k: C.c_void_p(a[k].ctypes.data) if a[k].size else None
        d.spheres, d.sphere_mat, d.n_spheres = ptr("spheres"), ptr("sphere_mat"), a["sphere_mat"].size
        d.tri_v, d.tri_n, d.tri_mat, d.n_tris = ptr("tri_v"), ptr("tri_n"), ptr("tri_mat"), a["tri_mat"].size
        d.cubes, d.cube_mat, d.n_cubes = ptr("cubes"), ptr("cube_mat"), a["cube_mat"].size
        d.media, d.n_media = ptr("media"), a["media"].size // C.sizeof(capi.Medium)
        d.ops, d.n_ops = ptr("ops"), a["ops"].size // C.sizeof(capi.XformOp)
        d.objects, d.n_objects = ptr("objects"), a["objects"].size // C.sizeof(capi.Object)
        d.materials, d.n_materials = ptr("materials"), a["materials"].size // C.sizeof(capi.Material)
        d.textures, d.n_textures = ptr("textures"), a["textures"].size // C.sizeof(capi.Texture)
        d.texels, d.texel_bytes = ptr("texels"), a["texels"].size
        d.groups, d.n_groups = ptr("groups"), a["groups"].size // 8
        return d
