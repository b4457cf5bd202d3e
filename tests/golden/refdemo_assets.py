"""A working directory for the reference's own scene code (scene_management.hpp reads `assets/...` relative to the cwd): the reference's assets/ tree by
symlink plus a generated teapot.obj — the real one is a blob missing from the reference checkout (.MISSING_LARGE_BLOBS), without which the reference
itself dereferences a null mesh; a small lathe body stands in for it (synthetic input of the same shape).  Build container only."""
import os

import numpy as np

REF = "/root/reference"


def asset_dir(tmp):
    """a working directory with the reference's assets/ tree (symlinks) plus a generated teapot.obj"""
    for sub in ("bump_maps", "textures"):
        os.makedirs(os.path.join(tmp, "assets"), exist_ok=True)
        os.symlink(os.path.join(REF, "assets", sub), os.path.join(tmp, "assets", sub))
    os.makedirs(os.path.join(tmp, "assets", "models"))
    for f in os.listdir(os.path.join(REF, "assets", "models")):
        os.symlink(os.path.join(REF, "assets", "models", f), os.path.join(tmp, "assets", "models", f))
    with open(os.path.join(tmp, "assets", "models", "teapot.obj"), "w") as f:   # a small lathe body: quads, no normals
        n, rings = 24, [(0.0, 0.0), (1.2, 0.0), (1.6, 0.8), (1.3, 1.6), (0.5, 2.0), (0.0, 2.1)]
        for r, y in rings:
            for k in range(n):
                a = 2 * np.pi * k / n
                f.write("v %.6f %.6f %.6f\n" % (r * np.cos(a), y, r * np.sin(a)))
        for j in range(len(rings) - 1):
            for k in range(n):
                a, b = j * n + k + 1, j * n + (k + 1) % n + 1
                f.write("f %d %d %d %d\n" % (a, b, b + n, a + n))
    return tmp


