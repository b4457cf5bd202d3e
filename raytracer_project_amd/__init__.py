"""raytracer_project_amd — MI355X-native path-tracing integrator behind the reference's scene/render API.

Only what the hot path needs lives here:
  csrc/     hand-written HIP kernels (gfx950), host BVH builder, the C ABI (include/zr_capi.h) and the
            BASELINE.json scenes compiled against the drop-in C++ API (include/zenith/zenith.hpp)
  capi.py   ctypes bindings of that C ABI for Python hosts (bench.py, tests, the multi-GPU driver)
  multi.py  pixel-tile sharding across GPUs, one process per GPU, accumulator reduce over RCCL
"""
from . import capi  # noqa: F401
from .capi import Camera, Context, Counters, DemoScene, Env, Region, Scene, ZrError  # noqa: F401
