"""ctypes view of the C ABI (include/zr_capi.h) — the Python host side of the drop-in.

The reference has no Python; this module exists because the measurement harness (bench.py), the parity
tests and the multi-GPU host (one process per GPU, torch.distributed over RCCL) are Python.  It binds
exactly the entry points include/zr_capi.h declares and adds no arithmetic of its own.

There is no CPU fallback: if libzr_hip.so is missing, `load()` raises; if no HIP device is present,
`Context()` raises with the library's own message.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZR_LIB: development override (scripts/ab_flags.sh, scripts/wave_profile.sh build experimental variants out of tree)
LIB_PATH = os.environ.get("ZR_LIB") or os.path.join(_HERE, "csrc", "libzr_hip.so")
SCENES_LIB_PATH = os.path.join(_HERE, "csrc", "libzr_scenes.so")

ZR_OK, ZR_E_INVALID, ZR_E_DEVICE, ZR_E_STATE, ZR_E_CANCELLED, ZR_E_NOMEM = 0, -1, -2, -3, -4, -5
NO_TEXTURE = 0xFFFFFFFF


class XformOp(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("mat", C.c_uint32), ("a", C.c_double * 3)]


class Object(C.Structure):
    _fields_ = [("type", C.c_uint32), ("index", C.c_uint32), ("chain_first", C.c_uint32), ("chain_count", C.c_uint32)]


class Medium(C.Structure):
    _fields_ = [("boundary_type", C.c_uint32), ("boundary_index", C.c_uint32), ("chain_first", C.c_uint32),
                ("chain_count", C.c_uint32), ("mat", C.c_uint32), ("pad_", C.c_uint32), ("neg_inv_density", C.c_double)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("tex", C.c_uint32), ("bump_tex", C.c_uint32), ("pad_", C.c_uint32),
                ("param", C.c_double), ("bump_strength", C.c_double), ("tint", C.c_double * 3)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("odd", C.c_uint32), ("even", C.c_uint32), ("width", C.c_uint32),
                ("height", C.c_uint32), ("pad_", C.c_uint32), ("texel_offset", C.c_uint64), ("inv_scale", C.c_double),
                ("color", C.c_double * 3)]


class Env(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("hdr_texture", C.c_uint32), ("background_color", C.c_double * 3),
                ("intensity", C.c_double), ("hdri_rotation", C.c_double), ("hdri_tilt", C.c_double),
                ("hdri_roll", C.c_double), ("sun_direction", C.c_double * 3), ("sun_color", C.c_double * 3),
                ("sun_intensity", C.c_double), ("sun_size", C.c_double)]


class Camera(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("max_depth", C.c_int32), ("vfov", C.c_double), ("lookfrom", C.c_double * 3), ("lookat", C.c_double * 3),
                ("vup", C.c_double * 3), ("defocus_angle", C.c_double), ("focus_dist", C.c_double)]

    def copy(self):
        c = Camera()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(Camera))
        return c


class Region(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("w", C.c_int32), ("h", C.c_int32), ("tile_size", C.c_int32),
                ("tile_mod", C.c_int32), ("tile_rem", C.c_int32), ("tile_skew", C.c_int32)]


class PostParams(C.Structure):
    """zr_post_params: post_processor's fields (color_processing.hpp:46-75); defaults are the reference's"""
    _fields_ = [("exposure", C.c_float), ("saturation", C.c_float), ("contrast", C.c_float), ("hue_shift", C.c_float),
                ("vignette_intensity", C.c_float), ("bloom_threshold", C.c_float), ("bloom_intensity", C.c_float), ("bloom_radius", C.c_int32),
                ("color_balance", C.c_double * 3), ("sharpen_amount", C.c_double), ("use_aces_tone_mapping", C.c_int32),
                ("use_bloom", C.c_int32), ("use_sharpening", C.c_int32), ("debug_red", C.c_int32), ("debug_green", C.c_int32),
                ("debug_blue", C.c_int32), ("debug_luminance", C.c_int32), ("debug_bvh", C.c_int32)]

    @classmethod
    def defaults(cls, **kw):
        p = cls(0.5, 1.0, 1.0, 0.0, 1.0, 1.0, 0.3, 4, (C.c_double * 3)(1.0, 1.0, 1.0), 0.2, 0, 0, 0, 0, 0, 0, 0, 0)
        for k, v in kw.items():
            if k == "color_balance":
                p.color_balance = (C.c_double * 3)(*v)
            elif k == "debug":
                p.debug_red, p.debug_green, p.debug_blue, p.debug_luminance, p.debug_bvh = [int(x) for x in v]
            else:
                setattr(p, k, v)
        return p


class ImageStats(C.Structure):
    _fields_ = [("average_luminance", C.c_float), ("max_luminance", C.c_float), ("histogram", C.c_int32 * 256)]


class Counters(C.Structure):
    _fields_ = [("primary_samples", C.c_uint64), ("segments", C.c_uint64), ("nodes_tested", C.c_uint64),
                ("spheres_tested", C.c_uint64), ("triangles_tested", C.c_uint64), ("cubes_tested", C.c_uint64),
                ("media_tested", C.c_uint64), ("hits", C.c_uint64), ("rng_draws", C.c_uint64),
                ("node_execs", C.c_uint64), ("node_lanes", C.c_uint64), ("leaf_execs", C.c_uint64), ("leaf_lanes", C.c_uint64),
                ("shade_execs", C.c_uint64), ("shade_lanes", C.c_uint64), ("rounds", C.c_uint64),
                ("extend_ms", C.c_double), ("shade_ms", C.c_double), ("kernel_ms", C.c_double), ("path", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}

    def algorithmic_bytes(self):
        """32 B per child box tested, 72 B per triangle (9 f64 vertices), 32 B per sphere,
        48 B per cube (6 f64: this layout keeps half extents + centre only), 76 B of shading data per hit."""
        return (32 * self.nodes_tested + 72 * self.triangles_tested + 32 * self.spheres_tested
                + 48 * self.cubes_tested + 76 * self.hits)


class AovParams(C.Structure):
    _fields_ = [("z_depth_max_dist", C.c_double)]


class Hit(C.Structure):
    _fields_ = [("p", C.c_double * 3), ("normal", C.c_double * 3), ("tangent", C.c_double * 3),
                ("bitangent", C.c_double * 3), ("t", C.c_double), ("u", C.c_double), ("v", C.c_double),
                ("mat", C.c_uint32), ("front_face", C.c_uint32)]


HIT_DTYPE = np.dtype([("p", "<f8", 3), ("normal", "<f8", 3), ("tangent", "<f8", 3), ("bitangent", "<f8", 3),
                      ("t", "<f8"), ("u", "<f8"), ("v", "<f8"), ("mat", "<u4"), ("front_face", "<u4")])
assert HIT_DTYPE.itemsize == C.sizeof(Hit)


class ScatterOut(C.Structure):
    _fields_ = [("attenuation", C.c_double * 3), ("origin", C.c_double * 3), ("direction", C.c_double * 3), ("emitted", C.c_double * 3),
                ("scattered", C.c_uint32), ("draws", C.c_uint32)]


SCATTER_DTYPE = np.dtype([("attenuation", "<f8", 3), ("origin", "<f8", 3), ("direction", "<f8", 3), ("emitted", "<f8", 3),
                          ("scattered", "<u4"), ("draws", "<u4")])
assert SCATTER_DTYPE.itemsize == C.sizeof(ScatterOut)


class SceneDesc(C.Structure):
    _fields_ = [("spheres", C.c_void_p), ("sphere_mat", C.c_void_p), ("n_spheres", C.c_uint64),
                ("tri_v", C.c_void_p), ("tri_n", C.c_void_p), ("tri_mat", C.c_void_p), ("n_tris", C.c_uint64),
                ("cubes", C.c_void_p), ("cube_mat", C.c_void_p), ("n_cubes", C.c_uint64),
                ("media", C.c_void_p), ("n_media", C.c_uint64),
                ("ops", C.c_void_p), ("n_ops", C.c_uint64),
                ("objects", C.c_void_p), ("n_objects", C.c_uint64),
                ("materials", C.c_void_p), ("n_materials", C.c_uint64),
                ("textures", C.c_void_p), ("n_textures", C.c_uint64),
                ("texels", C.c_void_p), ("texel_bytes", C.c_uint64),
                ("groups", C.c_void_p), ("n_groups", C.c_uint64)]


_lib = None
_scenes = None

# every symbol include/zr_capi.h declares (tests check that the library exports all of them)
CAPI_SYMBOLS = [
    "zr_abi_version", "zr_last_error", "zr_create", "zr_destroy", "zr_scene_create", "zr_scene_destroy",
    "zr_scene_set_spheres", "zr_scene_set_triangles", "zr_scene_set_cubes", "zr_scene_set_media",
    "zr_scene_set_xform_ops", "zr_scene_set_objects", "zr_scene_set_groups", "zr_scene_set_materials", "zr_scene_set_textures",
    "zr_scene_set_all", "zr_scene_set_all_borrowed", "zr_scene_commit", "zr_scene_stats", "zr_scene_traversal_stack", "zr_scene_builder", "zr_render", "zr_render_device", "zr_render_aov", "zr_render_passes", "zr_trace_paths", "zr_post_process", "zr_analyze_frame", "zr_get_counters",
    "zr_get_kernel_times", "zr_trace", "zr_kat_scatter", "zr_kat_texture", "zr_kat_background", "zr_kat_camera_rays", "zr_comm_unique_id", "zr_comm_create", "zr_comm_reduce_frame", "zr_comm_gather_frame", "zr_comm_destroy",
]


def load():
    """Loads libzr_hip.so (raises OSError with a build hint if it is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    lib.zr_abi_version.restype = i32
    lib.zr_last_error.restype = C.c_char_p
    lib.zr_create.restype = vp; lib.zr_create.argtypes = [i32]
    lib.zr_destroy.argtypes = [vp]
    lib.zr_scene_create.restype = vp; lib.zr_scene_create.argtypes = [vp]
    lib.zr_scene_destroy.argtypes = [vp]
    lib.zr_scene_set_all.argtypes = [vp, C.POINTER(SceneDesc)]
    lib.zr_scene_set_all_borrowed.argtypes = [vp, C.POINTER(SceneDesc)]
    lib.zr_scene_set_spheres.argtypes = [vp, vp, vp, C.c_size_t]
    lib.zr_scene_set_triangles.argtypes = [vp, vp, vp, vp, C.c_size_t]
    lib.zr_scene_set_cubes.argtypes = [vp, vp, vp, C.c_size_t]
    lib.zr_scene_set_media.argtypes = [vp, vp, C.c_size_t]
    lib.zr_scene_set_xform_ops.argtypes = [vp, vp, C.c_size_t]
    lib.zr_scene_set_objects.argtypes = [vp, vp, C.c_size_t]
    lib.zr_scene_set_materials.argtypes = [vp, vp, C.c_size_t]
    lib.zr_scene_set_textures.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
    lib.zr_scene_commit.argtypes = [vp]
    lib.zr_scene_stats.argtypes = [vp, C.POINTER(u64 * 4)]
    lib.zr_scene_traversal_stack.argtypes = [vp]; lib.zr_scene_traversal_stack.restype = C.c_uint32
    lib.zr_scene_builder.argtypes = [vp]; lib.zr_scene_builder.restype = C.c_char_p
    lib.zr_render.argtypes = [vp, vp, C.POINTER(Camera), C.POINTER(Env), u64, C.POINTER(Region), i32, vp, vp, vp]
    lib.zr_render_device.argtypes = [vp, vp, C.POINTER(Camera), C.POINTER(Env), u64, C.POINTER(Region), i32, vp, vp]
    lib.zr_render_aov.argtypes = [vp, vp, C.POINTER(Camera), u64, C.POINTER(Region), C.POINTER(AovParams), vp, vp, vp]
    lib.zr_post_process.argtypes = [vp, C.POINTER(PostParams), vp, i32, i32, i32, i32, vp]
    lib.zr_analyze_frame.argtypes = [vp, vp, C.c_size_t, C.POINTER(ImageStats)]
    lib.zr_trace_paths.argtypes = [vp, vp, C.POINTER(Camera), u64, vp, i32, i32, vp]
    lib.zr_render_passes.argtypes = [vp, vp, C.POINTER(Camera), C.POINTER(Env), u64, C.POINTER(Region), vp, vp, vp]
    lib.zr_get_counters.argtypes = [vp, C.POINTER(Counters)]
    lib.zr_get_kernel_times.argtypes = [vp, C.POINTER(C.c_float), i32]
    lib.zr_trace.argtypes = [vp, vp, vp, C.c_size_t, C.c_double, C.c_double, u64, u64, C.c_uint32, vp]
    lib.zr_kat_scatter.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.zr_kat_texture.argtypes = [vp, vp, C.c_uint32, vp, C.c_size_t, vp]
    lib.zr_kat_background.argtypes = [vp, vp, C.POINTER(Env), vp, C.c_size_t, vp]
    lib.zr_kat_camera_rays.argtypes = [vp, C.POINTER(Camera), u64, vp, C.c_size_t, vp]
    lib.zr_comm_unique_id.argtypes = [vp]
    lib.zr_comm_create.restype = vp; lib.zr_comm_create.argtypes = [vp, i32, i32, vp]
    lib.zr_comm_reduce_frame.argtypes = [vp, vp, C.c_size_t, i32, vp]
    lib.zr_comm_gather_frame.argtypes = [vp, vp, i32, i32, C.POINTER(Region), i32, vp]
    lib.zr_comm_destroy.argtypes = [vp]
    _lib = lib
    return lib


def load_scenes():
    global _scenes
    if _scenes is not None:
        return _scenes
    load()
    if not os.path.exists(SCENES_LIB_PATH):
        raise OSError(f"{SCENES_LIB_PATH} not built: run __graft_entry__.build()")
    s = C.CDLL(SCENES_LIB_PATH)
    s.zrs_build.restype = C.c_void_p; s.zrs_build.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    s.zrs_free.argtypes = [C.c_void_p]
    s.zrs_desc.restype = C.POINTER(SceneDesc); s.zrs_desc.argtypes = [C.c_void_p]
    s.zrs_camera.restype = C.POINTER(Camera); s.zrs_camera.argtypes = [C.c_void_p]
    s.zrs_env.restype = C.POINTER(Env); s.zrs_env.argtypes = [C.c_void_p]
    s.zrs_seed.restype = C.c_uint64; s.zrs_seed.argtypes = [C.c_void_p]
    s.zrs_warnings.restype = C.c_char_p; s.zrs_warnings.argtypes = [C.c_void_p]
    s.zrs_render_dropin.restype = C.c_int
    s.zrs_render_dropin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(Counters)]
    _scenes = s
    return s


class ZrError(RuntimeError):
    pass


def _check(rc, allow_cancel=False):
    if rc == ZR_OK or (allow_cancel and rc == ZR_E_CANCELLED):
        return rc
    raise ZrError(f"zr error {rc}: {load().zr_last_error().decode()}")


class DemoScene:
    """One of the BASELINE.json scenes, built through the drop-in C++ scene API and flattened."""

    def __init__(self, name, *args):
        s = load_scenes()
        a = list(args) + [0] * (4 - len(args))
        self._h = s.zrs_build(name.encode(), *[int(x) for x in a[:4]])
        if not self._h:
            raise ValueError(f"unknown scene {name!r}")
        self.name = name
        self.desc = s.zrs_desc(self._h).contents
        self.camera = s.zrs_camera(self._h).contents.copy()
        self.env = s.zrs_env(self._h).contents
        self.seed = int(s.zrs_seed(self._h))
        self.warnings = s.zrs_warnings(self._h).decode()
        ids = (C.c_uint32 * 64)()
        s.zrs_kat_textures.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.kat_textures = [int(ids[k]) for k in range(min(64, s.zrs_kat_textures(self._h, ids, 64)))]

    def render_dropin(self, width=0, height=0, spp=0, device=0):
        """camera::render(world, env, post, flag) of include/zenith/zenith.hpp, end to end."""
        w = width or self.camera.image_width
        h = height or self.camera.image_height
        out = np.zeros((h, w, 3), dtype=np.float64)
        ctr = Counters()
        rc = load_scenes().zrs_render_dropin(self._h, width, height, spp, device, out.ctypes.data, C.byref(ctr))
        if rc != 0:
            raise ZrError(f"drop-in render failed: {load().zr_last_error().decode()}")
        return out, ctr

    def render_dropin_threads(self, n, width=0, height=0, spp=0, device=0):
        """n drop-in renders, each on a fresh host thread, one after the other (the reference's render-restart pattern,
        main.cpp:1520-1531): (last frame, device contexts the process has created so far)"""
        w = width or self.camera.image_width
        h = height or self.camera.image_height
        out = np.zeros((h, w, 3), dtype=np.float64)
        lib = load_scenes()
        lib.zrs_render_dropin_threads.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        rc = lib.zrs_render_dropin_threads(self._h, width, height, spp, device, n, out.ctypes.data)
        if rc < 0:
            raise ZrError(f"drop-in render failed: {load().zr_last_error().decode()}")
        return out, rc

    def dropin_virtuals(self, rays8, seed, pixel=0x7ACE):
        """bvh_node(world).hit + rec.mat->emitted / scatter through the drop-in classes (one device launch per call):
        (recs[n,16], scat[n,14]) in the layout of `zenith_ref kat <scene> hits`"""
        rays8 = np.ascontiguousarray(rays8, dtype=np.float64).reshape(-1, 8)
        recs = np.zeros((len(rays8), 16)); scat = np.zeros((len(rays8), 14))
        lib = load_scenes()
        lib.zrs_dropin_virtuals.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        if lib.zrs_dropin_virtuals(self._h, rays8.ctypes.data, len(rays8), seed, pixel, recs.ctypes.data, scat.ctypes.data) != 0:
            raise ZrError("drop-in hit()/scatter() failed (see stderr)")
        return recs, scat

    def dropin_frame_to_rgb8(self, spp=0, device=0):
        """render (auto-exposure, reflection split) + post stack through include/zenith/zenith.hpp: (rgb8, reflection8, exposure)"""
        w, h = self.camera.image_width, self.camera.image_height
        a = np.zeros((h, w, 3), dtype=np.uint8); b = np.zeros((h, w, 3), dtype=np.uint8)
        ex = C.c_float(0)
        lib = load_scenes()
        lib.zrs_dropin_frame_to_rgb8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        rc = lib.zrs_dropin_frame_to_rgb8(self._h, spp, device, a.ctypes.data, b.ctypes.data, C.byref(ex))
        if rc != 0:
            raise ZrError(f"drop-in frame pipeline failed ({rc}): {load().zr_last_error().decode()}")
        return a, b, ex.value

    def close(self):
        if self._h:
            load_scenes().zrs_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    def __init__(self, device=0):
        self.lib = load()
        self._c = self.lib.zr_create(int(device))
        if not self._c:
            raise ZrError(self.lib.zr_last_error().decode())
        self.device = device

    def close(self):
        if self._c:
            self.lib.zr_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def post_process(self, params, frame, is_data_pass=False, apply_gamma=True):
        """camera::process_framebuffer_to_image up to the PNG encoder: (H, W, 3) float64 -> (H, W, 3) uint8"""
        frame = np.ascontiguousarray(frame, dtype=np.float64)
        h, w = frame.shape[:2]
        out = np.zeros((h, w, 3), dtype=np.uint8)
        _check(self.lib.zr_post_process(self._c, C.byref(params), frame.ctypes.data, w, h, int(is_data_pass), int(apply_gamma), out.ctypes.data))
        return out

    def analyze_frame(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.float64)
        st = ImageStats()
        _check(self.lib.zr_analyze_frame(self._c, frame.ctypes.data, frame.size // 3, C.byref(st)))
        return st

    def kat_camera_rays(self, camera, seed, requests):
        """camera::initialize + get_ray: (n, 7) = origin, direction, draws"""
        req = np.ascontiguousarray(requests, dtype=np.int32).reshape(-1, 3)
        out = np.zeros((len(req), 7))
        _check(self.lib.zr_kat_camera_rays(self._c, C.byref(camera), C.c_uint64(seed), req.ctypes.data, len(req), out.ctypes.data))
        return out

    def counters(self):
        c = Counters()
        _check(self.lib.zr_get_counters(self._c, C.byref(c)))
        return c

    def kernel_times_ms(self, cap=256):
        buf = (C.c_float * cap)()
        n = self.lib.zr_get_kernel_times(self._c, buf, cap)
        if n < 0:
            _check(n)
        return [buf[i] for i in range(min(n, cap))]


class Scene:
    def __init__(self, ctx, desc):
        self.ctx = ctx
        self.lib = ctx.lib
        self._s = self.lib.zr_scene_create(ctx._c)
        if not self._s:
            raise ZrError(self.lib.zr_last_error().decode())
        # the description's arrays outlive this call (the caller holds them): no need for the library to copy 200 MB of triangles
        _check(self.lib.zr_scene_set_all_borrowed(self._s, C.byref(desc)))
        _check(self.lib.zr_scene_commit(self._s))

    def stats(self):
        out = (C.c_uint64 * 4)()
        _check(self.lib.zr_scene_stats(self._s, C.byref(out)))
        return {"bvh_pairs": out[0], "bvh_depth": out[1], "objects": out[2], "device_bytes": out[3],
                "traversal_stack": int(self.lib.zr_scene_traversal_stack(self._s)), "builder": self.lib.zr_scene_builder(self._s).decode()}

    def render(self, camera, env, seed, region=None, count=False, out=None):
        h, w = camera.image_height, camera.image_width
        if out is None:
            out = np.zeros((h, w, 3), dtype=np.float64)
        assert out.dtype == np.float64 and out.flags.c_contiguous and out.shape == (h, w, 3)
        rp = C.byref(region) if region is not None else None
        _check(self.lib.zr_render(self.ctx._c, self._s, C.byref(camera), C.byref(env), C.c_uint64(seed), rp,
                                  1 if count else 0, out.ctypes.data, None, None))
        return out

    def render_aov(self, camera, seed, z_depth_max_dist, region=None):
        """first-hit albedo / normal / z-depth passes (camera.hpp:464-488): three (H, W, 3) float64 frames"""
        h, w = camera.image_height, camera.image_width
        outs = [np.zeros((h, w, 3), dtype=np.float64) for _ in range(3)]
        ap = AovParams(float(z_depth_max_dist))
        rp = C.byref(region) if region is not None else None
        _check(self.lib.zr_render_aov(self.ctx._c, self._s, C.byref(camera), C.c_uint64(seed), rp, C.byref(ap),
                                      outs[0].ctypes.data, outs[1].ctypes.data, outs[2].ctypes.data))
        return outs

    PATH_RECORD = 17

    def trace_paths(self, camera, seed, requests, max_segments):
        """per-segment records of the primary samples (px, py, sample): (n, max_segments, 17) float64, see zr_trace_paths"""
        req = np.ascontiguousarray(requests, dtype=np.int32).reshape(-1, 3)
        out = np.zeros((req.shape[0], max_segments, self.PATH_RECORD), dtype=np.float64)
        _check(self.lib.zr_trace_paths(self.ctx._c, self._s, C.byref(camera), C.c_uint64(seed), req.ctypes.data, req.shape[0], max_segments,
                                       out.ctypes.data))
        return out

    def render_passes(self, camera, env, seed, region=None):
        """beauty / reflection / refraction frames with the split enabled (camera.hpp:490-517): three (H, W, 3) float64 frames"""
        h, w = camera.image_height, camera.image_width
        outs = [np.zeros((h, w, 3), dtype=np.float64) for _ in range(3)]
        rp = C.byref(region) if region is not None else None
        _check(self.lib.zr_render_passes(self.ctx._c, self._s, C.byref(camera), C.byref(env), C.c_uint64(seed), rp,
                                         outs[0].ctypes.data, outs[1].ctypes.data, outs[2].ctypes.data))
        return outs

    def render_device(self, camera, env, seed, d_ptr, stream=0, region=None, count=False):
        rp = C.byref(region) if region is not None else None
        _check(self.lib.zr_render_device(self.ctx._c, self._s, C.byref(camera), C.byref(env), C.c_uint64(seed), rp,
                                         1 if count else 0, C.c_void_p(d_ptr), C.c_void_p(stream)))

    def trace(self, rays, tmin=0.001, tmax=float("inf"), seed=1, pixel=0x7ACE, bounce=0):
        rays = np.ascontiguousarray(rays, dtype=np.float64)
        n = rays.shape[0]
        out = np.zeros(n, dtype=HIT_DTYPE)
        _check(self.lib.zr_trace(self.ctx._c, self._s, rays.ctypes.data, n, tmin, tmax, C.c_uint64(seed),
                                 C.c_uint64(pixel), bounce, out.ctypes.data))
        return out

    # ---- per-function known-answer entry points (zr_kat_*) ----
    def kat_scatter(self, rays, hits, keys, first_draw=None):
        """material::scatter + emitted for (ray, hit record) pairs -> array of SCATTER_DTYPE"""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        fd = np.ascontiguousarray(first_draw, dtype=np.uint64) if first_draw is not None else None
        out = np.zeros(len(rays), dtype=SCATTER_DTYPE)
        _check(self.lib.zr_kat_scatter(self.ctx._c, self._s, rays.ctypes.data, hits.ctypes.data, keys.ctypes.data,
                                       fd.ctypes.data if fd is not None else None, len(rays), out.ctypes.data))
        return out

    def kat_texture(self, tex, uvp):
        uvp = np.ascontiguousarray(uvp, dtype=np.float64).reshape(-1, 5)
        out = np.zeros((len(uvp), 3))
        _check(self.lib.zr_kat_texture(self.ctx._c, self._s, int(tex), uvp.ctypes.data, len(uvp), out.ctypes.data))
        return out

    def kat_background(self, env, dirs):
        dirs = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        out = np.zeros((len(dirs), 3))
        _check(self.lib.zr_kat_background(self.ctx._c, self._s, C.byref(env), dirs.ctypes.data, len(dirs), out.ctypes.data))
        return out

    def close(self):
        if self._s:
            self.lib.zr_scene_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
