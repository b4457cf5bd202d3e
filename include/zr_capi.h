/* zr_capi.h — C ABI of the MI355X path-tracing integrator ("zr" = Zenith replacement).
 *
 * The reference (jarek1992/raytracer_project) has no FFI: its seam is the C++ virtual API
 *   hittable::hit / bounding_box          /root/reference/hittable.hpp:29-36
 *   material::scatter / emitted           /root/reference/material.hpp:7-31
 *   texture::value                        /root/reference/texture.hpp:6-10
 *   camera::render(world, env, post, flag) /root/reference/camera.hpp:236
 * and its objects keep all data private.  The drop-in therefore has two layers:
 *   (1) include/zenith/ — C++20 classes with the reference's names, constructors and signatures whose
 *       camera::render() flattens the world and calls
 *   (2) this C ABI — plain pointers and sizes, no C++ or torch types — implemented by
 *       raytracer_project_amd/csrc (libzr_hip.so).  It is what a cgo/JNI/ctypes binding would bind.
 *
 * Every entry point names the reference interface it replaces.  All arithmetic on the path is FP64,
 * as in the reference (vec3 is 3 x double, /root/reference/vec3.hpp:7-115).
 *
 * Ownership: every pointer is caller-owned; the library copies on set_*.  A zr_ctx is bound to one HIP
 * device and must be used from one host thread at a time.  Functions returning int return 0 on
 * success and a negative ZR_E_* code otherwise; zr_last_error() describes the last failure of the
 * calling thread.  Nothing throws across this boundary.
 */
#ifndef ZR_CAPI_H
#define ZR_CAPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZR_ABI_VERSION 3

enum {
    ZR_OK = 0,
    ZR_E_INVALID = -1,  /* bad argument / inconsistent scene */
    ZR_E_DEVICE = -2,   /* HIP runtime error (no device, launch failure, out of memory) */
    ZR_E_STATE = -3,    /* call order violated (e.g. render before commit) */
    ZR_E_CANCELLED = -4, /* render_flag went false; partial results were written */
    ZR_E_NOMEM = -5      /* device memory exhausted where the library could not work around it (it first shrinks its own buffers) */
};

/* ---- primitive / object kinds ------------------------------------------------------------- */
enum {
    ZR_PRIM_SPHERE = 0,   /* sphere            /root/reference/sphere.hpp:5-89 */
    ZR_PRIM_TRIANGLE = 1, /* triangle          /root/reference/triangle.hpp:5-110 */
    ZR_PRIM_CUBE = 2,     /* cube              /root/reference/cube.hpp:7-143 */
    ZR_PRIM_MEDIUM = 3,   /* constant_medium   /root/reference/constant_medium.hpp:24-87 */
    ZR_PRIM_GROUP = 6     /* a run of triangles (zr_group) placed as ONE object: what a wrapper around a mesh is in the reference
                             (translate / rotate_* / scale / material_instance holding a model or a bvh_node, scene_management.hpp:
                             112-117): the run is stored and built once however many objects name it (two-level BVH) */
};

/* instance wrappers (applied outermost first to the ray, innermost first to the hit record) */
enum {
    ZR_OP_TRANSLATE = 0, /* translate          /root/reference/translate.hpp:6-41   a = offset */
    ZR_OP_ROTATE_X = 1,  /* rotate_x (degrees) /root/reference/rotate_x.hpp:7-80    a[0]=sin a[1]=cos */
    ZR_OP_ROTATE_Y = 2,  /* rotate_y (radians) /root/reference/rotate_y.hpp:7-83    a[0]=sin a[1]=cos */
    ZR_OP_ROTATE_Z = 3,  /* rotate_z (degrees) /root/reference/rotate_z.hpp:7-77    a[0]=sin a[1]=cos */
    ZR_OP_SCALE = 4,     /* scale              /root/reference/scale.hpp:5-45       a = factors */
    ZR_OP_MATERIAL = 5   /* material_instance  /root/reference/material_instance.hpp:5-41  mat = id */
};

typedef struct zr_xform_op {
    uint32_t kind; /* ZR_OP_* */
    uint32_t mat;  /* ZR_OP_MATERIAL: material id */
    double a[3];
} zr_xform_op;

/* one entry of the world list (what hittable_list::add received, hittable_list.hpp:24-27) */
typedef struct zr_object {
    uint32_t type;        /* ZR_PRIM_* */
    uint32_t index;       /* index into that type's array */
    uint32_t chain_first; /* first wrapper op (outermost) in the op array */
    uint32_t chain_count; /* number of wrapper ops, 0 = bare primitive */
} zr_object;

/* triangles [first_triangle, first_triangle + triangle_count) of the triangle arrays, in their own (object) space; they are not
 * world-list entries themselves.  A zr_object of type ZR_PRIM_GROUP with index = position in the group array places the run
 * under that object's wrapper chain; several objects may place the same group. */
typedef struct zr_group {
    uint32_t first_triangle;
    uint32_t triangle_count;
} zr_group;

/* constant_medium(boundary, density, tex|color): boundary is a sphere or cube entry that is not
 * itself part of the world list; `mat` is the id of a ZR_MAT_ISOTROPIC material. */
typedef struct zr_medium {
    uint32_t boundary_type;  /* ZR_PRIM_SPHERE or ZR_PRIM_CUBE */
    uint32_t boundary_index;
    uint32_t chain_first;    /* wrappers applied to the boundary (inside the medium) */
    uint32_t chain_count;
    uint32_t mat;
    uint32_t pad_;
    double neg_inv_density;  /* -1/density, constant_medium.hpp:30,37 */
} zr_medium;

/* ---- materials and textures --------------------------------------------------------------- */
enum {
    ZR_MAT_LAMBERTIAN = 0, /* material.hpp:58-108 */
    ZR_MAT_METAL = 1,      /* material.hpp:111-163   param = fuzz (already clamped to <= 1) */
    ZR_MAT_DIELECTRIC = 2, /* material.hpp:166-242   param = refraction index, tint = albedo colour */
    ZR_MAT_LIGHT = 3,      /* diffuse_light, material.hpp:245-279 */
    ZR_MAT_ISOTROPIC = 4   /* isovolumetric, constant_medium.hpp:9-22 */
};

#define ZR_NO_TEXTURE 0xFFFFFFFFu

typedef struct zr_material {
    uint32_t kind;          /* ZR_MAT_* */
    uint32_t tex;           /* albedo / emission texture id (unused by dielectric) */
    uint32_t bump_tex;      /* bump map texture id or ZR_NO_TEXTURE (material.hpp:35-54) */
    uint32_t pad_;
    double param;           /* fuzz or refraction index */
    double bump_strength;
    double tint[3];         /* dielectric albedo */
} zr_material;

enum {
    ZR_TEX_SOLID = 0,     /* solid_color     texture.hpp:86-102 */
    ZR_TEX_CHECKER = 1,   /* checker_texture texture.hpp:104-133 */
    ZR_TEX_IMAGE_U8 = 2,  /* image_texture, 8-bit RGB   texture.hpp:12-84 */
    ZR_TEX_IMAGE_F32 = 3  /* image_texture, float RGB (HDR) */
};

typedef struct zr_texture {
    uint32_t kind;      /* ZR_TEX_* */
    uint32_t odd, even; /* checker: child texture ids */
    uint32_t width, height; /* image; 0 x 0 = the reference's "missing image" cyan fallback */
    uint32_t pad_;
    uint64_t texel_offset;  /* byte offset of the first texel inside the texel blob */
    double inv_scale;       /* checker */
    double color[3];        /* solid */
} zr_texture;

/* ---- environment: EnvironmentSettings, /root/reference/environment.hpp:8-76 ---------------- */
enum { ZR_ENV_PHYSICAL_SUN = 0, ZR_ENV_HDR_MAP = 1, ZR_ENV_SOLID_COLOR = 2 };

typedef struct zr_env {
    uint32_t mode;        /* ZR_ENV_* (same order as EnvironmentSettings::Mode) */
    uint32_t hdr_texture; /* texture id of an IMAGE_F32 texture, or ZR_NO_TEXTURE */
    double background_color[3];
    double intensity;
    double hdri_rotation, hdri_tilt, hdri_roll; /* radians */
    double sun_direction[3];
    double sun_color[3];
    double sun_intensity;
    double sun_size;
} zr_env;

/* ---- camera: the public fields camera::render reads, /root/reference/camera.hpp:26-57 ------- */
typedef struct zr_camera {
    int32_t image_width, image_height;
    int32_t samples_per_pixel;
    int32_t max_depth;
    double vfov;
    double lookfrom[3], lookat[3], vup[3];
    double defocus_angle;
    double focus_dist;
} zr_camera;

/* which part of the frame one call renders.  Pixels are visited in 2-D tiles of tile_size x tile_size;
 * tile t (row-major tile index) is rendered iff t % tile_mod == tile_rem — the interleaved pixel-tile
 * sharding across GPUs.  The rectangle [x0,x0+w) x [y0,y0+h) further restricts the
 * pixels (used by parity tests); w = h = 0 means the full frame. */
typedef struct zr_region {
    int32_t x0, y0, w, h;
    int32_t tile_size; /* 0 = default (32) */
    int32_t tile_mod;  /* 0 or 1 = every tile */
    int32_t tile_rem;
    int32_t tile_skew; /* 0: tile t (row-major index) belongs to the part t % tile_mod; s > 0: tile (tx, ty) belongs to (tx + s * ty) % tile_mod — a lattice
                          instead of the near-vertical stripes t % tile_mod makes when the tiles per row are a multiple of tile_mod / 2 (1920 / 32 = 60 tiles per
                          row over 8 ranks: a rank then owns the same two tile columns in every row, and the ranks under the middle of the picture are 4 %
                          slower than the others; with skew 3: round 4, profiles/r4_shards.txt).  (This word was padding until round 4: 0 keeps the old map.) */
} zr_region;

/* counters of the last zr_render on a context (collected only when `collect_counters` was set:
 * the counting kernel variant is slower and is never the timed one) */
typedef struct zr_counters {
    uint64_t primary_samples;
    uint64_t segments;       /* closest-hit queries = "ray·bounces" */
    uint64_t nodes_tested;   /* BVH child boxes tested */
    uint64_t spheres_tested;
    uint64_t triangles_tested;
    uint64_t cubes_tested;
    uint64_t media_tested;
    uint64_t hits;           /* segments that found a surface / medium event */
    uint64_t rng_draws;      /* main-stream draws */
    /* wave-scheduler statistics of the EXTEND kernel (counting build): executions of the NODE / LEAF phase and the lanes that were
     * ready at each execution (lanes / (64 * execs) = SIMT utilisation of the phase); shade_* are unused by the pipeline */
    uint64_t node_execs, node_lanes, leaf_execs, leaf_lanes, shade_execs, shade_lanes;
    uint64_t rounds;         /* kernel variant 2: EXTEND/SHADE rounds of the last render */
    double extend_ms, shade_ms; /* kernel variant 2: device time of the EXTEND / SHADE launches of the last render */
    double kernel_ms;        /* device time of the render kernels of the last call (hipEvents) */
    uint64_t path;           /* which kernels rendered the last frame: 0 pixel-group megakernel (fallback), 2 streaming pipeline
                                (EXTEND / SHADE rounds), 3 fused small-scene kernel (worlds of at most ZR_FUSED_MAX objects) — ABI 3 */
} zr_counters;

/* hit record returned by zr_trace (debug / known-answer entry): hit_record, hittable.hpp:9-26 */
typedef struct zr_hit {
    double p[3], normal[3], tangent[3], bitangent[3];
    double t, u, v;
    uint32_t mat;       /* material id, 0xFFFFFFFF on a miss */
    uint32_t front_face;
} zr_hit;

/* a whole flattened world as borrowed arrays (layouts as in the zr_scene_set_* calls below) */
typedef struct zr_scene_desc {
    const double* spheres;      const uint32_t* sphere_mat;   uint64_t n_spheres;
    const double* tri_v;        const double* tri_n;          const uint32_t* tri_mat; uint64_t n_tris;
    const double* cubes;        const uint32_t* cube_mat;     uint64_t n_cubes;
    const zr_medium* media;     uint64_t n_media;
    const zr_xform_op* ops;     uint64_t n_ops;
    const zr_object* objects;   uint64_t n_objects; /* 0 = implicit world list */
    const zr_material* materials; uint64_t n_materials;
    const zr_texture* textures; uint64_t n_textures;
    const void* texels;         uint64_t texel_bytes;
    const zr_group* groups;     uint64_t n_groups;   /* ABI 2 */
} zr_scene_desc;

typedef struct zr_ctx zr_ctx;
typedef struct zr_scene zr_scene;

/* ---- lifetime ----------------------------------------------------------------------------- */
int zr_abi_version(void);
const char* zr_last_error(void);
zr_ctx* zr_create(int device_ordinal); /* NULL on failure (no HIP device => fails loudly) */
void zr_destroy(zr_ctx*);

/* ---- scene: replaces building a hittable_list / bvh_node of reference objects -------------- */
zr_scene* zr_scene_create(zr_ctx*);
void zr_scene_destroy(zr_scene*);
/* sphere(center, radius, mat): 4 doubles per sphere = cx, cy, cz, radius (raw ctor argument) */
int zr_scene_set_spheres(zr_scene*, const double* cxyz_r, const uint32_t* mat, size_t n);
/* triangle(a,b,c,n0,n1,n2,mat): 9 doubles of vertices, 9 doubles of vertex normals */
int zr_scene_set_triangles(zr_scene*, const double* v9, const double* n9, const uint32_t* mat, size_t n);
/* cube: 12 doubles = half_extents, center, min_p, max_p exactly as the cube members (cube.hpp:92-97) */
int zr_scene_set_cubes(zr_scene*, const double* hcmm12, const uint32_t* mat, size_t n);
int zr_scene_set_media(zr_scene*, const zr_medium*, size_t n);
int zr_scene_set_xform_ops(zr_scene*, const zr_xform_op*, size_t n);
/* the world list.  If never called, every sphere/triangle/cube/medium that is not a medium boundary
 * is a bare top-level object. */
int zr_scene_set_objects(zr_scene*, const zr_object*, size_t n);
/* runs of triangles that ZR_PRIM_GROUP objects place (needs an explicit world list: the implicit one would also list the
 * runs' triangles as bare objects) */
int zr_scene_set_groups(zr_scene*, const zr_group*, size_t n);
int zr_scene_set_materials(zr_scene*, const zr_material*, size_t n);
int zr_scene_set_textures(zr_scene*, const zr_texture*, size_t n, const void* texel_blob, size_t texel_bytes);
/* all of the above in one call */
int zr_scene_set_all(zr_scene*, const zr_scene_desc*);
/* the same without the copies: the library keeps POINTERS to the geometry arrays of `desc` (spheres, triangles, cubes, media, ops,
 * objects, texels — 216 MB for a million triangles) and reads them during the next zr_scene_commit, after which it forgets them.
 * The caller keeps them valid and unchanged until that commit has returned.  The small tables (materials, textures) are copied. */
int zr_scene_set_all_borrowed(zr_scene*, const zr_scene_desc*);
/* builds the flattened BVH (replaces bvh_node's constructor, bvh.hpp:11-44) and uploads to HBM */
int zr_scene_commit(zr_scene*);
/* sizes of the committed scene: out[0]=bvh nodes (child-pair records), [1]=max depth, [2]=objects,
 * [3]=device bytes */
int zr_scene_stats(const zr_scene*, uint64_t out[4]);
/* traversal-stack bound of the committed scene: the exact worst-case number of entries one ray's stack can hold while
 * walking the 4-wide tree (the recursion depth of bvh_node::hit, bvh.hpp:46-54, has no bound in the reference; here the
 * per-wave spill slabs are sized from this number, so no scene can overrun them); 0 = not committed */
uint32_t zr_scene_traversal_stack(const zr_scene*);
/* which builder made the committed tree: "host (binned SAH)" (zr_bvh.cpp) or "device (PLOC)" (zr_build.hip).  The tree is
 * built on the device for worlds of at least ZR_BVH_DEVICE_MIN world-list entries; ZR_BVH_BUILD=device|host forces one.  Replaces
 * bvh_node's constructor (/root/reference/bvh.hpp:11-44), which the reference runs on every render restart (main.cpp:1492-1500). */
const char* zr_scene_builder(const zr_scene*);

/* ---- render: replaces camera::render's sample loop (camera.hpp:236-248, 404-579) ----------- */
/* Fills out_rgb[(j*W+i)*3 + c] (host memory, W*H*3 doubles, row 0 = top, mean over spp — the layout of
 * camera::render_accumulator) for the pixels of `region`; other pixels are left untouched.
 * `keep_going` (may be NULL) is polled between kernel batches and has render_flag's polarity
 * (camera.hpp:441): *keep_going == 0 stops the render (ZR_E_CANCELLED, finished batches are written).
 * `rows_done` (may be NULL) is advanced like camera::lines_rendered (camera.hpp:548-552): while the frame renders it holds
 * H x the finished fraction of the frame's samples (the pipeline finishes samples all over the frame, not row by row) and
 * it is set to H at the end (camera.hpp:576-578).  When rows_done is given, out_rgb is also refreshed a few times per
 * second (ZR_PREVIEW_PERIOD_S, default 0.2) with the mean of the samples finished so far — the live preview the
 * reference's GUI reads from render_accumulator mid-render (main.cpp:1576); the final image does not depend on it. */
int zr_render(zr_ctx*, const zr_scene*, const zr_camera*, const zr_env*, uint64_t seed,
              const zr_region* region, int collect_counters,
              double* out_rgb, volatile const uint8_t* keep_going, volatile int* rows_done);
/* Same, but the accumulator stays in HBM: d_out_rgb is a device pointer to W*H*3 doubles and the kernels run on
 * `hip_stream` (a hipStream_t, NULL = default stream), ordered after work already enqueued there.  The call BLOCKS until
 * the frame is complete: the streaming pipeline's round loop reads the active-path count back every few rounds, so the
 * stream is synchronised internally and is idle on return (no D2H copy of the frame takes place).  This is what the
 * multi-GPU host uses before the RCCL exchange. */
int zr_render_device(zr_ctx*, const zr_scene*, const zr_camera*, const zr_env*, uint64_t seed,
                     const zr_region* region, int collect_counters,
                     void* d_out_rgb, void* hip_stream);
/* ---- first-hit AOV passes: the albedo / normal / z-depth part of render_rows (camera.hpp:433, 464-488, 521-541) ---- */
/* For the first min(clamp(spp / 8, 64, 1024), spp) samples of every pixel the primary hit contributes
 *   albedo  += rec.mat->get_albedo(rec)                       (material.hpp:29-31,99-102,154-156,226-229,266-275)
 *   normal  += (unit(rec.normal) . (u, v, w) + 1) / 2          camera space; a miss adds (0.5, 0.5, 1.0)
 *   z-depth += 1 - clamp(rec.t / z_depth_max_dist, 0, 1)       as a grey colour
 * and the sums are divided by the number of those samples.  Same seeds and the same primary rays as zr_render.
 * Output buffers are W*H*3 doubles in host memory, laid out like out_rgb; a NULL buffer skips that pass. */
typedef struct zr_aov_params {
    double z_depth_max_dist; /* post_processor::z_depth_max_dist */
} zr_aov_params;
int zr_render_aov(zr_ctx*, const zr_scene*, const zr_camera*, uint64_t seed, const zr_region* region, const zr_aov_params*,
                  double* out_albedo, double* out_normal, double* out_zdepth);

/* The render with the reflection / refraction split enabled (camera.hpp:490-517, use_reflection || use_refraction): per
 * primary sample, after the beauty path, the first hit is scattered a second time (fresh draws: the sample's stream just
 * continues) and a second path of max_depth - 1 bounces is traced; its radiance, luma-clamped to 2.0 (0.2126 |c|), times
 * the attenuation goes to the reflection frame when the scattered direction is within acos(0.9) of the mirror direction,
 * else to the refraction frame when it points below the surface.  Frames are W*H*3 doubles, means over spp (light_scale,
 * camera.hpp:531-533); a NULL buffer skips that frame.  out_beauty equals zr_render's image.  zr_get_counters afterwards
 * reports primary samples, segments (both paths), hits and RNG draws of the call. */
int zr_render_passes(zr_ctx*, const zr_scene*, const zr_camera*, const zr_env*, uint64_t seed, const zr_region* region,
                     double* out_beauty, double* out_reflection, double* out_refraction);

/* ---- after the sample loop -------------------------------------------------------------------
 * zr_post_process = camera::process_framebuffer_to_image up to the PNG encoder (camera.hpp:701-780): optional bloom
 * (bloom.hpp:18-68 on the 2^exposure-scaled frame), optional sharpening (color_processing.hpp:207-227), then per pixel
 * x 2^exposure, post_processor::process (color_processing.hpp:76-147: x exposure, colour balance, contrast about 0.18,
 * vignette, hue/saturation in HSV, ACES, debug views, clamp, gamma 1/2.2) and (unsigned char)(255.999 c).  A data pass
 * (albedo, normals, z-depth, reflection, refraction frames) is only clamped and, if apply_gamma, gamma-corrected.
 * frame_rgb: W*H*3 doubles (host); out_rgb8: W*H*3 bytes (host).  Byte-exact with the reference. */
typedef struct zr_post_params {      /* post_processor, color_processing.hpp:46-75 */
    float exposure, saturation, contrast, hue_shift, vignette_intensity;
    float bloom_threshold, bloom_intensity;
    int32_t bloom_radius;
    double color_balance[3];
    double sharpen_amount;
    int32_t use_aces_tone_mapping, use_bloom, use_sharpening;
    int32_t debug_red, debug_green, debug_blue, debug_luminance, debug_bvh;
} zr_post_params;
int zr_post_process(zr_ctx*, const zr_post_params*, const double* frame_rgb, int width, int height, int is_data_pass, int apply_gamma,
                    uint8_t* out_rgb8);
/* post_processor::analyze_framebuffer (color_processing.hpp:150-183): maximum luminance, log2-mean luminance and the
 * 256-bin log-luminance histogram that auto-exposure (apply_auto_exposure, 186-205) and the GUI plot read */
typedef struct zr_image_stats {
    float average_luminance, max_luminance;
    int32_t histogram[256];
} zr_image_stats;
int zr_analyze_frame(zr_ctx*, const double* frame_rgb, size_t n_pixels, zr_image_stats* out);

/* Known answers for whole paths: walks the primary sample (px, py, sample) of each request on the device and records
 * every segment, ZR_PATH_RECORD doubles each: ray origin, direction | hit flag, t, material id | scattered flag,
 * attenuation rgb | emission rgb | main-stream RNG draws consumed so far.  requests = n * 3 ints; out = n * max_segments *
 * ZR_PATH_RECORD doubles (zero-filled past the path's end).  Same arithmetic as zr_render; the environment plays no role
 * (a miss ends the record list). */
#define ZR_PATH_RECORD 17
int zr_trace_paths(zr_ctx*, const zr_scene*, const zr_camera*, uint64_t seed, const int32_t* requests, int n, int max_segments, double* out);

/* counters + device time of the last render on this context (synchronises the context's stream) */
int zr_get_counters(zr_ctx*, zr_counters*);
/* drains the log of render-kernel launch durations (ms, measured with HIP events on the launch stream) recorded
 * since the previous call: copies the newest min(cap, n) of them, oldest first, and returns n */
int zr_get_kernel_times(zr_ctx*, float* ms, int cap);

/* ---- multi-GPU: the one collective of a frame, for hosts that stay C++ (no torch) ---------------------------- */
/* One process per GPU.  Each rank renders its interleaved tiles with zr_render_device(region.tile_mod = nranks,
 * region.tile_rem = rank, region.tile_size) into a W*H*3 double frame in HBM.  zr_comm_gather_frame then packs the pixels of
 * the rank's own tiles (1/nranks of the frame) and sends them to `root`, which receives the other ranks' shares (one grouped
 * ncclSend / ncclRecv exchange: only the root needs the frame) and scatters them into its frame: each rank's share crosses one
 * xGMI link once instead of a whole frame of mostly zeros going through a ring reduce, and no arithmetic touches the pixels.
 * `region` is the zr_region the rank rendered with: the exchange moves tile t from rank t % nranks, so anything but the whole
 * frame with tile_mod = nranks and tile_rem = rank is refused (ZR_E_INVALID) rather than silently overwriting rendered pixels
 * (ABI 3; ABI 2 took the bare tile size).  zr_comm_reduce_frame is the round-1 form: the
 * zero-initialised frames summed onto `root` in place (ncclReduce, ncclDouble, ncclSum); tiles are disjoint, so it is exact too.
 * The reference has no counterpart: it shards rows over std::threads in one address space (camera.hpp:557-573).
 * librccl.so is loaded lazily (dlopen) by these entry points only. */
#define ZR_COMM_ID_BYTES 128
typedef struct zr_comm zr_comm;
int zr_comm_unique_id(unsigned char id[ZR_COMM_ID_BYTES]);  /* rank 0 creates it and ships it to the other ranks */
zr_comm* zr_comm_create(zr_ctx*, int nranks, int rank, const unsigned char id[ZR_COMM_ID_BYTES]);
int zr_comm_reduce_frame(zr_comm*, void* d_frame, size_t n_doubles, int root, void* hip_stream);
int zr_comm_gather_frame(zr_comm*, void* d_frame, int W, int H, const zr_region* region, int root, void* hip_stream);
void zr_comm_destroy(zr_comm*);

/* ---- known-answer entry: world.hit(r, interval(tmin,tmax), rec) for a batch of rays ---------- */
/* rays: 6 doubles each (origin, direction).  The medium draw of ray k (zr_rng.h) is keyed by
 * zr_stream_key(seed, pixel, k) and `bounce`. */
int zr_trace(zr_ctx*, const zr_scene*, const double* rays6, size_t n, double tmin, double tmax,
             uint64_t seed, uint64_t pixel, uint32_t bounce, zr_hit* out);

/* ---- per-function known-answer entry points -------------------------------------------------------------------------
 * The reference's seam is its virtual API (hittable::hit hittable.hpp:29-36, material::scatter / emitted material.hpp:7-31,
 * texture::value texture.hpp:6-10) plus camera::get_ray / get_background_color (camera.hpp:784-794, 828-925).  zr_trace above
 * answers hittable::hit; the four calls below answer the others, n calls per launch, with the device arithmetic of the
 * render kernels.  They serve the parity tests (hand-placed edge cases against the genuine reference) and the drop-in
 * classes' hit() / scatter() / emitted() in include/zenith/zenith.hpp, which stay callable without any CPU evaluation. */
typedef struct zr_scatter_out {
    double attenuation[3], origin[3], direction[3];  /* material::scatter's outputs (zero when it returned false) */
    double emitted[3];                                /* material::emitted(rec.u, rec.v, rec.p) */
    uint32_t scattered;                               /* scatter's return value */
    uint32_t draws;                                   /* random_double() calls it made */
} zr_scatter_out;
/* material::scatter(r_in, rec, attenuation, scattered) and emitted for n (ray, hit record) pairs; recs[k].mat indexes the
 * scene's materials; draws come from the contract stream keys[k] (include/zr_rng.h), starting at draw first_draw[k] (NULL = 0) */
int zr_kat_scatter(zr_ctx*, const zr_scene*, const double* rays6, const zr_hit* recs, const uint64_t* keys, const uint64_t* first_draw,
                   size_t n, zr_scatter_out* out);
/* texture::value(u, v, p) of the scene's texture `texture_id`; uvp5 = n x (u, v, px, py, pz); out = n x rgb */
int zr_kat_texture(zr_ctx*, const zr_scene*, uint32_t texture_id, const double* uvp5, size_t n, double* out_rgb);
/* camera::get_background_color(ray(., dir), env) for n directions (env.hdr_texture indexes the scene's textures) */
int zr_kat_background(zr_ctx*, const zr_scene*, const zr_env*, const double* dirs3, size_t n, double* out_rgb);
/* camera::get_ray(i, j) after camera::initialize() for n requests (i, j, sample); out = n x (origin, direction, draws used) */
int zr_kat_camera_rays(zr_ctx*, const zr_camera*, uint64_t seed, const int32_t* requests3, size_t n, double* out7);

#ifdef __cplusplus
}
#endif
#endif /* ZR_CAPI_H */
