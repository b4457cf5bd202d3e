// zr_device_types.h — HBM data layout shared by the host (zr_host.cpp) and the kernels (zr_device.h).
#pragma once
#include <stdint.h>
#include "../../include/zr_capi.h"

#define ZR_KIND_WRAPPED 4u /* leaf kind: object with a wrapper chain (index into DScene::wrapped) */
#define ZR_STACK_DEPTH 48  /* builder guarantees tree depth <= ZR_STACK_DEPTH */
#define ZR_MAX_CHAIN 8

namespace zr {

// ---- HBM layout ----------------------------------------------------------------------------------
// One BVH record = one sibling PAIR (64 B, one 64-B aligned fetch): both children's boxes plus their
// links, so a single load decides both descents.
struct alignas(64) NodePair {
    float lo[2][3];
    float hi[2][3];
    uint32_t child[2];  // internal: index of the child's own NodePair; leaf: first primitive in its kind's array
    uint32_t meta[2];   // 0 = internal; else ((kind + 1) << 16) | count
};
static_assert(sizeof(NodePair) == 64, "NodePair must be 64 bytes");

// 4-wide node (128 B = one L2 line): the children's FP32 boxes as six float4 rows + links + leaf meta.  Walked by
// the EXTEND kernel of variant 2: half as many dependent fetches per ray as the pair records, one line each.
struct alignas(128) NodeQuad {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    uint32_t child[4];  // as NodePair::child
    uint32_t meta[4];   // as NodePair::meta; an unused slot is a leaf with count 0
};
static_assert(sizeof(NodeQuad) == 128, "NodeQuad must be 128 bytes");

struct DMedium {
    uint32_t btype, bindex, chain_first, chain_count;
    uint32_t mat, id;
    double neg_inv_density;
};
struct DWrapped { uint32_t type, index, chain_first, chain_count; };

struct DScene {
    const NodePair* nodes;
    const NodeQuad* quads;
    const double* spheres;      // 4 per sphere: cx cy cz max(0, r)
    const uint32_t* sphere_mat;
    const double* tri_v;        // 9 per triangle
    const double* tri_n;        // 9 per triangle
    const uint32_t* tri_mat;
    const double* cubes;        // 6 per cube: half extents, centre
    const uint32_t* cube_mat;
    const DMedium* media;
    const DWrapped* wrapped;
    const zr_xform_op* ops;
    const zr_material* mats;
    const zr_texture* texs;
    const unsigned char* texels;
    uint32_t n_mats;
    uint32_t root_meta;  // unused (root is pair 0)
};

struct DCamera {
    double center[3], pixel00[3], du[3], dv[3], disk_u[3], disk_v[3];
    int32_t W, H, spp, max_depth;
    int32_t defocus;  // !(defocus_angle <= 0)
    int32_t pad_;
};

// environment with everything ray-independent folded on the host (camera.hpp:874-895,914)
struct DEnv {
    uint32_t mode, hdr_tex;
    double solid[3];                 // background_color * intensity
    double cy, sy, cp, sp, cr, sr;   // cos/sin of hdri_rotation, tilt, roll
    double intensity;
    double sun[3];                   // unit sun direction
    double horizon[3], zenith[3];
    double sky_scale, sky_exposure;  // intensity*1.5, exposure
    double sun_thr;
    double sun_add[3];               // s_color * sun_intensity * visibility (alpha applied per ray), zero when disc disabled
    int32_t sun_on, pad_;
};

}  // namespace zr
