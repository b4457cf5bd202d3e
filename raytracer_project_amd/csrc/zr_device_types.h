// zr_device_types.h — HBM data layout shared by the host (zr_commit.cpp, zr_render.cpp) and the kernels (zr_device.h).
#pragma once
#include <stdint.h>
#include "../../include/zr_capi.h"

#define ZR_KIND_WRAPPED 4u /* leaf kind: object with a wrapper chain (index into DScene::wrapped) */
#define ZR_KIND_PCUBE 5u   /* leaf kind: a "placed" cube — cube -> [rotate_y] -> translate, the way every cube of the reference's scenes enters the
                              world (scene_management.hpp:132-139): the two wrapper parameters travel with the cube (DScene::pcubes) */
#define ZR_KIND_INSTANCE 6u /* leaf kind (= ZR_PRIM_GROUP): a run of triangles with a pair-BVH of its own, placed under a wrapper chain (DScene::insts).
                              A hit on one is reported as kind = ZR_KIND_INSTANCE | instance << 8, index = the triangle. */
#ifndef ZR_TRI_STRIDE
#define ZR_TRI_STRIDE 9    /* doubles between two records of DScene::tri_v (9 are used) */
#endif
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

// 4-wide nodes of the EXTEND kernel (variant 2).
//  * NodeQ, 64 bytes = four 16-byte loads per visit: the children's boxes are 8-bit quantised on the node's own
//    grid: plane = origin[axis] + q * scale[axis] (as real numbers; the kernel never rounds it, see EXTEND), and the
//    builder picks the q's so that this box CONTAINS the true box (lower planes rounded down, upper planes up).  Box tests were already conservative; a looser box only adds visits.
//  * NodeF, plain FP32 boxes: the ROOT only.  It travels in the kernel arguments (scalar registers), so the first
//    step of every ray costs no memory access, and it is the node where a grid hurts most (a ground sphere next to
//    the mesh: the mesh's box on a 2000-unit grid swallowed the camera and every ray paid 2.4 extra visits).
// Measured on MI355X (scripts/dev/randread.hip): a wave's divergent 16-byte loads cost the L1 about one clock per
// lane and instruction, and a random record costs the fabric one request whatever its size.
//   q[0..2] = lower x, y, z planes, q[3..5] = upper; byte c of each word belongs to child c.
//   ref[c]: ZR_REF_EMPTY, an inner node's index, or ZR_REF_LEAF | kind << 28 | (count - 1) << 24 | first primitive.
#define ZR_PCUBE_STRIDE 16 /* doubles per placed-cube record (128 bytes) */
#define ZR_REF_LEAF 0x80000000u
#define ZR_REF_EMPTY 0xFFFFFFFFu
struct alignas(64) NodeQ {
    float origin[3];
    float scale[3];
    uint32_t q[6];
    uint32_t ref[4];
};
static_assert(sizeof(NodeQ) == 64, "NodeQ must be 64 bytes");
struct NodeF {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    uint32_t ref[4];
};

struct DMedium {
    uint32_t btype, bindex, chain_first, chain_count;
    uint32_t mat, id;
    double neg_inv_density;
};
struct DWrapped { uint32_t type, index, chain_first, chain_count; };
struct DInstance { uint32_t chain_first, chain_count, root /* pair index of the run's subtree in DScene::nodes */, pad_ /* = qroot: the run's root among DScene::quads */; };

struct DScene {
    const NodePair* nodes;
    const NodeQ* quads;
    const double* spheres;      // 4 per sphere: cx cy cz max(0, r)
    const uint32_t* sphere_mat;
    const double* tri_v;        // 9 per triangle
    const double* tri_s;        // shading record, 20 doubles (160 B, two 128-B lines at any 32-B phase) per triangle:
                                // [0..8] vertices, [9..17] vertex normals (need not be unit), [18] low word = material id,
                                // [19] low word bit 0 = front_face forced true (triangle baked from under a translate / rotate_y)
    const double* cubes;        // 6 per cube: half extents, centre
    const uint32_t* cube_mat;
    const double* pcubes;       // placed cubes, ZR_PCUBE_STRIDE (16) per cube: half extents, centre | translate offset | rotate_y sin, cos, has_rotation | scale x, y, z, has_scale
    const uint32_t* pcube_mat;
    const DMedium* media;
    const DWrapped* wrapped;
    const DInstance* insts;
    const zr_xform_op* ops;
    const zr_material* mats;
    const zr_texture* texs;
    const unsigned char* texels;
    uint32_t n_mats;
    uint32_t mat_kinds;  // bit k set: a material of kind k exists (SHADE sorts by kind only when more than one does)
    uint32_t shade_lean; // 1: SHADE may run its lean build (bare triangles / spheres, lambertian / metal / dielectric / light over solid colours, no bump maps)
    uint32_t leaf_cnt[8]; // leaf objects per kind (the first leaf_cnt[k] records of kind k's array): the fused small-scene kernel tests them all
    NodeF root;          // variant 2: the root of the 4-wide tree
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
