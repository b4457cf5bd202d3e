// zr_launch.h — host-visible launch interface of zr_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/zr_capi.h"

#define ZR_BLOCK 256

namespace zr {

struct DScene;
struct DCamera;
struct DEnv;

// which pixels one launch covers: the tiles `tiles[0..n_tiles)` (row-major tile ids), clipped to the
// rectangle [x0,x1) x [y0,y1)
struct WorkDesc {
    const int32_t* tiles;
    int32_t n_tiles, tile_size, tiles_x;
    int32_t x0, y0, x1, y1;
    int32_t lanes_per_pixel;
};

hipError_t launch_render(const DScene& sc, const DCamera& cam, const DEnv& env, uint64_t seed, const WorkDesc& wd, double* out,
                         unsigned long long* gctr, bool count, hipStream_t stream);
// variant 2: streaming wavefront pipeline (zr_stream.hip)
struct StreamTimer {  // host-provided HIP-event recorder; kind: 0 init, 1 extend, 2 shade, 3 reduce
    virtual void begin(hipStream_t, int kind) = 0;
    virtual void end(hipStream_t, int kind) = 0;
    virtual ~StreamTimer() = default;
};
// host-provided progress sink of the round loop (camera::lines_rendered and the live preview of camera.hpp:548-552 / main.cpp:1576):
// after every host synchronisation of the loop, report() gets the fraction of the frame's samples that are finished;
// when wants_frame() said yes just before, `out` holds the mean of the samples finished so far (partial sums / spp)
struct StreamProgress {
    virtual bool wants_frame() = 0;
    virtual void report(double finished_fraction, bool frame_reduced) = 0;
    virtual ~StreamProgress() = default;
};
#define ST_MAX_POOLS 8   /* sub-pools of the slot pool, one HIP stream each */
uint32_t stream_overflow_levels(uint32_t stack_demand);
size_t stream_overflow_bytes(int blocks, uint32_t levels);
size_t stream_ctl_words();
int stream_extend_blocks();
size_t stream_pool_bytes(uint32_t P);
hipError_t stream_render(const DScene& sc, const DCamera& cam, const DEnv& env, uint64_t seed, void* pool, uint32_t P, uint32_t spp,
                         uint32_t n_pix, const uint32_t* d_pixels, double* d_samples, unsigned int* d_ctl, void* d_overflow, uint32_t ovf_levels, int extend_blocks,
                         double* out, unsigned long long* gctr, bool count, hipStream_t* streams, int n_pools, hipEvent_t ev, StreamTimer* timer,
                         unsigned int* h_active, volatile const uint8_t* keep_going, int* rounds_out, int leaf_level, int mode = 0, void* d_kend = nullptr,
                         void* d_cls = nullptr, double* out2 = nullptr, unsigned long long* d_cpart = nullptr, StreamProgress* progress = nullptr,
                         void* drain_pool = nullptr, uint32_t drain_slots = 0, uint32_t unit_chunk = 0);
// The objects of a small world as the fused kernel wants them: IN THE KERNEL ARGUMENTS.  The kernarg segment is read with scalar
// loads, so an object's record reaches every lane of a wave through SGPRs — no vector memory instruction, no VGPRs per lane for
// data that is the same in all of them (reading the records through the scene's pointers, the compiler issued 255 vector loads
// in the loop and spilled; the pointers come out of a struct, so it cannot prove the addresses uniform and read-only).
// rec: sphere cx cy cz r | triangle 9 vertices | cube 6 | placed cube 16 (ZR_PCUBE_STRIDE) | plain medium: boundary (sphere 4 / cube 6), [6] = -1/density,
// [7] = id bits, [8] = boundary type bits.
#define ZR_FUSED_OBJECTS 16
struct FusedObjs {
    uint32_t n, pad_;
    uint32_t kind[ZR_FUSED_OBJECTS];    // leaf kind (ZR_PRIM_* / ZR_KIND_PCUBE); media here are plain ones only
    uint32_t index[ZR_FUSED_OBJECTS];   // index in that kind's array (what a hit reports)
    double rec[ZR_FUSED_OBJECTS][16];   // (a placed cube's record is the longest: ZR_PCUBE_STRIDE)
};
int fused_blocks();
hipError_t fused_render_frame(const DScene& sc, const DCamera& cam, const DEnv& env, uint64_t seed, uint32_t spp, uint32_t n_pix, const uint32_t* d_pixels, double* d_samples,
                              unsigned int* d_ctl, int blocks, double* out, unsigned long long* gctr, bool count, int level, hipStream_t stream, StreamTimer* timer,
                              const FusedObjs& objs, volatile const uint8_t* keep_going = nullptr, StreamProgress* progress = nullptr, int* parts_done = nullptr);
hipError_t stream_trace(const DScene& sc, const double* d_rays, uint32_t n, uint64_t seed, uint64_t pixel, uint32_t bounce, zr_hit* d_out,
                        void* pool, unsigned int* d_ctl, void* d_overflow, uint32_t ovf_levels, int extend_blocks, unsigned long long* gctr, int leaf_level,
                        hipStream_t stream);
hipError_t launch_aov(const DScene& sc, const DCamera& cam, uint64_t seed, const WorkDesc& wd, int aux, double zmax, const double* uvw9,
                      double* out_albedo, double* out_normal, double* out_zdepth, hipStream_t stream);
hipError_t launch_passes(const DScene& sc, const DCamera& cam, const DEnv& env, uint64_t seed, const WorkDesc& wd, double* out_beauty,
                         double* out_reflection, double* out_refraction, unsigned long long* gctr, hipStream_t stream);
hipError_t launch_post(const double* d_frame, int W, int H, const zr_post_params& pp, int is_data_pass, int apply_gamma, double ev, double* d_tmp0,
                       double* d_tmp1, double* d_tmp2, uint8_t* d_out, hipStream_t stream);
hipError_t launch_analyze(const double* d_frame, size_t n, double* d_part_log, float* d_part_max, int* d_hist, hipStream_t stream);
#define ZR_PATH_REC 17
hipError_t launch_path_records(const DScene& sc, const DCamera& cam, uint64_t seed, const int32_t* req, int n_req, int max_seg, double* out,
                               hipStream_t stream);
hipError_t launch_kat_scatter(const DScene& sc, const double* rays, const zr_hit* recs, const uint64_t* keys, const uint64_t* first_draw, size_t n,
                              zr_scatter_out* out, hipStream_t stream);
hipError_t launch_kat_texture(const DScene& sc, uint32_t tex, const double* uvp, size_t n, double* out, hipStream_t stream);
hipError_t launch_kat_background(const DScene& sc, const DEnv& env, const double* dirs, size_t n, double* out, hipStream_t stream);
hipError_t launch_kat_camera_rays(const DCamera& cam, uint64_t seed, const int32_t* req, size_t n, double* out, hipStream_t stream);
hipError_t launch_trace(const DScene& sc, const double* rays, size_t n, double tmin, double tmax, uint64_t seed, uint64_t pixel,
                        uint32_t bounce, zr_hit* out, hipStream_t stream);

}  // namespace zr
