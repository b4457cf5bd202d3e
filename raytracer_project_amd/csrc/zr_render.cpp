// zr_render.cpp — the render side of the C ABI (include/zr_capi.h): camera frame and sky constants, the round loop of the streaming pipeline, the fused
// small-scene kernel, the pixel-group fallback, AOV and split passes, the post stack, counters and the known-answer entry points.
#include "zr_host_internal.h"

namespace {

// camera::initialize, camera.hpp:358-399
void make_camera(const zr_camera& c, zr::DCamera& d) {
    int W = c.image_width < 1 ? 1 : c.image_width, H = c.image_height < 1 ? 1 : c.image_height;
    double aspect = double(W) / H;
    H3 center = h3(c.lookfrom), lookat = h3(c.lookat), vup = h3(c.vup);
    double theta = c.vfov * kPi / 180.0;
    double h = std::tan(theta / 2);
    double vh = 2 * h * c.focus_dist;
    double vw = vh * aspect;
    H3 w = unit(center - lookat);
    H3 u = unit(cross(vup, w));
    H3 v = cross(w, u);
    H3 vu = vw * u;
    H3 vv = vh * -v;
    H3 du = vu / W;
    H3 dv = vv / H;
    H3 ul = center - (c.focus_dist * w) - vu / 2 - vv / 2;
    H3 p00 = ul + 0.5 * (du + dv);
    double rad = c.focus_dist * std::tan((c.defocus_angle / 2) * kPi / 180.0);
    st3(d.center, center); st3(d.pixel00, p00); st3(d.du, du); st3(d.dv, dv);
    st3(d.disk_u, u * rad); st3(d.disk_v, v * rad);
    d.W = W; d.H = H; d.spp = c.samples_per_pixel < 1 ? 1 : c.samples_per_pixel; d.max_depth = c.max_depth;
    d.defocus = !(c.defocus_angle <= 0) ? 1 : 0;
    d.pad_ = 0;
}

// ray-independent part of get_background_color, camera.hpp:832-834, 844-858, 874-895, 914-918
void make_env(const zr_env& e, zr::DEnv& d) {
    std::memset(&d, 0, sizeof d);
    d.mode = e.mode; d.hdr_tex = e.hdr_texture; d.intensity = e.intensity;
    H3 bg = h3(e.background_color) * e.intensity;
    st3(d.solid, bg);
    d.cy = std::cos(e.hdri_rotation); d.sy = std::sin(e.hdri_rotation);
    d.cp = std::cos(e.hdri_tilt); d.sp = std::sin(e.hdri_tilt);
    d.cr = std::cos(e.hdri_roll); d.sr = std::sin(e.hdri_roll);
    H3 sun = unit(h3(e.sun_direction));
    double sh = sun.y;
    double ah = sh - 0.05;
    double sky_exposure = clampd(ah * 8.0 + 1.4, 0.0, 1.0);
    double day = clampd(ah * 10.0 + 1.1, 0.0, 1.0);
    double sunset_i = clampd(1.0 - std::fabs(ah + 0.05) * 30.0, 0.0, 1.0);
    double sunset = (ah > -0.1) ? sunset_i : 0.0;
    if (sh < 0) sunset *= (sh * 10.0 + 1.0);
    sunset = clampd(sunset, 0.0, 1.0);
    H3 zen = H3{0.01, 0.03, 0.1} * (1.0 - day) + H3{0.2, 0.5, 1.0} * day;
    H3 hor = H3{0.05, 0.02, 0.01} * (1.0 - day) + H3{0.6, 0.8, 1.0} * day;
    hor = hor * (1.0 - sunset) + H3{1.0, 0.35, 0.1} * sunset;
    st3(d.sun, sun); st3(d.horizon, hor); st3(d.zenith, zen);
    d.sky_scale = e.intensity * 1.5; d.sky_exposure = sky_exposure;
    d.sun_thr = 1.0 - (e.sun_size * 0.001);
    d.sun_on = ah > -0.1 ? 1 : 0;
    H3 scol = h3(e.sun_color) * (1.0 - sunset) + H3{1.0, 0.3, 0.1} * sunset;
    double vis = clampd(sh * 5.0 + 1.0, 0.0, 1.0);
    st3(d.sun_add, (scol * e.sun_intensity) * vis);
}

struct Plan {
    int W, H, ts, tiles_x, tiles_y, x0, y0, x1, y1, lanes;
    std::vector<int32_t> tiles;
};

int make_plan(const zr_camera& cam, const zr_region* region, Plan& p) {
    p.W = cam.image_width < 1 ? 1 : cam.image_width;
    p.H = cam.image_height < 1 ? 1 : cam.image_height;
    p.ts = 32; int mod = 1, rem = 0, skew = 0;
    p.x0 = 0; p.y0 = 0; p.x1 = p.W; p.y1 = p.H;
    if (region) {
        if (region->tile_size > 0) p.ts = region->tile_size;
        if (region->tile_mod > 1) { mod = region->tile_mod; rem = region->tile_rem; skew = region->tile_skew; }
        if (region->w > 0 && region->h > 0) { p.x0 = region->x0; p.y0 = region->y0; p.x1 = region->x0 + region->w; p.y1 = region->y0 + region->h; }
    }
    if (p.x0 < 0 || p.y0 < 0 || p.x1 > p.W || p.y1 > p.H || rem < 0 || rem >= mod || skew < 0 || p.ts > 1024)
        return fail(ZR_E_INVALID, "region outside the %dx%d frame or bad tile parameters", p.W, p.H);
    p.tiles_x = (p.W + p.ts - 1) / p.ts; p.tiles_y = (p.H + p.ts - 1) / p.ts;
    p.tiles.clear();
    for (int ty = p.y0 / p.ts; ty <= (p.y1 - 1) / p.ts; ty++)
        for (int tx = p.x0 / p.ts; tx <= (p.x1 - 1) / p.ts; tx++) {
            int t = ty * p.tiles_x + tx;
            const int part = skew > 0 ? (int)(((long long)tx + (long long)skew * ty) % mod) : t % mod;   // zr_region::tile_skew
            if (part == rem) p.tiles.push_back(t);
        }
    int spp = cam.samples_per_pixel < 1 ? 1 : cam.samples_per_pixel;
    p.lanes = 64; while (p.lanes > spp) p.lanes >>= 1;
    return ZR_OK;
}

int resolve_times(zr_ctx* c);

// spill slabs of the EXTEND traversal stack, sized for the deepest tree this context has met (never below 36 levels, the
// fixed size of round 1): one slab per resident wave and sub-pool
int ensure_stack_slabs(zr_ctx* c, const zr_scene* s) {
    const uint32_t need = std::max<uint32_t>(36u, zr::stream_overflow_levels(s->stack_demand));
    if (need <= c->st_ovf_levels && c->d_st_overflow.p) return ZR_OK;
    HIP_OK(hipDeviceSynchronize());
    c->d_st_overflow.release();
    int rc = c->d_st_overflow.alloc(ST_MAX_POOLS * zr::stream_overflow_bytes(c->st_blocks, need));
    if (rc) { c->st_ovf_levels = 0; return rc; }
    c->st_ovf_levels = need;
    return ZR_OK;
}

struct HostTimer : zr::StreamTimer {
    zr_ctx* c; hipEvent_t cur_a = nullptr; bool ok = true;
    explicit HostTimer(zr_ctx* c) : c(c) {}
    hipEvent_t get() {
        if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) { ok = false; return nullptr; }
        return e;
    }
    void begin(hipStream_t st, int) override { cur_a = get(); if (cur_a) (void)hipEventRecord(cur_a, st); }
    void end(hipStream_t st, int kind) override {
        hipEvent_t b = get();
        if (!cur_a || !b) return;
        (void)hipEventRecord(b, st);
        zr_ctx::Pending pe{}; pe.a = cur_a; pe.b = b; pe.render_id = c->render_id; pe.kind = kind;
        c->pending.push_back(pe);
        cur_a = nullptr;
    }
};

// variant 2: streaming wavefront pipeline (zr_stream.hip).  Synchronises the stream internally (the round loop
// needs the active-slot count), so zr_render_device returns with the frame complete.
// mode 0: the render; 1 / 2: beauty pass and replay pass of the reflection / refraction split (zr_stream.hip, stream_shade)
int render_stream(zr_ctx* c, const zr_scene* s, const zr::DCamera& dc, const zr::DEnv& de, uint64_t seed, const Plan& plan, int count,
                  double* d_out, hipStream_t stream, volatile const uint8_t* keep_going, int mode = 0, double* d_out2 = nullptr,
                  zr::StreamProgress* progress = nullptr) {
    // pixel list (cached per plan)
    std::vector<int32_t> key = {plan.W, plan.H, plan.ts, plan.x0, plan.y0, plan.x1, plan.y1, (int32_t)plan.tiles.size(),
                                plan.tiles.empty() ? -1 : plan.tiles.front(), plan.tiles.empty() ? -1 : plan.tiles.back(),
                                (int32_t)env_double("ZR_STREAM_BOTTOM_UP", 1)};
    if (plan.W > 65535 || plan.H > 65535) return fail(ZR_E_INVALID, "kernel variant 2 supports frames up to 65535 x 65535");
    if (key != c->pix_key || !c->d_pixels.p) {
        std::vector<uint32_t> pix;
        pix.reserve((size_t)plan.tiles.size() * plan.ts * plan.ts);
        for (int32_t t : plan.tiles) {
            int tx = (t % plan.tiles_x) * plan.ts, ty = (t / plan.tiles_x) * plan.ts;
            int xa = std::max(tx, plan.x0), xb = std::min(tx + plan.ts, plan.x1), ya = std::max(ty, plan.y0), yb = std::min(ty + plan.ts, plan.y1);
            for (int y = ya; y < yb; y++) for (int x = xa; x < xb; x++) pix.push_back((uint32_t)x | ((uint32_t)y << 16));
        }
        // Work units are handed out in pixel-list order, and when they run out the frame DRAINS: the paths still alive need up to
        // max_depth more rounds, each with fewer rays than the chip wants (10 rounds = 16 ms of a 415 ms cfg3 frame, 10 of the
        // 60 ms of a rank's 1/8 share).  The drain is as long as the paths started last, so the list runs BOTTOM-UP: the top
        // of a frame is where the sky is, and a sky sample ends in one round.  The image does not depend on the order (every
        // sample is written once and reduced in a fixed order).
        if (env_double("ZR_STREAM_BOTTOM_UP", 1) != 0) std::reverse(pix.begin(), pix.end());
        int rc = c->d_pixels.upload(pix);
        if (rc) return rc;
        c->pix_key = key;
    }
    const uint32_t n_pix = (uint32_t)c->d_pixels.n;
    if (c->pending.size() > 65536) { int rr = resolve_times(c); if (rr) return rr; }
    c->render_id++; c->last_stream = stream; c->last_counted = count != 0; c->last_rounds = 0;
    HIP_OK(hipMemsetAsync(c->d_ctr.p, 0, 48 * sizeof(unsigned long long), stream));
    if (n_pix == 0) return ZR_OK;
    const uint32_t spp = (uint32_t)dc.spp;
    const uint64_t units = (uint64_t)n_pix * spp;   // one work unit per primary sample
    if (units > 0xFFFFFFFFull) return fail(ZR_E_INVALID, "frame too large for kernel variant 2 (pixels x spp must fit 32 bits); shard it (zr_region) or set ZR_KERNEL=0");
    // slot pool: large enough to fill the chip every round, small enough that the frame takes dozens of rounds (a
    // rank that owns 1/8 of the tiles must not degenerate into one shrinking batch)
    int rc;
    // per-sample radiance first: without it this pipeline cannot run at all (the caller falls back to the pixel-group kernel)
    const size_t samples_n = (size_t)units * 3;
    if (c->d_partial.n < samples_n) {
        HIP_OK(hipStreamSynchronize(stream));
        if (c->d_partial.alloc(samples_n) != ZR_OK) return fail(ZR_E_NOMEM, "no device memory for the per-sample radiance buffer (%zu bytes)", samples_n * sizeof(double));
    }
    // A world of a handful of objects is rendered by the FUSED kernel (zr_stream.hip: fused_render): every object tested per
    // segment, the path in registers, no tree, no slot pool.  Testing all objects costs time in proportion to their number, the
    // pipeline about the same per segment whatever the scene: the switch-over is ZR_FUSED_MAX objects.  A caller that polls
    // (cancellation, lines_rendered, live preview) gets the frame in sixteen launches with the poll between them; the split passes
    // stay on the pipeline.
    if (mode == 0 && s->fused_ok && s->leaf_level <= 2 && s->leaf_objects > 0 && (double)s->leaf_objects <= env_double("ZR_FUSED_MAX", ZR_FUSED_OBJECTS) &&
        env_double("ZR_FUSED", 1) != 0) {
        if (c->fused_blocks == 0) c->fused_blocks = zr::fused_blocks();
        HostTimer ftimer(c);
        int parts = 1;
        if (keep_going || progress) HIP_OK(hipMemsetAsync(c->d_partial.p, 0, samples_n * sizeof(double), stream));   // a cancelled frame / a preview reduces what exists
        hipError_t fe = zr::fused_render_frame(s->ds, dc, de, seed, spp, n_pix, c->d_pixels.p, c->d_partial.p, c->d_ctl.p, c->fused_blocks, d_out, c->d_ctr.p, count != 0,
                                               s->leaf_level <= 1 ? 1 : 2, stream, &ftimer, s->fused, keep_going, progress, &parts);
        if (fe != hipSuccess) return fail(ZR_E_DEVICE, "fused small-scene kernel failed: %s", hipGetErrorString(fe));
        c->last_rounds = (uint64_t)(parts < 0 ? -parts : parts); c->last_path = 3;
        HIP_OK(hipStreamSynchronize(stream));
        if (parts < 0) return fail(ZR_E_CANCELLED, "render cancelled after %d of 16 parts", -parts);
        return ZR_OK;
    }
    c->last_path = 2;
    if ((rc = ensure_stack_slabs(c, s))) return rc;
    // slot pool: large enough to fill the chip every round, small enough that the frame takes dozens of rounds (a
    // rank that owns 1/8 of the tiles must not degenerate into one shrinking batch)
    const bool affine = env_double("ZR_STREAM_AFFINE", 0) != 0;   // measured: -18 % L2 requests, -14 % misses, frame time +1 % (profiles/r3_affine_ab.txt): off
    uint32_t P = 0, unit_chunk = 0, drain_slots = 0;
    size_t drain_at = 0;
    for (uint32_t cap = c->st_slots;; cap /= 2) {
        P = cap / 64 * 64;
        // slots = units / 8 where the lean kernels run as two sub-pools (cfg2: 93.2 ms at 8, 98.9 at 4, 115 at 2), units / 3 for one pool of the general builds, whose
        // launches are worth making larger (demo: 128.1 ms at 8, 126.0 at 4, 125.1 at 3: profiles/r4_experiments_ab.txt)
        const bool lean_two_pools = s->leaf_level == 0 && s->ds.shade_lean != 0 && mode == 0;
        uint64_t want = std::max<uint64_t>(units / (uint64_t)std::max(1.0, env_double("ZR_STREAM_UNITS_PER_SLOT", lean_two_pools ? 8 : 3)), 1u << 20);
        want = want / 64 * 64;
        if (want < P) P = (uint32_t)want;
        if (units < P) P = (uint32_t)((units + 63) / 64 * 64);
        // XCD-affine hand-out of the work units (zr_stream.hip: st_unit_of): chunks of `unit_chunk` units — by default the samples of
        // 1024 consecutive pixels of the tile-ordered list, i.e. one 32 x 32 tile — belong to one shard, hence to one XCD's L2
        unit_chunk = 0;
        if (affine) {
            const uint32_t round = 64u * 256u;   // a pool is whole rounds of ST_SHARDS SHADE blocks
            P = std::max<uint32_t>(round, P / round * round);
            if (units < P) P = (uint32_t)((units + round - 1) / round * round);
            uint64_t G = (uint64_t)std::max(1.0, env_double("ZR_STREAM_CHUNK_PX", 1024)) * spp;
            G = std::min<uint64_t>(G, units / (64u * 8u));   // every shard gets at least eight chunks (small frames: smaller chunks)
            unit_chunk = (uint32_t)std::max<uint64_t>(256, std::min<uint64_t>(G, 1u << 30));
        }
        // the slot pool, its sub-pools' rounding, and behind them the small pool the survivors of a frame's drain are moved to
        // (zr_stream.hip: stream_compact).  Sized for the P this frame uses and only ever grown: a 64 x 64 test frame or a one-ray
        // device_hit() does not reserve the 13.7 GB a 1080p frame at 512 spp wants ("Device memory")
        drain_slots = P / 16 / 256 * 256 + 256;
        drain_at = zr::stream_pool_bytes(P) + 65536 * ST_MAX_POOLS;
        const size_t pool_need = drain_at + zr::stream_pool_bytes(drain_slots);
        if (c->d_pool.n >= pool_need) break;
        HIP_OK(hipStreamSynchronize(stream));
        if (c->d_pool.alloc(pool_need) == ZR_OK) break;
        // a pool that cannot be had is retried at half the size: the frame takes more rounds, the image is the same
        if (cap <= (1u << 20)) return fail(ZR_E_NOMEM, "no device memory for a slot pool of %u paths (%zu bytes)", P, pool_need);
        std::fprintf(stderr, "[zr] no device memory for a pool of %u path slots (%zu bytes): retrying with half\n", P, pool_need);
    }
    const bool use_drain = env_double("ZR_STREAM_DRAIN_POOL", 1) != 0;
    if (keep_going || progress) HIP_OK(hipMemsetAsync(c->d_partial.p, 0, samples_n * sizeof(double), stream));  // a cancelled frame / a preview reduces what exists
    if (mode != 0) {
        if (c->d_kend.n < units * 2) { if ((rc = c->d_kend.alloc(units * 2))) return rc; }
        if (c->d_cls.n < units) { if ((rc = c->d_cls.alloc(units))) return rc; }
        if (mode == 2) HIP_OK(hipMemsetAsync(c->d_cls.p, 0, units, stream));
        const size_t cp = ((size_t)c->st_slots / 256 + ST_MAX_POOLS + 1) * 4;
        if (c->d_cpart.n < cp) { if ((rc = c->d_cpart.alloc(cp))) return rc; }
    }
    HostTimer timer(c);
    int rounds = 0;
    hipStream_t streams[ST_MAX_POOLS];
    streams[0] = stream;
    for (int k = 1; k < ST_MAX_POOLS; k++) streams[k] = c->sub[k];
    const bool sharded = (size_t)plan.tiles.size() < (size_t)plan.tiles_x * plan.tiles_y;
    // Two sub-pools, a fraction of a round apart on two streams, let one pool's SHADE run beside the other's EXTEND.  Until round 3 that paid on a rank's share only
    // (the whole frame: 366.6 against 365.7 ms): SHADE needed 124 registers and found no room beside EXTEND's waves.  The lean builds of both kernels use 80
    // (zr_stream.hip), a SIMD holds three waves of each, and a whole cfg3 frame gains 3.5 % with 64 Mi slots, 5.4 % with 128 Mi (profiles/r4_experiments_ab.txt); the
    // general builds (demo: 128 + 117 registers) do not fit beside each other and lose 2 %: one pool for those
    const bool lean_pair = s->leaf_level == 0 && s->ds.shade_lean != 0 && mode == 0;
    const int pools = c->st_pools > 0 ? c->st_pools : ((sharded || lean_pair) ? 2 : 1);
    hipError_t e = zr::stream_render(s->ds, dc, de, seed, c->d_pool.p, P, spp, n_pix, c->d_pixels.p, c->d_partial.p, c->d_ctl.p,
                                     c->d_st_overflow.p, c->st_ovf_levels, c->st_blocks, d_out, c->d_ctr.p, count != 0, streams, pools, c->st_event, &timer, c->h_active,
                                     keep_going, &rounds, s->leaf_level, mode, mode ? (void*)c->d_kend.p : nullptr, mode ? (void*)c->d_cls.p : nullptr, d_out2, mode ? c->d_cpart.p : nullptr, progress,
                                     use_drain ? (void*)((unsigned char*)c->d_pool.p + drain_at) : nullptr, drain_slots, unit_chunk);
    if (e != hipSuccess) return fail(ZR_E_DEVICE, "streaming pipeline failed: %s", hipGetErrorString(e));
    c->last_rounds = (uint64_t)(rounds < 0 ? -rounds : rounds);
    HIP_OK(hipStreamSynchronize(stream));
    if (rounds < 0) return fail(ZR_E_CANCELLED, "render cancelled after %d rounds", -rounds);
    return ZR_OK;
}

int enqueue_render(zr_ctx* c, const zr_scene* s, const zr_camera* cam, const zr_env* env, uint64_t seed, const Plan& plan, int count,
                   double* d_out, hipStream_t stream, volatile const uint8_t* keep_going, volatile int* rows_done, zr::StreamProgress* progress = nullptr) {
    c->last_rounds = 0;
    zr::DCamera dc; make_camera(*cam, dc);
    zr::DEnv de; make_env(*env, de);
    if (de.mode > ZR_ENV_SOLID_COLOR) return fail(ZR_E_INVALID, "unknown environment mode %u", de.mode);
    if (de.mode == ZR_ENV_HDR_MAP && de.hdr_tex != ZR_NO_TEXTURE) {
        if (de.hdr_tex >= s->textures.size()) return fail(ZR_E_INVALID, "environment texture id out of range");
    }
    std::vector<int32_t> tiles = plan.tiles;
    int rc = c->d_tiles.upload(tiles);
    if (rc) return rc;
    HIP_OK(hipMemsetAsync(c->d_ctr.p, 0, 48 * sizeof(unsigned long long), stream));
    // the streaming pipeline packs bounce counters into 8 bits, work units into 32 bits and leaf references into 24 + 4
    // bits; frames or scenes beyond that are rendered by the pixel-group megakernel below (slower, same results)
    uint64_t stream_units = 0;
    for (int32_t t : plan.tiles) {
        int tx = (t % plan.tiles_x) * plan.ts, ty = (t / plan.tiles_x) * plan.ts;
        int xa = std::max(tx, plan.x0), xb = std::min(tx + plan.ts, plan.x1), ya = std::max(ty, plan.y0), yb = std::min(ty + plan.ts, plan.y1);
        if (xb > xa && yb > ya) stream_units += (uint64_t)(xb - xa) * (yb - ya) * (uint64_t)dc.spp;
    }
    if (c->variant == 2 && dc.max_depth <= 250 && s->quad_ok && stream_units <= 0xFFFFFFFFull && plan.W <= 65535 && plan.H <= 65535) {
        int r2 = render_stream(c, s, dc, de, seed, plan, count, d_out, stream, keep_going, 0, nullptr, progress);
        if (rows_done && r2 == ZR_OK) *rows_done = plan.H;
        if (r2 != ZR_E_NOMEM) return r2;
        // the pipeline's buffers (24 bytes per primary sample + the slot pool) do not fit beside what else lives on the device: the
        // pixel-group kernel below needs neither
        std::fprintf(stderr, "[zr] %s: rendering this frame with the pixel-group kernel (same results, slower)\n", zr_host::last_error());
        HIP_OK(hipMemsetAsync(c->d_ctr.p, 0, 48 * sizeof(unsigned long long), stream));
    } else if (c->variant == 2 && !c->warned_fallback) {   // said once per context: the frame is rendered, by the slower kernel
        c->warned_fallback = true;
        std::fprintf(stderr, "[zr] frame outside the streaming pipeline's packing limits (max_depth %d > 250, %llu work units > 2^32, %d x %d px > 65535, "
                             "or a scene with more than 2^24 primitives of a kind): rendered by the pixel-group kernel — same results, about six times slower\n",
                     dc.max_depth, (unsigned long long)stream_units, plan.W, plan.H);
    }
    c->last_path = 0;
    // one launch per frame unless the caller wants progress / cancellation, which need batch boundaries
    const bool interactive = keep_going || rows_done;
    const int batch = std::max(1, (int)env_double("ZR_BATCH_TILES", interactive ? 256 : (double)(1 << 30)));
    size_t n_batches = (tiles.size() + batch - 1) / batch;
    if (c->pending.size() > 4096) { int rr = resolve_times(c); if (rr) return rr; }
    c->render_id++; c->last_stream = stream; c->last_counted = count != 0;
    auto get_event = [&](hipEvent_t& e) -> int {
        if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); return ZR_OK; }
        HIP_OK(hipEventCreate(&e));
        return ZR_OK;
    };
    for (size_t b = 0; b < n_batches; b++) {
        if (keep_going && *keep_going == 0) {
            HIP_OK(hipStreamSynchronize(stream));
            return fail(ZR_E_CANCELLED, "render cancelled after %zu of %zu batches", b, n_batches);
        }
        zr::WorkDesc wd;
        wd.tiles = c->d_tiles.p + b * batch;
        wd.n_tiles = (int32_t)std::min<size_t>(batch, tiles.size() - b * batch);
        wd.tile_size = plan.ts; wd.tiles_x = plan.tiles_x;
        wd.x0 = plan.x0; wd.y0 = plan.y0; wd.x1 = plan.x1; wd.y1 = plan.y1;
        wd.lanes_per_pixel = plan.lanes;
        zr_ctx::Pending pe{}; pe.render_id = c->render_id; pe.kind = 1;
        if ((rc = get_event(pe.a)) || (rc = get_event(pe.b))) return rc;
        HIP_OK(hipEventRecord(pe.a, stream));
        HIP_OK(zr::launch_render(s->ds, dc, de, seed, wd, d_out, c->d_ctr.p, count != 0, stream));
        HIP_OK(hipEventRecord(pe.b, stream));
        c->pending.push_back(pe);
        if (keep_going || rows_done) {
            // progress / cancellation need the batch to have finished (camera.hpp:441,548-552)
            HIP_OK(hipStreamSynchronize(stream));
            if (rows_done) {
                int last_tile = tiles[std::min(tiles.size(), (b + 1) * (size_t)batch) - 1];
                int rows = std::min(plan.H, (last_tile / plan.tiles_x) * plan.ts);
                if (rows > *rows_done) *rows_done = rows;
            }
        }
    }
    return ZR_OK;
}

int resolve_times(zr_ctx* c) {
    if (c->pending.empty()) return ZR_OK;
    bool fresh = false;
    for (auto& p : c->pending) {
        HIP_OK(hipEventSynchronize(p.b));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, p.a, p.b));
        if (p.kind == c->log_kind) c->log.push_back(ms);  // default: the dominant kernel's launches (render_* / stream_extend)
        if (p.render_id == c->render_id) {
            if (!fresh) { c->last_render_ms = 0; c->last_extend_ms = 0; c->last_shade_ms = 0; fresh = true; }
            c->last_render_ms += ms;
            if (p.kind == 1) c->last_extend_ms += ms;
            if (p.kind == 2) c->last_shade_ms += ms;
        }
        c->pool.push_back(p.a); c->pool.push_back(p.b);
    }
    c->pending.clear();
    if (c->log.size() > (1u << 20)) c->log.erase(c->log.begin(), c->log.begin() + (c->log.size() - (1u << 20)));
    return ZR_OK;
}

}  // namespace

extern "C" {

int zr_render_device(zr_ctx* c, const zr_scene* s, const zr_camera* cam, const zr_env* env, uint64_t seed, const zr_region* region,
                     int collect_counters, void* d_out_rgb, void* hip_stream) {
    if (!c || !s || !cam || !env || !d_out_rgb) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_render");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    HIP_OK(hipSetDevice(c->device));
    Plan plan;
    int rc = make_plan(*cam, region, plan);
    if (rc) return rc;
    // default stream requested: use the legacy null stream so that callers' stream-ordered work (torch) sees it
    hipStream_t st = (hipStream_t)hip_stream;
    return enqueue_render(c, s, cam, env, seed, plan, collect_counters, (double*)d_out_rgb, st, nullptr, nullptr);
}

int zr_render(zr_ctx* c, const zr_scene* s, const zr_camera* cam, const zr_env* env, uint64_t seed, const zr_region* region,
              int collect_counters, double* out_rgb, volatile const uint8_t* keep_going, volatile int* rows_done) {
    if (!c || !s || !cam || !env || !out_rgb) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_render");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    HIP_OK(hipSetDevice(c->device));
    Plan plan;
    int rc = make_plan(*cam, region, plan);
    if (rc) return rc;
    const size_t npx = (size_t)plan.W * plan.H;
    if ((rc = c->d_out.alloc(npx * 3))) return rc;
    HIP_OK(hipMemsetAsync(c->d_out.p, 0, npx * 3 * sizeof(double), c->stream));
    if (rows_done) *rows_done = 0;
    // a whole frame goes straight into the caller's buffer; a region through a staging copy (only its pixels may be touched)
    const bool whole = (size_t)plan.tiles.size() == (size_t)plan.tiles_x * plan.tiles_y && plan.x0 == 0 && plan.y0 == 0 && plan.x1 == plan.W && plan.y1 == plan.H;
    std::vector<double> frame(whole ? 0 : npx * 3);
    auto copy_out = [&]() -> int {   // the region's pixels of the device frame -> the caller's buffer
        if (whole) { HIP_OK(hipMemcpy(out_rgb, c->d_out.p, npx * 3 * sizeof(double), hipMemcpyDeviceToHost)); return ZR_OK; }
        HIP_OK(hipMemcpy(frame.data(), c->d_out.p, frame.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int32_t t : plan.tiles) {
            int tx = (t % plan.tiles_x) * plan.ts, ty = (t / plan.tiles_x) * plan.ts;
            int xa = std::max(tx, plan.x0), xb = std::min(tx + plan.ts, plan.x1), ya = std::max(ty, plan.y0), yb = std::min(ty + plan.ts, plan.y1);
            for (int y = ya; y < yb; y++)
                if (xb > xa) std::memcpy(out_rgb + ((size_t)y * plan.W + xa) * 3, frame.data() + ((size_t)y * plan.W + xa) * 3, (size_t)(xb - xa) * 3 * sizeof(double));
        }
        return ZR_OK;
    };
    // Progress as the reference's callers see it: lines_rendered advances while the frame renders (camera.hpp:548-552) and the
    // GUI reads render_accumulator mid-render (main.cpp:1576).  The pipeline finishes samples all over the frame rather than
    // row by row, so `rows_done` = H x the finished fraction of the samples (H only at the very end), and a few times per second
    // out_rgb receives the mean of the samples finished so far (every pixel brightens towards its final value).
    struct Preview : zr::StreamProgress {
        volatile int* rows; int H; double last = 0; std::function<int()> copy; double period;
        static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
        bool wants_frame() override { return now() - last >= period; }
        void report(double f, bool reduced) override {
            const int r = std::min(H - 1, std::max(0, (int)(f * H)));
            if (r > *rows) *rows = r;
            if (reduced) { (void)copy(); last = now(); }
        }
    } preview;
    preview.rows = rows_done; preview.H = plan.H; preview.copy = copy_out; preview.period = env_double("ZR_PREVIEW_PERIOD_S", 0.2); preview.last = Preview::now();
    int rrc = enqueue_render(c, s, cam, env, seed, plan, collect_counters, c->d_out.p, c->stream, keep_going, rows_done, rows_done ? &preview : nullptr);
    if (rrc != ZR_OK && rrc != ZR_E_CANCELLED) return rrc;
    std::string cancel_msg = zr_host::last_error();
    HIP_OK(hipStreamSynchronize(c->stream));
    if ((rc = copy_out())) return rc;
    if (rrc == ZR_E_CANCELLED) return fail(rrc, "%s", cancel_msg.c_str());
    if (rows_done) *rows_done = plan.H;  // camera.hpp:576-578
    return ZR_OK;
}

int zr_render_aov(zr_ctx* c, const zr_scene* s, const zr_camera* cam, uint64_t seed, const zr_region* region, const zr_aov_params* ap,
                  double* out_albedo, double* out_normal, double* out_zdepth) {
    if (!c || !s || !cam || !ap) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_render_aov");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    if (!out_albedo && !out_normal && !out_zdepth) return ZR_OK;
    HIP_OK(hipSetDevice(c->device));
    Plan plan;
    int rc = make_plan(*cam, region, plan);
    if (rc) return rc;
    zr::DCamera dc; make_camera(*cam, dc);
    // camera basis u, v, w exactly as camera::initialize builds it (camera.hpp:380-382)
    H3 w = unit(h3(cam->lookfrom) - h3(cam->lookat));
    H3 u = unit(cross(h3(cam->vup), w));
    H3 v = cross(w, u);
    double uvw[9] = {u.x, u.y, u.z, v.x, v.y, v.z, w.x, w.y, w.z};
    const int spp = dc.spp;
    const int aux_sample = std::min(std::max(spp / 8, 64), 1024);   // std::clamp(spp / 8, 64, 1024), camera.hpp:433
    const int aux = std::min(aux_sample, spp);                      // camera.hpp:535
    const size_t npx = (size_t)plan.W * plan.H;
    DevBuf<double> d_a, d_n, d_z;
    if (out_albedo) { if ((rc = d_a.alloc(npx * 3))) return rc; HIP_OK(hipMemsetAsync(d_a.p, 0, npx * 24, c->stream)); }
    if (out_normal) { if ((rc = d_n.alloc(npx * 3))) return rc; HIP_OK(hipMemsetAsync(d_n.p, 0, npx * 24, c->stream)); }
    if (out_zdepth) { if ((rc = d_z.alloc(npx * 3))) return rc; HIP_OK(hipMemsetAsync(d_z.p, 0, npx * 24, c->stream)); }
    std::vector<int32_t> tiles = plan.tiles;
    if ((rc = c->d_tiles.upload(tiles))) return rc;
    zr::WorkDesc wd;
    wd.tiles = c->d_tiles.p; wd.n_tiles = (int32_t)tiles.size(); wd.tile_size = plan.ts; wd.tiles_x = plan.tiles_x;
    wd.x0 = plan.x0; wd.y0 = plan.y0; wd.x1 = plan.x1; wd.y1 = plan.y1;
    wd.lanes_per_pixel = 64; while (wd.lanes_per_pixel > aux) wd.lanes_per_pixel >>= 1;
    HIP_OK(zr::launch_aov(s->ds, dc, seed, wd, aux, ap->z_depth_max_dist, uvw, d_a.p, d_n.p, d_z.p, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    std::vector<double> frame(npx * 3);
    auto copy_out = [&](DevBuf<double>& d, double* out) -> int {
        if (!out) return ZR_OK;
        HIP_OK(hipMemcpy(frame.data(), d.p, frame.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int32_t t : plan.tiles) {
            int tx = (t % plan.tiles_x) * plan.ts, ty = (t / plan.tiles_x) * plan.ts;
            int xa = std::max(tx, plan.x0), xb = std::min(tx + plan.ts, plan.x1), ya = std::max(ty, plan.y0), yb = std::min(ty + plan.ts, plan.y1);
            for (int y = ya; y < yb; y++)
                if (xb > xa) std::memcpy(out + ((size_t)y * plan.W + xa) * 3, frame.data() + ((size_t)y * plan.W + xa) * 3, (size_t)(xb - xa) * 3 * sizeof(double));
        }
        return ZR_OK;
    };
    if ((rc = copy_out(d_a, out_albedo)) || (rc = copy_out(d_n, out_normal)) || (rc = copy_out(d_z, out_zdepth))) return rc;
    return ZR_OK;
}

int zr_render_passes(zr_ctx* c, const zr_scene* s, const zr_camera* cam, const zr_env* env, uint64_t seed, const zr_region* region,
                     double* out_beauty, double* out_reflection, double* out_refraction) {
    if (!c || !s || !cam || !env) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_render_passes");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    if (!out_beauty && !out_reflection && !out_refraction) return ZR_OK;
    HIP_OK(hipSetDevice(c->device));
    Plan plan;
    int rc = make_plan(*cam, region, plan);
    if (rc) return rc;
    zr::DCamera dc; make_camera(*cam, dc);
    zr::DEnv de; make_env(*env, de);
    if (de.mode > ZR_ENV_SOLID_COLOR) return fail(ZR_E_INVALID, "unknown environment mode %u", de.mode);
    if (de.mode == ZR_ENV_HDR_MAP && de.hdr_tex != ZR_NO_TEXTURE && de.hdr_tex >= s->textures.size()) return fail(ZR_E_INVALID, "environment texture id out of range");
    const size_t npx = (size_t)plan.W * plan.H;
    DevBuf<double> d_b, d_r, d_f;
    if (out_beauty) { if ((rc = d_b.alloc(npx * 3))) return rc; HIP_OK(hipMemsetAsync(d_b.p, 0, npx * 24, c->stream)); }
    if (out_reflection) { if ((rc = d_r.alloc(npx * 3))) return rc; HIP_OK(hipMemsetAsync(d_r.p, 0, npx * 24, c->stream)); }
    if (out_refraction) { if ((rc = d_f.alloc(npx * 3))) return rc; HIP_OK(hipMemsetAsync(d_f.p, 0, npx * 24, c->stream)); }
    std::vector<int32_t> tiles = plan.tiles;
    if ((rc = c->d_tiles.upload(tiles))) return rc;
    zr::WorkDesc wd;
    wd.tiles = c->d_tiles.p; wd.n_tiles = (int32_t)tiles.size(); wd.tile_size = plan.ts; wd.tiles_x = plan.tiles_x;
    wd.x0 = plan.x0; wd.y0 = plan.y0; wd.x1 = plan.x1; wd.y1 = plan.y1;
    wd.lanes_per_pixel = 64; while (wd.lanes_per_pixel > dc.spp) wd.lanes_per_pixel >>= 1;
    const uint64_t stream_units = (uint64_t)plan.tiles.size() * plan.ts * plan.ts * (uint64_t)dc.spp;
    const bool streaming = c->variant == 2 && s->quad_ok && 2 * dc.max_depth <= 250 && stream_units <= 0xFFFFFFFFull && plan.W <= 65535 &&
                           plan.H <= 65535 && env_double("ZR_PASSES_STREAM", 1) != 0;
    if (streaming) {
        // two runs of the streaming pipeline: the beauty pass records where every sample's stream stopped, the replay pass traces
        // the camera ray again and runs the second path from there (stream_shade MODE 1 / 2)
        unsigned long long ha[16], hb[16];
        if ((rc = render_stream(c, s, dc, de, seed, plan, 0, d_b.p, c->stream, nullptr, 1))) return rc;
        HIP_OK(hipMemcpy(ha, c->d_ctr.p, sizeof ha, hipMemcpyDeviceToHost));
        if ((rc = render_stream(c, s, dc, de, seed, plan, 0, d_r.p, c->stream, nullptr, 2, d_f.p))) return rc;
        HIP_OK(hipMemcpy(hb, c->d_ctr.p, sizeof hb, hipMemcpyDeviceToHost));
        // counted by SHADE in both passes (EXTEND runs uninstrumented): samples, segments, hits, draws
        unsigned long long h[16] = {0};
        h[0] = (unsigned long long)c->d_pixels.n * (unsigned long long)dc.spp;   // every sample of the region, once
        h[1] = ha[1] + hb[1]; h[7] = ha[7] + hb[7]; h[8] = ha[8] + hb[8];
        HIP_OK(hipMemcpy(c->d_ctr.p, h, sizeof h, hipMemcpyHostToDevice));
        c->last_counted = true;
    } else {
        c->render_id++; c->last_counted = true; c->last_rounds = 0;
        HIP_OK(hipMemsetAsync(c->d_ctr.p, 0, 48 * sizeof(unsigned long long), c->stream));
        HIP_OK(zr::launch_passes(s->ds, dc, de, seed, wd, d_b.p, d_r.p, d_f.p, c->d_ctr.p, c->stream));
    }
    HIP_OK(hipStreamSynchronize(c->stream));
    std::vector<double> frame(npx * 3);
    auto copy_out = [&](DevBuf<double>& d, double* out) -> int {
        if (!out) return ZR_OK;
        HIP_OK(hipMemcpy(frame.data(), d.p, frame.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int32_t t : plan.tiles) {
            int tx = (t % plan.tiles_x) * plan.ts, ty = (t / plan.tiles_x) * plan.ts;
            int xa = std::max(tx, plan.x0), xb = std::min(tx + plan.ts, plan.x1), ya = std::max(ty, plan.y0), yb = std::min(ty + plan.ts, plan.y1);
            for (int y = ya; y < yb; y++)
                if (xb > xa) std::memcpy(out + ((size_t)y * plan.W + xa) * 3, frame.data() + ((size_t)y * plan.W + xa) * 3, (size_t)(xb - xa) * 3 * sizeof(double));
        }
        return ZR_OK;
    };
    if ((rc = copy_out(d_b, out_beauty)) || (rc = copy_out(d_r, out_reflection)) || (rc = copy_out(d_f, out_refraction))) return rc;
    return ZR_OK;
}

int zr_post_process(zr_ctx* c, const zr_post_params* pp, const double* frame, int W, int H, int is_data_pass, int apply_gamma, uint8_t* out) {
    if (!c || !pp || !frame || !out) return fail(ZR_E_INVALID, "null argument");
    if (W < 2 || H < 2 || (size_t)W * H > (1ull << 31)) return fail(ZR_E_INVALID, "frame size %d x %d not supported", W, H);
    if (pp->use_bloom && (pp->bloom_radius < 0 || pp->bloom_radius > 4096)) return fail(ZR_E_INVALID, "bloom radius out of range");
    HIP_OK(hipSetDevice(c->device));
    const size_t n = (size_t)W * H;
    DevBuf<double> d_frame, t0, t1, t2; DevBuf<uint8_t> d_out;
    int rc;
    if ((rc = d_frame.alloc(n * 3)) || (rc = d_out.alloc(n * 3))) return rc;
    const bool bloom = !is_data_pass && pp->use_bloom, sharpen = !is_data_pass && pp->use_sharpening;
    if (bloom && ((rc = t0.alloc(n * 3)) || (rc = t1.alloc(n * 3)))) return rc;
    if (sharpen && (rc = t2.alloc(n * 3))) return rc;
    HIP_OK(hipMemcpyAsync(d_frame.p, frame, n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const double ev = std::pow(2.0, (double)pp->exposure);   // camera.hpp:711
    HIP_OK(zr::launch_post(d_frame.p, W, H, *pp, is_data_pass, apply_gamma, ev, t0.p, t1.p, t2.p, d_out.p, c->stream));
    HIP_OK(hipMemcpyAsync(out, d_out.p, n * 3, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return ZR_OK;
}

int zr_analyze_frame(zr_ctx* c, const double* frame, size_t n, zr_image_stats* out) {
    if (!c || !frame || !out) return fail(ZR_E_INVALID, "null argument");
    if (n == 0 || n > (1ull << 31)) return fail(ZR_E_INVALID, "pixel count not supported");
    HIP_OK(hipSetDevice(c->device));
    const size_t blocks = (n + 255) / 256;
    DevBuf<double> d_frame, d_log; DevBuf<float> d_max; DevBuf<int> d_hist;
    int rc;
    if ((rc = d_frame.alloc(n * 3)) || (rc = d_log.alloc(blocks)) || (rc = d_max.alloc(blocks)) || (rc = d_hist.alloc(256))) return rc;
    HIP_OK(hipMemcpyAsync(d_frame.p, frame, n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_OK(zr::launch_analyze(d_frame.p, n, d_log.p, d_max.p, d_hist.p, c->stream));
    std::vector<double> plog(blocks); std::vector<float> pmax(blocks);
    HIP_OK(hipMemcpyAsync(plog.data(), d_log.p, blocks * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipMemcpyAsync(pmax.data(), d_max.p, blocks * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipMemcpyAsync(out->histogram, d_hist.p, 256 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    double total = 0.0; float mx = 0.0f;
    for (size_t b = 0; b < blocks; b++) { total += plog[b]; if (pmax[b] > mx) mx = pmax[b]; }
    out->max_luminance = mx;
    out->average_luminance = std::pow(2.0f, static_cast<float>(total / (double)n));   // color_processing.hpp:180
    return ZR_OK;
}

int zr_trace_paths(zr_ctx* c, const zr_scene* s, const zr_camera* cam, uint64_t seed, const int32_t* requests, int n, int max_segments, double* out) {
    if (!c || !s || !cam || (n > 0 && (!requests || !out))) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_trace_paths");
    if (n <= 0 || max_segments <= 0) return ZR_OK;
    static_assert(ZR_PATH_RECORD == ZR_PATH_REC, "record size");
    HIP_OK(hipSetDevice(c->device));
    zr::DCamera dc; make_camera(*cam, dc);
    for (int k = 0; k < n; k++)
        if (requests[3 * k] < 0 || requests[3 * k] >= dc.W || requests[3 * k + 1] < 0 || requests[3 * k + 1] >= dc.H || requests[3 * k + 2] < 0)
            return fail(ZR_E_INVALID, "path request %d outside the frame", k);
    DevBuf<int32_t> d_req; DevBuf<double> d_out;
    std::vector<int32_t> r(requests, requests + (size_t)n * 3);
    int rc;
    if ((rc = d_req.upload(r))) return rc;
    const size_t words = (size_t)n * max_segments * ZR_PATH_RECORD;
    if ((rc = d_out.alloc(words))) return rc;
    HIP_OK(zr::launch_path_records(s->ds, dc, seed, d_req.p, n, max_segments, d_out.p, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(out, d_out.p, words * sizeof(double), hipMemcpyDeviceToHost));
    return ZR_OK;
}

int zr_get_counters(zr_ctx* c, zr_counters* out) {
    if (!c || !out) return fail(ZR_E_INVALID, "null argument");
    HIP_OK(hipSetDevice(c->device));
    int rc = resolve_times(c);
    if (rc) return rc;
    std::memset(out, 0, sizeof *out);
    out->kernel_ms = c->last_render_ms;
    out->extend_ms = c->last_extend_ms; out->shade_ms = c->last_shade_ms; out->rounds = c->last_rounds; out->path = (uint64_t)c->last_path;
    unsigned long long h[16];
    HIP_OK(hipMemcpy(h, c->d_ctr.p, sizeof h, hipMemcpyDeviceToHost));
    if (h[15] != 0) return fail(ZR_E_DEVICE, "render kernel hit its iteration cap on %llu task(s): results are incomplete", h[15]);
    if (c->last_counted || env_double("ZR_RAW_COUNTERS", 0) != 0) {
        out->primary_samples = h[0]; out->segments = h[1]; out->nodes_tested = h[2]; out->spheres_tested = h[3];
        out->triangles_tested = h[4]; out->cubes_tested = h[5]; out->media_tested = h[6]; out->hits = h[7]; out->rng_draws = h[8];
        out->node_execs = h[9]; out->node_lanes = h[10]; out->leaf_execs = h[11]; out->leaf_lanes = h[12]; out->shade_execs = h[13]; out->shade_lanes = h[14];
    }
    if (std::getenv("ZR_LANE_HISTOGRAM")) {   // development aid (a -DZR_WAVE_PROFILE build fills them): EXTEND's iterations per phase by ready lanes, 8 buckets of 8 lanes
        unsigned long long hh[24];
        HIP_OK(hipMemcpy(hh, c->d_ctr.p + 16, sizeof hh, hipMemcpyDeviceToHost));
        const char* names[3] = {"NODE", "LEAF", "FETCH"};
        for (int ph = 0; ph < 3; ph++) {
            unsigned long long tot = 0;
            for (int b = 0; b < 8; b++) tot += hh[ph * 8 + b];
            std::fprintf(stderr, "[zr] %-5s iterations by ready lanes (1-8 ... 57-64):", names[ph]);
            for (int b = 0; b < 8; b++) std::fprintf(stderr, " %5.1f%%", tot ? 100.0 * (double)hh[ph * 8 + b] / (double)tot : 0.0);
            std::fprintf(stderr, "   of %llu\n", tot);
        }
    }
    return ZR_OK;
}

int zr_get_kernel_times(zr_ctx* c, float* ms, int cap) {
    if (!c) return fail(ZR_E_INVALID, "null argument");
    HIP_OK(hipSetDevice(c->device));
    int rc = resolve_times(c);
    if (rc) return rc;
    int total = (int)c->log.size();
    int n = std::min(total, std::max(cap, 0));
    for (int k = 0; k < n; k++) ms[k] = c->log[c->log.size() - n + k];
    c->log.clear();
    return total;
}

int zr_trace(zr_ctx* c, const zr_scene* s, const double* rays6, size_t n, double tmin, double tmax, uint64_t seed, uint64_t pixel,
             uint32_t bounce, zr_hit* out) {
    if (!c || !s || (n && (!rays6 || !out))) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_trace");
    HIP_OK(hipSetDevice(c->device));
    DevBuf<double> d_rays; DevBuf<zr_hit> d_hits;
    std::vector<double> r(rays6, rays6 + n * 6);
    int rc;
    if ((rc = d_rays.upload(r))) return rc;
    if ((rc = d_hits.alloc(n))) return rc;
    // Two engines answer the same question: the pair-BVH walk of variants 0/1 and — for the render interval
    // [0.001, inf) — the EXTEND kernel of the streaming pipeline.  ZR_TRACE_ENGINE=pairs|extend picks one (tests run
    // both); by default the engine of the active render variant is used.
    const char* eng = std::getenv("ZR_TRACE_ENGINE");
    const bool can_extend = c->variant == 2 && s->quad_ok && tmin == 0.001 && tmax == HUGE_VAL && n < (1u << 30);
    if (eng && std::strcmp(eng, "extend") == 0 && !can_extend)
        return fail(ZR_E_INVALID, "ZR_TRACE_ENGINE=extend needs ZR_KERNEL=2, tmin = 0.001, tmax = inf and a scene within the 4-wide tree's limits");
    if (can_extend && !(eng && std::strcmp(eng, "pairs") == 0)) {
        DevBuf<unsigned char> pool;
        if ((rc = ensure_stack_slabs(c, s))) return rc;
        if ((rc = pool.alloc(zr::stream_pool_bytes((uint32_t)n) + 65536))) return rc;
        HIP_OK(hipMemsetAsync(c->d_ctr.p, 0, 48 * sizeof(unsigned long long), c->stream));
        HIP_OK(zr::stream_trace(s->ds, d_rays.p, (uint32_t)n, seed, pixel, bounce, d_hits.p, pool.p, c->d_ctl.p, c->d_st_overflow.p, c->st_ovf_levels, c->st_blocks,
                                c->d_ctr.p, s->leaf_level, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        unsigned int capped = 0;
        HIP_OK(hipMemcpy(&capped, c->d_ctl.p + 2, sizeof capped, hipMemcpyDeviceToHost));
        if (capped) return fail(ZR_E_DEVICE, "EXTEND hit its iteration cap on %u wave(s)", capped);
    } else {
        HIP_OK(zr::launch_trace(s->ds, d_rays.p, n, tmin, tmax, seed, pixel, bounce, d_hits.p, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    if (n) HIP_OK(hipMemcpy(out, d_hits.p, n * sizeof(zr_hit), hipMemcpyDeviceToHost));
    return ZR_OK;
}

int zr_kat_scatter(zr_ctx* c, const zr_scene* s, const double* rays6, const zr_hit* recs, const uint64_t* keys, const uint64_t* first_draw,
                   size_t n, zr_scatter_out* out) {
    if (!c || !s || (n && (!rays6 || !recs || !keys || !out))) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_kat_scatter");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    if (n == 0) return ZR_OK;
    HIP_OK(hipSetDevice(c->device));
    DevBuf<double> d_rays; DevBuf<zr_hit> d_recs; DevBuf<uint64_t> d_keys, d_first; DevBuf<zr_scatter_out> d_out;
    int rc;
    if ((rc = d_rays.upload(std::vector<double>(rays6, rays6 + n * 6))) || (rc = d_recs.upload(std::vector<zr_hit>(recs, recs + n))) ||
        (rc = d_keys.upload(std::vector<uint64_t>(keys, keys + n))) || (rc = d_out.alloc(n))) return rc;
    if (first_draw && (rc = d_first.upload(std::vector<uint64_t>(first_draw, first_draw + n)))) return rc;
    HIP_OK(zr::launch_kat_scatter(s->ds, d_rays.p, d_recs.p, d_keys.p, first_draw ? d_first.p : nullptr, n, d_out.p, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(out, d_out.p, n * sizeof(zr_scatter_out), hipMemcpyDeviceToHost));
    return ZR_OK;
}

int zr_kat_texture(zr_ctx* c, const zr_scene* s, uint32_t texture_id, const double* uvp5, size_t n, double* out_rgb) {
    if (!c || !s || (n && (!uvp5 || !out_rgb))) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_kat_texture");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    if (texture_id >= s->textures.size()) return fail(ZR_E_INVALID, "texture id %u out of range", texture_id);
    if (n == 0) return ZR_OK;
    HIP_OK(hipSetDevice(c->device));
    DevBuf<double> d_in, d_out;
    int rc;
    if ((rc = d_in.upload(std::vector<double>(uvp5, uvp5 + n * 5))) || (rc = d_out.alloc(n * 3))) return rc;
    HIP_OK(zr::launch_kat_texture(s->ds, texture_id, d_in.p, n, d_out.p, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(out_rgb, d_out.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return ZR_OK;
}

int zr_kat_background(zr_ctx* c, const zr_scene* s, const zr_env* env, const double* dirs3, size_t n, double* out_rgb) {
    if (!c || !s || !env || (n && (!dirs3 || !out_rgb))) return fail(ZR_E_INVALID, "null argument");
    if (!s->committed) return fail(ZR_E_STATE, "zr_scene_commit must precede zr_kat_background");
    if (s->ctx != c) return fail(ZR_E_INVALID, "scene belongs to another context");
    zr::DEnv de; make_env(*env, de);
    if (de.mode > ZR_ENV_SOLID_COLOR) return fail(ZR_E_INVALID, "unknown environment mode %u", de.mode);
    if (de.mode == ZR_ENV_HDR_MAP && de.hdr_tex != ZR_NO_TEXTURE && de.hdr_tex >= s->textures.size()) return fail(ZR_E_INVALID, "environment texture id out of range");
    if (n == 0) return ZR_OK;
    HIP_OK(hipSetDevice(c->device));
    DevBuf<double> d_in, d_out;
    int rc;
    if ((rc = d_in.upload(std::vector<double>(dirs3, dirs3 + n * 3))) || (rc = d_out.alloc(n * 3))) return rc;
    HIP_OK(zr::launch_kat_background(s->ds, de, d_in.p, n, d_out.p, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(out_rgb, d_out.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return ZR_OK;
}

int zr_kat_camera_rays(zr_ctx* c, const zr_camera* cam, uint64_t seed, const int32_t* requests3, size_t n, double* out7) {
    if (!c || !cam || (n && (!requests3 || !out7))) return fail(ZR_E_INVALID, "null argument");
    if (n == 0) return ZR_OK;
    zr::DCamera dc; make_camera(*cam, dc);
    for (size_t k = 0; k < n; k++)
        if (requests3[3 * k] < 0 || requests3[3 * k] >= dc.W || requests3[3 * k + 1] < 0 || requests3[3 * k + 1] >= dc.H || requests3[3 * k + 2] < 0)
            return fail(ZR_E_INVALID, "camera-ray request %zu outside the frame", k);
    HIP_OK(hipSetDevice(c->device));
    DevBuf<int32_t> d_req; DevBuf<double> d_out;
    int rc;
    if ((rc = d_req.upload(std::vector<int32_t>(requests3, requests3 + n * 3))) || (rc = d_out.alloc(n * 7))) return rc;
    HIP_OK(zr::launch_kat_camera_rays(dc, seed, d_req.p, n, d_out.p, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(out7, d_out.p, n * 7 * sizeof(double), hipMemcpyDeviceToHost));
    return ZR_OK;
}

}  // extern "C"
