// zr_commit.cpp — the scene side of the C ABI (include/zr_capi.h): setters, validation, zr_scene_commit (BVH build on the host or on the device, flattening,
// upload), scene statistics.  See zr_host_internal.h for the layout of the host side.
#include "zr_flatten.h"

namespace {

int finish_commit(zr_scene* s, const CommitSummary& cs, size_t n_objs) {
    int rc;
    if (std::getenv("ZR_QUANT_STATS")) std::fprintf(stderr, "[zr] 4-wide nodes: %zu quantised (64 B) + FP32 root; %zu children kept closed for the grid\n", cs.n_quads, cs.kept_closed);
    s->quad_ok = cs.quant_ok && cs.n_quads < (1u << 31) && cs.max_leaf <= 16 && cs.n_sph < (1u << 24) && cs.n_tri < (1u << 24) && cs.n_cube < (1u << 24) &&
                 cs.n_media < (1u << 24) && cs.n_wrapped < (1u << 24) && cs.n_pcube < (1u << 24) && cs.n_insts < (1u << 24);
    if ((rc = s->d_ops.upload(s->ops.data(), s->ops.size()))) return rc;
    {
        // zr_material::pad_ on the device copy: the material reads u/v/tangent (image texture anywhere in its
        // texture tree, or a bump map) -> the kernels compute those hit-record fields only then
        std::vector<zr_material> mats = s->materials;
        auto tex_uses_uv = [&](uint32_t id) {
            std::vector<uint32_t> todo{id}; int guard = 0;
            while (!todo.empty() && guard++ < 4096) {
                uint32_t t = todo.back(); todo.pop_back();
                if (t >= s->textures.size()) continue;
                const zr_texture& tx = s->textures[t];
                if (tx.kind >= ZR_TEX_IMAGE_U8) return true;
                if (tx.kind == ZR_TEX_CHECKER) { todo.push_back(tx.odd); todo.push_back(tx.even); }
            }
            return guard >= 4096;
        };
        for (zr_material& m : mats) m.pad_ = (m.bump_tex != ZR_NO_TEXTURE || (m.kind != ZR_MAT_DIELECTRIC && tex_uses_uv(m.tex))) ? 1u : 0u;
        if ((rc = s->d_mats.upload(mats))) return rc;
    }
    if ((rc = s->d_texs.upload(s->textures))) return rc;
    if ((rc = s->d_texels.upload(s->texels.data(), s->texels.size()))) return rc;

    zr::DScene& d = s->ds;
    d.nodes = s->d_nodes.p; d.quads = s->d_quads.p;
    d.spheres = s->d_spheres.p; d.sphere_mat = s->d_sphere_mat.p;
    d.tri_v = s->d_tri_v.p; d.tri_s = s->d_tri_s.p;
    d.cubes = s->d_cubes.p; d.cube_mat = s->d_cube_mat.p;
    d.pcubes = s->d_pcubes.p; d.pcube_mat = s->d_pcube_mat.p;
    d.media = s->d_media.p; d.wrapped = s->d_wrapped.p; d.insts = s->d_insts.p; d.ops = s->d_ops.p;
    d.mats = s->d_mats.p; d.texs = s->d_texs.p; d.texels = s->d_texels.p;
    d.n_mats = (uint32_t)s->materials.size();
    d.mat_kinds = 0;
    for (const zr_material& m : s->materials) d.mat_kinds |= 1u << m.kind;
    d.root = cs.root;
    s->leaf_objects = 0;
    for (int k = 0; k < 8; k++) { d.leaf_cnt[k] = cs.leaf_cnt[k]; s->leaf_objects += cs.leaf_cnt[k]; }
    // a small world's objects for the fused kernel's arguments (zr_launch.h: FusedObjs): read back from the arrays just built,
    // whichever builder made them (a few hundred bytes)
    s->fused_ok = false;
    if (s->leaf_objects > 0 && s->leaf_objects <= ZR_FUSED_OBJECTS && cs.leaf_cnt[ZR_KIND_INSTANCE] == 0) {
        zr::FusedObjs fo{};
        auto take = [&](uint32_t kind, const double* d_src, size_t stride, size_t doubles) -> int {
            for (uint32_t i = 0; i < cs.leaf_cnt[kind]; i++) {
                fo.kind[fo.n] = kind; fo.index[fo.n] = i;
                HIP_OK(hipMemcpy(fo.rec[fo.n], d_src + (size_t)i * stride, doubles * sizeof(double), hipMemcpyDeviceToHost));
                fo.n++;
            }
            return ZR_OK;
        };
        if ((rc = take(ZR_PRIM_SPHERE, s->d_spheres.p, 4, 4)) || (rc = take(ZR_PRIM_TRIANGLE, s->d_tri_v.p, ZR_TRI_STRIDE, 9)) ||
            (rc = take(ZR_PRIM_CUBE, s->d_cubes.p, 6, 6)) || (rc = take(ZR_KIND_PCUBE, s->d_pcubes.p, ZR_PCUBE_STRIDE, ZR_PCUBE_STRIDE))) return rc;
        std::vector<zr::DMedium> hm(cs.leaf_cnt[ZR_PRIM_MEDIUM]);
        if (!hm.empty()) HIP_OK(hipMemcpy(hm.data(), s->d_media.p, hm.size() * sizeof(zr::DMedium), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < hm.size(); i++) {
            if (hm[i].chain_count != 0) continue;   // a wrapped boundary: tested through the scene's arrays (level 2)
            fo.kind[fo.n] = ZR_PRIM_MEDIUM; fo.index[fo.n] = i;
            const bool sph = hm[i].btype == ZR_PRIM_SPHERE;
            HIP_OK(hipMemcpy(fo.rec[fo.n], sph ? s->d_spheres.p + (size_t)hm[i].bindex * 4 : s->d_cubes.p + (size_t)hm[i].bindex * 6, (sph ? 4 : 6) * sizeof(double), hipMemcpyDeviceToHost));
            fo.rec[fo.n][6] = hm[i].neg_inv_density;
            const uint64_t idb = hm[i].id, tb = hm[i].btype;
            std::memcpy(&fo.rec[fo.n][7], &idb, 8); std::memcpy(&fo.rec[fo.n][8], &tb, 8);
            fo.n++;
        }
        bool scaled = false;   // the fused kernel's placed-cube code carries no scale (zr_device.h pcube_ray<false>): such a world takes the pipeline
        for (uint32_t i = 0; i < fo.n; i++) if (fo.kind[i] == ZR_KIND_PCUBE && fo.rec[i][15] != 0.0) scaled = true;
        s->fused = fo; s->fused_ok = !scaled;
    }
    {   // which build of the EXTEND kernel this world needs (zr_stream.hip)
        if (cs.n_insts) s->leaf_level = 3;   // placed runs of triangles: the build with the nested walk
        else if (cs.n_wrapped || !cs.plain_media) s->leaf_level = 2;
        else if (cs.n_cube || cs.n_pcube || cs.n_media) s->leaf_level = 1;
        else s->leaf_level = 0;
        const int force = (int)env_double("ZR_EXTEND_LEVEL", -1);
        if (force > s->leaf_level && force <= 3) s->leaf_level = force;
        // SHADE's lean build (zr_device.h: lean_rec / lean_shade): a world of bare triangles and spheres whose materials are lambertian / metal / dielectric / light
        // over solid-colour textures, no bump maps — nothing in it reads u, v, a tangent, an image or a wrapper chain
        bool lean = s->leaf_level == 0 && env_double("ZR_SHADE_LEAN", 1) != 0;
        for (const zr_material& m : s->materials) {
            if (m.kind != ZR_MAT_LAMBERTIAN && m.kind != ZR_MAT_METAL && m.kind != ZR_MAT_DIELECTRIC && m.kind != ZR_MAT_LIGHT) lean = false;
            if (m.bump_tex != ZR_NO_TEXTURE) lean = false;
            if (m.kind != ZR_MAT_DIELECTRIC && (m.tex >= s->textures.size() || s->textures[m.tex].kind != ZR_TEX_SOLID)) lean = false;
        }
        d.shade_lean = lean ? 1u : 0u;
    }
    s->stack_demand = cs.stack_demand;
    if (std::getenv("ZR_QUANT_STATS")) std::fprintf(stderr, "[zr] 4-wide tree: depth %d, worst-case traversal stack %u entries\n", cs.quad_depth, s->stack_demand);
    s->stats[0] = cs.n_pairs; s->stats[1] = (uint64_t)cs.max_depth; s->stats[2] = n_objs;
    s->stats[3] = cs.n_pairs * sizeof(zr::NodePair) + cs.n_quads * sizeof(zr::NodeQ) + (cs.n_sph * 4 + cs.n_tri * (ZR_TRI_STRIDE + 20) + cs.n_cube * 6 + cs.n_pcube * ZR_PCUBE_STRIDE) * 8 +
                  (cs.n_sph + cs.n_cube) * 4 + s->texels.size();
    s->builder = cs.builder;
    if (std::getenv("ZR_COMMIT_HASH")) {   // development / test aid: a hash of the committed 4-wide node array and pair records — the tree AND its layout in memory
        (void)hipDeviceSynchronize();
        auto fnv = [](const void* dev, size_t bytes) -> unsigned long long {
            std::vector<unsigned char> h(bytes);
            unsigned long long x = 1469598103934665603ull;
            if (bytes && hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost) == hipSuccess) for (unsigned char b : h) { x ^= b; x *= 1099511628211ull; }
            return x;
        };
        std::fprintf(stderr, "[zr] commit hash (%s): quads %016llx (%zu), pairs %016llx (%zu)\n", cs.builder, fnv(s->d_quads.p, cs.n_quads * sizeof(zr::NodeQ)), cs.n_quads,
                     fnv(s->d_nodes.p, cs.n_pairs * sizeof(zr::NodePair)), cs.n_pairs);
    }
    s->committed = true;
    if (s->borrowed) {   // the caller's arrays are not read again: forget them (a second commit needs a new zr_scene_set_*)
        s->spheres.drop(); s->sphere_mat.drop(); s->tri_v.drop(); s->tri_n.drop(); s->tri_mat.drop(); s->cubes.drop(); s->cube_mat.drop();
        s->media.drop(); s->ops.drop(); s->objects.drop(); s->texels.drop(); s->objects_set = false; s->borrowed = false; s->released = true;
    }
    return ZR_OK;
}


// ---- the commit with the tree built ON THE DEVICE (zr_build.hip) ------------------------------------------------------------------
// The scene's arrays go to the device as they are; boxes, Morton keys, sort, PLOC merging, leaf collapse, the 4-wide quantised
// nodes, the pair records and the primitive records in leaf order are all produced there.  The host classifies the world-list
// entries (a pass over 16-byte records), finishes the few compound objects (media, wrapped objects: each drags inner primitives
// behind the leaf ranges) and sizes the final arrays.  ZR_E_STATE from here means "this input is for the host builder" (a tree
// deeper than the traversal stack, coordinates beyond 1e18): the caller falls back.
constexpr int ZR_FALLBACK_HOST = 1;
int commit_device(zr_scene* s, const std::vector<zr_object>& objs, bool commit_stats, CommitSummary& cs) {
    auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_phase = now_s();
    auto phase = [&](const char* what) { if (commit_stats) { const double t = now_s(); std::fprintf(stderr, "[zr] commit(device): %-22s %.1f ms\n", what, (t - t_phase) * 1e3); t_phase = t; } };
    const uint32_t n = (uint32_t)objs.size();
    hipStream_t st = s->ctx->stream;
    int rc;
    // every group's tree is a build of its own (a few dozen launches and a handful of synchronisations, ~1 ms however small the run):
    // a world of very many small groups is the host builder's, which builds them on its threads
    if ((double)s->groups.size() > env_double("ZR_BVH_DEVICE_MAX_GROUPS", 256)) {
        std::fprintf(stderr, "[zr] device BVH build: %zu groups of triangles: host builder\n", s->groups.size());
        return ZR_FALLBACK_HOST;
    }
    // 1. classification + array sizes
    std::vector<uint8_t> code(n);
    const bool bake = env_double("ZR_BAKE_TRIANGLES", 1) != 0;
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        unsigned hw = std::thread::hardware_concurrency();
        const int T = n < 65536 ? 1 : (int)std::max(1u, std::min(16u, hw));
        std::vector<std::array<uint32_t, 8>> part((size_t)T, std::array<uint32_t, 8>{});
        auto work = [&](int t, size_t k0, size_t k1) {
            std::array<uint32_t, 8> c{};
            for (size_t k = k0; k < k1; k++) { uint32_t kind; uint8_t bk; classify_object(*s, objs[k], bake, kind, bk); code[k] = (uint8_t)(kind | (bk << 4)); c[kind & 7]++; }
            part[(size_t)t] = c;
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(work, t, (size_t)n * t / T, (size_t)n * (t + 1) / T);
        work(0, 0, (size_t)n / T);
        for (auto& x : th) x.join();
        for (auto& c : part) for (int k = 0; k < 8; k++) cnt[k] += c[k];
    }
    size_t x_sph = 0, x_tri = 0, x_cube = 0, x_media = 0;   // primitives inside media and wrapper chains: behind the leaf ranges
    auto count_inner = [&](uint32_t type, uint32_t idx, auto&& self) -> void {
        if (type == ZR_PRIM_SPHERE) x_sph++; else if (type == ZR_PRIM_TRIANGLE) x_tri++; else if (type == ZR_PRIM_CUBE) x_cube++;
        else { x_media++; self(s->media[idx].boundary_type, s->media[idx].boundary_index, self); }
    };
    if (cnt[ZR_PRIM_MEDIUM] + cnt[ZR_KIND_WRAPPED])
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t kind = code[k] & 7u;
            if (kind == ZR_PRIM_MEDIUM) count_inner(s->media[objs[k].index].boundary_type, s->media[objs[k].index].boundary_index, count_inner);
            else if (kind == ZR_KIND_WRAPPED) count_inner(objs[k].type, objs[k].index, count_inner);
        }
    size_t group_tris = 0;
    for (const zr_group& g : s->groups) group_tris += g.triangle_count;
    const size_t n_sph = cnt[ZR_PRIM_SPHERE] + x_sph, n_tri = cnt[ZR_PRIM_TRIANGLE] + group_tris + x_tri, n_cube = cnt[ZR_PRIM_CUBE] + x_cube;
    const size_t n_pcube = cnt[ZR_KIND_PCUBE], n_media = cnt[ZR_PRIM_MEDIUM] + x_media, n_wrapped = cnt[ZR_KIND_WRAPPED], n_insts = cnt[ZR_KIND_INSTANCE];
    phase("classify");
    // 2. the scene as given -> device (freed with this call), the final primitive arrays allocated.  The large arrays are pinned for the
    // copy (hipHostRegister: 2.4 ms per 160 MB on the GPU box, then 57 GB/s instead of the ~10 GB/s of a first pageable copy,
    // profiles/r3_affine_ab.txt) and travel asynchronously on the build's stream, under the classification above... and below
    DevBuf<double> r_sph, r_tri_v, r_tri_n, r_cubes, r_gbox;
    DevBuf<uint32_t> r_sph_mat, r_tri_mat, r_cube_mat, d_inst_group, d_run_demand, d_run_root, d_run_qroot;
    DevBuf<zr_medium> r_media; DevBuf<zr_object> r_objs; DevBuf<uint8_t> r_code;
    // (an early return leaves copies in flight: they are waited for before their source pages are unpinned)
    struct Pinned { hipStream_t st; std::vector<void*> p; ~Pinned() { if (!p.empty()) (void)hipStreamSynchronize(st); for (void* q : p) (void)hipHostUnregister(q); } } pinned{st, {}};
    auto send = [&](auto& buf, const auto* src, size_t count) -> int {
        using T = std::remove_cv_t<std::remove_pointer_t<decltype(src)>>;
        int r = buf.alloc(count);
        if (r || count == 0) return r;
        const size_t bytes = count * sizeof(T);
        if (bytes >= (4u << 20) && hipHostRegister((void*)src, bytes, hipHostRegisterDefault) == hipSuccess) {
            pinned.p.push_back((void*)src);
            HIP_OK(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, st));
        } else {
            (void)hipGetLastError();
            HIP_OK(hipMemcpy(buf.p, src, bytes, hipMemcpyHostToDevice));
        }
        return ZR_OK;
    };
    if ((rc = send(r_tri_v, s->tri_v.data(), s->tri_v.size())) || (rc = send(r_tri_n, s->tri_n.data(), s->tri_n.size())) ||
        (rc = send(r_objs, objs.data(), objs.size())) || (rc = send(r_tri_mat, s->tri_mat.data(), s->tri_mat.size())) ||
        (rc = send(r_sph, s->spheres.data(), s->spheres.size())) || (rc = send(r_sph_mat, s->sphere_mat.data(), s->sphere_mat.size())) ||
        (rc = send(r_cubes, s->cubes.data(), s->cubes.size())) || (rc = send(r_cube_mat, s->cube_mat.data(), s->cube_mat.size())) ||
        (rc = send(r_media, s->media.data(), s->media.size())) || (rc = s->d_ops.upload(s->ops.data(), s->ops.size())) || (rc = send(r_code, code.data(), code.size()))) return rc;
    if ((rc = s->d_spheres.alloc(n_sph * 4)) || (rc = s->d_sphere_mat.alloc(n_sph)) || (rc = s->d_tri_v.alloc(n_tri * ZR_TRI_STRIDE)) || (rc = s->d_tri_s.alloc(n_tri * 20)) ||
        (rc = s->d_cubes.alloc(n_cube * 6)) || (rc = s->d_cube_mat.alloc(n_cube)) || (rc = s->d_pcubes.alloc(n_pcube * ZR_PCUBE_STRIDE)) || (rc = s->d_pcube_mat.alloc(n_pcube)) ||
        (rc = s->d_insts.alloc(n_insts)) || (rc = d_inst_group.alloc(n_insts))) return rc;
    phase("upload as given");
    zr::BuildSceneIn in;
    in.spheres = r_sph.p; in.sphere_mat = r_sph_mat.p; in.tri_v = r_tri_v.p; in.tri_n = r_tri_n.p; in.tri_mat = r_tri_mat.p;
    in.cubes = r_cubes.p; in.cube_mat = r_cube_mat.p; in.media = r_media.p; in.ops = s->d_ops.p;
    zr::BuildParams bp;
    bp.ct = (float)env_double("ZR_BVH_COST_TRAVERSE", 1.0);
    const double ck[8] = {env_double("ZR_BVH_COST_SPHERE", 1.0), env_double("ZR_BVH_COST_TRI", 1.5), env_double("ZR_BVH_COST_CUBE", 1.0),
                          env_double("ZR_BVH_COST_MEDIUM", 3.0), env_double("ZR_BVH_COST_WRAPPED", 3.0), env_double("ZR_BVH_COST_PCUBE", 1.5),
                          env_double("ZR_BVH_COST_GROUP", 16.0), 1};
    for (int k = 0; k < 8; k++) bp.ck[k] = (float)ck[k];
    bp.max_leaf = (int)env_double("ZR_BVH_MAX_LEAF", 4);
    const int big = (int)env_double("ZR_BVH_MAX_LEAF_BIG", 1);
    const int leaf_cap[8] = {0, 0, big, big, big, big, 1, 0};
    for (int k = 0; k < 8; k++) bp.leaf_cap[k] = leaf_cap[k];
    bp.open_ratio = (float)env_double("ZR_BVH_OPEN_RATIO", 1.25);
    bp.radius = (int)env_double("ZR_BVH_PLOC_RADIUS", 16);
    // PLOC stops at n / 64 clusters (4096 ... 65536) and the host's SAH builder arranges those: the larger the SAH-built top, the closer
    // the walk comes to the host tree's, and the longer the host's pass takes (cfg3 EXTEND per frame against the host tree's: no top
    // +8.3 %, 16384 clusters +2.6 %, 65536 +2.3 %, for 17 / 19 / 28 ms of commit — the reference commits once per frame, so the default
    // is the 16384 a million objects get; profiles/r3_builders.txt).  ZR_BVH_TOP overrides (0: PLOC to the root)
    {
        const double top_env = env_double("ZR_BVH_TOP", -1);
        bp.top_clusters = top_env >= 0 ? (int)top_env : (int)std::min<size_t>(65536, std::max<size_t>(4096, (size_t)n / 64));
    }
    zr::BuildPrimOut out;
    out.spheres = s->d_spheres.p; out.sphere_mat = s->d_sphere_mat.p; out.tri_v = s->d_tri_v.p; out.tri_s = s->d_tri_s.p;
    out.cubes = s->d_cubes.p; out.cube_mat = s->d_cube_mat.p; out.pcubes = s->d_pcubes.p; out.pcube_mat = s->d_pcube_mat.p;
    out.insts = s->d_insts.p; out.inst_group = d_inst_group.p;
    auto builder = std::make_shared<zr::DeviceBuilder>(st);
    // explicit outcome (hipErrorInvalidValue used to be the signal): the builder says when the input is the host builder's business; running out
    // of memory is too — the device build's footprint is several times the host path's, whose arrays this function's buffers make room for when it returns
    auto build_fail = [&](hipError_t e) {
        if (builder->wants_host()) { std::fprintf(stderr, "[zr] device BVH build: %s\n", builder->error()); return (int)ZR_FALLBACK_HOST; }
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            std::fprintf(stderr, "[zr] device BVH build: out of device memory (%s): host builder\n", builder->error());
            builder.reset();
            return (int)ZR_FALLBACK_HOST;
        }
        return fail(ZR_E_DEVICE, "device BVH build failed: %s (%s)", hipGetErrorString(e), builder->error());
    };
    // 3. the groups' trees (two-level BVH: one tree per shared run of triangles, in its own space)
    const size_t ng = s->groups.size();
    std::vector<zr::BuiltTree> runs(ng);
    std::vector<double> gbox(ng * 6);
    std::vector<uint32_t> run_demand(ng), run_tri_base(ng);
    {
        size_t at = cnt[ZR_PRIM_TRIANGLE];
        for (size_t g = 0; g < ng; g++) {
            const zr_group& grp = s->groups[g];
            run_tri_base[g] = (uint32_t)at; at += grp.triangle_count;
            zr::BuildPrimOut go = out;
            go.base[ZR_PRIM_TRIANGLE] = run_tri_base[g];
            hipError_t e = builder->build(in, nullptr, nullptr, grp.first_triangle, grp.triangle_count, bp, true, go, nullptr, ZR_STACK_DEPTH - 2, false, runs[g]);
            if (e != hipSuccess) return build_fail(e);
            for (int k = 0; k < 6; k++) gbox[g * 6 + k] = runs[g].box[k];
            run_demand[g] = runs[g].demand;
        }
    }
    if (ng) { if ((rc = r_gbox.upload(gbox)) || (rc = d_run_demand.upload(run_demand))) return rc; in.group_box = r_gbox.p; }
    phase("groups' trees");
    // 4. the world's tree
    zr::BuiltTree world;
    world.want_boxes = std::getenv("ZR_BUILD_CHECK") != nullptr;
    {
        hipError_t e = builder->build(in, r_objs.p, r_code.p, 0, n, bp, false, out, ng ? d_run_demand.p : nullptr, ZR_STACK_DEPTH - 2, commit_stats, world);
        if (e != hipSuccess) return build_fail(e);
    }
    if (commit_stats)
        std::fprintf(stderr, "[zr] device build: boxes+keys %.2f, sort %.2f, PLOC %.2f (%u iterations), order %.2f, 4-wide %.2f, pairs %.2f, emit %.2f ms; depth %u, %u pairs, %u quads\n",
                     world.ms[0], world.ms[1], world.ms[2], world.ploc_iterations, world.ms[3], world.ms[4], world.ms[5], world.ms[6], world.depth, world.n_pairs, world.n_quads);
    if (world.want_boxes) {   // self-check: every object's device box must contain the box the host's Boxer computes for it
        std::vector<zr::BuildBox> gb(ng);
        for (size_t g = 0; g < ng; g++) for (int k = 0; k < 3; k++) { gb[g].lo[k] = gbox[g * 6 + k]; gb[g].hi[k] = gbox[g * 6 + 3 + k]; }
        Boxer boxer{*s, &gb};
        size_t bad = 0;
        for (uint32_t k = 0; k < n; k++) {
            const zr::BuildBox hb = boxer.chain(objs[k].type, objs[k].index, objs[k].chain_first, objs[k].chain_count);
            const float* d = &world.dbg_boxes[(size_t)k * 8];
            bool ok = true;
            for (int a = 0; a < 3; a++) if (!((double)d[a] <= hb.lo[a]) || !((double)d[4 + a] >= hb.hi[a])) ok = false;
            if (!ok && bad++ < 8)
                std::fprintf(stderr, "[zr] BUILD_CHECK: object %u (type %u, chain %u): device box [%g %g %g | %g %g %g] does not contain the host's [%g %g %g | %g %g %g]\n", k, objs[k].type,
                             objs[k].chain_count, d[0], d[1], d[2], d[4], d[5], d[6], hb.lo[0], hb.lo[1], hb.lo[2], hb.hi[0], hb.hi[1], hb.hi[2]);
        }
        if (bad) return fail(ZR_E_DEVICE, "ZR_BUILD_CHECK: %zu of %u object boxes computed on the device do not contain the host's", bad, n);
    }
    for (int k = 0; k < 8; k++)
        if (world.cnt[k] != cnt[k]) return fail(ZR_E_DEVICE, "device BVH build: %u leaf primitives of kind %d, expected %u (internal error)", world.cnt[k], k, cnt[k]);
    phase("world tree");
    // 5. the scene's node arrays at their exact sizes: the world's records first, then every group's
    size_t n_pairs = world.n_pairs, n_quads = world.n_quads;
    std::vector<uint32_t> run_root(ng), run_qroot(ng);
    for (size_t g = 0; g < ng; g++) { run_root[g] = (uint32_t)n_pairs; run_qroot[g] = (uint32_t)n_quads; n_pairs += runs[g].n_pairs; n_quads += runs[g].n_quads; }
    if ((rc = s->d_nodes.alloc(n_pairs)) || (rc = s->d_quads.alloc(n_quads))) return rc;
    {
        hipError_t e = builder->relocate(world, s->d_nodes.p, 0, s->d_quads.p, 0);
        for (size_t g = 0; g < ng && e == hipSuccess; g++) e = builder->relocate(runs[g], s->d_nodes.p, run_root[g], s->d_quads.p, run_qroot[g]);
        if (e == hipSuccess && n_insts) {
            if ((rc = d_run_root.upload(run_root)) || (rc = d_run_qroot.upload(run_qroot))) return rc;
            e = builder->patch_instances(s->d_insts.p, d_inst_group.p, (uint32_t)n_insts, d_run_root.p, d_run_qroot.p);
        }
        if (e != hipSuccess) return fail(ZR_E_DEVICE, "device BVH build: %s", hipGetErrorString(e));
    }
    // 6. compound objects on the host: a medium's boundary, the object inside a wrapper chain (Flattener's own routines, on arrays
    // whose untouched pages cost nothing; only what they wrote is uploaded)
    bool plain_media = true;
    if ((rc = s->d_media.alloc(n_media)) || (rc = s->d_wrapped.alloc(n_wrapped))) return rc;
    if (n_media + n_wrapped) {
        static const zr::BuildResult no_tree;
        Flattener fl{*s, objs, no_tree};
        fl.spheres.allocate(n_sph * 4); fl.sphere_mat.allocate(n_sph);
        fl.tri_v.allocate(n_tri * ZR_TRI_STRIDE); fl.tri_s.allocate(n_tri * 20);
        fl.cubes.allocate(n_cube * 6); fl.cube_mat.allocate(n_cube);
        fl.media.allocate(n_media); fl.wrapped.allocate(n_wrapped);
        fl.n_sph = cnt[ZR_PRIM_SPHERE]; fl.n_tri = cnt[ZR_PRIM_TRIANGLE] + group_tris; fl.n_cube = cnt[ZR_PRIM_CUBE]; fl.n_media = cnt[ZR_PRIM_MEDIUM];
        const size_t b_sph = fl.n_sph, b_tri = fl.n_tri, b_cube = fl.n_cube;
        std::vector<std::pair<uint32_t, uint32_t>> todo;   // (index in its kind's array, object): leaf order, media before wrapped objects
        for (int pass = 0; pass < 2; pass++) {
            todo.clear();
            for (size_t k = 0; k + 1 < world.compound.size(); k += 2) {
                const uint32_t oi = world.compound[k], di = world.compound[k + 1];
                if (((code[oi] & 7u) == ZR_PRIM_MEDIUM) == (pass == 0)) todo.emplace_back(di, oi);
            }
            std::sort(todo.begin(), todo.end());
            for (const auto& [di, oi] : todo) {
                const zr_object& o = objs[oi];
                if (pass == 0) fl.put_medium(di, o.index);
                else {
                    zr::DWrapped w{};
                    w.type = o.type; w.chain_first = o.chain_first; w.chain_count = o.chain_count;
                    w.index = fl.append_inner(o.type, o.index);
                    fl.wrapped[di] = w;
                }
            }
        }
        if (fl.n_sph != n_sph || fl.n_tri != n_tri || fl.n_cube != n_cube || fl.n_media != n_media)
            return fail(ZR_E_DEVICE, "device BVH build: compound objects do not add up (internal error)");
        for (size_t k = 0; k < n_media; k++) if (fl.media[k].chain_count != 0) plain_media = false;
        auto up = [&](void* dst, const void* src, size_t bytes) -> int { if (bytes) HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st)); return ZR_OK; };
        if ((rc = up(s->d_spheres.p + b_sph * 4, &fl.spheres[b_sph * 4], (n_sph - b_sph) * 32)) || (rc = up(s->d_sphere_mat.p + b_sph, &fl.sphere_mat[b_sph], (n_sph - b_sph) * 4)) ||
            (rc = up(s->d_tri_v.p + b_tri * ZR_TRI_STRIDE, &fl.tri_v[b_tri * ZR_TRI_STRIDE], (n_tri - b_tri) * ZR_TRI_STRIDE * 8)) ||
            (rc = up(s->d_tri_s.p + b_tri * 20, &fl.tri_s[b_tri * 20], (n_tri - b_tri) * 160)) ||
            (rc = up(s->d_cubes.p + b_cube * 6, &fl.cubes[b_cube * 6], (n_cube - b_cube) * 48)) || (rc = up(s->d_cube_mat.p + b_cube, &fl.cube_mat[b_cube], (n_cube - b_cube) * 4)) ||
            (rc = up(s->d_media.p, fl.media.data(), n_media * sizeof(zr::DMedium))) || (rc = up(s->d_wrapped.p, fl.wrapped.data(), n_wrapped * sizeof(zr::DWrapped)))) return rc;
        HIP_OK(hipStreamSynchronize(st));   // (the staging arrays die with this block)
    }
    HIP_OK(hipStreamSynchronize(st));
    phase("node arrays + compound");
    cs.root = world.root; cs.quant_ok = world.quant_ok;
    for (const zr::BuiltTree& r : runs) cs.quant_ok = cs.quant_ok && r.quant_ok;
    cs.n_pairs = n_pairs; cs.n_quads = n_quads; cs.n_sph = n_sph; cs.n_tri = n_tri; cs.n_cube = n_cube; cs.n_pcube = n_pcube;
    cs.n_media = n_media; cs.n_wrapped = n_wrapped; cs.n_insts = n_insts; cs.plain_media = plain_media;
    cs.stack_demand = world.demand; cs.quad_depth = (int)world.quad_depth; cs.max_depth = (int)world.depth; cs.max_leaf = bp.max_leaf;
    for (int k = 0; k < 8; k++) cs.leaf_cnt[k] = cnt[k];
    cs.builder = "device (PLOC)";
    {   // the scratch arena, the trees' local records and the as-given copies: freed off the caller's clock
        struct Trash { std::shared_ptr<zr::DeviceBuilder> b; DevBuf<double> a0, a1, a2, a3, a4; DevBuf<uint32_t> u0, u1, u2, u3, u4, u5, u6; DevBuf<zr_medium> m; DevBuf<zr_object> o; DevBuf<uint8_t> c; int device; };
        auto t = std::make_shared<Trash>();
        t->device = s->ctx->device;
        t->b = std::move(builder);
        std::swap(t->a0, r_sph); std::swap(t->a1, r_tri_v); std::swap(t->a2, r_tri_n); std::swap(t->a3, r_cubes); std::swap(t->a4, r_gbox);
        std::swap(t->u0, r_sph_mat); std::swap(t->u1, r_tri_mat); std::swap(t->u2, r_cube_mat); std::swap(t->u3, d_inst_group); std::swap(t->u4, d_run_demand);
        std::swap(t->u5, d_run_root); std::swap(t->u6, d_run_qroot); std::swap(t->m, r_media); std::swap(t->o, r_objs); std::swap(t->c, r_code);
        s->ctx->free_later([t]() mutable { (void)hipSetDevice(t->device); t.reset(); });
    }
    phase("release");
    return ZR_OK;
}

}  // namespace

extern "C" {

zr_scene* zr_scene_create(zr_ctx* c) {
    if (!c) { fail(ZR_E_INVALID, "null context"); return nullptr; }
    zr_scene* s = new zr_scene();
    s->ctx = c; s->device = c->device;
    return s;
}
void zr_scene_destroy(zr_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);   // (not through s->ctx: the context may be gone)
    delete s;
}

// A borrowed commit drops every geometry view (`released`): the scene can be committed again only after ALL of them were given
// again — zr_scene_set_all / zr_scene_set_all_borrowed, or each of the six geometry setters (SET_* bits).  A lone
// zr_scene_set_materials must not re-arm the commit: it would build an empty world and return ZR_OK.
enum { SET_SPHERES = 1, SET_TRIANGLES = 2, SET_CUBES = 4, SET_MEDIA = 8, SET_OPS = 16, SET_OBJECTS = 32, SET_ALL_GEOMETRY = 63 };
#define CHECK_SCENE(s) do { if (!(s)) return fail(ZR_E_INVALID, "null scene"); (s)->committed = false; } while (0)
#define GEOMETRY_SET(s, bit) do { if ((s)->released) { (s)->reset_mask |= (bit); if (((s)->reset_mask & SET_ALL_GEOMETRY) == SET_ALL_GEOMETRY) { (s)->released = false; (s)->reset_mask = 0; } } } while (0)

int zr_scene_set_spheres(zr_scene* s, const double* p, const uint32_t* mat, size_t n) {
    CHECK_SCENE(s);
    if (n && (!p || !mat)) return fail(ZR_E_INVALID, "null sphere arrays");
    s->spheres.copy(p, n * 4); s->sphere_mat.copy(mat, n);
    GEOMETRY_SET(s, SET_SPHERES);
    return ZR_OK;
}
int zr_scene_set_triangles(zr_scene* s, const double* v9, const double* n9, const uint32_t* mat, size_t n) {
    CHECK_SCENE(s);
    if (n && (!v9 || !n9 || !mat)) return fail(ZR_E_INVALID, "null triangle arrays");
    s->tri_v.copy(v9, n * 9); s->tri_n.copy(n9, n * 9); s->tri_mat.copy(mat, n);
    GEOMETRY_SET(s, SET_TRIANGLES);
    return ZR_OK;
}
int zr_scene_set_cubes(zr_scene* s, const double* q, const uint32_t* mat, size_t n) {
    CHECK_SCENE(s);
    if (n && (!q || !mat)) return fail(ZR_E_INVALID, "null cube arrays");
    s->cubes.copy(q, n * 12); s->cube_mat.copy(mat, n);
    GEOMETRY_SET(s, SET_CUBES);
    return ZR_OK;
}
int zr_scene_set_media(zr_scene* s, const zr_medium* m, size_t n) {
    CHECK_SCENE(s);
    if (n && !m) return fail(ZR_E_INVALID, "null media array");
    s->media.copy(m, n);
    GEOMETRY_SET(s, SET_MEDIA);
    return ZR_OK;
}
int zr_scene_set_xform_ops(zr_scene* s, const zr_xform_op* o, size_t n) {
    CHECK_SCENE(s);
    if (n && !o) return fail(ZR_E_INVALID, "null op array");
    s->ops.copy(o, n);
    GEOMETRY_SET(s, SET_OPS);
    return ZR_OK;
}
int zr_scene_set_objects(zr_scene* s, const zr_object* o, size_t n) {
    CHECK_SCENE(s);
    if (n && !o) return fail(ZR_E_INVALID, "null object array");
    s->objects.copy(o, n); s->objects_set = n > 0;
    GEOMETRY_SET(s, SET_OBJECTS);
    return ZR_OK;
}
int zr_scene_set_groups(zr_scene* s, const zr_group* g, size_t n) {
    CHECK_SCENE(s);
    if (n && !g) return fail(ZR_E_INVALID, "null group array");
    s->groups.assign(g, g + n);
    return ZR_OK;
}
int zr_scene_set_materials(zr_scene* s, const zr_material* m, size_t n) {
    CHECK_SCENE(s);
    if (n && !m) return fail(ZR_E_INVALID, "null material array");
    s->materials.assign(m, m + n);
    return ZR_OK;
}
int zr_scene_set_textures(zr_scene* s, const zr_texture* t, size_t n, const void* blob, size_t bytes) {
    CHECK_SCENE(s);
    if ((n && !t) || (bytes && !blob)) return fail(ZR_E_INVALID, "null texture arrays");
    s->textures.assign(t, t + n);
    s->texels.copy((const unsigned char*)blob, bytes);
    return ZR_OK;
}
int zr_scene_set_all(zr_scene* s, const zr_scene_desc* d) {
    CHECK_SCENE(s);
    if (!d) return fail(ZR_E_INVALID, "null scene description");
    int rc;
    if ((rc = zr_scene_set_spheres(s, d->spheres, d->sphere_mat, d->n_spheres))) return rc;
    if ((rc = zr_scene_set_triangles(s, d->tri_v, d->tri_n, d->tri_mat, d->n_tris))) return rc;
    if ((rc = zr_scene_set_cubes(s, d->cubes, d->cube_mat, d->n_cubes))) return rc;
    if ((rc = zr_scene_set_media(s, d->media, d->n_media))) return rc;
    if ((rc = zr_scene_set_xform_ops(s, d->ops, d->n_ops))) return rc;
    if ((rc = zr_scene_set_objects(s, d->objects, d->n_objects))) return rc;
    if ((rc = zr_scene_set_groups(s, d->groups, d->n_groups))) return rc;
    if ((rc = zr_scene_set_materials(s, d->materials, d->n_materials))) return rc;
    return zr_scene_set_textures(s, d->textures, d->n_textures, d->texels, d->texel_bytes);
}

int zr_scene_set_all_borrowed(zr_scene* s, const zr_scene_desc* d) {
    CHECK_SCENE(s);
    if (!d) return fail(ZR_E_INVALID, "null scene description");
    if ((d->n_spheres && (!d->spheres || !d->sphere_mat)) || (d->n_tris && (!d->tri_v || !d->tri_n || !d->tri_mat)) || (d->n_cubes && (!d->cubes || !d->cube_mat)) ||
        (d->n_media && !d->media) || (d->n_ops && !d->ops) || (d->n_objects && !d->objects) || (d->texel_bytes && !d->texels))
        return fail(ZR_E_INVALID, "null array in the scene description");
    s->spheres.borrow(d->spheres, d->n_spheres * 4); s->sphere_mat.borrow(d->sphere_mat, d->n_spheres);
    s->tri_v.borrow(d->tri_v, d->n_tris * 9); s->tri_n.borrow(d->tri_n, d->n_tris * 9); s->tri_mat.borrow(d->tri_mat, d->n_tris);
    s->cubes.borrow(d->cubes, d->n_cubes * 12); s->cube_mat.borrow(d->cube_mat, d->n_cubes);
    s->media.borrow(d->media, d->n_media);
    s->ops.borrow(d->ops, d->n_ops);
    s->objects.borrow(d->objects, d->n_objects); s->objects_set = d->n_objects > 0;
    s->texels.borrow((const unsigned char*)d->texels, d->texel_bytes);
    s->borrowed = true; s->released = false; s->reset_mask = 0;
    int rc;
    if ((rc = zr_scene_set_groups(s, d->groups, d->n_groups))) return rc;
    if ((rc = zr_scene_set_materials(s, d->materials, d->n_materials))) return rc;   // the small tables are copied: render calls validate against them
    if (d->n_textures && !d->textures) return fail(ZR_E_INVALID, "null texture array");
    s->textures.assign(d->textures, d->textures + d->n_textures);
    return ZR_OK;
}

int zr_scene_commit(zr_scene* s) {
    if (!s) return fail(ZR_E_INVALID, "null scene");
    if (s->released) return fail(ZR_E_STATE, "the arrays given to zr_scene_set_all_borrowed were released by the previous commit: set the scene again");
    s->committed = false;
    HIP_OK(hipSetDevice(s->ctx->device));
    // the world list
    std::vector<zr_object> objs;
    if (s->objects_set) objs.assign(s->objects.begin(), s->objects.end());
    else {
        std::vector<char> sb(s->sphere_mat.size(), 0), cb(s->cube_mat.size(), 0);
        for (const zr_medium& m : s->media) {
            if (m.boundary_type == ZR_PRIM_SPHERE && m.boundary_index < sb.size()) sb[m.boundary_index] = 1;
            if (m.boundary_type == ZR_PRIM_CUBE && m.boundary_index < cb.size()) cb[m.boundary_index] = 1;
        }
        for (uint32_t k = 0; k < s->sphere_mat.size(); k++) if (!sb[k]) objs.push_back({ZR_PRIM_SPHERE, k, 0, 0});
        for (uint32_t k = 0; k < s->tri_mat.size(); k++) objs.push_back({ZR_PRIM_TRIANGLE, k, 0, 0});
        for (uint32_t k = 0; k < s->cube_mat.size(); k++) if (!cb[k]) objs.push_back({ZR_PRIM_CUBE, k, 0, 0});
        for (uint32_t k = 0; k < s->media.size(); k++) objs.push_back({ZR_PRIM_MEDIUM, k, 0, 0});
    }
    const bool commit_stats = std::getenv("ZR_COMMIT_STATS") != nullptr;
    auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_phase = now_s();
    auto phase = [&](const char* what) { if (commit_stats) { const double t = now_s(); std::fprintf(stderr, "[zr] commit: %-22s %.1f ms\n", what, (t - t_phase) * 1e3); t_phase = t; } };
    int rc = validate(*s, objs);
    if (rc) return rc;
    phase("world list + validate");
    if (s->media.size() > 65535) return fail(ZR_E_INVALID, "at most 65535 media (RNG key layout, zr_rng.h)");
    {   // which builder.  ZR_BVH_BUILD=device | host forces one; otherwise worlds of at least ZR_BVH_DEVICE_MIN entries (131072: from
        // there on the device build's top is arranged by SAH, zr_build.h) are built on the device: cfg3's 1M triangles commit in
        // 19 ms instead of 106 and the frame takes 1 % longer than on the host's tree (EXTEND alone 2.6 %) — profiles/r3_builders.txt;
        // a small world is built faster by the host than a few dozen kernel launches take
        const char* bm = std::getenv("ZR_BVH_BUILD");
        const bool force_dev = bm && std::strcmp(bm, "device") == 0, force_host = bm && std::strcmp(bm, "host") == 0;
        const bool use_dev = !force_host && !objs.empty() && objs.size() < (1u << 30) && (force_dev || (double)objs.size() >= env_double("ZR_BVH_DEVICE_MIN", 131072));
        if (use_dev) {
            CommitSummary cs;
            rc = commit_device(s, objs, commit_stats, cs);
            if (rc == ZR_OK) return finish_commit(s, cs, objs.size());
            // the device build holds the scene as given, its arena and the final arrays at once: when one of ITS allocations does not fit, the host path
            // (whose staging lives in host memory) may still commit the world
            const bool oom = rc == ZR_E_DEVICE && std::strstr(zr_host::last_error(), "out of device memory") != nullptr;
            if (rc != ZR_FALLBACK_HOST && !oom) return rc;
            if (oom) std::fprintf(stderr, "[zr] device BVH build: %s: host builder\n", zr_host::last_error());
            phase("device build refused");
        }
    }

    // two-level BVH: every group of triangles gets a tree of its own, in its own space, once — however many objects place it
    std::vector<zr::BuildResult> runs(s->groups.size());
    std::vector<zr::BuildBox> group_box(s->groups.size());
    {
        Boxer tri_boxer{*s};
        const double ck_tri[8] = {1, 1, 1, 1, 1, 1, 1, 1};
        for (size_t g = 0; g < s->groups.size(); g++) {
            const zr_group& grp = s->groups[g];
            std::vector<zr::BuildBox> tb(grp.triangle_count);
            std::vector<uint32_t> tk(grp.triangle_count, ZR_PRIM_TRIANGLE);
            zr::BuildBox all; for (int a = 0; a < 3; a++) { all.lo[a] = kInf; all.hi[a] = -kInf; }
            for (uint32_t k = 0; k < grp.triangle_count; k++) {
                tb[k] = tri_boxer.prim(ZR_PRIM_TRIANGLE, grp.first_triangle + k);
                for (int a = 0; a < 3; a++) { all.lo[a] = std::fmin(all.lo[a], tb[k].lo[a]); all.hi[a] = std::fmax(all.hi[a], tb[k].hi[a]); }
            }
            group_box[g] = all;
            zr::build_bvh(tb, tk, 4, ZR_STACK_DEPTH - 2, 1.0, ck_tri, runs[g]);
            if (runs[g].max_depth >= ZR_STACK_DEPTH - 1) return fail(ZR_E_INVALID, "group %zu: BVH depth %d exceeds the traversal stack", g, runs[g].max_depth);
        }
    }
    // boxes + kinds
    Boxer boxer{*s, &group_box};
    std::vector<zr::BuildBox> boxes(objs.size());
    std::vector<uint32_t> kinds(objs.size());
    std::vector<uint8_t> baked(objs.size(), 0);
    const bool bake = env_double("ZR_BAKE_TRIANGLES", 1) != 0;
    std::atomic<size_t> bad_box{(size_t)-1};
    {
        unsigned hw = std::thread::hardware_concurrency();
        const size_t nobj = objs.size();
        const int T = nobj < 65536 ? 1 : (int)std::max(1u, std::min(16u, hw));
        auto work = [&](size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; k++) {
            const zr_object& o = objs[k];
            boxes[k] = boxer.chain(o.type, o.index, o.chain_first, o.chain_count);
            classify_object(*s, o, bake, kinds[k], baked[k]);
            for (int a = 0; a < 3; a++)
                if (!std::isfinite(boxes[k].lo[a]) || !std::isfinite(boxes[k].hi[a])) { size_t want = (size_t)-1; bad_box.compare_exchange_strong(want, k); }
        }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(work, nobj * t / T, nobj * (t + 1) / T);
        work(0, nobj / T);
        for (auto& x : th) x.join();
    }
    if (bad_box.load() != (size_t)-1) return fail(ZR_E_INVALID, "object %zu has a non-finite bounding box", bad_box.load());
    zr::BuildResult br;
    double ck[8] = {env_double("ZR_BVH_COST_SPHERE", 1.0), env_double("ZR_BVH_COST_TRI", 1.5), env_double("ZR_BVH_COST_CUBE", 1.0),
                    env_double("ZR_BVH_COST_MEDIUM", 3.0), env_double("ZR_BVH_COST_WRAPPED", 3.0), env_double("ZR_BVH_COST_PCUBE", 1.5),
                    env_double("ZR_BVH_COST_GROUP", 16.0), 1};
    int max_leaf = (int)env_double("ZR_BVH_MAX_LEAF", 4);
    // cubes, media and wrapped objects are few, large and dear to test: one per leaf, so that a ray only tests those whose own box it enters
    const int big = (int)env_double("ZR_BVH_MAX_LEAF_BIG", 1);
    const int leaf_cap[8] = {0, 0, big, big, big, big, 1, 0};   // a placement is always a leaf of its own (EXTEND enters it as a whole)
    phase("boxes");
    zr::build_bvh(boxes, kinds, max_leaf, ZR_STACK_DEPTH - 2, env_double("ZR_BVH_COST_TRAVERSE", 1.0), ck, br, leaf_cap);
    phase("binned-SAH build");
    if (br.max_depth >= ZR_STACK_DEPTH - 1) return fail(ZR_E_INVALID, "BVH depth %d exceeds the traversal stack", br.max_depth);

    // the flattener lives on the heap: it is handed, with everything else that is large, to a thread that frees it (below)
    std::shared_ptr<Flattener> flp(new Flattener{*s, objs, br});
    Flattener& fl = *flp;
    fl.baked = &baked;
    fl.runs = runs.empty() ? nullptr : &runs;
    fl.open_ratio = env_double("ZR_BVH_OPEN_RATIO", 1.25);
    // the primitive arrays (a quarter of a gigabyte for a million triangles) go to the device while the host still plans and
    // numbers the 4-wide nodes: a thread of its own, joined before the node arrays follow
    int up_rc = ZR_OK;
    std::string up_err;
    std::thread uploader;
    const int device = s->ctx ? s->ctx->device : 0;
    fl.after_primitives = [&]() {
        uploader = std::thread([&, device]() {
            auto go = [&]() -> int {
                HIP_OK(hipSetDevice(device));
                int r;
                if ((r = s->d_spheres.upload(fl.spheres))) return r;
                if ((r = s->d_sphere_mat.upload(fl.sphere_mat))) return r;
                if ((r = s->d_tri_v.upload(fl.tri_v))) return r;
                if ((r = s->d_tri_s.upload(fl.tri_s))) return r;
                if ((r = s->d_cubes.upload(fl.cubes))) return r;
                if ((r = s->d_cube_mat.upload(fl.cube_mat))) return r;
                if ((r = s->d_pcubes.upload(fl.pcubes))) return r;
                if ((r = s->d_pcube_mat.upload(fl.pcube_mat))) return r;
                if ((r = s->d_media.upload(fl.media))) return r;
                if ((r = s->d_wrapped.upload(fl.wrapped))) return r;
                return ZR_OK;
            };
            up_rc = go();
            if (up_rc != ZR_OK) up_err = zr_host::last_error();   // the error text is per thread
        });
    };
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{uploader};   // no path leaves the thread running
    fl.run();
    phase("flatten + quantise");

    if ((rc = s->d_nodes.upload(fl.pairs))) return rc;
    if ((rc = s->d_quads.upload(fl.quads))) return rc;
    if ((rc = s->d_insts.upload(fl.insts))) return rc;   // (after the groups' nodes were numbered: a placement names its group's root)
    if (uploader.joinable()) uploader.join();
    else if (fl.after_primitives) {   // a world without nodes returned from run() before the hook: upload the (empty) arrays here
        fl.after_primitives(); if (uploader.joinable()) uploader.join();
    }
    if (up_rc != ZR_OK) return fail(up_rc, "%s", up_err.c_str());
    CommitSummary cs;
    cs.root = fl.root; cs.quant_ok = fl.quant_ok; cs.n_pairs = fl.pairs.size(); cs.n_quads = fl.quads.size();
    cs.n_sph = fl.sphere_mat.size(); cs.n_tri = fl.tri_s.size() / 20; cs.n_cube = fl.cube_mat.size(); cs.n_pcube = fl.pcube_mat.size();
    cs.n_media = fl.media.size(); cs.n_wrapped = fl.wrapped.size(); cs.n_insts = fl.insts.size();
    cs.plain_media = true;   // media whose boundary is an unwrapped sphere or cube
    for (size_t k = 0; k < fl.media.size(); k++) if (fl.media[k].chain_count != 0) cs.plain_media = false;
    cs.stack_demand = fl.stack_demand(); cs.quad_depth = fl.quad_depth; cs.max_depth = br.max_depth; cs.max_leaf = max_leaf; cs.kept_closed = fl.n_kept_closed;
    for (int k = 0; k < 8; k++) cs.leaf_cnt[k] = fl.cnt[k];
    cs.builder = "host (binned SAH)";
    if ((rc = finish_commit(s, cs, objs.size()))) return rc;
    phase("upload");
    {   // unmapping half a gigabyte of staging arrays takes tens of milliseconds: not on the caller's clock
        struct Trash { std::shared_ptr<Flattener> fl; zr::BuildResult br; std::vector<zr::BuildBox> boxes; std::vector<zr_object> objs;
                       std::vector<uint32_t> kinds; std::vector<uint8_t> baked; std::vector<zr::BuildResult> runs; };
        auto t = std::make_shared<Trash>();
        fl.after_primitives = nullptr;   // (it captures locals of this call)
        t->fl = std::move(flp); t->br = std::move(br); t->boxes = std::move(boxes); t->objs = std::move(objs); t->kinds = std::move(kinds); t->baked = std::move(baked); t->runs = std::move(runs);
        s->ctx->free_later([t]() mutable { t.reset(); });
    }
    phase("release");
    return ZR_OK;
}

int zr_scene_stats(const zr_scene* s, uint64_t out[4]) {
    if (!s || !s->committed) return fail(ZR_E_STATE, "scene not committed");
    std::memcpy(out, s->stats, sizeof s->stats);
    return ZR_OK;
}

uint32_t zr_scene_traversal_stack(const zr_scene* s) { return s && s->committed ? s->stack_demand : 0u; }
const char* zr_scene_builder(const zr_scene* s) { return s && s->committed ? s->builder : ""; }

}  // extern "C"
