// zr_wavefront.hip — render kernel variant 1: "wave scheduler" persistent kernel for gfx950.
//
// One workgroup = one wave64.  A wave owns an 8x8 pixel sub-tile at a time (task queue: one global atomic per
// task) and streams that sub-tile's 64*spp primary samples through its 64 lanes as a per-wave state machine:
//
//     NEED  --regenerate-->  NODE  --leaf found-->  LEAF  --leaf done-->  NODE ... --stack empty-->  SHADE
//       ^                                                                                              |
//       +------------------------- path ended (miss / absorbed / depth / roulette) --------------------+
//                                  path continues (scatter): new ray, back to NODE
//
// Every loop iteration the wave counts, with __ballot, how many lanes are ready for each phase and runs ONE
// phase — the one with the most ready lanes — so the long FP64 code of a phase (BVH pair test, triangle test,
// shading) always executes with a well-filled exec mask instead of serialising one lane's traversal step behind
// another lane's shading.  Lanes whose path ends take the next sample of the sub-tile in the same shading phase
// (ballot + prefix popcount = ray regeneration), so no lane waits for the longest path of its wave.
//   * BVH stack: per lane, 16 entries {pair index, entry distance} in LDS laid out [level][lane] (conflict-free
//     ds_read/write_b64), deeper levels spill to a per-wave slab in HBM; popped entries farther than the current
//     closest hit are culled without touching memory.
//   * accumulator: finished samples are added to the sub-tile's 64 x double3 slots in LDS (ds_add_f64), the
//     pixel means are written once per task — no global atomics.  A wave executes in lockstep, so the order of
//     those adds, and therefore the image, is bit-reproducible run to run and independent of which wave picks
//     which task.
#include "zr_device.h"
#include "zr_launch.h"

namespace zr {

#ifndef WF_MIN_WAVES
#define WF_MIN_WAVES 2 /* waves per SIMD the register allocator must leave room for */
#endif
#define WF_LDS_STACK 16
#define WF_OVERFLOW (ZR_STACK_DEPTH - WF_LDS_STACK)

enum { ST_NEED = 0, ST_NODE = 1, ST_LEAF = 2, ST_SHADE = 3, ST_DONE = 4 };

struct StackEntry { uint32_t node; float tn; };

template <bool COUNT>
__global__ __launch_bounds__(64, WF_MIN_WAVES) void render_wavefront(DScene sc, DCamera cam, DEnv env, uint64_t seed, WorkDesc wd,
                                                           double* __restrict__ out, unsigned long long* __restrict__ gctr,
                                                           unsigned int* __restrict__ task_counter, StackEntry* __restrict__ overflow,
                                                           unsigned int n_tasks, int sub_x) {
    __shared__ double acc[64 * 3];
    __shared__ StackEntry lstack[WF_LDS_STACK * 64];
    __shared__ unsigned char slot_of[64];

    const int lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    StackEntry* gstack = overflow + (size_t)blockIdx.x * WF_OVERFLOW * 64 + lane;
    const double INF = __builtin_huge_val();
    const uint32_t NONE = 0xFFFFFFFFu;
    const int sub_per_tile = sub_x * sub_x;
    const int spp = cam.spp;
    const int depth_inner = cam.max_depth - 1;  // ray_color(scattered, depth - 1), camera.hpp:1000

    // counters (COUNT build only)
    uint32_t c_nodes = 0, c_sph = 0, c_tri = 0, c_cube = 0, c_med = 0, c_seg = 0, c_hits = 0, c_samp = 0;
    unsigned long long c_draws = 0;
    // scheduler statistics (COUNT build only, wave-uniform): executions of each phase and lanes ready at each
    unsigned long long s_exec[3] = {0, 0, 0}, s_lanes[3] = {0, 0, 0};

    for (;;) {
        unsigned int task = 0;
        if (lane == 0) task = atomicAdd(task_counter, 1u);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= n_tasks) break;
        const int tile = wd.tiles[task / sub_per_tile];
        const int sub = (int)(task % sub_per_tile);
        const int tx0 = (tile % wd.tiles_x) * wd.tile_size, ty0 = (tile / wd.tiles_x) * wd.tile_size;
        const int bx = tx0 + (sub % sub_x) * 8, by = ty0 + (sub / sub_x) * 8;
        const int mypx = bx + (lane & 7), mypy = by + (lane >> 3);
        const bool valid = mypx >= wd.x0 && mypx < wd.x1 && mypy >= wd.y0 && mypy < wd.y1 && mypx < tx0 + wd.tile_size && mypy < ty0 + wd.tile_size;
        const unsigned long long vmask = __ballot(valid);
        const int nvalid = __popcll(vmask);
        if (nvalid == 0) continue;
        __syncthreads();  // previous task's flush is complete (single wave: orders the LDS accesses)
        acc[lane * 3 + 0] = 0.0; acc[lane * 3 + 1] = 0.0; acc[lane * 3 + 2] = 0.0;
        if (valid) slot_of[__popcll(vmask & lt_mask)] = (unsigned char)lane;
        __syncthreads();
        const unsigned int total = (unsigned int)nvalid * (unsigned int)spp;
        unsigned int next_q = 0;

        // ---- lane state --------------------------------------------------------------------------------
        int st = ST_NEED;
        Ray ray; ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1);
        double idx_ = 0, idy_ = 0, idz_ = 0, ox_ = 0, oy_ = 0, oz_ = 0;
        double tbest = INF;
        uint32_t kbest = NONE, ibest = 0, cur = NONE;
        int sp = 0;
        uint32_t pa_first = 0, pa_meta = 0, pb_first = 0, pb_meta = 0, pend_i = 0;
        V3 L = mk(0, 0, 0), beta = mk(1, 1, 1), att0 = mk(1, 1, 1);
        Rng g; g.key = 0; g.k = 0; g.bounce = 0;
        int b_inner = 0;       // loop index of ray_color's for-loop (camera.hpp:933)
        bool first = true;     // the segment in flight is the primary one
        int slot = 0;

        auto start_traversal = [&]() {
            slab_constants(ray.d.x, ray.o.x, idx_, ox_); slab_constants(ray.d.y, ray.o.y, idy_, oy_); slab_constants(ray.d.z, ray.o.z, idz_, oz_);
            tbest = INF; kbest = NONE; cur = 0; sp = 0; pa_meta = 0; pb_meta = 0; pend_i = 0;
            st = ST_NODE;
            if (COUNT) c_seg++;
        };
        auto pop_next = [&]() {  // cur = nearest-first deferred node that can still matter, or NONE
            cur = NONE;
            while (sp > 0) {
                sp--;
                StackEntry e = sp < WF_LDS_STACK ? lstack[sp * 64 + lane] : gstack[(size_t)(sp - WF_LDS_STACK) * 64];
                if ((double)e.tn <= tbest) { cur = e.node; break; }
            }
        };

        // every iteration advances at least one lane, so the loop terminates; the cap only bounds a logic error
        const unsigned long long iter_cap = (unsigned long long)total * 4096ull + (1ull << 22);
        unsigned long long iter = 0;
        for (; iter < iter_cap; iter++) {
            const uint32_t lkind = (pa_meta >> 16) - 1u;  // kind of the pending leaf (valid in ST_LEAF)
            const int n1 = __popcll(__ballot(st == ST_NODE));
            const int n2t = __popcll(__ballot(st == ST_LEAF && lkind == ZR_PRIM_TRIANGLE));
            const int n2s = __popcll(__ballot(st == ST_LEAF && lkind == ZR_PRIM_SPHERE));
            const int n2g = __popcll(__ballot(st == ST_LEAF)) - n2t - n2s;  // cubes, media, wrapped objects
            const int n3 = __popcll(__ballot(st == ST_SHADE));
            const int n0 = __popcll(__ballot(st == ST_NEED));
            const bool work_left = next_q < total;
            const int n3e = n3 + (work_left ? n0 : 0);
            const int n2 = n2t > n2s ? (n2t > n2g ? n2t : n2g) : (n2s > n2g ? n2s : n2g);  // best leaf sub-phase
            if (n1 + n2 + n3e == 0) break;

            if (n1 >= n2 && n1 >= n3e) {
                // ================= NODE phase: one sibling-pair record per lane =================
                if (COUNT) { s_exec[0]++; s_lanes[0] += n1; }
                if (st == ST_NODE) {
                    const NodePair* np = sc.nodes + cur;
                    const float4 q0 = reinterpret_cast<const float4*>(np)[0];
                    const float4 q1 = reinterpret_cast<const float4*>(np)[1];
                    const float4 q2 = reinterpret_cast<const float4*>(np)[2];
                    const uint4 q3 = reinterpret_cast<const uint4*>(np)[3];
                    if (COUNT) c_nodes += 2;
                    double tn0, tf0, tn1, tf1;
                    {
                        double a0 = fma((double)q0.x, idx_, -ox_), a1 = fma((double)q1.z, idx_, -ox_);
                        double b0 = fma((double)q0.y, idy_, -oy_), b1 = fma((double)q1.w, idy_, -oy_);
                        double c0 = fma((double)q0.z, idz_, -oz_), c1 = fma((double)q2.x, idz_, -oz_);
                        tn0 = fmax(fmax(fmin(a0, a1), fmin(b0, b1)), fmax(fmin(c0, c1), 0.001));
                        tf0 = fmin(fmin(fmax(a0, a1), fmax(b0, b1)), fmin(fmax(c0, c1), tbest));
                    }
                    {
                        double a0 = fma((double)q0.w, idx_, -ox_), a1 = fma((double)q2.y, idx_, -ox_);
                        double b0 = fma((double)q1.x, idy_, -oy_), b1 = fma((double)q2.z, idy_, -oy_);
                        double c0 = fma((double)q1.y, idz_, -oz_), c1 = fma((double)q2.w, idz_, -oz_);
                        tn1 = fmax(fmax(fmin(a0, a1), fmin(b0, b1)), fmax(fmin(c0, c1), 0.001));
                        tf1 = fmin(fmin(fmax(a0, a1), fmax(b0, b1)), fmin(fmax(c0, c1), tbest));
                    }
                    bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
                    // empty leaves (count 0) never count as hit
                    if (q3.z != 0 && (q3.z & 0xFFFFu) == 0) h0 = false;
                    if (q3.w != 0 && (q3.w & 0xFFFFu) == 0) h1 = false;
                    // visit order: nearer child first
                    const bool swp = h0 && h1 && tn1 < tn0;
                    const uint32_t cA = swp ? q3.y : q3.x, cB = swp ? q3.x : q3.y;
                    const uint32_t mA = swp ? q3.w : q3.z, mB = swp ? q3.z : q3.w;
                    const bool hA = swp ? h1 : h0, hB = swp ? h0 : h1;
                    const double tnB = swp ? tn0 : tn1;
                    uint32_t next = NONE;
                    pa_meta = 0; pb_meta = 0; pend_i = 0;
                    if (hA) {
                        if (mA != 0) { pa_first = cA; pa_meta = mA; } else next = cA;
                    }
                    if (hB) {
                        if (mB != 0) {
                            if (pa_meta == 0) { pa_first = cB; pa_meta = mB; } else { pb_first = cB; pb_meta = mB; }
                        } else if (next == NONE) {
                            next = cB;
                        } else {
                            StackEntry e; e.node = cB; e.tn = __double2float_rd(tnB);
                            if (sp < WF_LDS_STACK) lstack[sp * 64 + lane] = e; else gstack[(size_t)(sp - WF_LDS_STACK) * 64] = e;
                            sp++;
                        }
                    }
                    cur = next;
                    if (pa_meta != 0) st = ST_LEAF;
                    else {
                        if (cur == NONE) pop_next();
                        if (cur == NONE) st = ST_SHADE;
                    }
                }
            } else if (n2 >= n3e) {
                // ================= LEAF phase: one primitive per lane, grouped by kind =================
                // the leaf kind with the most waiting lanes runs; the other kinds keep waiting
                const uint32_t kind = lkind;
                const bool is_leaf = st == ST_LEAF;
                const bool any_tri = n2t == n2;
                const bool any_sph = !any_tri && n2s == n2;
                if (COUNT) { s_exec[1]++; s_lanes[1] += n2; }
                bool tested = false;
                if (any_tri) {
                    if (is_leaf && kind == ZR_PRIM_TRIANGLE) {
                        double t;
                        if (COUNT) c_tri++;
                        if (triangle_t(sc.tri_v + (size_t)(pa_first + pend_i) * 9, ray, 0.001, tbest, t)) { tbest = t; kbest = kind; ibest = pa_first + pend_i; }
                        tested = true;
                    }
                } else if (any_sph) {
                    if (is_leaf && kind == ZR_PRIM_SPHERE) {
                        double t;
                        if (COUNT) c_sph++;
                        if (sphere_t(sc.spheres + (size_t)(pa_first + pend_i) * 4, ray, 0.001, tbest, t)) { tbest = t; kbest = kind; ibest = pa_first + pend_i; }
                        tested = true;
                    }
                } else if (is_leaf && kind != ZR_PRIM_TRIANGLE && kind != ZR_PRIM_SPHERE) {
                    double t;
                    if (COUNT) {
                        uint32_t kk = kind;
                        if (kk == ZR_KIND_WRAPPED) kk = sc.wrapped[pa_first + pend_i].type;
                        if (kk == ZR_PRIM_SPHERE) c_sph++; else if (kk == ZR_PRIM_TRIANGLE) c_tri++; else if (kk == ZR_PRIM_CUBE) c_cube++; else c_med++;
                    }
                    if (object_t(sc, kind, pa_first + pend_i, ray, 0.001, tbest, g, t)) { tbest = t; kbest = kind; ibest = pa_first + pend_i; }
                    tested = true;
                }
                if (tested) {
                    pend_i++;
                    if (pend_i >= (pa_meta & 0xFFFFu)) {
                        pa_first = pb_first; pa_meta = pb_meta; pb_meta = 0; pend_i = 0;
                        if (pa_meta == 0) {
                            if (cur == NONE) pop_next();
                            st = cur != NONE ? ST_NODE : ST_SHADE;
                        }
                    }
                }
            } else {
                // ================= SHADE phase: finish segments, then regenerate =================
                if (COUNT) { s_exec[2]++; s_lanes[2] += n3; }
                if (st == ST_SHADE) {
                    g.bounce++;
                    bool ended = false;   // path ended: `contrib` is added to the pixel
                    V3 contrib = mk(0, 0, 0);
                    if (kbest == NONE) {
                        V3 bg = background(sc, env, ray.d);
                        if (first) contrib = bg;                     // camera.hpp:520
                        else contrib = att0 * (L + beta * bg);       // camera.hpp:941, 1000
                        ended = true;
                    } else {
                        if (COUNT) c_hits++;
                        Rec rec;
                        object_rec(sc, kbest, ibest, ray, tbest, rec);
                        V3 em = emitted(sc, rec);
                        V3 att; Ray nr;
                        const bool sc_ok = scatter(sc, ray, rec, att, nr, g);
                        if (first) {
                            // ray_color_from_hit, camera.hpp:989-1004: L0 goes straight to the pixel sum
                            if (em.x != 0.0 || em.y != 0.0 || em.z != 0.0) {
                                atomicAdd(&acc[slot * 3 + 0], em.x); atomicAdd(&acc[slot * 3 + 1], em.y); atomicAdd(&acc[slot * 3 + 2], em.z);
                            }
                            if (!sc_ok) ended = true;
                            else {
                                att0 = att; L = mk(0, 0, 0); beta = mk(1, 1, 1); b_inner = 0; first = false;
                                if (depth_inner <= 0) ended = true;  // ray_color with depth <= 0 returns 0
                                else { ray = nr; start_traversal(); }
                            }
                        } else {
                            // body of ray_color's loop, camera.hpp:944-983
                            L = L + beta * em;
                            bool stop = !sc_ok;
                            if (!stop) {
                                beta = beta * att;
                                if (b_inner > 10) {
                                    if (len(beta) < 0.0001) stop = true;
                                    else {
                                        double p = fmax(fmax(beta.x, beta.y), beta.z);
                                        p = clampd(p, 0.05, 0.95);
                                        if (g.next() > p) stop = true;
                                        else beta = beta * (1 / p);
                                    }
                                }
                            }
                            if (!stop) { b_inner++; if (b_inner >= depth_inner) stop = true; }
                            if (stop) { contrib = att0 * L; ended = true; }
                            else { ray = nr; start_traversal(); }
                        }
                    }
                    if (ended) {
                        if (contrib.x != 0.0 || contrib.y != 0.0 || contrib.z != 0.0) {
                            atomicAdd(&acc[slot * 3 + 0], contrib.x); atomicAdd(&acc[slot * 3 + 1], contrib.y); atomicAdd(&acc[slot * 3 + 2], contrib.z);
                        }
                        if (COUNT) c_draws += g.k;
                        st = ST_NEED;
                    }
                }
                // regeneration: lanes without a path take the next samples of the sub-tile
                const unsigned long long need = __ballot(st == ST_NEED);
                if (need != 0ull) {
                    if (next_q < total) {
                        const unsigned int q = next_q + (unsigned int)__popcll(need & lt_mask);
                        if (st == ST_NEED && q < total) {
                            const unsigned int j = q / (unsigned int)spp;
                            const unsigned int s = q - j * (unsigned int)spp;
                            slot = slot_of[j];
                            const int px = bx + (slot & 7), py = by + (slot >> 3);
                            g.key = zr_stream_key(seed, (uint64_t)py * (uint64_t)cam.W + (uint64_t)px, (uint64_t)s);
                            g.k = 0; g.bounce = 0;
                            ray = camera_ray(cam, px, py, g);
                            first = true;
                            if (COUNT) c_samp++;
                            start_traversal();
                        }
                        const unsigned int nneed = (unsigned int)__popcll(need);
                        next_q = (total - next_q < nneed) ? total : next_q + nneed;
                    }
                    if (st == ST_NEED && next_q >= total) st = ST_DONE;
                }
            }
        }
        if (iter >= iter_cap && lane == 0) atomicAdd(&gctr[15], 1ull);  // reported by zr_get_counters as an error
        __syncthreads();
        if (valid) {
            const double scale = 1.0 / spp;  // camera.hpp:437,531
            double* o = out + ((size_t)mypy * cam.W + mypx) * 3;
            o[0] = acc[lane * 3 + 0] * scale; o[1] = acc[lane * 3 + 1] * scale; o[2] = acc[lane * 3 + 2] * scale;
        }
    }
    if (COUNT) {
        atomicAdd(&gctr[0], (unsigned long long)c_samp);
        atomicAdd(&gctr[1], (unsigned long long)c_seg);
        atomicAdd(&gctr[2], (unsigned long long)c_nodes);
        atomicAdd(&gctr[3], (unsigned long long)c_sph);
        atomicAdd(&gctr[4], (unsigned long long)c_tri);
        atomicAdd(&gctr[5], (unsigned long long)c_cube);
        atomicAdd(&gctr[6], (unsigned long long)c_med);
        atomicAdd(&gctr[7], (unsigned long long)c_hits);
        atomicAdd(&gctr[8], c_draws);
        if (lane == 0) for (int k = 0; k < 3; k++) { atomicAdd(&gctr[9 + 2 * k], s_exec[k]); atomicAdd(&gctr[10 + 2 * k], s_lanes[k]); }
    }
}

size_t wavefront_overflow_bytes(int blocks) { return (size_t)blocks * WF_OVERFLOW * 64 * sizeof(StackEntry); }

int wavefront_max_blocks() {
    int dev = 0, cus = 256, per_cu = 8;
    if (hipGetDevice(&dev) == hipSuccess) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_wavefront<false>, 64, 0) != hipSuccess || per_cu < 1) per_cu = 8;
    return cus * per_cu;
}

hipError_t launch_render_wavefront(const DScene& sc, const DCamera& cam, const DEnv& env, uint64_t seed, const WorkDesc& wd, double* out,
                                   unsigned long long* gctr, bool count, unsigned int* task_counter, void* overflow, int max_blocks,
                                   hipStream_t stream) {
    const int sub_x = (wd.tile_size + 7) / 8;
    const long long n_tasks = (long long)wd.n_tiles * sub_x * sub_x;
    if (n_tasks <= 0) return hipSuccess;
    if (n_tasks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(task_counter, 0, sizeof(unsigned int), stream);
    if (e != hipSuccess) return e;
    const int blocks = (int)(n_tasks < max_blocks ? n_tasks : max_blocks);
    dim3 grid((unsigned)blocks), block(64);
    if (count)
        hipLaunchKernelGGL(render_wavefront<true>, grid, block, 0, stream, sc, cam, env, seed, wd, out, gctr, task_counter, (StackEntry*)overflow,
                           (unsigned int)n_tasks, sub_x);
    else
        hipLaunchKernelGGL(render_wavefront<false>, grid, block, 0, stream, sc, cam, env, seed, wd, out, gctr, task_counter, (StackEntry*)overflow,
                           (unsigned int)n_tasks, sub_x);
    return hipGetLastError();
}

}  // namespace zr
