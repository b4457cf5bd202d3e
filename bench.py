#!/usr/bin/env python3
"""bench.py — Msamples/s (ray·bounces = closest-hit segments per second) of the per-pixel sample loop on MI355X.

  python bench.py --gpus 1 --steps K --warmup W [--workload cfg3|cfg2|cfg5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full frame of the workload (every pixel, every sample, all bounces).  Default workload = cfg3 of
BASELINE.json — the 1 000 000-triangle mesh + HDRI IBL at 1920x1080, 512 spp, depth 10 — because that is the
configuration the metric's roofline target ("1M-triangle BVH traversal at 1 GPU") and the multi-GPU config (cfg4,
"same 1M-tri scene pixel-tiled") are quoted on; cfg2 (configs[1]) and cfg5 are selectable with --workload.
With N > 1 the frame's 32x32 pixel tiles are interleaved over the ranks (scene replicated, weak in memory but the
total work is fixed => "strong" scaling) and the double3 accumulator is sum-reduced to rank 0 over RCCL every step.

The timed region starts with the scene (BVH, vertices, normals, HDRI) resident in HBM and contains: zeroing the
accumulator, the render kernel(s), the RCCL reduce.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    "cfg3": ("cfg3", (), "1M-triangle torus-knot mesh + 4096x2048 HDRI IBL, 1920x1080, 512 spp, depth 10"),
    "cfg2": ("cfg2", (), "Shirley random spheres (485 spheres), 1280x720, 256 spp, depth 50"),
    "cfg5": ("cfg5", (), "Cornell box: cubes + constant_medium + dielectric, 600x600, 1024 spp, depth 50"),
    "cfg1": ("cfg1", (), "3 Lambertian spheres, 400x225, 16 spp, depth 8"),
    "cfg3w": ("cfg3w", (), "cfg3 with the mesh placed as the reference places meshes: 1M triangles under material_instance -> rotate_x -> "
                           "rotate_z -> translate, 1920x1080, 512 spp, depth 10"),
    "demo": ("demo", (), "the reference's demo workload (scene_management.hpp:103-236): ~900 scaled/rotated/re-materialed instances, "
                         "a wrapped 3840-triangle mesh, fog volume, 1280x720, 128 spp, depth 10"),
}
# bounded CPU-baseline samples (zenith_ref `time` arguments: xstep ystep spp), sized for ~10-30 s on 16 host threads
CPU_SAMPLE = {"cfg3": (4, 4, 128), "cfg2": (4, 4, 48), "cfg5": (4, 4, 48), "cfg1": (1, 1, 16), "demo": (4, 4, 32), "cfg3w": (4, 4, 128)}


def cpu_baseline(workload, threads):
    """The reference's CPU arithmetic on a bounded sample of the same workload, on this box's host cores."""
    from oracle import zr_oracle_py as zo
    xs, ys, spp = CPU_SAMPLE[workload]
    scene = WORKLOADS[workload][0]
    if zo.ref_available():
        r = zo.ref_run("time", scene, xs, ys, spp, threads)
        return {"value": round(r["mseg_per_s"], 4), "unit": "Msamples/s", "cores": threads, "kind": "reference",
                "sample": f"every {xs}th column x {ys}th row of the frame at {spp} spp ({r['primary']} primary samples, "
                          f"{r['segments']} segments, {r['render_s']:.1f} s; reference BVH build {r['bvh_build_s']:.1f} s not counted); "
                          "genuine reference hit/scatter/BVH code, camera loop restated (camera.hpp needs OpenImageDenoise)"}
    # fallback: the CPU restatement (a port)
    from raytracer_project_amd import capi
    ds = capi.DemoScene(scene)
    cam = ds.camera.copy()
    cam.samples_per_pixel = spp
    osc = zo.OracleScene(ds.desc)
    w, h = cam.image_width // xs, cam.image_height // ys
    reg = capi.Region(0, 0, w, h, 0, 0, 0, 0)
    t0 = time.perf_counter()
    _, ctr, _, _ = osc.render(cam, ds.env, ds.seed, reg, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": round(ctr.segments / dt * 1e-6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"top-left {w}x{h} pixels at {spp} spp ({ctr.segments} segments, {dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (invalidates the headline number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from raytracer_project_amd import capi, multi

    rank, local, world = multi.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    if os.environ.get("ZR_BENCH_ONE_DEVICE"):   # rehearsal of the N > 1 path on a one-GPU box (gloo): every rank on device 0
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    scene_name, scene_args, descr = WORKLOADS[args.workload]
    t0 = time.perf_counter()
    ds = capi.DemoScene(scene_name, *scene_args)
    t_scene = time.perf_counter() - t0
    cam = ds.camera.copy()
    if args.spp > 0:
        cam.samples_per_pixel = args.spp
    ctx = capi.Context(local)
    t0 = time.perf_counter()
    sc = capi.Scene(ctx, ds.desc)           # BVH build + upload: reported, not timed
    t_commit = time.perf_counter() - t0
    stats = sc.stats()

    H, W = cam.image_height, cam.image_width
    acc = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
    region = multi.tile_region(capi, rank, world)
    if os.environ.get("ZR_BENCH_SHARD_OF"):  # development aid: time one rank's share of an N-way sharded frame on one GPU
        region = multi.tile_region(capi, 0, int(os.environ["ZR_BENCH_SHARD_OF"]))
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step(count=False):
        acc.zero_()
        sc.render_device(cam, ds.env, ds.seed, acc.data_ptr(), stream, region, count)
        multi.reduce_frame(acc, world)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    # counting pass (instrumented kernel, never timed): segments and algorithmic bytes of one step
    step(count=True)
    torch.cuda.synchronize(dev)
    ctr = ctx.counters()
    seg_local = ctr.segments
    # the dominant kernel of the streaming pipeline is EXTEND (BVH walk): it owns the box / primitive bytes; the 76 B of
    # shading data per hit belong to the SHADE kernel and are left out of the EXTEND roofline
    bytes_local = ctr.algorithmic_bytes() - (76 * ctr.hits if int(os.environ.get("ZR_KERNEL", "2")) == 2 else 0)
    segments, alg_bytes, primary = multi.all_reduce_values([seg_local, bytes_local, ctr.primary_samples], world, dev)
    checksum = float(acc.sum().item()) if rank == 0 else 0.0

    for _ in range(args.warmup):
        step()
    ctx.kernel_times_ms()  # drain the launch log
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    dt = multi.all_reduce_values([dt], world, dev, op="max")[0]
    launches = ctx.kernel_times_ms(1 << 20)  # dominant kernel's launches (HIP events on the launch stream), timed steps only
    variant = int(os.environ.get("ZR_KERNEL", "2"))
    kernel_name = {0: "render_pixels", 1: "render_wavefront", 2: "stream_extend"}.get(variant, "?")

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = segments * args.steps / dt * 1e-6
        k_ms = sum(launches) / max(1, len(launches))
        launches_per_step = max(1, len(launches) // max(1, args.steps))
        # roofline of the dominant kernel: this rank's algorithmic bytes per launch / mean launch duration
        # (= bytes of one step / summed duration of that kernel's launches in one step)
        achieved = (bytes_local / launches_per_step) / (k_ms * 1e-3) * 1e-9 if k_ms > 0 else 0.0
        traffic = None
        try:  # HBM bytes per launch of the dominant kernel from the committed FETCH_SIZE pass (separate --pmc run)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r1_v2g_cfg3_traffic.json")))
            if tj["workload"] == args.workload and world == 1 and args.spp == 0:
                kj = tj["kernels"][kernel_name]
                traffic = int(kj["hbm_read_bytes_per_launch"] + kj.get("hbm_write_bytes_per_launch", 0))   # reads (x2 corrected) + writes
        except Exception:
            traffic = None
        out = {
            "metric": "Msamples/sec (rays·bounces)", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {descr}", "image": [W, H], "spp": cam.samples_per_pixel,
                       "max_depth": cam.max_depth, "segments_per_step": int(segments), "primary_per_step": int(primary),
                       "segments_per_primary": round(segments / max(primary, 1), 4),
                       "parallelism": f"pixel-tiles x{world}" if world > 1 else "single GPU",
                       "bvh_pairs": stats["bvh_pairs"], "bvh_depth": stats["bvh_depth"], "objects": stats["objects"],
                       "scene_build_s": round(t_scene, 3), "bvh_build_upload_s": round(t_commit, 3),
                       "frame_checksum": checksum},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": kernel_name, "kernel_ms": round(k_ms, 4), "launches_timed": len(launches),
                         "kernel_ms_per_step": round(sum(launches) / max(1, args.steps), 3),
                         "algorithmic_bytes_per_launch": int(bytes_local / launches_per_step),
                         "model": "32 B/child box tested + 72 B/triangle + 32 B/sphere + 48 B/cube + 76 B/hit (SURVEY §8d)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload, min(16, os.cpu_count() or 1))
            except Exception as e:  # the baseline is reported, never required
                out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
