#!/usr/bin/env python3
"""bench.py — frames/s (+ Mfragments/s) of the MI355X raster + fragment-shade stage on BASELINE.json's configs[1]
(spot_triangulated_good.obj, 1024x1024, TEXTURE shader + 2 point lights).

A "step" = one pass of the hot path (setup → band binning → visibility raster → shading) over one batch of F synthetic
frames per GPU (frame i is the spot mesh rotated by 10*i degrees, the reference's own per-frame variation), inputs
(post-MVP triangle streams, lights, texture) already resident in HBM, output = the reference's framebuffer layout
(z + 3 planar float colour planes per frame) in HBM.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1: one process per GPU; every frame's 32-row bands are dealt round-robin to the ranks (srz_set_shard), each rank
renders its bands of F*N frames (per-GPU pixel work fixed → weak scaling), and an RCCL all-gather over xGMI plus a
de-interleave copy reassemble every full framebuffer on every rank (the exchange step BASELINE.json's north_star names).

Prints ONE JSON line (rank 0).  PyTorch is only plumbing here (device buffers, streams, torch.distributed).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "software-rasterizer_amd"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_MEASURED_COPY_GBS = 6290.0


def cpu_baseline(workload_name, budget_s=12.0, max_frames=100000):
    """The CPU oracle's OpenMP builds (oracle/srz_oracle.c: the same per-pixel code as the checker) timed on this host's
    cores on a bounded sample of the same workload, in the two shapes a CPU can run it:
      rows   — one frame at a time, its rows dealt in bands to the threads (the reference's own structure: it parallelises
               inside a frame, `src/Rasterizer.cpp:217`),
      frames — whole frames dealt to the threads, each with private planes (the throughput shape, = what the GPU batch does).
    The better of the two is `value`; both are in `sample`.  kind = "port": the reference itself cannot be built in this
    image (Oracle).  This is the ONLY place bench.py touches oracle/."""
    sys.path.insert(0, REPO)
    from oracle import oracle  # noqa: E402
    from srz import scenes
    wl = scenes.WORKLOADS[workload_name]()
    frames = [wl.frame(i) for i in range(36)]
    for slot, tex in enumerate(wl.texture_arrays):
        oracle.texture_set(slot, tex)
    planes = oracle.new_planes(wl.width, wl.height)
    ncpu = os.cpu_count() or 1
    # ---- rows: pick the team size that serves one frame best (short trials), then a bounded run ------------------------
    cands = sorted({t for t in (4, 8, 16, 32, 64) if 1 <= t <= ncpu})  # (a GPU box grants ~16 cores of its host to one GPU)
    best_t, best_rate = cands[0], 0.0
    for t in cands:
        rc, _ = oracle.draw_omp(frames[0], planes, band=8, threads=t)  # warm-up at this team size
        assert rc == 0
        k, t0 = 0, time.perf_counter()
        while k < 400 and time.perf_counter() - t0 < 0.4:
            oracle.draw_omp(frames[k % len(frames)], planes, band=8, threads=t)
            k += 1
        rate = k / (time.perf_counter() - t0)
        if rate > best_rate:
            best_t, best_rate = t, rate
    n, t0, rows_threads = 0, time.perf_counter(), best_t
    while n < max_frames and (time.perf_counter() - t0) < budget_s / 3:
        rc, rows_threads = oracle.draw_omp(frames[n % len(frames)], planes, band=8, threads=best_t)  # FUSED_CLEAR: clear + draw
        n += 1
    rows_dt = time.perf_counter() - t0
    rows_rate, rows_n = n / rows_dt, n
    t1 = time.perf_counter()
    rc, _, _ = oracle.draw(frames[0], planes, want_stats=False)
    single = time.perf_counter() - t1
    # ---- frames: whole frames per thread; calibrate the team size on ~4 frames per thread, then a bounded run ----------
    fcands = sorted({t for t in (8, 16, 32, 64) if 1 <= t <= ncpu})
    fbest_t, fbest_rate = fcands[0], 0.0
    for t in fcands:
        k = 4 * t
        t0 = time.perf_counter()
        rc, _ = oracle.draw_frames_omp(frames, k, threads=t)
        assert rc == 0
        rate = k / (time.perf_counter() - t0)
        if rate > fbest_rate:
            fbest_t, fbest_rate = t, rate
    fn = int(min(max_frames, max(fbest_t, fbest_rate * budget_s / 2)))
    t0 = time.perf_counter()
    rc, frames_threads = oracle.draw_frames_omp(frames, fn, threads=fbest_t)
    frames_dt = time.perf_counter() - t0
    frames_rate = fn / frames_dt
    use_frames = frames_rate >= rows_rate
    return {"value": frames_rate if use_frames else rows_rate, "unit": "frames/s",
            "cores": int(frames_threads if use_frames else rows_threads), "kind": "port",
            "shape": "frames" if use_frames else "rows",
            "sample": f"{workload_name}, clear+draw per frame, rotation 10 deg/frame; rows: {rows_n} frames in {rows_dt:.1f} s on "
                      f"{rows_threads} threads = {rows_rate:.0f} frames/s (bands of 8 rows, best of {cands}); frames: {fn} frames in "
                      f"{frames_dt:.1f} s on {frames_threads} threads = {frames_rate:.0f} frames/s (best of {fcands}); "
                      f"single-thread oracle: {1.0 / single:.1f} frames/s",
            "host_cpus": ncpu}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=256, help="frames per step per GPU")
    ap.add_argument("--workload", default="spot_texture_1024")
    ap.add_argument("--scope", choices=["raster", "draw"], default="raster",
                    help="raster: post-MVP triangle streams resident in HBM (BASELINE's hot path); "
                         "draw: meshes + per-frame matrices resident, the vertex stage (k_vertex) is timed too")
    ap.add_argument("--exchange", choices=["planes", "bgr8"], default="planes",
                    help="N>1 only. planes: all-gather the 4 float planes (16 B/px, the reference's framebuffer); "
                         "bgr8: resolve to 8-bit on the device first and all-gather display()'s image (3 B/px)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import srz
    from srz import abi, parallel, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched through torch.distributed.run with N ranks")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # ---- inputs: built by the product's host layer (C++ Scene / ObjLoader / vertex stage), uploaded once ----------
    wl = scenes.WORKLOADS[args.workload]()
    n_frames = args.frames * world
    uniq = [wl.frame(i) for i in range(min(n_frames, 36))]  # 36 distinct rotations (10 deg steps)
    frames = [uniq[i % len(uniq)] for i in range(n_frames)]
    ctx = srz.Context(local_rank, rank, world)
    wl.upload_textures(ctx)
    n_tris_frame = frames[0].n_tris
    if args.scope == "draw":  # same frames, given as meshes + matrices: the device runs the vertex stage every step
        wl.upload_meshes(ctx)
        suniq = [wl.scene_frame(i) for i in range(min(n_frames, 36))]
        frames = [suniq[i % len(suniq)] for i in range(n_frames)]
    fs = ctx.frameset(frames)
    stats = fs.stats()  # counting variant of the kernels, run once, outside the timed region
    if world > 1:
        st = torch.tensor([stats["fragments"], stats["visible"], stats["shaded"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(st)
        frag_total, vis_total = int(st[0]), int(st[1])
    else:
        frag_total, vis_total = stats["fragments"], stats["visible"]
    algo_bytes = fs.algorithmic_bytes()

    out = torch.empty(fs.out_shape, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    gathered = full = bgr = None
    if world > 1:
        bpr = fs.local_rows // 32
        if args.exchange == "planes":
            gathered = torch.empty((world,) + tuple(fs.out_shape), dtype=torch.float32, device="cuda")
            full = torch.empty((n_frames, 4, bpr * world * 32, fs.width), dtype=torch.float32, device="cuda")
        else:  # display()'s 8-bit image: one "plane" of W*3 bytes per row
            bgr = torch.empty((n_frames, 1, fs.local_rows, fs.width * 3), dtype=torch.uint8, device="cuda")
            gathered = torch.empty((world,) + tuple(bgr.shape), dtype=torch.uint8, device="cuda")
            full = torch.empty((n_frames, 1, bpr * world * 32, fs.width * 3), dtype=torch.uint8, device="cuda")

    def step():
        fs.render(out.data_ptr(), fs.out_bytes, abi.FUSED_CLEAR, stream)
        if world > 1:
            # band b = local_band*world + rank  →  row-major planes (rows beyond `height` are all-gather padding)
            if args.exchange == "planes":
                parallel.all_gather_frames(out, world, gathered, full)
            else:
                fs.resolve8(out.data_ptr(), bgr.data_ptr(), bgr.numel(), stream)
                parallel.all_gather_frames(bgr, world, gathered, full)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.set_kernel_timing(True)
    ctx.kernel_time_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    kt = ctx.kernel_time_ms(reset=True)
    ctx.set_kernel_timing(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])

    if rank == 0:
        fps = n_frames * args.steps / dt
        pipeline_s = kt["total_ms"] * 1e-3
        achieved = algo_bytes / pipeline_s / 1e9 if pipeline_s > 0 else 0.0
        traffic = None
        tfile = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "frames_per_sec", "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "width": fs.width, "height": fs.height,
                       "frames_per_step": n_frames, "frames_per_step_per_gpu": args.frames,
                       "triangles_per_frame": n_tris_frame, "lights": 2, "scope": args.scope,
                       "sharding": "whole frames on 1 GPU" if world == 1 else
                       f"32-row bands round-robin over {world} GPUs + RCCL all-gather of {args.exchange} + de-interleave (timed)"},
            "mfragments_per_sec": frag_total / fs.n_frames * fps / 1e6,
            "fragments_per_frame": frag_total / fs.n_frames, "visible_pixels_per_frame": vis_total / fs.n_frames,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "frac_of_measured_copy_6290": achieved / HBM_MEASURED_COPY_GBS,
                         "kernel": "hot path = k_setup + k_bands + k_raster + k_shade in line, k_clear beside k_raster/k_shade on a second stream (one launch each per step)",
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "launch_ms": kt["total_ms"], "k_setup_bands_ms": kt["bin_ms"], "k_raster_ms": kt["raster_ms"],
                         "k_shade_ms": kt["shade_ms"], "launches_timed": kt["launches"],
                         "note": "rank 0's shard; HIP events on the launch stream; algorithmic bytes = "
                                 "16*W*rows + 96*N_tri + 24*N_lights + min(3*texW*texH, 3*textured_px) per frame"},
            "reference_published": {"fps": 58.6, "note": "README.md:619-629, i7-12800HX/MSVC, spot+crate 5-mesh scene, "
                                    "draw() incl. vertex stage — different workload/hardware, not reproducible here"},
        }
        if world > 1:  # report the render and the exchange separately (rank 0's view)
            step_ms = dt / args.steps * 1e3
            res["multi_gpu"] = {"render_ms_per_step": kt["total_ms"], "exchange_ms_per_step": max(0.0, step_ms - kt["total_ms"]),
                                "exchange": args.exchange,
                                "bytes_received_per_rank_per_step": (world - 1) * (out.numel() * 4 if args.exchange == "planes" else bgr.numel()),
                                "note": "exchange = RCCL all-gather of every rank's band shard + de-interleave to row-major; "
                                        "xGMI-bound by construction (every rank receives (N-1)/N of every framebuffer)"}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_budget_s)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
