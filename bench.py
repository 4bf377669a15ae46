#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-tracing core (BASELINE.json metric: Mrays/s incl. bounces).

Workload at N=1 (BASELINE.json configs[1]): scenes/scene.txt (the reference's scene.txt: monkey + two spheres +
plane, 2 889 triangles, cape_hill_1k.hdr environment) at 1920x1080, 64 spp, MAX_BOUNCES 6.  One "step" = one
full pass of the hot path over that batch: reset accumulators, render 1920x1080x64 paths with rt_render
(closest-hit traversal + shading + accumulation), and -- for N > 1 -- combine accumulators over RCCL.

N > 1 (one process per GPU, launched by torch.distributed.run):
  --partition samples (default): rank r renders sample slice [r*spp, (r+1)*spp) of every pixel, then one RCCL
      sum-reduce of the float accumulators to rank 0 (BASELINE config 5 pattern).  Per-GPU work is fixed -> "weak".
  --partition rows: image rows interleaved in bands of 16 over the ranks, RCCL gather to rank 0
      (BASELINE config 4 pattern).  Total work fixed -> "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def algorithmic_bytes(st, n_pixels):
    """B = 56*N_inner + 8*N_leaf + 36*N_tri + S, counted on the reference's logical data.
    S = per hit 64 B (rest of the 25-float record) + 44 B material, 12 B per texel fetched (material textures
    and environment), 36 B per light-triangle pdf test, + 12 B per pixel accumulator write per launch."""
    return (56 * st.inner_visits + 8 * st.leaf_visits + 36 * st.tri_tests + (64 + 44) * st.hits +
            12 * st.texel_fetches + 36 * st.light_tests + 12 * n_pixels)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="scene.txt")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=6)
    ap.add_argument("--partition", choices=["samples", "rows"], default="samples")
    ap.add_argument("--mode", choices=["wavefront", "megakernel", "lockstep"], default="megakernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=16, help="spp of the bounded CPU-baseline sample")
    ap.add_argument("--synth-instances", type=int, default=0, help="use the synthetic instanced field instead of --scene")
    ap.add_argument("--bvh", choices=["reference", "sah"], default="reference",
                    help="sah = the NON-PARITY fast mode (binned-SAH tree, depth <= 24, leaves <= 2), reported separately")
    args = ap.parse_args()

    # torch first: its bundled HIP runtime has the same SONAME as /opt/rocm's, so librt_amd.so binds to
    # the one already loaded and torch tensors / RCCL and our kernels share one runtime.
    import torch
    import torch.distributed as dist
    import numpy as np
    import importlib
    from __graft_entry__ import load_package
    rt = load_package()
    D = importlib.import_module("ray_tracer_amd.distributed")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if args.synth_instances > 0:
        scene = rt.Scene.synth_field(args.synth_instances, env_path=os.path.join(rt.ASSET_DIR, "cape_hill_1k.hdr"),
                                     env_upsample=4)
        workload = f"synthetic field of {args.synth_instances} dragon instances ({scene.n_tris} triangles)"
    else:
        scene = rt.Scene.load(args.scene)
        workload = f"{args.scene} ({scene.n_tris} triangles, {scene.desc.n_nodes} BVH nodes)"
    if args.bvh == "sah":
        scene.rebuild_bvh("sah")
        workload += f" [non-parity SAH tree, {scene.desc.n_nodes} nodes]"
    W, H, spp = args.width, args.height, args.spp

    r = rt.Renderer(local_rank)
    r.upload(scene)
    p = rt.default_params(scene.camera, spp=spp, width=W, height=H, max_bounces=args.bounces)
    p.flags |= rt.MODE_FLAGS[args.mode]
    if world > 1:
        D.partition_params(p, rank, world, args.partition, spp)
    rows = rt.band_rows_of(p)
    n_floats = rows * W * 3

    # accumulators live in a torch tensor so RCCL can work on them in place (rt_bind_accum)
    accum = torch.zeros(n_floats, dtype=torch.float32, device="cuda")
    r.bind_accum(accum.data_ptr(), n_floats)
    row_gather = D.RowGather(W, H, rank, world, "cuda") if (world > 1 and args.partition == "rows") else None

    def step(params):
        accum.zero_()
        torch.cuda.current_stream().synchronize()
        st = r.render(params)  # blocking: launches on the context's stream and waits
        if world > 1:
            if args.partition == "samples":
                D.combine_samples(accum, dst=0)
            else:
                row_gather(accum)
        return st

    # one untimed launch with counters on: ray / node / triangle counts are deterministic for fixed inputs
    p_stats = rt.RenderParams.from_buffer_copy(p)
    p_stats.flags |= rt.F_STATS
    st_counts = step(p_stats)
    # one untimed launch with per-launch HIP events around the trace kernel (wavefront mode): its summed duration
    p_prof = rt.RenderParams.from_buffer_copy(p)
    p_prof.flags |= rt.F_PROFILE
    st_prof = step(p_prof)

    for _ in range(args.warmup):
        step(p)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # fused modes: HIP events around the dominant kernel of every timed step (two events per launch, on the
    # render stream); the wavefront pipeline has ~800 launches per frame and keeps its separate profile run
    p_timed = rt.RenderParams.from_buffer_copy(p)
    if args.mode != "wavefront":
        p_timed.flags |= rt.F_PROFILE
    t0 = time.perf_counter()
    kernel_ms, dom_ms_steps, dom_launches = [], [], 0
    for _ in range(args.steps):
        st = step(p_timed)
        kernel_ms.append(st.kernel_ms)
        dom_ms_steps.append(st.trace_ms)
        dom_launches = st.trace_launches
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([st_counts.rays, st_counts.paths], dtype=torch.float64, device="cuda")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_rays, total_paths = float(cnt[0].item()), float(cnt[1].item())
    else:
        total_rays, total_paths = float(st_counts.rays), float(st_counts.paths)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mrays = total_rays / (elapsed / args.steps) / 1e6
        k_ms = float(np.mean(kernel_ms))          # all kernels of one frame (HIP events on the render stream)
        abytes = algorithmic_bytes(st_counts, rows * W)
        if args.mode == "wavefront" and st_prof.trace_launches > 0:
            # dominant kernel = rt_trace_kernel: its algorithmic bytes (nodes + leaves + triangles) over the
            # summed duration of its launches in one frame, per launch on both sides of the ratio
            dom_kernel = "rt_trace_kernel"
            dom_bytes = 56 * st_counts.inner_visits + 8 * st_counts.leaf_visits + 36 * st_counts.tri_tests
            dom_ms = st_prof.trace_ms
            launches = st_prof.trace_launches
        elif args.mode == "megakernel" and dom_launches > 0:
            # path-parallel fused kernel: everything but the 12 bytes per pixel of the accumulator update, which
            # rt_sum_samples_kernel does; duration of its launches alone, averaged over the timed steps
            dom_kernel, dom_bytes = "rt_render_paths_kernel", abytes - 12 * rows * W
            dom_ms, launches = float(np.mean(dom_ms_steps)), dom_launches
        else:
            dom_kernel, dom_bytes, dom_ms, launches = "rt_render_kernel", abytes, k_ms, 1
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path) and args.synth_instances == 0 and args.scene == "scene.txt" and args.bvh == "reference" and (W, H, spp) == (1920, 1080, 64):
            try:
                traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/sec (incl. bounces)",
            "value": round(mrays, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong" if (world > 1 and args.partition == "rows") else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "bundled scene geometry/textures, synthetic camera batch" if args.synth_instances == 0 else "synthetic",
            "config": {"workload": workload, "width": W, "height": H, "spp_per_gpu": spp, "max_bounces": args.bounces,
                       "partition": "none" if world == 1 else args.partition, "mode": args.mode,
                       "wavefront_iterations": int(st_counts.iterations),
                       "rays_per_step": int(total_rays), "paths_per_step": int(total_paths),
                       "spp_per_sec_1080p": round(total_paths / (elapsed / args.steps) / (1920 * 1080), 2)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": dom_kernel, "launches_per_step": launches,
                         "avg_launch_ms": round(dom_ms / launches, 4),
                         "algorithmic_bytes_per_launch": int(dom_bytes / launches),
                         "frame_kernels_ms": round(k_ms, 3), "frame_algorithmic_bytes": int(abytes),
                         "frame_algorithmic_GBps": round(abytes / (k_ms * 1e-3) / 1e9, 1),
                         "bytes_per_ray": round(abytes / max(st_counts.rays, 1), 1)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(rt, scene, args)
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(rt, scene, args):
    """The oracle (CPU restatement of the reference trace/shade code) timed on this box's host cores, on a
    bounded sample of the same workload (same scene, resolution and bounce depth, fewer spp)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob  # baseline leg only
    cores = max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    p = rt.default_params(scene.camera, spp=args.cpu_spp, width=args.width, height=args.height,
                          max_bounces=args.bounces, flags=rt.F_DEFAULT | rt.F_STATS)
    t0 = time.perf_counter()
    _, st = ob.render(scene, p, n_threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(st.rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"same scene at {args.width}x{args.height}, {args.cpu_spp} of {args.spp} spp, {st.rays} rays in {dt:.1f} s"}


if __name__ == "__main__":
    main()
