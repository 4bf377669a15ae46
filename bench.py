#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-tracing core (BASELINE.json metric: Mrays/s incl. bounces).

One "step" = one full pass of the hot path over one batch: reset accumulators, render with rt_render (closest-hit
traversal + shading + accumulation) and -- for N > 1 -- combine the accumulators over RCCL.

Workloads (`--config`, BASELINE.json configs; explicit flags override a preset):
  c2   (default) scenes/scene.txt 1920x1080, 64 spp, 6 bounces            -- the configuration the metric is quoted on
  c3   scene2.txt 1920x1080, 256 spp             c3is  scene2_is.txt (960 importance-sampled light triangles), 256 spp
  c4   synthetic 10M-triangle field, 3840x2160, 128 spp, image rows over the ranks, RCCL gather      (strong scaling)
  c5   Cornell box 1920x1080, 4096 spp IN TOTAL, sample slices of 4096/N spp per rank, RCCL sum-reduce (strong scaling)
c2 / c3 with N > 1 keep `--spp` samples PER RANK (sample slices, weak scaling).

N > 1: one process per GPU.  Under torch.distributed.run the ranks come from the environment; started plainly as
`python bench.py --gpus N`, bench.py starts the N ranks itself (a torch.distributed.run child process, before
anything in this process touches the GPU) and exits with its code.

Prints ONE JSON line on rank 0.  `roofline.frac` is the largest THROUGHPUT fraction of the dominant kernel -- valu, hbm,
lds or l2, achieved rate over that unit's peak -- and `bound` names it, or says "latency" when none exceeds 0.5 (then
`wait_frac` = SQ_WAIT_ANY / SQ_WAVE_CYCLES and `lane_util` say where the time goes).  See DESIGN.md section 4:
  valu  useful vector lane-operations (SQ_THREAD_CYCLES_VALU, active lanes only) per second against the chip's issue
        peak for plain f32 vector instructions (tools/micro/valu_peak.hip)           = issue_util x lane_util
  hbm   measured HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction) per second against 8 TB/s
  lds / l2  bytes the walk requests from the LDS node window / from L1-L2 (device counters of this run) per second
        against 150 TB/s / 34.5 TB/s
  td    (diagnostic only, never `frac`) busy fraction of the texture-data return path (TD_TD_BUSY / elapsed cycles per CU)
The counters are a STORED measurement (profiles/r*_pmc_<workload>.json, written by tools/round_profile.sh with
rocprofv3 in separate --pmc passes); counts per launch are deterministic for a workload and a build, the launch
duration is measured live with HIP events.  `algorithmic_GBps` keeps SURVEY 8(d)'s figure (bytes of the REFERENCE's
walk per second): it is served by the LDS node cache and L2 and may exceed the HBM peak -- it is not a roofline.
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
CLOCK_GHZ = 2.4              # peak engine clock
N_SIMD = 256 * 4
# cycles one SIMD needs per plain f32 wave64 instruction at its best: 2 (SIMD-32; MI355X_MICROARCH.md).  Measured with
# >= 2 waves per SIMD: 2.2-2.4 (profiles/r02_valu_peak.txt, tools/micro/valu_peak.hip); the nominal 2 is the peak used.
VALU_CYCLES_PER_WAVE_INST = 2.0
VALU_PEAK_TLANEOPS = N_SIMD * 64 / VALU_CYCLES_PER_WAVE_INST * CLOCK_GHZ * 1e9 / 1e12

PRESETS = {
    "c2": dict(scene="scene.txt", width=1920, height=1080, spp=64, partition="samples", total_spp=False),
    "c3": dict(scene="scene2.txt", width=1920, height=1080, spp=256, partition="samples", total_spp=False),
    "c3is": dict(scene="scene2_is.txt", width=1920, height=1080, spp=256, partition="samples", total_spp=False),
    "c4": dict(synth_instances=628, width=3840, height=2160, spp=128, partition="rows", total_spp=False),
    "c5": dict(scene="scene_cornellbox.txt", width=1920, height=1080, spp=4096, partition="samples", total_spp=True),
}


def algorithmic_bytes(st, n_pixels):
    """B = 56*N_inner + 8*N_leaf + 36*N_tri + S, counted on the reference's logical data.
    S = per hit 64 B (rest of the 25-float record) + 44 B material, 12 B per texel fetched (material textures
    and environment), 36 B per light-triangle pdf test, + 12 B per pixel accumulator write per launch."""
    return (56 * st.inner_visits + 8 * st.leaf_visits + 36 * st.tri_tests + (64 + 44) * st.hits +
            12 * st.texel_fetches + 36 * st.light_tests + 12 * n_pixels)


def source_sha():
    """Hash of the HIP sources the render kernels are built from: a stored counter file is stale once it changes."""
    h = hashlib.sha256()
    for f in ("rt_core.hip", "rt_device.h", "rt_pmath.h", "rt_pool_kernel.h"):
        with open(os.path.join(ROOT, "ray-tracer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def workload_key(args):
    base = f"synth{args.synth_instances}" if args.synth_instances > 0 else os.path.splitext(os.path.basename(args.scene))[0]
    return f"{base}_{args.width}x{args.height}_{args.spp}spp_{args.bounces}b_{args.mode}_{args.bvh}"


def stored_pmc(key):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{key}.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        d["file"] = os.path.relpath(files[-1], ROOT)
        return d
    except Exception:
        return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: c2 40 -- a timed region of about a second --, c3 / c3is / c4 10, c5 5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default: c2 5, the others 2)")
    ap.add_argument("--config", choices=sorted(PRESETS), default="c2")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel (per rank, except --config c5 / --total-spp: in total)")
    ap.add_argument("--total-spp", action="store_true", default=None, help="--spp is the job's total: each rank renders spp/N")
    ap.add_argument("--bounces", type=int, default=6)
    ap.add_argument("--partition", choices=["samples", "rows"], default=None)
    ap.add_argument("--mode", choices=["wavefront", "megakernel", "lockstep", "pool"], default="megakernel")
    ap.add_argument("--walk", choices=["shortcut", "reference"], default="shortcut",
                    help="fused kernel: shortcut = the library's surface-area tree over the reference's leaves first, the reference's own "
                         "tree for the rays whose result is not certified (default; same sums bit for bit); reference = RT_F_REFWALK")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=32, help="spp of the bounded CPU-baseline sample")
    ap.add_argument("--synth-instances", type=int, default=None, help="use the synthetic instanced field instead of --scene")
    ap.add_argument("--bvh", choices=["reference", "sah"], default="reference",
                    help="sah = the NON-PARITY fast mode (binned-SAH tree, depth <= 24, leaves <= 2), reported separately")
    args = ap.parse_args()
    pre = PRESETS[args.config]
    if args.scene is not None and args.synth_instances is None:
        args.synth_instances = 0
    for k, v in pre.items():
        if getattr(args, k, None) is None:
            setattr(args, k, v)
    if args.steps is None:
        args.steps = {"c2": 40, "c5": 5}.get(args.config, 10)
    if args.warmup is None:
        args.warmup = 5 if args.config == "c2" else 2
    if args.scene is None:
        args.scene = "scene.txt"
    if args.synth_instances is None:
        args.synth_instances = 0
    return args


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as a child torch.distributed.run (this process has
    not touched the GPU and does not), hand its output through and exit with its code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

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
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: a scaling run must have one rank per GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"RCCL sees {dist.get_world_size()} ranks, --gpus says {args.gpus}")

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
    W, H = args.width, args.height
    if args.total_spp and args.partition == "samples":
        if args.spp % world:
            raise SystemExit(f"--spp {args.spp} in total does not divide over {world} ranks")
        spp = args.spp // world           # c5: 4096 spp in total, 4096 / N per rank
    else:
        spp = args.spp                    # samples: per rank (weak); rows: every rank renders all samples of its rows
    strong = world > 1 and (args.partition == "rows" or args.total_spp)

    r = rt.Renderer(local_rank)
    r.upload(scene)
    p = rt.default_params(scene.camera, spp=spp, width=W, height=H, max_bounces=args.bounces)
    p.flags |= rt.MODE_FLAGS[args.mode]
    if args.walk == "reference":
        p.flags |= rt.F_REFWALK
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
    # ... and one with the counters of the walk the timed steps really do (the shortcut tree, where the scene has one)
    p_dev = rt.RenderParams.from_buffer_copy(p_stats)
    p_dev.flags |= rt.F_SHORTCUT_STATS
    st_dev = step(p_dev) if args.mode == "megakernel" and args.walk == "shortcut" else st_counts
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
        step_s = elapsed / max(args.steps, 1)
        ms_per_step = step_s * 1e3
        mrays = total_rays / step_s / 1e6
        k_ms = float(np.mean(kernel_ms))          # all kernels of one frame (HIP events on the render stream)
        abytes = algorithmic_bytes(st_counts, rows * W)
        if args.mode == "wavefront" and st_prof.trace_launches > 0:
            # dominant kernel = rt_trace_kernel: its algorithmic bytes (nodes + leaves + triangles) over the
            # summed duration of its launches in one frame, per launch on both sides of the ratio
            dom_kernel = "rt_trace_kernel"
            dom_bytes = 56 * st_counts.inner_visits + 8 * st_counts.leaf_visits + 36 * st_counts.tri_tests
            dom_ms = st_prof.trace_ms
            launches = st_prof.trace_launches
        elif args.mode in ("megakernel", "pool") and dom_launches > 0:
            # path-parallel fused kernel: everything but the 12 bytes per pixel of the accumulator update, which
            # rt_sum_samples_kernel does; duration of its launches alone, averaged over the timed steps
            dom_kernel = "rt_render_pool_kernel" if args.mode == "pool" else "rt_render_paths_kernel"
            dom_bytes = abytes - 12 * rows * W
            dom_ms, launches = float(np.mean(dom_ms_steps)), dom_launches
        else:
            dom_kernel, dom_bytes, dom_ms, launches = "rt_render_kernel", abytes, k_ms, 1
        launch_s = dom_ms * 1e-3 / launches       # average duration of one launch of the dominant kernel, measured live
        roof = {"kernel": dom_kernel, "launches_per_step": launches, "avg_launch_ms": round(dom_ms / launches, 4),
                "frame_kernels_ms": round(k_ms, 3)}
        # what the device requests per launch, by the level that serves it (RT_F_STATS device counters): inner items
        # from the LDS node cache, the other inner items and every triangle test from L1/L2 (HBM only on a miss)
        lds_b = 64 * st_dev.dev_lds_inner_visits / launches
        l2_b = (64 * (st_dev.dev_inner_visits - st_dev.dev_lds_inner_visits) + 48 * st_dev.dev_tri_tests +
                (64 + 48 + 48) * st_dev.hits + 12 * st_dev.texel_fetches + 64 * st_dev.dev_light_tests) / launches
        roof["device_requests"] = {
            "walk": args.walk if args.mode == "megakernel" else "reference",
            "inner_items_visited": int(st_dev.dev_inner_visits), "of_those_from_lds": int(st_dev.dev_lds_inner_visits),
            "wide_visits": int(st_dev.dev_wide_visits), "triangle_tests_run": int(st_dev.dev_tri_tests), "reference_walk_triangle_tests": int(st_counts.tri_tests),
            "reference_walk_inner_visits": int(st_counts.inner_visits), "shortcut_rewalks": int(st_dev.shortcut_rewalks),
            "scene_has_shortcut_tree": bool(st_dev.dev_inner_visits != st_counts.dev_inner_visits or st_dev.shortcut_rewalks),
            "light_tests_run": int(st_dev.dev_light_tests), "reference_walk_light_tests": int(st_counts.light_tests),
            "lds_GBps": round(lds_b / launch_s / 1e9, 1), "lds_peak_GBps": 150000.0,
            "l2_GBps": round(l2_b / launch_s / 1e9, 1), "l2_peak_GBps": 34500.0}
        roof["algorithmic_bytes_per_launch"] = int(dom_bytes / launches)
        roof["algorithmic_GBps"] = round(dom_bytes / launches / launch_s / 1e9, 1)   # the reference's walk; NOT a roofline (cache resident)
        roof["bytes_per_ray"] = round(abytes / max(st_counts.rays, 1), 1)
        pmc = stored_pmc(workload_key(args)) if world == 1 else None
        fracs = {}
        if pmc:
            c = pmc["per_launch"]
            stale = pmc.get("source_sha") != source_sha()
            roof["pmc"] = {"file": pmc["file"], "git_head": pmc.get("git_head"), "stored_kernel_ms": pmc.get("kernel_trace_avg_ms"),
                           "stale": stale, "note": "counts per launch stored by tools/round_profile.sh; duration measured in this run"}
            if "SQ_INSTS_VALU" in c and "SQ_THREAD_CYCLES_VALU" in c:
                insts, lane_ops = c["SQ_INSTS_VALU"], c["SQ_THREAD_CYCLES_VALU"]
                issue_util = insts * VALU_CYCLES_PER_WAVE_INST / (N_SIMD * CLOCK_GHZ * 1e9 * launch_s)
                lane_util = lane_ops / (64.0 * insts)
                roof["valu"] = {"wave_insts_per_launch": insts, "lane_ops_per_launch": lane_ops,
                                "issue_util": round(issue_util, 4), "lane_util": round(lane_util, 4),
                                "achieved_Tlaneops": round(lane_ops / launch_s / 1e12, 3), "peak_Tlaneops": round(VALU_PEAK_TLANEOPS, 2),
                                "frac": round(issue_util * lane_util, 4),
                                # what four waves per SIMD (this kernel's occupancy) were measured to reach on plain f32 instructions:
                                # 2.36 cycles per wave-instruction (profiles/r02_valu_peak.txt), against the nominal 2 of `issue_util`
                                "issue_util_of_4wave_peak": round(issue_util * 2.36 / VALU_CYCLES_PER_WAVE_INST, 4)}
                fracs["valu"] = issue_util * lane_util
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                hbm_bytes = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0   # KB; gfx950: FETCH_SIZE counts half (MI355X_MICROARCH.md)
                roof["hbm"] = {"measured_bytes_per_launch": int(hbm_bytes), "measured_GBps": round(hbm_bytes / launch_s / 1e9, 1),
                               "peak_GBps": HBM_PEAK_GBS, "frac": round(hbm_bytes / launch_s / 1e9 / HBM_PEAK_GBS, 4)}
                fracs["hbm"] = hbm_bytes / launch_s / 1e9 / HBM_PEAK_GBS
                roof["traffic"] = int(hbm_bytes)
            if "TD_TD_BUSY_sum" in c and "GRBM_GUI_ACTIVE" in c:
                # texture-data return path of the vector memory pipeline (one per CU): busy cycles / elapsed cycles.  A divergent
                # 16-byte request occupies it ~1-2.5 cycles per lane (tools/micro/gather_bench.hip): at ~1 it is the limit
                cyc = c["GRBM_GUI_ACTIVE"] / 8.0            # rocprofv3 sums GRBM over the 8 XCDs
                td = c["TD_TD_BUSY_sum"] / (256.0 * cyc)
                roof["td"] = {"busy_frac": round(td, 4), "ta_busy_frac": round(c.get("TA_TA_BUSY_sum", 0.0) / (256.0 * cyc), 4),
                              "vmem_read_wave_insts_per_launch": c.get("SQ_INSTS_VMEM_RD"), "note": "stored ratio (same pass as the counters)"}
            if "SQ_LDS_IDX_ACTIVE" in c and "GRBM_GUI_ACTIVE" in c:
                roof["lds"] = {"busy_frac": round(c["SQ_LDS_IDX_ACTIVE"] / (256.0 * c["GRBM_GUI_ACTIVE"] / 8.0), 4),
                               "bank_conflict_share": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), 4)}
            for k in ("TCC_HIT_sum", "TCC_MISS_sum", "TD_TD_BUSY_sum", "TA_TA_BUSY_sum", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES",
                      "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_SALU"):
                if k in c:
                    roof.setdefault("counters_per_launch", {})[k] = c[k]
        if pmc:
            # LDS and L1/L2 request rates of the walk really run (device counters of this run) against their peaks
            fracs["lds"] = roof["device_requests"]["lds_GBps"] / roof["device_requests"]["lds_peak_GBps"]
            fracs["l2"] = roof["device_requests"]["l2_GBps"] / roof["device_requests"]["l2_peak_GBps"]
        if fracs:
            # `frac` is a THROUGHPUT fraction: the largest of valu / hbm / lds / l2 (achieved rate over that unit's peak).
            # When none of them exceeds 0.5 no unit is the limit: the kernel waits (dependent node fetches, divergent
            # lanes) and `bound` says "latency", with the share of wave-cycles spent waiting and the lane utilisation.
            # td.busy_frac (share of time the texture-data path has work pending) stays a diagnostic, never `frac`.
            nearest = max(fracs, key=fracs.get)
            ach = {"valu": lambda: (roof["valu"]["achieved_Tlaneops"], roof["valu"]["peak_Tlaneops"], "Tlane-op/s"),
                   "hbm": lambda: (roof["hbm"]["measured_GBps"], HBM_PEAK_GBS, "GB/s"),
                   "lds": lambda: (roof["device_requests"]["lds_GBps"], roof["device_requests"]["lds_peak_GBps"], "GB/s"),
                   "l2": lambda: (roof["device_requests"]["l2_GBps"], roof["device_requests"]["l2_peak_GBps"], "GB/s")}[nearest]()
            roof.update(bound=nearest if fracs[nearest] > 0.5 else "latency", nearest_limit=nearest, achieved=ach[0], peak=ach[1], unit=ach[2],
                        frac=round(fracs[nearest], 4), fracs={k: round(v, 4) for k, v in fracs.items()}, counters="stored")
            c = pmc["per_launch"]
            if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
                roof["wait_frac"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
            if "valu" in roof:
                roof["lane_util"] = roof["valu"]["lane_util"]
        else:
            # no stored counters for this workload: no fraction is claimed (run tools/round_profile.sh <config>)
            roof.update(bound="unmeasured", achieved=None, peak=None, unit=None, frac=None, traffic=None)
        roof.setdefault("traffic", None)
        out = {
            "metric": "Mrays/sec (incl. bounces)",
            "value": round(mrays, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "bundled scene geometry/textures, synthetic camera batch" if args.synth_instances == 0 else "synthetic",
            "config": {"workload": workload, "preset": args.config, "width": W, "height": H, "spp_per_gpu": spp,
                       "spp_total": spp * world if args.partition == "samples" else spp, "max_bounces": args.bounces,
                       "partition": "none" if world == 1 else args.partition, "ranks_seen_by_rccl": world, "mode": args.mode, "walk": args.walk if args.mode == "megakernel" else "reference",
                       "wavefront_iterations": int(st_counts.iterations),
                       "rays_per_step": int(total_rays), "paths_per_step": int(total_paths),
                       "spp_per_sec_1080p": round(total_paths / step_s / (1920 * 1080), 2)},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(rt, scene, args)
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(rt, scene, args):
    """The oracle (CPU restatement of the reference trace/shade code) timed on this box's host cores, on a
    bounded sample of the same workload (same scene and bounce depth; fewer spp and, for large frames, a smaller
    frame, sized for 10-30 s of CPU work)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob  # baseline leg only
    # every core this process may run on, capped at the share the GPU pool allots one box (it exports OMP_NUM_THREADS=16 and
    # kills runs that oversubscribe; os.cpu_count() there is the whole machine's)
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(usable, int(os.environ.get("OMP_NUM_THREADS", str(usable)))))
    w, h, spp = args.width, args.height, min(args.cpu_spp, args.spp)
    if args.config in ("c3", "c3is"):         # heavier per ray than c2: quarter frame, 16 spp
        w, h, spp = w // 2, h // 2, min(16, spp)
    elif args.config == "c4":                 # 10M triangles, 150 per leaf in the oracle's walk: quarter frame, 4 spp
        w, h, spp = w // 2, h // 2, min(4, spp)
    p = rt.default_params(scene.camera, spp=spp, width=w, height=h, max_bounces=args.bounces, flags=rt.F_DEFAULT | rt.F_STATS)
    t0 = time.perf_counter()
    _, st = ob.render(scene, p, n_threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(st.rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "cores_visible": usable, "kind": "port",
            "sample": f"same scene at {w}x{h}, {spp} of {args.spp} spp, {st.rays} rays in {dt:.1f} s"}


if __name__ == "__main__":
    main()
