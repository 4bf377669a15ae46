#!/usr/bin/env python3
"""GPU box: differential fuzz of whole renders (every material type, textures, importance-sampled lights, options) against
the oracle, on the random scenes of tests/test_gpu_parity.py.  usage: tools/gpu_fuzz_render.py [n] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import oracle_binding as ob
import test_gpu_parity as T
rt = ob.rt; ob.lib()
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
gpu = rt.Renderer(0)
bad = 0
rewalks = short_rays = 0
visits = {"shortcut": 0, "megakernel": 0}
t0 = time.time()
for it in range(n_scenes):
    seed = seed0 + it
    os.environ["RT_SUB_MIN_TRIS"] = "0" if it % 3 else "256"
    lf = [None, 0.1, 0.4, 0.9][it % 4]
    sc, rng = T.random_scene(rt, seed, light_frac=lf)
    gpu.upload(sc)
    flags = rt.F_BVH | int(rng.choice([0, rt.F_ANTIALIAS])) | int(rng.choice([0, rt.F_DEFOCUS])) | rt.F_STATS | int(rng.choice([0, 0, rt.F_VOLUME]))
    p = rt.default_params(sc.camera, spp=int(rng.integers(1, 5)), width=int(rng.integers(9, 120)), height=int(rng.integers(9, 90)),
                          max_bounces=int(rng.integers(1, 9)), flags=flags)
    p.cam.aperture = float(rng.choice([0.0, 0.08])); p.cam.focus_distance = 5.0
    p.seed = int(rng.integers(0, 1 << 31)); p.sample_begin = int(rng.integers(0, 1000))
    o, ost = ob.render(sc, p, n_threads=16)
    modes = ["megakernel", "pool"] + (["lockstep", "wavefront"] if it % 5 == 0 and not (flags & rt.F_VOLUME) else [])
    for mode in modes + ["shortcut"]:
        q = rt.RenderParams.from_buffer_copy(p)
        q.flags |= rt.MODE_FLAGS["megakernel"] | rt.F_SHORTCUT_STATS if mode == "shortcut" else rt.MODE_FLAGS[mode]
        gpu.reset(); st = gpu.render(q)
        same = np.array_equal(gpu.readback(), o)
        walk = [] if mode == "shortcut" else ["inner_visits", "leaf_visits", "tri_tests"]   # (the shortcut walk's own visits are not the reference's)
        cnt = all(getattr(st, f) == getattr(ost, f) for f in ["rays", "hits", "texel_fetches", "light_tests"] + walk)
        rewalks += st.shortcut_rewalks; short_rays += st.rays if mode == "shortcut" else 0
        if mode in ("shortcut", "megakernel"): visits[mode] += st.dev_inner_visits
        if not (same and cnt):
            bad += 1
            print(f"MISMATCH seed {seed} mode {mode} lights {len(sc.lights())} tris {sc.n_tris} image {same} counters {cnt}", flush=True)
    if it % 10 == 9:
        print(f"{it + 1} scenes, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
gpu.close()
print("render fuzz done:", n_scenes, "scenes,", bad, "mismatches;", rewalks, "shortcut rewalks in", short_rays, "rays; inner items visited",
      visits["shortcut"], "on the shortcut walk,", visits["megakernel"], "on the reference's")
sys.exit(1 if bad else 0)
