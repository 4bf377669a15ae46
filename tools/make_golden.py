#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/rt_oracle.cpp).

The reference has no tests and cannot be built in this image (DESIGN.md "Oracle"), so these are NOT
reference-produced vectors: they pin the oracle (and, through the GPU tests, the HIP path) against
regressions.  Inputs are regenerated from seeds; outputs are stored.  Run from the repo root:
    make && python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402

rt = ob.rt
OUT = os.path.join(ROOT, "tests", "golden")

SCENES = ["scene_cornellbox.txt", "scene.txt", "scene2.txt", "scene2_is.txt"]
IMG = {"scene_cornellbox.txt": (48, 8, 4), "scene.txt": (48, 8, 6), "scene2.txt": (32, 4, 6), "scene2_is.txt": (24, 2, 6)}


def primary_rays(scene, w, h):
    p = rt.default_params(scene.camera, spp=1, width=w, height=h, flags=rt.F_SKYBOX | rt.F_BVH)
    ys, xs = np.mgrid[0:h, 0:w]
    rec = np.stack([xs.ravel(), ys.ravel(), np.zeros(w * h, np.int64)], 1).astype(np.uint32)
    return ob.debug_op(rt.OP_CAMERA_RAY, rec, p)


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in SCENES:
        sc = rt.Scene.load(name)
        rays = primary_rays(sc, 64, 64)
        prim, t, pos, nrm, cnt = ob.trace_rays(sc, rays, rt.F_BVH, want_counts=True)
        size, spp, bounces = IMG[name]
        p = rt.default_params(sc.camera, spp=spp, width=size, height=size, max_bounces=bounces,
                              flags=rt.F_DEFAULT | rt.F_STATS)
        acc, st = ob.render(sc, p, n_threads=8)
        nodes = sc.nodes()
        np.savez_compressed(
            os.path.join(OUT, name.replace(".txt", "") + ".npz"),
            rays=rays, prim=prim, t=t, normal=nrm, counts=cnt,
            img_size=size, img_spp=spp, img_bounces=bounces, img_raw_sum=acc,
            img_stats=np.array([st.paths, st.rays, st.inner_visits, st.leaf_visits, st.tri_tests, st.hits,
                                st.texel_fetches, st.light_tests], np.uint64),
            n_tris=sc.n_tris, n_nodes=sc.desc.n_nodes,
            node_children=np.stack([nodes["child_a"], nodes["child_b"], nodes["tri_offs"], nodes["tri_len"]], 1)
            if sc.desc.n_nodes <= 64 else np.zeros((0, 4), np.int32),
            node_hash=np.frombuffer(__import__("hashlib").sha256(nodes.tobytes()).digest(), np.uint8),
            tri_hash=np.frombuffer(__import__("hashlib").sha256(sc.tris().tobytes()).digest(), np.uint8))
        print(name, "hits", int((prim >= 0).sum()), "rays", st.rays)


if __name__ == "__main__":
    main()
