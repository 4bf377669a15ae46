#!/usr/bin/env python3
"""Micro-benchmark of the persistent trace kernel through the rt_trace_rays hook: run under
rocprofv3 --kernel-trace and read the rt_trace_kernel<true> durations (one dispatch per n)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rt = load_package()
sc = rt.Scene.load(sys.argv[1] if len(sys.argv) > 1 else "scene.txt")
r = rt.Renderer(0); r.upload(sc)
rng = np.random.default_rng(0); tris = sc.tris()
def rays(n):
    k = rng.integers(0, tris.shape[0], n); b = rng.dirichlet([1, 1, 1], n)
    o = tris[k, 0:3] * b[:, :1] + tris[k, 3:6] * b[:, 1:2] + tris[k, 6:9] * b[:, 2:3]
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o + 1e-5 * d, d], 1).astype(np.float32)
for n in [1 << 21, 1 << 21, 1 << 21]:
    r.trace_rays(rays(n), rt.F_BVH)
    print("n", n, flush=True)
