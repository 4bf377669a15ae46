#!/usr/bin/env python3
"""GPU box: wall time of the config-4 scene pipeline: host assembly (instances, BVH), upload (layout conversion + H2D), first frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first: one HIP runtime)
from __graft_entry__ import load_package
rt = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 628
t0 = time.perf_counter()
sc = rt.Scene.synth_field(n, env_path=os.path.join(rt.ASSET_DIR, "cape_hill_1k.hdr"), env_upsample=4)
t1 = time.perf_counter()
r = rt.Renderer(0)
t2 = time.perf_counter()
r.upload(sc)
t3 = time.perf_counter()
p = rt.default_params(sc.camera, spp=1, width=3840, height=2160, flags=rt.F_DEFAULT)
r.render(p)
t4 = time.perf_counter()
st = r.render(p)
print(f"{sc.n_tris} triangles: host assembly {t1 - t0:.2f} s, context {t2 - t1:.2f} s, upload {t3 - t2:.2f} s, "
      f"first frame (1 spp) {t4 - t3:.3f} s, second {st.kernel_ms:.1f} ms")
