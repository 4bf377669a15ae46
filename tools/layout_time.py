#!/usr/bin/env python3
"""Anywhere (no GPU): stage times of the host-side layout build (RT_UPLOAD_TIMING=1) on the config-4 field, and a hash of
the result -- a change of the builders that is meant to keep the layout must keep the hash.   usage: tools/layout_time.py [instances]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RT_UPLOAD_TIMING", "1")
from __graft_entry__ import load_package
rt = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 628
sc = rt.Scene.synth_field(n, env_path=os.path.join(rt.ASSET_DIR, "cape_hill_1k.hdr"), env_upsample=4)
t0 = time.perf_counter()
info, items, places = rt.device_layout(sc, light_items=True)
t1 = time.perf_counter()
h = hashlib.sha256()
h.update(items.tobytes())
h.update(places.tobytes())
print(f"{sc.n_tris} triangles: layout {t1 - t0:.3f} s, {items.shape[0]} items, depth {info.depth}, sha256 {h.hexdigest()[:16]}")
