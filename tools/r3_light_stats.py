#!/usr/bin/env python3
"""GPU box: device counters of the shortcut walk on scene2_is.txt (light tree on / off via RT_NO_LIGHT_TREE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from __graft_entry__ import load_package
rt = load_package()
sc = rt.Scene.load(sys.argv[1] if len(sys.argv) > 1 else "scene2_is.txt")
r = rt.Renderer(0)
r.upload(sc)
p = rt.default_params(sc.camera, spp=8, width=960, height=540, flags=rt.F_DEFAULT | rt.F_STATS | rt.F_SHORTCUT_STATS)
st = r.render(p)
print({k: v for k, v in st.as_dict().items()})
p.flags = rt.F_DEFAULT
for _ in range(3):
    r.reset(); st = r.render(p)
print("kernel ms", st.kernel_ms)
