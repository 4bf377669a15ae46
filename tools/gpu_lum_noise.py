#!/usr/bin/env python3
"""GPU box: how much does the mean luminance of a 128 x 128 render move from seed to seed?  (Sizes the spp of
tests/test_gpu_converged.py against SURVEY's fixed 1 % bound: the bound is only meaningful where the estimator's own
seed-to-seed scatter is well below it.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rt = load_package()
LUMA = np.float64([0.2126, 0.7152, 0.0722])
r = rt.Renderer(0)
CASES = [("scene_cornellbox.txt", rt.F_DEFAULT, "as configured"), ("scene.txt", rt.F_DEFAULT, "as configured"),
         ("scene.txt", rt.F_DEFAULT & ~rt.F_SKYBOX, "flat backdrop instead of the HDR environment"),
         ("scene2.txt", rt.F_DEFAULT & ~rt.F_SKYBOX, "flat backdrop")]
if len(sys.argv) > 1:
    CASES = CASES[int(sys.argv[1]):]
for name, flags, what in CASES:
    sc = rt.Scene.load(name)
    r.upload(sc)
    print(name, what)
    for spp in (1024, 4096, 16384):
        lum = []
        for seed in range(100, 112):
            p = rt.default_params(sc.camera, spp=spp, width=128, height=128, seed=seed, flags=flags)
            r.reset(); r.render(p)
            lum.append(float((r.readback(rt.READ_AVERAGE).reshape(-1, 3) @ LUMA).mean()))
        lum = np.array(lum)
        print(f"{name} {spp} spp: mean luminance {lum.mean():.5f}, seed-to-seed std {lum.std(ddof=1) / lum.mean() * 100:.3f} %, max deviation of a pair {np.ptp(lum) / lum.mean() * 100:.3f} %")
