#!/usr/bin/env python3
"""Generate tests/golden/conv_<scene>.npz: the ORACLE side of the converged-image test on the scenes BASELINE.json
names as they are configured (Cornell box: its small light is found by chance, material.h:205-213; scene.txt under
the HDR environment, main.cu:374-383).

One seed of 1 024 spp scatters by 2-3 % in mean luminance on these scenes (fireflies), so a single pair of seeds
cannot carry SURVEY 8(c)(ii)'s fixed 1 % bound.  K independent oracle seeds can: the standard error of their mean is
sigma / sqrt(K).  Stored per scene: the K per-seed mean luminances, the mean image of the K seeds, and the images
of the first two seeds (refA / refB of the RMSE criterion).  Build container only (about 15 min on 8 cores):
    make && python tools/make_converged_golden.py [K]
The `-m gpu` test (tests/test_gpu_converged.py) renders OTHER seeds on the GPU and compares against these arrays;
nothing here is read from /root/reference and no GPU is needed."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402

rt = ob.rt
OUT = os.path.join(ROOT, "tests", "golden")
LUMA = np.float64([0.2126, 0.7152, 0.0722])
SIZE, SPP, BOUNCES = 128, 1024, 6
SEED0 = 100000                      # oracle seeds SEED0 + k; the GPU test uses 200000 + k


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    threads = int(os.environ.get("ORC_THREADS", "8"))
    for name in ("scene_cornellbox.txt", "scene.txt"):
        sc = rt.Scene.load(name)
        p = rt.default_params(sc.camera, spp=SPP, width=SIZE, height=SIZE, max_bounces=BOUNCES, flags=rt.F_DEFAULT)
        lum = np.zeros(K, np.float64)
        mean = np.zeros((SIZE, SIZE, 3), np.float64)
        first = []
        t0 = time.time()
        for k in range(K):
            q = rt.RenderParams.from_buffer_copy(p)
            q.seed = SEED0 + k
            acc, _ = ob.render(sc, q, n_threads=threads)
            img = acc / np.float32(SPP)
            lum[k] = float((img.reshape(-1, 3).astype(np.float64) @ LUMA).mean())
            mean += img
            if k < 2:
                first.append(img.astype(np.float32))
            if k % 8 == 7:
                print(f"{name}: {k + 1}/{K} seeds, {time.time() - t0:.0f} s, lum {lum[:k + 1].mean():.6f} "
                      f"+- {lum[:k + 1].std(ddof=1) / lum[:k + 1].mean() * 100:.2f} % per seed", flush=True)
        np.savez_compressed(os.path.join(OUT, "conv_" + name.replace(".txt", "") + ".npz"),
                            size=SIZE, spp=SPP, bounces=BOUNCES, flags=np.uint32(rt.F_DEFAULT), seed0=SEED0,
                            lum=lum, mean_image=(mean / K).astype(np.float32), ref_a=first[0], ref_b=first[1])


if __name__ == "__main__":
    main()
