#!/usr/bin/env python3
"""GPU box: differential fuzz of the closest-hit path (leaf sub-trees, tie rule, padded boxes) against the oracle.
Random triangle soups / lattices / coincident copies at random scales and tree depths, rays of every awkward kind.
usage: tools/gpu_fuzz.py [n_scenes] [first_seed]      prints one line per mismatch and a summary; exit 1 on mismatch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import oracle_binding as ob
rt = ob.rt; ob.lib()
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
os.environ["RT_SUB_MIN_TRIS"] = "0"
# how the shortcut tree is built, scene after scene: over runs (default) / over whole leaves / over a frontier 1-3 levels below the leaves /
# over single triangles without sub-trees (what scenes too small for sub-trees get)
SHORTCUT_MODES = [{}, {"RT_SHORT_LEAVES": "1", "RT_SHORT_CUT": "0"}, {"RT_SHORT_LEAVES": "1", "RT_SHORT_CUT": None}, {"RT_SUB_MIN_TRIS": "100000000"}]
gpu = rt.Renderer(0)
bad = 0
t_start = time.time()
for it in range(n_scenes):
    rng = np.random.default_rng(seed0 + it)
    for k_ in ("RT_SHORT_LEAVES", "RT_SHORT_CUT"):
        os.environ.pop(k_, None)
    os.environ["RT_SUB_MIN_TRIS"] = "0"
    for k_, v_ in SHORTCUT_MODES[it % 4].items():
        os.environ[k_] = v_ if v_ is not None else str(1 + it // 4 % 3)
    kind = rng.integers(0, 6)
    n = int(rng.integers(30, 6000))
    scale = float(10.0 ** rng.uniform(-3, 3))
    if kind == 0:      # soup
        v = rng.uniform(-2, 2, (n, 1, 3)) + rng.normal(0, rng.choice([0.01, 0.05, 0.3, 1.0]), (n, 3, 3))
    elif kind == 1:    # thin slivers and big triangles mixed
        v = rng.uniform(-2, 2, (n, 1, 3)) + rng.normal(0, 1, (n, 3, 3)) * rng.choice([1e-3, 0.1, 2.0], (n, 1, 1))
    elif kind == 2:    # axis-aligned lattice quads (flat boxes, shared edges and vertices)
        g = rng.integers(-4, 5, (n, 3)).astype(float) * 0.5
        ax = rng.integers(0, 3, n)
        e1 = np.eye(3)[(ax + 1) % 3] * 0.5; e2 = np.eye(3)[(ax + 2) % 3] * 0.5
        v = np.stack([g, g + e1, g + e1 + e2], 1)
    elif kind == 3:    # coincident copies
        m = max(n // 4, 8)
        base = rng.uniform(-2, 2, (m, 1, 3)) + rng.normal(0, 0.3, (m, 3, 3))
        v = np.concatenate([base] * 4)[rng.permutation(4 * m)]
    elif kind == 4:    # two or three differently triangulated sheets in ONE plane: near ties between leaves everywhere (shortcut rewalks)
        ax = int(rng.integers(0, 3)); h = float(rng.uniform(-1, 1)); vs = []
        for _ in range(int(rng.integers(2, 4))):
            nx, nz = int(rng.integers(3, 30)), int(rng.integers(3, 30))
            xs, zs = np.linspace(-2, 2, nx + 1), np.linspace(-2, 2, nz + 1)
            X, Z = np.meshgrid(xs, zs, indexing="ij")
            def P(i, j):
                q = np.zeros(X[:-1, :-1].shape + (3,)); q[..., ax] = h; q[..., (ax + 1) % 3] = X[i, j]; q[..., (ax + 2) % 3] = Z[i, j]
                return q.reshape(-1, 3)
            a, b, c, d_ = P(np.s_[:-1], np.s_[:-1]), P(np.s_[1:], np.s_[:-1]), P(np.s_[1:], np.s_[1:]), P(np.s_[:-1], np.s_[1:])
            vs += [np.stack([a, b, c], 1), np.stack([a, c, d_], 1)]
        v = np.concatenate(vs)
    else:              # mirror-image pairs in the plane x = y sharing the edge that lies in it: EXACT ties between leaves for rays in that plane
        m = max(n // 2, 8)
        pq = rng.uniform(-2, 2, (m, 2, 2)); r = rng.uniform(-2, 2, (m, 3)); r[:, 1] = r[:, 0] + rng.choice([-1, 1], m) * rng.uniform(0.3, 1.0, m)
        Pm, Qm = np.stack([pq[:, 0, 0], pq[:, 0, 0], pq[:, 0, 1]], 1), np.stack([pq[:, 1, 0], pq[:, 1, 0], pq[:, 1, 1]], 1)
        v = np.concatenate([np.stack([Pm, Qm, r], 1), np.stack([Pm, Qm, r[:, [1, 0, 2]]], 1)])
    v = v * scale
    n = len(v)
    t = np.zeros((n, 25), np.float32)
    t[:, 0:9] = v.reshape(n, 9)
    t[:, 9:18] = np.tile([0, 0, 1.0], (n, 3))
    mat = rt.Material(); mat.type = 3; mat.color = 0x808080; mat.image_offset[:] = [-1, -1, -1]
    depth = int(rng.choice([0, 1, 2, 4, 8, 16]))
    sc = rt.Scene.from_records(t, [mat], bvh_max_depth=depth, bvh_min_tris=int(rng.choice([1, 4])))
    vf = t[:, 0:9].reshape(n, 3, 3).astype(np.float64)
    k = 4000
    def on_tri(cnt):
        return np.einsum("kj,kjc->kc", rng.dirichlet([1, 1, 1], cnt), vf[rng.integers(0, n, cnt)])
    dirs = rng.normal(0, 1, (k, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    axis_dirs = np.eye(3)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1))
    far = rng.normal(0, 1, (k, 3)) * scale * float(10.0 ** rng.uniform(0, 4.5))
    toward = on_tri(k) - far; toward /= np.maximum(np.linalg.norm(toward, axis=1, keepdims=True), 1e-30)
    vert = vf[rng.integers(0, n, k), rng.integers(0, 3, k)]
    edge = vert + (vf[rng.integers(0, n, k), rng.integers(0, 3, k)] - vert) * 0.0    # (vertices again: shared corners)
    # rays inside the plane x = y (kind 5: they meet the shared edges; elsewhere just more rays)
    tgt = on_tri(k); tgt[:, 1] = tgt[:, 0]
    o2 = rng.uniform(-4, 4, (k, 2)) * scale
    org_p = np.stack([o2[:, 0], o2[:, 0], o2[:, 1]], 1).astype(np.float32)
    d_p = tgt.astype(np.float32) - org_p; d_p[:, 1] = d_p[:, 0]
    if kind == 5:      # aim at the shared edges themselves
        w = rng.uniform(0.05, 0.95, (k, 1)); e = rng.integers(0, n // 2, k)
        tgt = (vf[e, 0] * w + vf[e, 1] * (1 - w)).astype(np.float32); tgt[:, 1] = tgt[:, 0]
        d_p = tgt - org_p; d_p[:, 1] = d_p[:, 0]
    rays = np.concatenate([
        np.concatenate([org_p, d_p], 1),
        np.concatenate([on_tri(k) + 1e-5 * scale * dirs, dirs], 1),
        np.concatenate([on_tri(k), axis_dirs], 1),
        np.concatenate([vert, dirs * rng.uniform(0.1, 5.0, (k, 1))], 1),
        np.concatenate([far, toward], 1),
        np.concatenate([far, (vert - far) / np.maximum(np.linalg.norm(vert - far, axis=1, keepdims=True), 1e-30)], 1),   # aimed at vertices
        np.concatenate([np.round(far / (0.5 * scale)) * 0.5 * scale, axis_dirs], 1),
    ]).astype(np.float32)
    op, ot, opos = ob.trace_rays(sc, rays, rt.F_BVH)[:3]
    gpu.upload(sc)
    gp, gt, gpos = gpu.trace_rays(rays, rt.F_BVH)[:3]
    mism = np.flatnonzero((gp != op) | (gt != ot))
    if mism.size:
        bad += 1
        i = mism[0]
        print(f"MISMATCH seed {seed0 + it} kind {kind} n {n} depth {depth} scale {scale:.3g}: {mism.size} rays, first {i}: gpu {gp[i]} {gt[i]} oracle {op[i]} {ot[i]} ray {rays[i]}", flush=True)
    if it % 10 == 9:
        print(f"{it + 1} scenes, {bad} with mismatches, {time.time() - t_start:.0f} s", flush=True)
gpu.close()
print("fuzz done:", n_scenes, "scenes,", bad, "with mismatches")
sys.exit(1 if bad else 0)
