"""The device layout with its leaf sub-trees, checked WITHOUT a GPU (rt_layout_create, include/rt.h).

Structure: every triangle sits in the item stream exactly once and is reachable exactly once from the root; a
sub-tree holds the triangles of its reference leaf and nothing else; its padded boxes contain what hangs below
them; the reference's own nodes keep their boxes and ids.  Behaviour: a float32 re-enactment of the device
traversal (csrc/rt_device.h: visit_inner, visit_triangle, the tie rule and the pruning slack) on the item stream
finds the same triangle at the same distance as the oracle's walk over the reference's arrays."""
import numpy as np
import pytest

F = np.float32
EMPTY = 0x3ffffff0


def make_scene(rt, seed, n, depth, coincident=False):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-2, 2, (n, 1, 3))
    v = c + rng.normal(0, rng.choice([0.05, 0.3]), (n, 3, 3))
    if coincident:                                     # every triangle four times, copies far apart in the array
        v = np.concatenate([v] * 4)[rng.permutation(4 * n)]
        n = len(v)
    t = np.zeros((n, 25), np.float32)
    t[:, 0:9] = v.reshape(n, 9)
    t[:, 9:18] = np.tile([0, 0, 1.0], (n, 3))
    m = rt.Material(); m.type = 3; m.color = 0x808080; m.image_offset[:] = [-1, -1, -1]
    return rt.Scene.from_records(t, [m], bvh_max_depth=depth, bvh_min_tris=1), rng


def as_int(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def kinds_of(items, info):
    """Per inner item: 0 the reference's node, 1 sub-tree node, 2 two-child shortcut node, 3 / 4 first / second item of a
    wide (four-child) shortcut node (csrc/rt_device.h)."""
    kind = as_int(items[:info.n_inner, 15]).copy()
    lo = info.short_base
    kind[lo:lo + 2 * info.n_wide:2] = 3
    kind[lo + 1:lo + 2 * info.n_wide:2] = 4
    return kind


def wide_children(items, info, entry):
    """(entry, box) of the children of the wide node whose first item is `entry`."""
    w0, w1 = items[entry], items[entry + 1]
    nk = int(as_int(w1[12:13])[0])
    assert 2 <= nk <= 4 and int(as_int(w1[15:16])[0]) == 3
    out = []
    for i in range(4):
        box = np.array([w0[i], w0[4 + i], w0[8 + i], w1[i], w1[4 + i], w1[8 + i]], F)
        e = int(as_int(w0[12 + i:13 + i])[0])
        if i < nk:
            out.append((e, box))
        else:
            assert e == EMPTY and (box == F(3e38)).all()
    return out


def unpack(items, info):
    inner = items[:info.n_inner]
    tri = items[info.n_inner:]
    boxes = np.stack([np.stack([inner[:, 0], inner[:, 1], inner[:, 8], inner[:, 2], inner[:, 3], inner[:, 9]], 1),
                      np.stack([inner[:, 4], inner[:, 5], inner[:, 10], inner[:, 6], inner[:, 7], inner[:, 11]], 1)], 1)
    entries = as_int(inner[:, 12:14])
    third = as_int(inner[:, 14])
    is_sub = as_int(inner[:, 15]) == 1                                 # (2 = a node of the shortcut tree)
    is_sub[info.short_base:info.short_base + 2 * info.n_wide] = False  # (wide shortcut nodes: two items each, other fields)
    flags, leaf_end, ids = as_int(tri[:, 9]), as_int(tri[:, 10]), as_int(tri[:, 11])
    return boxes, entries, third, is_sub, flags, leaf_end, ids


@pytest.mark.parametrize("seed,n,depth,sweep_max", [(1, 900, 4, None), (2, 3000, 16, None), (3, 400, 1, None), (4, 2500, 2, "64"),
                                                   (5, 1500, 0, "100")])
def test_presorted_sweep_builds_the_same_sub_trees(rt, monkeypatch, seed, n, depth, sweep_max):
    """The sub-tree builder sorts a leaf's triangles once per axis and keeps the orders by stable partition
    (rt_core.hip, sweep_sub); RT_SUB_RESORT selects the form that sorts at every node.  Same item stream, byte for
    byte -- also below median splits (leaves above RT_SUB_SWEEP_MAX), where the sweep starts from a permuted range."""
    monkeypatch.setenv("RT_SUB_MIN_TRIS", "0")
    if sweep_max:
        monkeypatch.setenv("RT_SUB_SWEEP_MAX", sweep_max)
    sc, _ = make_scene(rt, seed, n, depth)
    info_a, items_a, places_a = rt.device_layout(sc)
    monkeypatch.setenv("RT_SUB_RESORT", "1")
    info_b, items_b, places_b = rt.device_layout(sc)
    assert info_a.n_sub == info_b.n_sub > 0 and info_a.n_inner == info_b.n_inner
    assert np.array_equal(as_int(items_a), as_int(items_b)) and np.array_equal(places_a, places_b)


@pytest.mark.parametrize("seed,n,depth,short_sweep_max,runs_max", [(1, 900, 4, None, None), (2, 3000, 16, None, None), (3, 400, 1, None, "0"),
                                                                    (4, 2500, 2, "16", None), (5, 1500, 0, "100", "0"), (6, 5000, 6, "7", None)])
def test_presorted_sweep_builds_the_same_shortcut_tree(rt, monkeypatch, seed, n, depth, short_sweep_max, runs_max):
    """The shortcut tree's builder sorts the leaves of a node of at most RT_SHORT_SWEEP_MAX once per axis and keeps the
    orders by stable partition (rt_core.hip, sweep_short); RT_SHORT_RESORT selects the form that sorts the records at
    every node.  Same item stream, byte for byte -- over runs and over the reference's leaves (RT_SHORT_RUNS_MAX=0), and
    below binned splits (nodes above the sweep bound), where the sweep starts from a partitioned range."""
    monkeypatch.setenv("RT_SUB_MIN_TRIS", "0")
    if short_sweep_max:
        monkeypatch.setenv("RT_SHORT_SWEEP_MAX", short_sweep_max)
    if runs_max:
        monkeypatch.setenv("RT_SHORT_RUNS_MAX", runs_max)
    sc, _ = make_scene(rt, seed, n, depth)
    info_a, items_a, places_a = rt.device_layout(sc)
    monkeypatch.setenv("RT_SHORT_RESORT", "1")
    info_b, items_b, places_b = rt.device_layout(sc)
    assert info_a.n_wide == info_b.n_wide > 0 and info_a.n_inner == info_b.n_inner and info_a.short_depth == info_b.short_depth
    assert np.array_equal(as_int(items_a), as_int(items_b)) and np.array_equal(places_a, places_b)


@pytest.mark.parametrize("binary", [False, True])
@pytest.mark.parametrize("seed,n,depth,by_leaves", [(1, 900, 4, False), (2, 3000, 16, False), (3, 400, 1, False), (1, 900, 4, True),
                                                   (2, 3000, 16, True), (4, 200, 3, "single"), (1, 900, 4, "cut2"), (3, 400, 1, "cut3")])
def test_item_stream_invariants(rt, monkeypatch, seed, n, depth, by_leaves, binary):
    if binary:
        monkeypatch.setenv("RT_SHORT_BINARY", "1")                     # the two-child form of the shortcut tree (experiments)
    single = by_leaves == "single"                                     # a scene too small for sub-trees: shortcut tree over single triangles
    cut = int(by_leaves[3:]) if isinstance(by_leaves, str) and by_leaves.startswith("cut") else 0
    by_leaves = by_leaves is True or cut > 0
    monkeypatch.setenv("RT_SUB_MIN_TRIS", "256" if single else "0")
    if by_leaves:                                                      # the shortcut tree over whole reference leaves (huge scenes),
        monkeypatch.setenv("RT_SHORT_LEAVES", "1")                     # or over what stands `cut` levels below them in their sub-trees
        monkeypatch.setenv("RT_SHORT_CUT", str(cut))
    sc, _ = make_scene(rt, seed, n, depth)
    info, items, places = rt.device_layout(sc)
    boxes, entries, third, is_sub, flags, leaf_end, ids = unpack(items, info)
    nodes, tris = sc.nodes(), sc.tris()
    assert info.n_tris == n and info.n_sub == is_sub.sum() and (single or info.n_sub > 0) and info.root_entry == 0
    assert bool(info.short_single) == single and (not single or binary or 3 * info.n_short + 1 >= n) and (not single or not binary or info.n_short == n - 1)
    assert (info.n_wide == 0) == binary and (binary or info.n_wide == info.n_short)
    assert sorted(places.tolist()) == list(range(n)) and np.array_equal(ids[places], np.arange(n))
    v = tris[:, :9].reshape(n, 3, 3)
    placed = v[ids]                                                    # triangle geometry in item order
    got = items[info.n_inner:, :9].reshape(n, 3, 3)
    assert np.array_equal(got[:, 0], placed[:, 0]) and np.array_equal(got[:, 1], placed[:, 1] - placed[:, 0])
    # reference nodes: boxes and ids as uploaded, children in the reference's A / B order
    ref_inner = [i for i in range(len(nodes)) if nodes[i]["child_a"] != -1]
    kind = kinds_of(items, info)                                       # 0 the reference's node, 1 sub-tree node, 2 shortcut node, 3 / 4 wide shortcut node
    by_ref = {int(third[k]): k for k in range(info.n_inner) if kind[k] == 0}
    assert sorted(by_ref) == ref_inner
    for i in ref_inner:
        for side, ch in enumerate((nodes[i]["child_a"], nodes[i]["child_b"])):
            assert np.array_equal(boxes[by_ref[i], side], np.concatenate([nodes[ch]["bbox_min"], nodes[ch]["bbox_max"]]))
    # one walk over the whole structure: every triangle once, sub-trees confined to their reference leaf
    seen = np.zeros(n, int)
    depth_seen = 0

    def leaf_run(first):
        p, out = first, []
        while True:
            out.append(p)
            if flags[p] & 1:
                return out
            p += 1

    def walk(entry, d, in_sub):
        nonlocal depth_seen
        depth_seen = max(depth_seen, d)
        if entry == EMPTY:
            return []
        if entry >= info.n_inner:
            run = leaf_run(entry - info.n_inner)
            for p in run:
                seen[p] += 1
                assert bool(flags[p] & 4) == in_sub
            return run
        sub = bool(is_sub[entry])
        assert sub or not in_sub                                       # no reference node below a sub-tree node
        below = []
        for side in range(2):
            got_below = walk(int(entries[entry, side]), d + 1, sub)
            if sub:                                                    # the padded box holds everything below it
                pts = placed[got_below].reshape(-1, 3)
                assert (pts >= boxes[entry, side, :3]).all() and (pts <= boxes[entry, side, 3:]).all()
                assert (boxes[entry, side, :3] <= pts.min(0) - 0.5 * info.box_pad).all()
                assert (boxes[entry, side, 3:] >= pts.max(0) + 0.5 * info.box_pad).all()
            below += got_below
        if sub and not in_sub:                                         # root of a sub-tree: one reference leaf, all of it
            ends = set(leaf_end[below].tolist())
            assert len(ends) == 1 and len(below) == third[entry] > 0
            e = ends.pop()
            assert sorted(ids[below].tolist()) == list(range(e - len(below), e))
        elif sub:
            assert third[entry] == 0
        return below

    walk(info.root_entry, 0, False)
    assert (seen == 1).all()
    assert depth_seen <= info.depth <= 24
    # the shortcut tree: every non-empty reference leaf hangs below it exactly once, with what the reference's tree has
    # below that leaf (its first triangle or its sub-tree); padded boxes hold everything below them
    assert info.n_short == (kind == (2 if binary else 3)).sum() > 0 and kind[info.short_root] == (2 if binary else 3)
    assert (kind == 4).sum() == info.n_wide and (binary or info.short_root == info.short_base)   # (the root has the biggest box)
    assert info.short_base + 1 <= info.n_inner and info.short_base <= 1024
    leaf_entries = {}
    for k in range(info.n_inner):
        if kind[k] == 0:
            for side, ch in enumerate((nodes[int(third[k])]["child_a"], nodes[int(third[k])]["child_b"])):
                if nodes[ch]["child_a"] == -1 and nodes[ch]["tri_len"] > 0:
                    leaf_entries[int(entries[k, side])] = ch
    seen[:] = 0
    depth_seen = 0
    reached = []

    def walk_short(entry, d):
        nonlocal depth_seen
        if entry >= info.n_inner and (not by_leaves or not binary):    # a run of triangles up to a 'last' flag (or ONE triangle); the wide
                                                                       # form has four-child nodes of its own all the way down to the runs
            depth_seen = max(depth_seen, d)
            run = [entry - info.n_inner] if info.short_single else leaf_run(entry - info.n_inner)
            assert info.short_single or entry == info.n_inner or flags[entry - info.n_inner - 1] & 1
            seen[run] += 1
            return run
        if by_leaves and binary and (entry >= info.n_inner or kind[entry] == 1):  # a leaf, or a node / sub-leaf of its sub-tree on the cut
            if entry in leaf_entries:
                reached.append(entry)
            if entry >= info.n_inner:
                return walk(entry, d, bool(flags[entry - info.n_inner] & 4))
            return walk(entry, d, third[entry] == 0)
        assert entry < info.n_inner and kind[entry] == (2 if binary else 3)
        below = []
        kids = [(int(entries[entry, side]), boxes[entry, side]) for side in range(2)] if binary else wide_children(items, info, entry)
        for k, (e, box) in enumerate(kids):
            got_below = walk_short(e, d + 1)
            pts = placed[got_below].reshape(-1, 3)
            assert (box[:3] <= pts.min(0) - 0.5 * info.box_pad).all()
            assert (box[3:] >= pts.max(0) + 0.5 * info.box_pad).all()
            below += got_below
        return below

    walk_short(info.short_root, 0)
    assert (seen == 1).all() and (not by_leaves or not binary or cut > 0 or sorted(reached) == sorted(leaf_entries))
    assert info.short_cut == cut and (cut == 0 or not binary or len(reached) < len(leaf_entries))
    assert binary or info.wide_sp_max >= 3
    assert (depth_seen <= info.short_depth <= 32) if binary else (info.short_depth <= 32 and info.wide_sp_max + 3 <= max(info.depth, info.short_depth))
    leaf_of = as_int(items[info.n_inner:, 12])                        # every triangle names its reference leaf
    for nd_i, nd in enumerate(nodes):
        if nd["child_a"] == -1 and nd["tri_len"] > 0:
            first, cnt = nd["tri_offs"] // 25, nd["tri_len"] // 25
            assert (leaf_of[first:first + cnt] == nd_i).all()
    # triangles outside sub-trees stand at their reference place, first / last flags per reference leaf
    for nd in nodes:
        if nd["child_a"] == -1 and nd["tri_len"] > 0:
            first, cnt = nd["tri_offs"] // 25, nd["tri_len"] // 25
            assert (leaf_end[first:first + cnt] == first + cnt).all()
            assert sorted(ids[first:first + cnt].tolist()) == list(range(first, first + cnt))
            if not (flags[first] & 4):
                assert np.array_equal(ids[first:first + cnt], np.arange(first, first + cnt))
                assert flags[first] & 2 and flags[first + cnt - 1] & 1


def trace_on_items(info, items, o, d, shortcut=False, nodes=None):
    """One ray through the item stream the way csrc/rt_device.h does it, in float32."""
    o, d = o.astype(F), d.astype(F)
    with np.errstate(all="ignore"):
        inv = F(1.0) / d
        osum = abs(o[0]) + abs(o[1]) + abs(o[2])
        slack = F(1e-5) * osum * osum if max(abs(o)) <= F(info.cull_far) else F(np.inf)
        inv_d2 = F(1.00001) / (d[0] * d[0] + d[1] * d[1] + d[2] * d[2])

        def box_dist(b):
            mn, mx = b[:3], b[3:]
            if (o >= mn).all() and (o <= mx).all():
                return F(0.0)
            tmin, tmax = F(0.0), F(1e13)
            for k in range(3):
                t1, t2 = (mn[k] - o[k]) * inv[k], (mx[k] - o[k]) * inv[k]
                tmin = np.fmax(tmin, np.fmin(t1, t2))
                tmax = np.fmin(tmax, np.fmax(t1, t2))
            return F(1e13) if tmax < tmin else tmin

        def dot(a, b):
            return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]

        def cross(a, b):
            return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], F)

        def fma(a, b, c):                                                 # (a * b is exact in float64)
            return F(np.float64(a) * np.float64(b) + np.float64(c))

        def search(start, fast):
            min_d2, hit, hit_t = F(1e12), -1, F(0)
            rival, best_leaf = F(1e12), -1
            sub_lim = (min_d2 * F(1.00002) + slack) * inv_d2
            stack, cur, eps = [], start, F(1e-9)
            wide_lo, wide_hi = info.short_base, info.short_base + 2 * info.n_wide
            while True:
                if cur == EMPTY:
                    pass
                elif fast and wide_lo <= cur < wide_hi:                    # rt_device.h: trav_step_wide
                    if len(stack) > info.wide_sp_max:
                        return -2, min_d2, hit_t, rival                    # gave up: the reference's walk decides
                    lim_t = np.sqrt(sub_lim, dtype=F) * F(1.00001)
                    noi = -(o * inv)
                    kids = []
                    w0, w1 = items[cur], items[cur + 1]
                    for i in range(4):
                        lo3 = [fma(w0[4 * k + i], inv[k], noi[k]) for k in range(3)]
                        hi3 = [fma(w1[4 * k + i], inv[k], noi[k]) for k in range(3)]
                        tmin = max(max(min(lo3[0], hi3[0]), min(lo3[1], hi3[1])), max(min(lo3[2], hi3[2]), F(0)))
                        tmax = min(min(max(lo3[0], hi3[0]), max(lo3[1], hi3[1])), min(max(lo3[2], hi3[2]), lim_t))
                        if tmin <= tmax:
                            kids.append((tmin, i, int(as_int(w0[12 + i:13 + i])[0])))
                    kids.sort()
                    for _, _, e in reversed(kids[1:]):
                        stack.append(e)
                    if kids:
                        cur = kids[0][2]
                        continue
                elif cur < info.n_inner:
                    w = items[cur]
                    bA = np.array([w[0], w[1], w[8], w[2], w[3], w[9]], F)
                    bB = np.array([w[4], w[5], w[10], w[6], w[7], w[11]], F)
                    dA, dB = box_dist(bA), box_dist(bB)
                    dA, dB = dA * dA, dB * dB
                    eA, eB = int(as_int(w[12:13])[0]), int(as_int(w[13:14])[0])
                    lim = sub_lim if as_int(w[15:16])[0] != 0 else min_d2
                    near, far, d_near, d_far = (eA, eB, dA, dB) if dA < dB else (eB, eA, dB, dA)
                    if d_far < lim:
                        stack.append(far)
                    if d_near < lim:
                        cur = near
                        continue
                else:
                    w = items[cur]
                    v0, e1, e2 = w[0:3], w[3:6], w[6:9]
                    p = cross(d, e2)
                    det = dot(e1, p)
                    if not (det > -eps and det < eps):
                        inv_det = F(1.0) / det
                        s = o - v0
                        u = inv_det * dot(s, p)
                        if not (u < 0 or u > 1):
                            q = cross(s, e1)
                            v = inv_det * dot(d, q)
                            if not (v < 0 or u + v > 1):
                                t = inv_det * dot(e2, q)
                                if t > eps:
                                    pos = o + d * t
                                    r = pos - o
                                    nd = r[0] * r[0] + r[1] * r[1] + r[2] * r[2]
                                    tie = False
                                    if nd == min_d2 and hit >= 0:
                                        held = items[info.n_inner + hit]
                                        tie = as_int(held[10:11])[0] == as_int(w[10:11])[0] and as_int(w[11:12])[0] < as_int(held[11:12])[0]
                                    this_leaf = int(as_int(w[10:11])[0])
                                    if nd < min_d2 or tie:
                                        if this_leaf != best_leaf:
                                            rival, best_leaf = np.fmin(rival, min_d2), this_leaf
                                        min_d2, hit, hit_t = nd, cur - info.n_inner, t
                                        sub_lim = (min_d2 * F(1.00002) + slack) * inv_d2
                                    elif this_leaf != best_leaf:
                                        rival = np.fmin(rival, nd)
                    if not (as_int(w[9:10])[0] & 1) and not (fast and info.short_single):
                        cur += 1                                           # (a shortcut search over single triangles pops after one test)
                        continue
                if not stack:
                    break
                cur = stack.pop()
            return hit, min_d2, hit_t, rival

        rewalked = False
        if shortcut:                                                   # rt_device.h: short_begin, short_certified
            root = nodes[0]
            root_ok = box_dist(np.concatenate([root["bbox_min"], root["bbox_max"]]).astype(F)) <= F(1e12)
            if not root_ok:
                return -1, F(-1), False
            usable = np.isfinite(inv).all() and np.isfinite(slack) and (info.n_wide == 0 or np.isfinite(o * inv).all())
            hit, min_d2, hit_t, rival = search(info.short_root if usable else info.root_entry, usable)
            if usable and hit == -2:                                   # the wide walk ran out of stack room
                rewalked = True
                hit, min_d2, hit_t, rival = search(info.root_entry, False)
            elif usable and hit >= 0:
                leaf = nodes[int(as_int(items[info.n_inner + hit][12:13])[0])]
                bd = box_dist(np.concatenate([leaf["bbox_min"], leaf["bbox_max"]]).astype(F))
                bd2 = bd * bd
                if not (bd2 < rival and min_d2 < rival and bd2 <= min_d2 * F(1.00001) + F(0.5) * slack):
                    rewalked = True
                    hit, min_d2, hit_t, rival = search(info.root_entry, False)
        else:
            hit, min_d2, hit_t, rival = search(info.root_entry, False)
        if hit < 0:
            return (-1, F(-1), rewalked) if shortcut else (-1, F(-1))
        r = (o + d * hit_t) - o
        out = int(as_int(items[info.n_inner + hit][11:12])[0]), np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], dtype=F)
        return out + (rewalked,) if shortcut else out


@pytest.mark.parametrize("seed,n,depth,coincident,by_leaves", [(11, 700, 3, False, False), (12, 1500, 16, False, False), (13, 150, 2, True, False),
                                                               (11, 700, 3, False, True), (13, 150, 2, True, True),
                                                               (14, 200, 3, False, "single"), (15, 40, 2, True, "single"),
                                                               (11, 700, 3, False, "cut2")])
def test_float32_walk_of_the_item_stream_equals_the_oracle(ob, rt, monkeypatch, seed, n, depth, coincident, by_leaves):
    single = by_leaves == "single"                                                 # (tiny scene: shortcut tree over single triangles)
    monkeypatch.setenv("RT_SUB_MIN_TRIS", "256" if single else "0")
    if by_leaves is True or by_leaves == "cut2":
        monkeypatch.setenv("RT_SHORT_LEAVES", "1")
        monkeypatch.setenv("RT_SHORT_CUT", "2" if by_leaves == "cut2" else "0")
    sc, rng = make_scene(rt, seed, n, depth, coincident)
    info, items, _ = rt.device_layout(sc)
    assert (info.n_sub > 0) != single and bool(info.short_single) == single
    k = 250
    v = sc.tris()[:, :9].reshape(-1, 3, 3).astype(np.float64)
    target = np.einsum("kj,kjc->kc", rng.dirichlet([1, 1, 1], k), v[rng.integers(0, len(v), k)])
    org = np.concatenate([rng.normal(0, 4, (k // 2, 3)),                           # outside, inside,
                          np.einsum("kj,kjc->kc", rng.dirichlet([1, 1, 1], k - k // 2), v[rng.integers(0, len(v), k - k // 2)])])
    d = target - org + 1e-9                                                        # ... and from points ON triangles
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::5] = np.eye(3)[rng.integers(0, 3, len(d[::5]))]                            # some along the axes
    d[1::7] *= 2.5                                                                 # some not unit vectors
    rays = np.concatenate([org, d], 1).astype(np.float32)
    prim, t = ob.trace_rays(sc, rays, rt.F_BVH)[:2]
    assert (prim >= 0).sum() > k // 3
    nodes, rewalks = sc.nodes(), 0
    assert info.n_short > 0
    for i in range(k):
        p, tt = trace_on_items(info, items, rays[i, :3], rays[i, 3:])
        assert p == prim[i] and (p < 0 or tt == t[i]), (i, p, prim[i], tt, t[i])
        # the shortcut tree first, the reference's tree where the certificate does not hold: the same answer
        p, tt, again = trace_on_items(info, items, rays[i, :3], rays[i, 3:], shortcut=True, nodes=nodes)
        assert p == prim[i] and (p < 0 or tt == t[i]), ("shortcut", i, p, prim[i], tt, t[i], again)
        rewalks += again
    # (coincident copies share a centroid, hence a leaf; in a random soup of overlapping triangles a few per cent of the wide
    # walks also run out of their 12 stack entries and give up -- 4 in a million on scene.txt)
    assert rewalks < k // 10, rewalks


def butterfly_scene(rt, seed, n):
    """Pairs of triangles that are mirror images in the plane x = y and share the edge that lies in it.  Swapping x and y
    maps every product and sum of the triangle test onto one with the same bits, so a ray inside that plane which meets
    the shared edge gets exactly the same distance from both triangles (or misses both): an exact tie between two
    triangles whose centroids, and reference leaves, differ."""
    rng = np.random.default_rng(seed)
    p, q = rng.uniform(-2, 2, (n, 2)), rng.uniform(-2, 2, (n, 2))                  # (x = y, z) of the edge's ends
    r = rng.uniform(-2, 2, (n, 3))
    r[:, 1] = r[:, 0] + rng.choice([-1, 1], n) * rng.uniform(0.3, 1.0, n)          # the third vertex, off the plane
    P, Q = np.stack([p[:, 0], p[:, 0], p[:, 1]], 1), np.stack([q[:, 0], q[:, 0], q[:, 1]], 1)
    a = np.stack([P, Q, r], 1)
    b = np.stack([P, Q, r[:, [1, 0, 2]]], 1)
    v = np.concatenate([a, b]).astype(np.float32)
    t = np.zeros((2 * n, 25), np.float32)
    t[:, 0:9] = v.reshape(2 * n, 9)
    t[:, 9:18] = np.tile([0, 0, 1.0], (2 * n, 3))
    m = rt.Material(); m.type = 3; m.color = 0x808080; m.image_offset[:] = [-1, -1, -1]
    sc = rt.Scene.from_records(t, [m], bvh_max_depth=16, bvh_min_tris=1)
    k = 3 * n
    w = rng.uniform(0.1, 0.9, (k, 1)).astype(np.float32)
    edge = np.tile(np.arange(n), 3)
    target = (P[edge] * w + Q[edge] * (1 - w)).astype(np.float32)
    target[:, 1] = target[:, 0]
    o2 = rng.uniform(-4, 4, (k, 2)).astype(np.float32)
    org = np.stack([o2[:, 0], o2[:, 0], o2[:, 1]], 1)
    d = target - org
    d[:, 1] = d[:, 0]
    return sc, np.concatenate([org, d], 1).astype(np.float32)


def test_exact_ties_across_leaves_walk_the_reference_tree(ob, rt, monkeypatch):
    """Which of two triangles at exactly the same distance the reference keeps depends on the order of ITS walk: the
    shortcut walk must not decide that (rival == best fails the certificate) and the reference's tree is walked."""
    monkeypatch.setenv("RT_SUB_MIN_TRIS", "0")
    sc, rays = butterfly_scene(rt, 21, 120)
    info, items, _ = rt.device_layout(sc)
    assert info.n_short > 0
    prim, t = ob.trace_rays(sc, rays, rt.F_BVH)[:2]
    nodes, rewalks, hits = sc.nodes(), 0, 0
    leaf_of = as_int(items[info.n_inner:, 12])
    for i in range(0, len(rays), 2):
        p, tt, again = trace_on_items(info, items, rays[i, :3], rays[i, 3:], shortcut=True, nodes=nodes)
        assert p == prim[i] and (p < 0 or tt == t[i]), (i, p, prim[i], tt, t[i], again)
        rewalks += again
        hits += p >= 0
    assert hits > 60 and rewalks > 20, (hits, rewalks)


@pytest.mark.parametrize("scale,dist", [(1e-3, 1.0), (1.0, 3e4), (1e3, 10.0), (30.0, 2e4)])
def test_float32_walk_at_extreme_scales(ob, rt, monkeypatch, scale, dist):
    """Tiny and huge scenes, origins far outside the range in which the padded boxes may cull (rt_layout_info.cull_far)."""
    monkeypatch.setenv("RT_SUB_MIN_TRIS", "0")
    rng = np.random.default_rng(int(scale * 3 + dist) % 997)
    n = 500
    v = (rng.normal(0, 1.0, (n, 1, 3)) + rng.normal(0, 0.1, (n, 3, 3))) * scale
    t = np.zeros((n, 25), np.float32)
    t[:, 0:9] = v.reshape(n, 9)
    t[:, 9:18] = np.tile([0, 0, 1.0], (n, 3))
    m = rt.Material(); m.type = 3; m.color = 0x808080; m.image_offset[:] = [-1, -1, -1]
    sc = rt.Scene.from_records(t, [m], bvh_max_depth=2, bvh_min_tris=1)
    info, items, _ = rt.device_layout(sc)
    assert info.n_sub > 0
    k = 160
    d0 = rng.normal(0, 1, (k, 3)); d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    org = d0 * dist * scale + rng.normal(0, 0.3 * scale, (k, 3))
    d = rng.normal(0, 0.6 * scale, (k, 3)) - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d], 1).astype(np.float32)
    prim, tt = ob.trace_rays(sc, rays, rt.F_BVH)[:2]
    assert (prim >= 0).sum() > k // 10
    nodes = sc.nodes()
    for i in range(k):
        p, td = trace_on_items(info, items, rays[i, :3], rays[i, 3:])
        assert p == prim[i] and (p < 0 or td == tt[i]), (i, p, prim[i])
        p, td, _ = trace_on_items(info, items, rays[i, :3], rays[i, 3:], shortcut=True, nodes=nodes)
        assert p == prim[i] and (p < 0 or td == tt[i]), ("shortcut", i, p, prim[i])


def test_no_shortcut_tree_over_boxes_that_are_not_nested(rt, monkeypatch):
    """The certificate of the shortcut walk relies on every reference box lying inside its parent's (rt_device.h).  The
    reference's builder makes them so; a caller's tree that does not gets no shortcut tree (and is walked as it is)."""
    monkeypatch.setenv("RT_SUB_MIN_TRIS", "0")
    sc, _ = make_scene(rt, 8, 600, 6)
    assert rt.device_layout(sc)[0].n_short > 0
    child = sc.desc.nodes[sc.desc.nodes[0].child_a]
    keep = child.bbox_max[1]
    child.bbox_max[1] = sc.desc.nodes[0].bbox_max[1] + 1.0            # sticks out of the root's box
    try:
        info = rt.device_layout(sc)[0]
        assert info.n_short == 0 and info.short_root == -1 and info.short_depth == 0 and info.n_sub > 0
    finally:
        child.bbox_max[1] = keep
    monkeypatch.setenv("RT_NO_SHORTCUT", "1")
    assert rt.device_layout(sc)[0].n_short == 0


def test_layout_errors_are_reported(rt):
    sc, _ = make_scene(rt, 5, 40, 3)
    sc.desc.n_tri_floats -= 1
    try:
        with pytest.raises(RuntimeError, match="multiple of 25"):
            rt.device_layout(sc)
    finally:
        sc.desc.n_tri_floats += 1


def test_light_tree_holds_every_light_once(rt, scenes):
    """The LIGHT TREE of a scene with enough importance-sampled triangles (csrc/rt_device.h): wide nodes inside the range of the
    wide shortcut nodes whose leaves are the light items behind the triangles -- every light exactly once, each item the
    light's triangle as the triangle test reads it (v0, e1, e2) with its normal, area and place in the reference's list, and
    every child box holding the triangles below it with the padding."""
    sc = scenes("scene2_is.txt")
    info, items, _ = rt.device_layout(sc, light_items=True)
    lights = sc.lights()
    n = len(lights)
    assert n == 960 == info.n_light_items and info.light_root >= 0
    lo = info.short_base
    assert lo <= info.light_root < lo + 2 * info.n_wide and (info.light_root - lo) % 2 == 0
    first = info.n_inner + info.n_tris + 1
    assert items.shape[0] == first + n and not items[first - 1].any()
    tris = sc.tris()
    seen = np.zeros(n, int)

    def walk(entry):
        if entry >= info.n_inner:
            j = entry - first
            assert 0 <= j < n
            w = items[entry]
            ref = int(as_int(w[11:12])[0])                                 # the light's place in the reference's list
            seen[ref] += 1
            rec = tris[lights[ref] // 25]
            v0, v1, v2 = rec[0:3], rec[3:6], rec[6:9]
            assert np.array_equal(w[0:3], v0) and np.array_equal(w[3:6], v1 - v0) and np.array_equal(w[6:9], v2 - v0)
            assert int(as_int(w[9:10])[0]) == 1                             # one light per leaf: "last" flag
            cr = np.cross((v1 - v0).astype(np.float64), (v2 - v0).astype(np.float64))
            assert abs(w[10] - 0.5 * np.linalg.norm(cr)) <= 1e-5 * w[10]    # area
            assert np.allclose(w[12:15], cr / np.linalg.norm(cr), atol=1e-5)
            return [np.stack([v0, v1, v2])]
        below = []
        for e, box in wide_children(items, info, entry):
            got = walk(e)
            pts = np.concatenate(got)
            assert (box[:3] <= pts.min(0) - 0.5 * info.box_pad).all() and (box[3:] >= pts.max(0) + 0.5 * info.box_pad).all()
            below += got
        return below

    walk(info.light_root)
    assert (seen == 1).all()
    # without enough lights for it to pay there is no light tree
    assert rt.device_layout(scenes("scene.txt"))[0].light_root == -1
