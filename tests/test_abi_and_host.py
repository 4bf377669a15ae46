"""CPU tests: the C-ABI libraries load and export every declared symbol, and the host scene assembly
(parser / OBJ loader / BVH builder / textures) behaves as the reference's host code does
(src/sceneParser.cpp, src/model.h, src/BVHcreator.cpp, src/noiseTexture.cpp, src/texture.cpp)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", txt)))


def test_core_library_exports_every_declared_symbol(rt):
    lib = rt.core_lib()  # loading needs no GPU
    names = declared_functions("rt.h")
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"librt_amd.so does not export {n}"
    assert sorted(rt.RT_SYMBOLS) == names


def test_scene_library_exports_every_declared_symbol(rt):
    lib = rt.scene_lib()
    names = declared_functions("rt_scene.h")
    for n in names:
        assert hasattr(lib, n), f"librt_scene.so does not export {n}"
    assert sorted(rt.SCENE_SYMBOLS) == names


def test_struct_layouts_match_reference_sizes(rt):
    # sizeof(BVHNode)=40 (BVHcreator.h:23-34), sizeof(Material)=48 (material.h:12-22), sizeof(SceneConfigs)=80
    assert C.sizeof(rt.BvhNode) == 40
    assert C.sizeof(rt.Material) == 48
    assert rt.Material.color.offset == 4 and rt.Material.image_offset.offset == 20
    assert rt.Material.image_width.offset == 32 and rt.Material.image_height.offset == 34
    assert rt.Material.textures_unused.offset == 40
    assert C.sizeof(rt.Camera) == 80


def test_headers_are_plain_c_and_agree_with_the_python_mirror(rt, tmp_path):
    """include/rt.h and include/rt_scene.h are the drop-in boundary for a C or C++ caller: they must
    compile as strict C99 on their own, and the structs a caller fills must have the sizes the ctypes mirror uses."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler on this machine")
    src = tmp_path / "abi.c"
    src.write_text(
        '#include "rt.h"\n#include "rt_scene.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        '    printf("%d %d %d %d %d %d %d\\n", (int)sizeof(rt_render_params), (int)sizeof(rt_render_stats), (int)sizeof(rt_layout_info),\n'
        "           (int)sizeof(rt_bvh_node), (int)sizeof(rt_material), (int)sizeof(rt_camera), RT_ABI_VERSION);\n"
        "    return 0;\n}\n")
    exe = tmp_path / "abi"
    r = subprocess.run([cc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sizes = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes[:6] == [C.sizeof(rt.RenderParams), C.sizeof(rt.RenderStats), C.sizeof(rt.LayoutInfo), C.sizeof(rt.BvhNode),
                         C.sizeof(rt.Material), C.sizeof(rt.Camera)]
    assert sizes[6] == rt.core_lib().rt_abi_version()


def test_no_gpu_means_loud_failure(rt):
    import subprocess, sys
    # in this container (no GPU) rt_create must fail, never fall back to a CPU path
    code = ("import sys; sys.path.insert(0, %r); from __graft_entry__ import load_package; rt = load_package()\n"
            "try:\n    rt.Renderer(0)\nexcept RuntimeError as e:\n    print('RAISED', e)\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert "RAISED" in out.stdout and "no CPU fallback" in out.stdout, out.stdout + out.stderr


# ---------------------------------------------------------------- scenes

def test_bundled_scene_sizes(scenes):
    """Triangle / node counts SURVEY.md 8 measured with the reference's own loader + createBVH."""
    c = scenes("scene_cornellbox.txt")
    assert (c.n_tris, c.desc.n_nodes, c.desc.n_mats, c.desc.n_light_tris) == (24, 17, 10, 0)
    s = scenes("scene.txt")
    assert (s.n_tris, s.desc.n_nodes) == (2889, 1877)
    nodes = s.nodes()
    assert int((nodes["child_a"] == -1).sum()) == 939          # leaves
    assert int(nodes["tri_len"].max()) <= 4 * 25               # <= 4 tris per leaf
    assert s.desc.n_tex_floats == 3932160                      # env 1024x512 + earth 1024x512 + noise 512x512
    assert (s.camera.env_width, s.camera.env_height) == (1024, 512)
    d = scenes("scene2.txt")
    assert d.n_tris == 15922 + 960 + 2 + 960                   # dragon (f v//n) + sphere + plane + sphere


def test_scene_defaults_and_colour_quantisation(scenes):
    s = scenes("scene_cornellbox.txt")
    mats = s.materials()
    # standard_mat_orange: COLOR 1.0 0.52 0.12 -> colorToInt truncation (vec3.h:111-116)
    assert mats[5].color == (255 << 16) | (int(0.52 * 255) << 8) | int(np.float32(0.12) * np.float32(255))
    assert mats[5].type == 5 and mats[2].type == 1 and abs(mats[2].p1 - 10.0) < 1e-6
    assert mats[0].color == 0  # no COLOR line: SceneAssets::Material::color is Vec3() = black
    assert all(m.image_offset[0] == -1 for m in mats)
    cam = s.camera
    assert (cam.width, cam.height, cam.num_samples) == (512, 512, 500)
    assert abs(cam.focal_length - 0.3) < 1e-7


def test_light_list_and_marker(scenes):
    s = scenes("scene2_is.txt")
    lights = s.lights()
    assert lights.size == 960 and np.all(lights % 25 == 0)
    tris = s.tris()
    ids = tris[lights // 25, 24].astype(np.int32)
    assert np.all(ids == 2)                      # light_mat; +1000 marker removed (sceneParser.cpp:462-466)
    assert np.all(tris[:, 24] < 1000)
    assert np.allclose(tris[:, 24] - np.floor(tris[:, 24]), 0.001, atol=2e-4)  # matId + 0.001 (model.h:162)


def test_unknown_command_is_an_error_not_an_exit(rt, tmp_path):
    p = tmp_path / "bad.txt"
    p.write_text("SET_OUTPUT_WIDTH 8\nFROBNICATE 1\n")
    with pytest.raises(RuntimeError, match="unknown command: FROBNICATE"):
        rt.Scene.load(str(p))
    with pytest.raises(RuntimeError, match="could not open"):
        rt.Scene.load(str(tmp_path / "missing.txt"))


def test_blank_line_closes_block_and_comments(rt, tmp_path):
    p = tmp_path / "s.txt"
    p.write_text("# comment\nDEF_MODEL m\nPATH plane.obj\n\nDEF_MATERIAL a\nTYPE 3\nCOLOR 0.5 0.5 0.5\n\n"
                 "DEF_OBJ o\nMODEL m\nMATERIAL a\n\nADD_OBJECT\nOBJECT o\nPOSITION 0 0 -3\nSCALE 2.0\n"
                 "IMPORTANCE_SAMPLE\n\nSET_OUTPUT_WIDTH 32\n")
    s = rt.Scene.load(str(p))
    assert s.n_tris == 2 and s.desc.n_light_tris == 2 and s.camera.width == 32
    assert s.camera.env_width == 0  # no SET_BACKGROUND -> constant background


# ---------------------------------------------------------------- OBJ loader

def obj_load(rt, name, pos=(0, 0, 0), scale=1.0, rot=(0, 0, 0), mat=3):
    L = rt.scene_lib()
    out = C.POINTER(C.c_float)()
    n = C.c_size_t()
    pos = (C.c_float * 3)(*pos)
    rot = (C.c_float * 3)(*rot)
    rc = L.rt_obj_load(os.path.join(rt.ASSET_DIR, name).encode(), pos, scale, rot, mat, C.byref(out), C.byref(n))
    assert rc == 0, L.rt_scene_last_error()
    a = np.ctypeslib.as_array(out, shape=(n.value,)).copy().reshape(-1, 25)
    L.rt_host_free(out)
    return a


def test_obj_plane_records(rt):
    a = obj_load(rt, "plane.obj", pos=(1, 2, 3), scale=2.0)
    assert a.shape == (2, 25)
    # f 2/1/1 3/2/1 1/3/1 : v2=(1,0,1) v3=(-1,0,-1) v1=(-1,0,1); v' = v*2 + (1,2,3)
    assert np.array_equal(a[0, 0:9], np.float32([3, 2, 5, -1, 2, 1, -1, 2, 5]))
    assert np.array_equal(a[0, 9:18], np.float32([0, 1, 0] * 3) * np.float32(1) + np.float32([-0.0, 0, -0.0] * 3))
    assert np.array_equal(a[0, 18:24], np.float32([1, 0, 0, 1, 0, 0]))
    assert a[0, 24] == np.float32(np.float64(np.float32(3)) + 0.001)


def test_obj_rotation_order_and_scale(rt):
    """v' = (R*v)*scale + T with R = Matrix::rotationMatrix (XYZ Euler, matrix.h:14-37); normals R*n only."""
    rot = (0.3, -1.1, 0.7)
    a = obj_load(rt, "cube.obj", pos=(0.5, -2, 4), scale=1.5, rot=rot)
    b = obj_load(rt, "cube.obj")
    cx, sx, cy, sy, cz, sz = [np.float64(f(np.float32(r))) for r in rot for f in (np.cos, np.sin)]
    R = np.array([[cy * cz, -cy * sz, sy],
                  [cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy],
                  [sx * sz - cx * sy * cz, sx * cz + cx * sy * sz, cx * cy]])
    v = b[:, 0:9].reshape(-1, 3).astype(np.float64)
    exp = (v @ R.T) * 1.5 + np.array([0.5, -2, 4])
    assert np.allclose(a[:, 0:9].reshape(-1, 3), exp, atol=2e-6)
    n = b[:, 9:18].reshape(-1, 3).astype(np.float64)
    assert np.allclose(a[:, 9:18].reshape(-1, 3), n @ R.T, atol=1e-6)
    assert np.array_equal(a[:, 18:24], b[:, 18:24])


def test_obj_v_slash_slash_n_faces(rt):
    a = obj_load(rt, "dragon.obj")
    assert a.shape == (15922, 25)
    assert np.all(a[:, 18:24] == 0)                         # no vt in the file -> uv = 0
    nl = np.linalg.norm(a[:, 9:12], axis=1)
    assert np.all(np.abs(nl - 1) < 1e-3)


# ---------------------------------------------------------------- BVH builder

def build(rt, tris, depth=16, min_tris=4):
    L = rt.scene_lib()
    t = np.ascontiguousarray(tris, np.float32).copy()
    nodes = C.POINTER(rt.BvhNode)()
    n = C.c_int32()
    rc = L.rt_bvh_build(t.ctypes.data_as(C.POINTER(C.c_float)), t.size, depth, min_tris, C.byref(nodes), C.byref(n))
    assert rc == 0
    out = [(tuple(nodes[i].bbox_min), tuple(nodes[i].bbox_max), nodes[i].child_a, nodes[i].child_b,
            nodes[i].tri_offs, nodes[i].tri_len) for i in range(n.value)]
    L.rt_host_free(nodes)
    return out, t.reshape(-1, 25)


def py_bvh(tris, max_depth=16, min_tris=4):
    """Independent pure-Python restatement of BVHcreator.cpp (splitNode :86-170, splitRecursive :173-184,
    BFSIndex :41-61, populateBVHNodes :209-237) used to cross-check the C++ builder."""
    f = np.float32
    T = np.asarray(tris, np.float32).reshape(-1, 25)

    def box(idx):
        mn = [f(1e12)] * 3
        mx = [f(-1e12)] * 3
        for i in idx:
            for v in range(3):
                for c in range(3):
                    mn[c] = min(mn[c], T[i, v * 3 + c])
                    mx[c] = max(mx[c], T[i, v * 3 + c])
        return mn, mx

    root = {"idx": list(range(T.shape[0])), "depth": 0, "a": None, "b": None}
    root["mn"], root["mx"] = box(root["idx"])

    def split(n):
        if n["depth"] >= max_depth or len(n["idx"]) <= min_tris or len(n["idx"]) <= 2:
            return
        w = [f(n["mx"][c] - n["mn"][c]) for c in range(3)]
        if w[0] >= w[1] and w[0] >= w[2]:
            ax = 0
        elif w[1] >= w[0] and w[1] >= w[2]:
            ax = 1
        else:
            ax = 2
        mean = f(0)
        for i in n["idx"]:
            mean = f(mean + f(f(T[i, ax] + T[i, 3 + ax]) + T[i, 6 + ax]))
        mean = f(mean / f(len(n["idx"]) * 3))
        A, B = [], []
        for i in n["idx"]:
            c = f(f(f(T[i, ax] + T[i, 3 + ax]) + T[i, 6 + ax]) / f(3.0))
            (A if c < mean else B).append(i)
        for name, idx in (("a", A), ("b", B)):
            ch = {"idx": idx, "depth": n["depth"] + 1, "a": None, "b": None}
            ch["mn"], ch["mx"] = box(idx)
            n[name] = ch
        n["idx"] = []
        split(n["a"])
        split(n["b"])

    split(root)
    order, q = [], [root]
    while q:
        n = q.pop(0)
        n["index"] = len(order)
        order.append(n)
        if n["a"]:
            q.append(n["a"])
            q.append(n["b"])
    out, perm, offs = [], [], 0
    for n in order:
        out.append((tuple(n["mn"]), tuple(n["mx"]), n["a"]["index"] if n["a"] else -1,
                    n["b"]["index"] if n["b"] else -1, offs, len(n["idx"]) * 25))
        perm += n["idx"]
        offs += len(n["idx"]) * 25
    return out, T[perm]


def random_soup(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-5, 5, (n, 1, 3))
    v = c + rng.normal(0, 0.3, (n, 3, 3))
    t = np.zeros((n, 25), np.float32)
    t[:, 0:9] = v.reshape(n, 9)
    t[:, 9:18] = np.tile([0, 1, 0], 3)
    t[:, 24] = np.arange(n) % 7 + 0.001
    return t


@pytest.mark.parametrize("n,seed", [(1, 0), (4, 1), (5, 2), (37, 3), (600, 4)])
def test_bvh_matches_python_restatement(rt, n, seed):
    tris = random_soup(n, seed)
    got_nodes, got_tris = build(rt, tris)
    exp_nodes, exp_tris = py_bvh(tris)
    assert len(got_nodes) == len(exp_nodes)
    for g, e in zip(got_nodes, exp_nodes):
        assert g[2:] == e[2:]
        assert np.array_equal(np.float32(g[0]), np.float32(e[0])) and np.array_equal(np.float32(g[1]), np.float32(e[1]))
    assert np.array_equal(got_tris, exp_tris)


@pytest.mark.parametrize("chunk", [1, 7, 64])
def test_bvh_parallel_chunks_change_nothing(rt, monkeypatch, chunk):
    """Big nodes count / scatter / bound their triangle range in parallel chunks (2^18 triangles by default, far
    above these sizes): forcing tiny chunks must give the same nodes and the same triangle order."""
    tris = random_soup(600, 4)
    exp_nodes, exp_tris = py_bvh(tris)
    monkeypatch.setenv("RT_BVH_CHUNK", str(chunk))
    got_nodes, got_tris = build(rt, tris)
    assert len(got_nodes) == len(exp_nodes)
    for g, e in zip(got_nodes, exp_nodes):
        assert g[2:] == e[2:]
        assert np.array_equal(np.float32(g[0]), np.float32(e[0])) and np.array_equal(np.float32(g[1]), np.float32(e[1]))
    assert np.array_equal(got_tris, exp_tris)
    big = random_soup(20000, 11)
    monkeypatch.delenv("RT_BVH_CHUNK")
    ref_nodes, ref_tris = build(rt, big)
    monkeypatch.setenv("RT_BVH_CHUNK", str(chunk * 13))
    got_nodes, got_tris = build(rt, big)
    assert got_nodes == ref_nodes and np.array_equal(got_tris, ref_tris)


@pytest.mark.parametrize("env", [{"RT_BVH_SINGLE_PASS": "1"}, {"RT_BVH_REGROUP": "0"}, {"RT_BVH_REGROUP": "40"},
                                 {"RT_BVH_REGROUP": "1000000000"}, {"RT_BVH_REGROUP": "300", "RT_BVH_CHUNK": "5"},
                                 {"RT_BVH_SINGLE_PASS": "1", "RT_BVH_REGROUP": "100"}])
def test_bvh_builder_paths_change_nothing(rt, monkeypatch, env):
    """The builder's three ways through a level -- a big node shared by the threads in chunks (boxes folded per chunk), one
    node per task split in a single pass (children's boxes and serial sums formed on the way), and either of them after
    the per-triangle data has been regrouped into id-array order -- give the nodes, boxes and triangle order of the
    Python restatement of BVHcreator.cpp, wherever the switch-over points are put."""
    for n, seed in ((600, 4), (5000, 9)):
        tris = random_soup(n, seed)
        exp_nodes, exp_tris = py_bvh(tris)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got_nodes, got_tris = build(rt, tris)
        for k in env:
            monkeypatch.delenv(k)
        assert len(got_nodes) == len(exp_nodes)
        for g, e in zip(got_nodes, exp_nodes):
            assert g[2:] == e[2:]
            assert np.array_equal(np.float32(g[0]), np.float32(e[0])) and np.array_equal(np.float32(g[1]), np.float32(e[1]))
        assert np.array_equal(got_tris, exp_tris)


def test_bvh_cornell_matches_python_restatement(rt, scenes):
    # rebuild from the (already reordered) cornell records: same records -> same tree shape
    s = scenes("scene_cornellbox.txt")
    got_nodes, got_tris = build(rt, s.tris())
    exp_nodes, exp_tris = py_bvh(s.tris())
    assert [g[2:] for g in got_nodes] == [e[2:] for e in exp_nodes]
    assert np.array_equal(got_tris, exp_tris)


def test_bvh_structure_invariants(rt):
    tris = random_soup(20000, 9)
    nodes, newtris = build(rt, tris)
    n = len(nodes)
    # BFS numbering: children after parents, A before B, consecutive
    seen = 1
    depth = {0: 0}
    for i, (mn, mx, a, b, offs, ln) in enumerate(nodes):
        if a != -1:
            assert a == seen and b == seen + 1 and ln == 0
            seen += 2
            depth[a] = depth[b] = depth[i] + 1
        else:
            assert b == -1
    assert seen == n
    assert max(depth.values()) <= 16
    # leaves tile the triangle array in BFS order
    leaf = [(offs, ln) for (_, _, a, _, offs, ln) in nodes if a == -1]
    assert leaf[0][0] == 0 and all(leaf[i][0] + leaf[i][1] == leaf[i + 1][0] for i in range(len(leaf) - 1))
    assert leaf[-1][0] + leaf[-1][1] == tris.size
    # every leaf box bounds its triangles; leaves above max depth hold <= 4 triangles
    for i, (mn, mx, a, b, offs, ln) in enumerate(nodes):
        if a == -1 and ln:
            v = newtris[offs // 25:(offs + ln) // 25, 0:9].reshape(-1, 3)
            assert np.all(v >= np.float32(mn)) and np.all(v <= np.float32(mx))
            assert ln // 25 <= 4 or depth[i] == 16
    # a permutation of the input
    assert sorted(map(bytes, newtris)) == sorted(map(bytes, tris))


def build_sah(rt, tris, depth=24, leaf=2):
    L = rt.scene_lib()
    t = np.ascontiguousarray(tris, np.float32).copy()
    nodes = C.POINTER(rt.BvhNode)()
    n = C.c_int32()
    L.rt_bvh_build_sah.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_int, C.c_int, C.POINTER(C.POINTER(rt.BvhNode)),
                                   C.POINTER(C.c_int32)]
    assert L.rt_bvh_build_sah(t.ctypes.data_as(C.POINTER(C.c_float)), t.size, depth, leaf, C.byref(nodes), C.byref(n)) == 0
    out = [(tuple(nodes[i].bbox_min), tuple(nodes[i].bbox_max), nodes[i].child_a, nodes[i].child_b,
            nodes[i].tri_offs, nodes[i].tri_len) for i in range(n.value)]
    L.rt_host_free(nodes)
    return out, t.reshape(-1, 25)


@pytest.mark.parametrize("n,depth,leaf", [(20000, 24, 2), (3000, 6, 2), (500, 24, 1), (1, 24, 2), (0, 24, 2)])
def test_sah_tree_invariants(rt, n, depth, leaf):
    """The non-parity fast tree (rt_bvh_build_sah) is a valid tree of the same layout: BFS numbering, leaves tile the
    triangle array, boxes bound, depth within the cap, leaves small wherever the depth budget allowed it."""
    tris = random_soup(n, 21) if n else np.zeros((0, 25), np.float32)
    if n >= 500:
        tris[100:140, 0:9] = tris[100, 0:9]          # 40 identical triangles: no split plane separates their centroids
    nodes, newtris = build_sah(rt, tris, depth, leaf)
    seen, dep = 1, {0: 0}
    for i, (mn, mx, a, b, offs, ln) in enumerate(nodes):
        if a != -1:
            assert a == seen and b == seen + 1 and ln == 0
            seen += 2
            dep[a] = dep[b] = dep[i] + 1
        else:
            assert b == -1
    assert seen == len(nodes) and max(dep.values()) <= depth
    leafs = [(i, offs, ln) for i, (_, _, a, _, offs, ln) in enumerate(nodes) if a == -1]
    assert leafs[0][1] == 0 and all(leafs[k][1] + leafs[k][2] == leafs[k + 1][1] for k in range(len(leafs) - 1))
    assert leafs[-1][1] + leafs[-1][2] == tris.size
    for i, offs, ln in leafs:
        if ln:
            v = newtris[offs // 25:(offs + ln) // 25, 0:9].reshape(-1, 3)
            assert np.all(v >= np.float32(nodes[i][0])) and np.all(v <= np.float32(nodes[i][1]))
        assert ln // 25 <= leaf or dep[i] == depth
    for i, (mn, mx, a, b, offs, ln) in enumerate(nodes):         # inner boxes contain their children's
        if a != -1:
            for ch in (a, b):
                if nodes[ch][5] or nodes[ch][2] != -1:
                    assert np.all(np.float32(nodes[ch][0]) >= np.float32(mn)) and np.all(np.float32(nodes[ch][1]) <= np.float32(mx))
    assert sorted(map(bytes, newtris)) == sorted(map(bytes, tris))
    if n == 20000:
        assert max(ln for _, _, ln in leafs) // 25 <= 2 and len(nodes) > 2 * n // 3


def test_scene_rebuild_keeps_lights_and_reference_tree(rt, scenes):
    sc = rt.Scene.load("scene2_is.txt")
    lights0 = sorted(bytes(sc.tris()[o // 25]) for o in np.ctypeslib.as_array(sc.desc.light_tris, shape=(sc.desc.n_light_tris,)))
    n0 = sc.desc.n_nodes
    sc.rebuild_bvh("sah")
    assert sc.desc.n_nodes > n0 and sc.desc.n_light_tris == len(lights0)
    lt = np.ctypeslib.as_array(sc.desc.light_tris, shape=(sc.desc.n_light_tris,))
    assert np.all(np.diff(lt) > 0) and np.all(lt % 25 == 0)
    assert sorted(bytes(sc.tris()[o // 25]) for o in lt) == lights0            # the same triangles, new offsets
    assert np.all(sc.tris()[:, 24] < 1000)                                      # markers removed again
    ref = scenes("scene.txt")
    a = rt.Scene.load("scene.txt")
    a.rebuild_bvh("reference")                                                  # same records, same order -> the same tree
    assert a.desc.n_nodes == ref.desc.n_nodes and np.array_equal(a.tris(), ref.tris())


def test_bvh_depth_cap_gives_fat_leaves(rt):
    tris = random_soup(3000, 5)
    nodes, _ = build(rt, tris, depth=4)
    assert max(ln for (_, _, a, _, _, ln) in nodes if a == -1) // 25 > 50


def test_empty_scene(rt):
    nodes, t = build(rt, np.zeros((0, 25), np.float32))
    assert len(nodes) == 1 and nodes[0][2] == -1 and nodes[0][5] == 0


# ---------------------------------------------------------------- textures

def test_noise_texture_formula(rt):
    """NoiseTexture::generate (noiseTexture.cpp:22-40) against a numpy restatement."""
    w = h = 64
    out = np.zeros((h, w, 3), np.float32)
    assert rt.scene_lib().rt_noise_texture(w, h, out.ctypes.data_as(C.POINTER(C.c_float))) == 0
    f = np.float32

    def four(x, y):
        s = np.sin
        return (s(x * f(1.52) + y * f(23.0)) + s(x * f(9.5) + y * f(10.0)) * f(0.9) +
                s(x * f(15.5) + y * f(23.0)) * f(0.7) + s(x * f(35.0) + y * f(2.0)) * f(0.5))

    ys, xs = np.mgrid[0:h, 0:w]
    u = (xs / w).astype(np.float32).astype(np.float64)
    v = (ys / h).astype(np.float32).astype(np.float64)
    u = u + four(u + 0.2, v + 0.9) * 0.1
    v = v + four(u + 0.1, v + 0.3) * 0.1
    hh = 1.0 / (1.0 + np.exp(-four(u, v) * 2.0))
    assert np.allclose(out[..., 0], hh, atol=5e-4)  # float vs double sin arguments
    assert np.array_equal(out[..., 0], out[..., 1]) and np.array_equal(out[..., 0], out[..., 2])
    assert 0.0 < out.min() and out.max() < 1.0


def test_hdr_loader_against_pillow_free_decode(rt):
    """RGBE decode: value = mantissa * 2^(e-136) (what stbi_loadf does), checked on the bundled 1k HDR."""
    L = rt.scene_lib()
    out = C.POINTER(C.c_float)()
    w, h = C.c_int32(), C.c_int32()
    path = os.path.join(rt.ASSET_DIR, "cape_hill_1k.hdr")
    assert L.rt_image_load(path.encode(), C.byref(out), C.byref(w), C.byref(h)) == 0
    assert (w.value, h.value) == (1024, 512)
    a = np.ctypeslib.as_array(out, shape=(h.value, w.value, 3)).copy()
    L.rt_host_free(out)
    assert np.all(np.isfinite(a)) and a.min() >= 0 and 0.05 < a.mean() < 50
    # independent pure-python RLE decode of a few scanlines
    raw = open(path, "rb").read()
    pos = raw.index(b"\n\n") + 2
    pos = raw.index(b"\n", pos) + 1
    for y in range(3):
        assert raw[pos] == 2 and raw[pos + 1] == 2
        pos += 4
        chans = []
        for _ in range(4):
            row = bytearray()
            while len(row) < 1024:
                c = raw[pos]; pos += 1
                if c > 128:
                    row += bytes([raw[pos]]) * (c - 128); pos += 1
                else:
                    row += raw[pos:pos + c]; pos += c
            chans.append(np.frombuffer(bytes(row), np.uint8).astype(np.float64))
        e = chans[3]
        sc = np.where(e > 0, np.ldexp(1.0, (e - 136).astype(int)), 0.0)
        exp = np.stack([chans[0] * sc, chans[1] * sc, chans[2] * sc], 1).astype(np.float32)
        assert np.array_equal(a[y], exp)


def test_ppm_roundtrip_and_gamma(rt, tmp_path):
    L = rt.scene_lib()
    img = np.linspace(0, 1, 4 * 3 * 3, dtype=np.float32).reshape(3, 4, 3)
    p = str(tmp_path / "o.ppm").encode()
    assert L.rt_image_save(p, img.ctypes.data_as(C.POINTER(C.c_float)), 4, 3) == 0
    out = C.POINTER(C.c_float)()
    w, h = C.c_int32(), C.c_int32()
    assert L.rt_image_load(p, C.byref(out), C.byref(w), C.byref(h)) == 0
    a = np.ctypeslib.as_array(out, shape=(3, 4, 3)).copy()
    L.rt_host_free(out)
    q = (img * 255).astype(np.uint8)                        # texture.cpp:55 truncating cast
    assert np.allclose(a, (q / 255.0) ** 2.2, atol=1e-6)    # stbi_loadf LDR->linear gamma 2.2
