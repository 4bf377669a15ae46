"""GPU parity of the device primitives, the shading entry, the committed golden fixtures and the error
behaviour of the C-ABI.  Everything goes through librt_amd.so; the oracle is only the checker."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rnd(n, k, seed, scale=2.0):
    return (np.random.default_rng(seed).normal(size=(n, k)) * scale).astype(np.float32)


def unit(v):
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


# ---- integer / pure IEEE arithmetic ops: bit-exact ---------------------------------------------------------

def test_rng_bit_exact(ob, rt, gpu):
    rng = np.random.default_rng(1)
    rec = rng.integers(0, 2 ** 32, size=(4096, 4), dtype=np.uint64).astype(np.uint32)
    assert same_bits(gpu.debug_op(rt.OP_RNG, rec), ob.debug_op(rt.OP_RNG, rec))


def test_portable_math_bit_exact(ob, rt, gpu):
    rng = np.random.default_rng(2)
    n = 20000
    for fn, x, y in [(0, rng.uniform(-20, 20, n), None), (0, rng.uniform(0, 6.3, n), None), (1, rng.uniform(-1, 1, n), None),
                     (2, rng.uniform(-1, 1, n), None), (3, rng.normal(size=n), rng.normal(size=n)),
                     (4, (rng.integers(0, 1 << 24, n) + 1) / 16777216.0, None)]:
        rec = np.zeros((n, 3), np.float32)
        rec[:, 0] = np.full(n, fn, np.int32).view(np.float32)
        rec[:, 1] = x
        if y is not None:
            rec[:, 2] = y
        assert same_bits(gpu.debug_op(rt.OP_PMATH, rec), ob.debug_op(rt.OP_PMATH, rec)), fn


def test_triangle_and_box_tests_bit_exact(ob, rt, gpu):
    n = 50000
    rec = np.concatenate([rnd(n, 3, 3), unit(rnd(n, 3, 4)), rnd(n, 9, 5)], 1)
    g, o = gpu.debug_op(rt.OP_TRI_INTERSECT, rec), ob.debug_op(rt.OP_TRI_INTERSECT, rec)
    assert np.array_equal(g[:, 0], o[:, 0]) and 0.03 < o[:, 0].mean() < 0.5
    hit = o[:, 0] == 1
    assert same_bits(g[hit], o[hit])
    d = unit(rnd(n, 3, 6))
    d[::7, 0] = 0.0          # zero direction components: inf / NaN slabs
    d[::11, 1] = 0.0
    c = rnd(n, 3, 7)
    h = np.abs(rnd(n, 3, 8, 1.0))
    rec = np.concatenate([rnd(n, 3, 9), d, c - h, c + h], 1)
    rec[::13, 0] = rec[::13, 6]   # origins exactly on a face
    assert same_bits(gpu.debug_op(rt.OP_AABB_DIST, rec), ob.debug_op(rt.OP_AABB_DIST, rec))


def test_sphere_floor_bit_exact(ob, rt, gpu):
    n = 20000
    rec = np.concatenate([rnd(n, 3, 10), unit(rnd(n, 3, 11)), rnd(n, 3, 12), np.abs(rnd(n, 1, 13, 1.0)) + 0.1], 1)
    g, o = gpu.debug_op(rt.OP_SPHERE, rec), ob.debug_op(rt.OP_SPHERE, rec)
    assert np.array_equal(g[:, 0], o[:, 0]) and same_bits(g[o[:, 0] == 1], o[o[:, 0] == 1])
    rec = np.concatenate([rnd(n, 3, 14), unit(rnd(n, 3, 15)), rnd(n, 1, 16)], 1)
    g, o = gpu.debug_op(rt.OP_FLOOR, rec), ob.debug_op(rt.OP_FLOOR, rec)
    assert np.array_equal(g[:, 0], o[:, 0]) and same_bits(g[o[:, 0] == 1], o[o[:, 0] == 1])


def test_light_sampling_primitives_bit_exact(ob, rt, gpu):
    n = 20000
    tri = rnd(n, 9, 20)
    rec = np.concatenate([rnd(n, 3, 21), unit(rnd(n, 3, 22)), tri], 1)
    assert same_bits(gpu.debug_op(rt.OP_ITRI_PDF, rec), ob.debug_op(rt.OP_ITRI_PDF, rec))
    r12 = np.random.default_rng(23).uniform(0, 1, (n, 2)).astype(np.float32)
    rec = np.concatenate([rnd(n, 3, 24), r12, tri], 1)
    gs, os_ = gpu.debug_op(rt.OP_ITRI_SAMPLE, rec), ob.debug_op(rt.OP_ITRI_SAMPLE, rec)
    assert same_bits(gs, os_)
    # a sampled direction hits its triangle: pdf through the GPU is positive and equals the oracle's
    rec2 = np.concatenate([rec[:, 0:3], gs, tri], 1)
    gp = gpu.debug_op(rt.OP_ITRI_PDF, rec2)
    assert same_bits(gp, ob.debug_op(rt.OP_ITRI_PDF, rec2)) and (gp > 0).mean() > 0.95
    quad = np.concatenate([rnd(n, 3, 25), rnd(n, 3, 26), rnd(n, 3, 27)], 1)
    rec = np.concatenate([rnd(n, 3, 28), unit(rnd(n, 3, 29)), quad], 1)
    assert same_bits(gpu.debug_op(rt.OP_ATTRACTOR_PDF, rec), ob.debug_op(rt.OP_ATTRACTOR_PDF, rec))
    rec = np.concatenate([rnd(n, 3, 30), r12, quad], 1)
    assert same_bits(gpu.debug_op(rt.OP_ATTRACTOR_SAMPLE, rec), ob.debug_op(rt.OP_ATTRACTOR_SAMPLE, rec))


def test_env_index_and_camera_rays_bit_exact(ob, rt, gpu, scenes):
    n = 20000
    d = unit(rnd(n, 3, 31))
    rec = np.concatenate([d, np.full((n, 1), 1024, np.float32), np.full((n, 1), 512, np.float32)], 1)
    assert same_bits(gpu.debug_op(rt.OP_ENV_INDEX, rec), ob.debug_op(rt.OP_ENV_INDEX, rec))
    sc = scenes("scene.txt")
    rng = np.random.default_rng(32)
    rec = np.stack([rng.integers(0, 1920, n), rng.integers(0, 1080, n), rng.integers(0, 4096, n)], 1).astype(np.uint32)
    for variant in range(3):
        p = rt.default_params(sc.camera, spp=1, width=1920, height=1080)
        if variant >= 1:
            p.cam.aperture = 0.15            # thin lens
            p.cam.rot[:] = [0.2, -0.4, 0.1]
        if variant == 2:
            p.cam.ang_vel[:] = [0.05, 0.1, -0.02]   # motion blur: per-sample rotation matrix on the device
            p.cam.vel[:] = [0.3, 0.0, -0.2]
        assert same_bits(gpu.debug_op(rt.OP_CAMERA_RAY, rec, p), ob.debug_op(rt.OP_CAMERA_RAY, rec, p)), variant


def test_tonemap_within_libm_tolerance(ob, rt, gpu):
    x = np.abs(rnd(20000, 3, 33, 1.5))
    g, o = gpu.debug_op(rt.OP_TONEMAP, x), ob.debug_op(rt.OP_TONEMAP, x)
    assert np.allclose(g, o, rtol=2e-6, atol=1e-7)       # powf: device libm vs glibc, a few ulp


# ---- Material::reflect on real hits --------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["scene_cornellbox.txt", "scene.txt", "scene2_is.txt"])
def test_shade_hits_bit_exact(ob, rt, gpu, scenes, name):
    sc = scenes(name)
    g = np.load(os.path.join(GOLD, name.replace(".txt", ".npz")))
    rays, prim = g["rays"], g["prim"]
    hit = prim >= 0
    prim_o, t, pos, nrm, _ = ob.trace_rays(sc, rays[hit])
    n = int(hit.sum())
    rec = np.zeros((n, 16), np.float32)
    rec[:, 0:3] = rays[hit, 0:3]
    rec[:, 3:6] = rays[hit, 3:6]
    rec[:, 6:9] = np.random.default_rng(5).uniform(0.2, 1.0, (n, 3))
    rec[:, 9] = 1.0
    rec[:, 10:13] = pos
    rec[:, 13] = prim_o.astype(np.int32).view(np.float32)
    rec[:, 14] = np.arange(n, dtype=np.uint32).view(np.float32)
    rec[:, 15] = (np.arange(n, dtype=np.uint32) % 7).view(np.float32)
    gpu.upload(sc)
    a, b = gpu.shade_hits(rec, 37811), ob.shade_hits(sc, rec, 37811)
    assert same_bits(a, b)
    mats = np.array([m.type for m in sc.materials()])
    used = set(mats[sc.tris()[prim_o, 24].astype(int)])
    assert len(used) >= 1


def test_shade_every_material_type_bit_exact(ob, rt, gpu):
    """One big triangle per material type 0..6 (+ an unknown type), textured Lambert included."""
    tex = np.random.default_rng(9).uniform(0, 1, (16, 16, 3)).astype(np.float32)
    mats = []
    for ty in [0, 1, 2, 3, 4, 5, 6, 9]:
        m = rt.Material(); m.type = ty; m.color = 0x80C040; m.p1 = 0.4; m.p2 = 0.2
        m.image_offset[:] = [-1, -1, -1]
        if ty == 3:
            m.image_offset[0] = 0; m.image_width = 16; m.image_height = 16
        mats.append(m)
    recs = np.zeros((len(mats) + 1, 25), np.float32)
    for i in range(len(mats) + 1):
        z = -2.0 - i
        recs[i, 0:9] = [-5, -5, z, 5, -5, z, 0, 5, z]
        recs[i, 9:18] = [0.1, 0.0, 1.0, -0.1, 0.1, 1.0, 0.0, -0.1, 1.0]
        recs[i, 18:24] = [0, 0, 0.9, 0, 0.5, 0.9]
        recs[i, 24] = min(i, len(mats) - 1) + 0.001
    recs[-1, 24] = 5 + 1000 + 0.001      # an importance-sampled light of material 5
    sc = rt.Scene.from_records(recs, mats, textures=tex)
    assert sc.desc.n_light_tris == 1
    tris = sc.tris()
    n = 4096
    rng = np.random.default_rng(10)
    tri = rng.integers(0, tris.shape[0], n)
    b = rng.dirichlet([1, 1, 1], n)
    pos = tris[tri, 0:3] * b[:, :1] + tris[tri, 3:6] * b[:, 1:2] + tris[tri, 6:9] * b[:, 2:3]
    d = unit(rnd(n, 3, 11))
    rec = np.zeros((n, 16), np.float32)
    rec[:, 0:3] = pos - d
    rec[:, 3:6] = d
    rec[:, 6:9] = 1.0
    rec[:, 9] = rng.choice([1.0, 1.4], n)
    rec[:, 10:13] = pos
    rec[:, 13] = tri.astype(np.int32).view(np.float32)
    rec[:, 14] = rng.integers(0, 1 << 20, n).astype(np.uint32).view(np.float32)
    rec[:, 15] = rng.integers(0, 64, n).astype(np.uint32).view(np.float32)
    gpu.upload(sc)
    assert same_bits(gpu.shade_hits(rec, 99), ob.shade_hits(sc, rec, 99))


# ---- golden fixtures ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("mode", ["wavefront", "megakernel", "lockstep"])
@pytest.mark.parametrize("name", ["scene_cornellbox", "scene", "scene2", "scene2_is"])
def test_gpu_matches_committed_golden(rt, gpu, scenes, name, mode):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    sc = scenes(name + ".txt")
    gpu.upload(sc)
    prim, t, pos, nrm, cnt = gpu.trace_rays(g["rays"], rt.F_BVH, want_counts=True)
    assert np.array_equal(prim, g["prim"])                                   # ids: bit-exact
    hit = prim >= 0
    assert np.all(np.abs(t[hit] - g["t"][hit]) <= 1e-5 * np.abs(g["t"][hit]))  # t: 1e-5 relative
    assert np.all(np.abs(nrm[hit] - g["normal"][hit]) <= 1e-5)               # normals: 1e-5
    assert np.array_equal(cnt, g["counts"])
    size, spp, bounces = int(g["img_size"]), int(g["img_spp"]), int(g["img_bounces"])
    flags = rt.F_DEFAULT | rt.F_STATS | rt.MODE_FLAGS[mode]
    p = rt.default_params(sc.camera, spp=spp, width=size, height=size, max_bounces=bounces, flags=flags)
    gpu.reset()
    st = gpu.render(p)
    img = gpu.readback()
    ref = g["img_raw_sum"]
    assert np.sqrt(np.mean((img - ref) ** 2)) / ref.mean() <= 2e-2           # the stated bar
    assert np.array_equal(img, ref)                                          # what is actually achieved
    assert [st.paths, st.rays, st.inner_visits, st.leaf_visits, st.tri_tests, st.hits, st.texel_fetches,
            st.light_tests] == list(map(int, g["img_stats"]))


# ---- options and edge cases --------------------------------------------------------------------------------

@pytest.mark.parametrize("mode", ["wavefront", "megakernel", "lockstep"])
def test_spheres_brute_force_no_aa_no_skybox(ob, rt, gpu, tmp_path, mode):
    txt = ("SET_OUTPUT_WIDTH 48\nSET_OUTPUT_HEIGHT 40\nSET_FOCAL_LENGTH 0.5\nSET_APERTURE_RADIUS 0.05\n\n"
           "DEF_IMAGE env\nPATH cape_hill_1k.hdr\n\nDEF_IMAGE earth\nPATH earth.ppm\n\nDEF_MODEL m\nPATH plane.obj\n\n"
           "DEF_MODEL c\nPATH cube.obj\n\nDEF_MATERIAL a\nTYPE 3\nCOLOR 0.5 0.5 0.5\nTEXTURE earth\n\n"
           "DEF_MATERIAL glow\nTYPE 1\nCOLOR 1 1 1\nP1 4\n\nDEF_MATERIAL glass\nTYPE 2\nP1 0.5\n\n"
           "DEF_OBJ o\nMODEL m\nMATERIAL glow\n\nDEF_OBJ g\nMODEL c\nMATERIAL glass\n\nSET_BACKGROUND env\n\n"
           "ADD_OBJECT\nOBJECT o\nPOSITION 0 -1 -4\nSCALE 4.0\n\nADD_OBJECT\nOBJECT g\nPOSITION 1.2 -0.3 -3.5\nROTATION 0.3 0.5 0.0\nSCALE 0.5\n\n"
           "ADD_SPHERE\nPOSITION 0 0 -3\nRADIUS 0.5\nMATERIAL a\n\nADD_SPHERE\nPOSITION -1.2 0.2 -4\nRADIUS 0.7\nMATERIAL a\n")
    f = tmp_path / "s.txt"
    f.write_text(txt)
    sc = rt.Scene.load(str(f), enable_spheres=True)
    gpu.upload(sc)
    m = rt.MODE_FLAGS[mode]
    for flags in [rt.F_DEFAULT | rt.F_SPHERES, rt.F_BVH | rt.F_SPHERES, rt.F_SKYBOX | rt.F_ANTIALIAS | rt.F_SPHERES,
                  rt.F_DEFAULT]:
        p = rt.default_params(sc.camera, spp=4, flags=flags | rt.F_STATS | m)
        gpu.reset()
        st = gpu.render(p)
        o, ost = ob.render(sc, p, n_threads=4)
        assert np.array_equal(gpu.readback(), o), flags
        assert (st.rays, st.tri_tests, st.hits, st.texel_fetches) == (ost.rays, ost.tri_tests, ost.hits, ost.texel_fetches)
        if mode == "megakernel":    # without counters the searches go down the shortcut tree (single triangles here): spheres come after it
            q = rt.RenderParams.from_buffer_copy(p); q.flags = flags | m
            gpu.reset(); gpu.render(q)
            assert np.array_equal(gpu.readback(), o), ("shortcut", flags)
            q.flags |= rt.F_STATS | rt.F_SHORTCUT_STATS
            gpu.reset(); ss = gpu.render(q)
            assert np.array_equal(gpu.readback(), o) and (ss.rays, ss.hits, ss.texel_fetches) == (ost.rays, ost.hits, ost.texel_fetches)


@pytest.mark.parametrize("mode", ["wavefront", "megakernel", "lockstep"])
def test_sphere_material_out_of_range_is_the_error_colour(ob, rt, gpu, tmp_path, mode):
    """A sphere record whose material id lies outside the material array (the reference would read out of bounds,
    main.cu:341) shades like a material without a texture -- Sphere::getTextureColor's "color of error"
    (sphere.h:64) -- in the oracle and in every schedule of the HIP path alike."""
    txt = ("SET_OUTPUT_WIDTH 40\nSET_OUTPUT_HEIGHT 32\nSET_FOCAL_LENGTH 0.5\n\nDEF_IMAGE earth\nPATH earth.ppm\n\n"
           "DEF_MODEL m\nPATH plane.obj\n\nDEF_MATERIAL a\nTYPE 3\nCOLOR 0.5 0.5 0.5\nTEXTURE earth\n\nDEF_OBJ o\nMODEL m\nMATERIAL a\n\n"
           "ADD_OBJECT\nOBJECT o\nPOSITION 0 -1 -4\nSCALE 4.0\n\n"
           "ADD_SPHERE\nPOSITION 0 0 -3\nRADIUS 0.6\nMATERIAL a\n\nADD_SPHERE\nPOSITION -1.2 0.2 -4\nRADIUS 0.7\nMATERIAL a\n")
    f = tmp_path / "s.txt"
    f.write_text(txt)
    sc = rt.Scene.load(str(f), enable_spheres=True)
    pre = np.ctypeslib.as_array(sc.desc.tris, shape=(sc.prefix,))
    imgs = []
    for mat in (0.0001, 99.0001, -3.0):
        pre[5] = mat                              # sphere 0: [N, (x, y, z, r, mat) ...]
        gpu.upload(sc)
        p = rt.default_params(sc.camera, spp=3, flags=rt.F_DEFAULT | rt.F_SPHERES | rt.F_STATS | rt.MODE_FLAGS[mode])
        gpu.reset()
        st = gpu.render(p)
        o, ost = ob.render(sc, p, n_threads=4)
        img = gpu.readback()
        assert np.array_equal(img, o), mat
        assert (st.rays, st.hits, st.texel_fetches) == (ost.rays, ost.hits, ost.texel_fetches)
        imgs.append(img)
    assert np.array_equal(imgs[1], imgs[2]) and not np.array_equal(imgs[0], imgs[1])
    centre = imgs[1][16, 20]                      # the untextured sphere fills the image centre: magenta paths only
    assert centre[1] == 0.0 and centre[0] > 0.0 and centre[2] > 0.0


@pytest.mark.parametrize("mode", ["wavefront", "megakernel", "lockstep"])
def test_motion_blur_and_ragged_image_sizes(ob, rt, gpu, scenes, mode):
    sc = scenes("scene.txt")
    gpu.upload(sc)
    m = rt.MODE_FLAGS[mode]
    for (w, h) in [(1, 1), (7, 3), (65, 9), (100, 33)]:
        p = rt.default_params(sc.camera, spp=3, width=w, height=h, flags=rt.F_DEFAULT | m)
        p.cam.vel[:] = [0.2, 0.1, 0.0]
        p.cam.ang_vel[:] = [0.0, 0.05, 0.02]
        gpu.reset()
        gpu.render(p)
        o, _ = ob.render(sc, p, n_threads=4)
        assert np.array_equal(gpu.readback(), o), (w, h)


def test_sample_chunks_of_the_path_kernel(ob, rt, gpu, scenes, monkeypatch):
    """The path-parallel kernel keeps one radiance per (pixel, sample) of a chunk; with a 1 MB buffer a 40-spp call
    runs in three chunks and a ragged image pads its edge tiles -- the per-pixel sums must not change by one bit."""
    sc = scenes("scene.txt")
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=40, width=93, height=61, flags=rt.F_DEFAULT | rt.F_STATS | rt.F_MEGAKERNEL)
    gpu.reset(); st = gpu.render(p); one = gpu.readback()
    monkeypatch.setenv("RT_SAMPLE_BUF_MB", "1")            # 12 x 8 tiles x 64 x 12 B = 73.7 KB per sample -> 14 spp per chunk
    gpu.reset(); st_c = gpu.render(p); chunked = gpu.readback()
    monkeypatch.delenv("RT_SAMPLE_BUF_MB")
    assert np.array_equal(one, chunked)
    assert (st.paths, st.rays, st.tri_tests) == (st_c.paths, st_c.rays, st_c.tri_tests) and st.paths == 93 * 61 * 40
    o, ost = ob.render(sc, p, n_threads=8)
    assert np.array_equal(one, o) and st.rays == ost.rays
    # progressive use: two calls of 20 samples accumulate like the lock-step kernel does
    q = rt.RenderParams.from_buffer_copy(p); q.sample_count = 20
    gpu.reset(); gpu.render(q); q.sample_begin = 20; gpu.render(q); two = gpu.readback()
    ql = rt.RenderParams.from_buffer_copy(q); ql.flags |= rt.F_LOCKSTEP; ql.sample_begin = 0
    gpu.reset(); gpu.render(ql); ql.sample_begin = 20; gpu.render(ql)
    assert np.array_equal(two, gpu.readback())

@pytest.mark.parametrize("name", ["scene.txt", "scene_cornellbox.txt", "scene2_is.txt"])
def test_material_table_in_device_memory(rt, gpu, scenes, monkeypatch, name):
    """A scene with more than 128 materials keeps its material table in device memory instead of the LDS (trace_cfg,
    RT_LDS_MATS_MAX); RT_NO_LDS_MATS forces that path on the bundled scenes: same sums bit for bit."""
    sc = scenes(name)
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=4, width=160, height=90, flags=rt.F_DEFAULT)
    monkeypatch.delenv("RT_NO_LDS_MATS", raising=False)
    gpu.reset(); gpu.render(p, want_stats=False); in_lds = gpu.readback()
    monkeypatch.setenv("RT_NO_LDS_MATS", "1")
    gpu.reset(); gpu.render(p, want_stats=False); in_mem = gpu.readback()
    monkeypatch.delenv("RT_NO_LDS_MATS")
    assert np.array_equal(in_lds, in_mem)


@pytest.mark.parametrize("name", ["scene.txt", "scene_cornellbox.txt", "scene2_is.txt"])
def test_upload_ways_give_the_same_scene(rt, gpu, scenes, monkeypatch, name):
    """rt_upload_scene has three ways to bring a scene to the device (csrc/rt_core.hip): everything copied at the end
    (RT_NO_EARLY_UPLOAD), cold data and textures copied by a thread of their own meanwhile (the default), and from a
    million triangles on the triangles' items too, written and copied while the shortcut tree is built and moved into
    place on the device (RT_EARLY_TRIS_MIN lowers that bound for the test).  Same sums bit for bit, stats included."""
    sc = scenes(name)
    p = rt.default_params(sc.camera, spp=4, width=160, height=90, flags=rt.F_DEFAULT | rt.F_STATS)
    results = []
    for env in ({}, {"RT_NO_EARLY_UPLOAD": "1"}, {"RT_EARLY_TRIS_MIN": "1"}, {"RT_EARLY_TRIS_MIN": "1", "RT_NO_HUGE_PAGES": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        gpu.upload(sc)
        for k in env:
            monkeypatch.delenv(k)
        gpu.reset()
        st = gpu.render(p)
        results.append((gpu.readback(), (st.rays, st.inner_visits, st.leaf_visits, st.tri_tests, st.dev_inner_visits, st.dev_wide_visits)))
    for img, counts in results[1:]:
        assert np.array_equal(results[0][0], img)
        assert counts == results[0][1]
    gpu.upload(sc)


@pytest.mark.parametrize("name", ["scene.txt", "scene2.txt", "scene2_is.txt"])
def test_lds_window_size_changes_nothing(ob, rt, gpu, scenes, monkeypatch, name):
    """The shortcut walk reads a wide node from the LDS window where it lies inside it and from device memory where it
    does not (csrc/rt_device.h, trav_step_wide; the window holds compact 112-byte records of wide nodes in the scenes
    without light queries, padded items in the others).  Whatever the window holds -- nothing, one node, an odd cut
    through the tree, as much as fits -- the sums are the same bit for bit, and they are the oracle's."""
    sc = scenes(name)
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=4, width=160, height=90, flags=rt.F_DEFAULT)
    images = []
    for cap in (None, "0", "1", "2", "77", "300", "1001"):
        if cap is None:
            monkeypatch.delenv("RT_LDS_NODES", raising=False)
        else:
            monkeypatch.setenv("RT_LDS_NODES", cap)       # (read by every render call: trace_cfg)
        gpu.reset(); gpu.render(p, want_stats=False); images.append(gpu.readback())
    monkeypatch.delenv("RT_LDS_NODES", raising=False)
    for img in images[1:]:
        assert np.array_equal(images[0], img)
    o, _ = ob.render(sc, p, n_threads=8)
    assert np.array_equal(images[0], o)



@pytest.mark.parametrize("depth", [2, 5, 8])
def test_big_leaves_get_sub_trees(ob, rt, gpu, depth):
    """Shallow trees over scene.txt: leaves of hundreds of triangles, which the device walks through a small tree
    of its own per leaf (triangles reordered inside the leaf, padded boxes, pruning with slack, ties by triangle
    id).  Sums and counters as the oracle's plain walk over every triangle of the leaf, in all three schedules."""
    sc = rt.Scene.load("scene.txt", bvh_max_depth=depth)
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=3, width=72, height=56, max_bounces=5, flags=rt.F_DEFAULT | rt.F_STATS | rt.F_MEGAKERNEL)
    gpu.reset(); st = gpu.render(p); img = gpu.readback()
    o, ost = ob.render(sc, p, n_threads=8)
    assert np.array_equal(img, o)
    assert (st.rays, st.inner_visits, st.leaf_visits, st.tri_tests, st.hits) == \
           (ost.rays, ost.inner_visits, ost.leaf_visits, ost.tri_tests, ost.hits)
    for m in ("lockstep", "wavefront"):
        q = rt.RenderParams.from_buffer_copy(p); q.flags = rt.F_DEFAULT | rt.F_STATS | rt.MODE_FLAGS[m]
        gpu.reset(); st2 = gpu.render(q)
        assert np.array_equal(gpu.readback(), img) and st2.tri_tests == st.tri_tests, m


@pytest.mark.parametrize("mode", ["wavefront", "megakernel", "lockstep"])
def test_device_actual_counters(rt, gpu, scenes, mode):
    """RT_F_STATS also reports what the device really did, next to the counters of the reference's walk: without
    sub-trees the two agree; with sub-trees the device visits more inner items and runs fewer triangle tests."""
    m = rt.MODE_FLAGS[mode]
    sc = scenes("scene_cornellbox.txt")                      # 24 triangles: no sub-trees
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=2, width=64, height=64, max_bounces=4, flags=rt.F_DEFAULT | rt.F_STATS | m)
    gpu.reset(); st = gpu.render(p)
    assert (st.dev_inner_visits, st.dev_tri_tests) == (st.inner_visits, st.tri_tests) and st.tri_tests > 0
    assert st.dev_lds_inner_visits == st.dev_inner_visits    # the whole tree fits the LDS node cache
    sc = rt.Scene.load("scene.txt", bvh_max_depth=5)         # leaves of ~100 triangles: every leaf gets a sub-tree
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=2, width=64, height=64, flags=rt.F_DEFAULT | rt.F_STATS | m)
    gpu.reset(); st = gpu.render(p)
    assert st.dev_inner_visits > st.inner_visits and 0 < st.dev_tri_tests < st.tri_tests
    assert 0 < st.dev_lds_inner_visits <= st.dev_inner_visits


VOLUME_SCENE = ("SET_OUTPUT_WIDTH 64\nSET_OUTPUT_HEIGHT 48\nSET_FOCAL_LENGTH 0.5\nSET_APERTURE_RADIUS 0.0\n\n"
                "DEF_IMAGE env\nPATH cape_hill_1k.hdr\n\nDEF_MODEL pl\nPATH plane.obj\n\nDEF_MODEL cu\nPATH cube.obj\n\n"
                "DEF_MODEL mo\nPATH monkey.obj\n\n"
                "DEF_MATERIAL floor\nTYPE 3\nCOLOR 0.7 0.7 0.7\n\nDEF_MATERIAL glow\nTYPE 1\nCOLOR 1 0.9 0.8\nP1 6\n\n"
                "DEF_MATERIAL fog\nTYPE 6\nCOLOR 0.9 0.6 0.3\nP1 1.5\n\nDEF_MATERIAL red\nTYPE 0\nCOLOR 0.8 0.2 0.2\nP1 0.2\nP2 0.3\n\n"
                "DEF_OBJ f\nMODEL pl\nMATERIAL floor\n\nDEF_OBJ l\nMODEL pl\nMATERIAL glow\n\nDEF_OBJ v\nMODEL cu\nMATERIAL fog\n\n"
                "DEF_OBJ m\nMODEL mo\nMATERIAL red\n\nSET_BACKGROUND env\n\n"
                "ADD_OBJECT\nOBJECT f\nPOSITION 0 -1 -4\nSCALE 5.0\n\nADD_OBJECT\nOBJECT l\nPOSITION 0 2.5 -4\nROTATION 3.14159 0 0\nSCALE 1.0\n\n"
                "ADD_OBJECT\nOBJECT v\nPOSITION 0 0 -3.5\nROTATION 0.2 0.4 0.0\nSCALE 0.9\n\n"
                "ADD_OBJECT\nOBJECT m\nPOSITION 0.2 -0.1 -3.6\nSCALE 0.4\n\n"
                "ADD_SPHERE\nPOSITION -1.4 -0.4 -3.5\nRADIUS 0.5\nMATERIAL floor\n")


def test_volume_scattering_matches_oracle(ob, rt, gpu, tmp_path):
    """ENABLE_VOLUME_SCATTERING (config.h:9, off in the reference): a type-6 cube around a monkey head; free-flight
    scattering inside it (checkScatteredVolume) with and without analytic spheres.  Fused kernel only."""
    f = tmp_path / "v.txt"
    f.write_text(VOLUME_SCENE)
    sc = rt.Scene.load(str(f), enable_spheres=True)
    gpu.upload(sc)
    base = None
    for flags in (rt.F_DEFAULT, rt.F_DEFAULT | rt.F_VOLUME, rt.F_DEFAULT | rt.F_VOLUME | rt.F_SPHERES):
        p = rt.default_params(sc.camera, spp=6, max_bounces=8, flags=flags | rt.F_STATS)
        gpu.reset(); st = gpu.render(p); img = gpu.readback()
        o, ost = ob.render(sc, p, n_threads=8)
        assert np.array_equal(img, o), flags
        assert (st.rays, st.hits, st.tri_tests, st.texel_fetches) == (ost.rays, ost.hits, ost.tri_tests, ost.texel_fetches)
        if base is None:
            base = img
        else:
            assert not np.array_equal(img, base)   # the volume is there
    for m in (rt.F_LOCKSTEP, rt.F_WAVEFRONT):
        p = rt.default_params(sc.camera, spp=1, flags=rt.F_DEFAULT | rt.F_VOLUME | m)
        with pytest.raises(RuntimeError, match="fused kernel only"):
            gpu.render(p)


def test_ray_marching_matches_oracle(ob, rt, gpu, tmp_path):
    """ENABLE_RAY_MARCHING (config.h:14): hits count only within |marchVelocity| of the ray origin, misses march
    under the point mass at the origin (marchRay) and see the flat backdrop beyond 20 units."""
    txt = VOLUME_SCENE.replace("POSITION 0 -1 -4\nSCALE 5.0", "POSITION 0 -0.04 -0.5\nSCALE 5.0")
    f = tmp_path / "m.txt"
    f.write_text(txt)
    sc = rt.Scene.load(str(f))
    gpu.upload(sc)
    for flags in (rt.F_DEFAULT | rt.F_RAYMARCH, rt.F_DEFAULT | rt.F_RAYMARCH | rt.F_VOLUME):
        p = rt.default_params(sc.camera, spp=4, max_bounces=25, flags=flags | rt.F_STATS)
        gpu.reset(); st = gpu.render(p); img = gpu.readback()
        o, ost = ob.render(sc, p, n_threads=8)
        assert np.array_equal(img, o)
        assert (st.rays, st.hits, st.tri_tests) == (ost.rays, ost.hits, ost.tri_tests)
        assert st.hits > 0 and st.rays > st.paths          # some rays hit the floor under the camera, the others march


def test_empty_scene_and_zero_bounces(ob, rt, gpu, scenes):
    mt = rt.Material(); mt.type = 3; mt.color = 0xFFFFFF; mt.image_offset[:] = [-1, -1, -1]
    sc = rt.Scene.from_records(np.zeros((0, 25), np.float32), [mt])
    gpu.upload(sc)
    for m in rt.MODE_FLAGS.values():
        p = rt.default_params(sc.camera, spp=3, width=20, height=10, flags=rt.F_DEFAULT | rt.F_STATS | m)
        gpu.reset()
        st = gpu.render(p)
        assert np.array_equal(gpu.readback(), np.broadcast_to(np.float32([1, 1, 0.05]) * np.float32(0.5) * 3, (10, 20, 3)))
        assert st.rays == st.paths == 600
    sc = scenes("scene_cornellbox.txt")
    gpu.upload(sc)
    for m in rt.MODE_FLAGS.values():
        p = rt.default_params(sc.camera, spp=2, width=16, height=16, max_bounces=0, flags=rt.F_DEFAULT | rt.F_STATS | m)
        gpu.reset()
        st = gpu.render(p)
        assert not gpu.readback().any() and st.rays == 0 and st.paths == 512
    p = rt.default_params(sc.camera, spp=0, width=16, height=16)
    gpu.reset()
    gpu.render(p)
    assert not gpu.readback().any()


def test_error_codes_instead_of_exit(rt, scenes):
    L = rt.core_lib()
    h = C.c_void_p()
    assert L.rt_create(C.byref(h), 0) == 0
    try:
        sc = scenes("scene_cornellbox.txt")
        p = rt.default_params(sc.camera, spp=1, width=8, height=8)
        assert L.rt_render(h, C.byref(p), None) == -3                       # RT_ERR_NO_SCENE
        assert b"before rt_upload_scene" in L.rt_last_error(h)
        assert L.rt_upload_scene(h, None) == -1
        assert L.rt_upload_scene(h, C.byref(sc.desc)) == 0
        bad = rt.RenderParams.from_buffer_copy(p); bad.cam.width = 0
        assert L.rt_render(h, C.byref(bad), None) == -1
        bad = rt.RenderParams.from_buffer_copy(p); bad.flags |= rt.F_RAYMARCH | rt.F_WAVEFRONT
        assert L.rt_render(h, C.byref(bad), None) == -1 and b"fused kernel only" in L.rt_last_error(h)
        bad = rt.RenderParams.from_buffer_copy(p); bad.band_rows, bad.band_stride, bad.band_index = 4, 2, 2
        assert L.rt_render(h, C.byref(bad), None) == -1
        # the wavefront pipeline packs (sample << 8 | bounce) into one word: larger counts are refused, not mis-rendered
        bad = rt.RenderParams.from_buffer_copy(p); bad.flags |= rt.F_WAVEFRONT; bad.max_bounces = 256
        assert L.rt_render(h, C.byref(bad), None) == -4 and b"255 bounces" in L.rt_last_error(h)
        bad = rt.RenderParams.from_buffer_copy(p); bad.flags |= rt.F_WAVEFRONT; bad.sample_count = 1 << 23
        assert L.rt_render(h, C.byref(bad), None) == -4
        bad = rt.RenderParams.from_buffer_copy(p); bad.cam.env_width, bad.cam.env_height = 4096, 2048   # larger than the pool
        assert L.rt_render(h, C.byref(bad), None) == -1
        out = np.zeros(8 * 8 * 3, np.float32)
        assert L.rt_readback(h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size, 0) == -1  # nothing rendered yet
        assert L.rt_render(h, C.byref(p), None) == 0
        assert L.rt_readback(h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size - 3, 0) == -1  # wrong size
        assert L.rt_readback(h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size, 0) == 0
        ok = rt.RenderParams.from_buffer_copy(p); ok.flags |= rt.F_WAVEFRONT; ok.max_bounces = 255   # the limit itself is fine
        assert L.rt_render(h, C.byref(ok), None) == 0
        # malformed BVH: child index pointing backwards (would loop forever) / bad triangle range
        nodes = (rt.BvhNode * sc.desc.n_nodes)(*[sc.desc.nodes[i] for i in range(sc.desc.n_nodes)])
        d2 = rt.SceneDesc.from_buffer_copy(sc.desc)
        nodes[0].child_a = 0
        d2.nodes = nodes
        assert L.rt_upload_scene(h, C.byref(d2)) == -1 and b"bad child" in L.rt_last_error(h)
        nodes[0].child_a = sc.desc.nodes[0].child_a
        leaf = next(i for i in range(sc.desc.n_nodes) if nodes[i].child_a == -1)
        nodes[leaf].tri_len = 10 ** 6 * 25
        assert L.rt_upload_scene(h, C.byref(d2)) == -1 and b"bad triangle range" in L.rt_last_error(h)
        # the good scene still renders after the failed uploads were rejected
        assert L.rt_upload_scene(h, C.byref(sc.desc)) == 0 and L.rt_render(h, C.byref(p), None) == 0
    finally:
        L.rt_destroy(h)


def test_bvh_deeper_than_stack_is_rejected(rt):
    """A degenerate chain deeper than the LDS stack must be refused (RT_ERR_LIMIT), not overflow."""
    depth = 40
    n = 2 * depth + 1
    nodes = (rt.BvhNode * n)()
    tris = np.zeros((depth + 1, 25), np.float32)
    for i in range(depth + 1):
        tris[i, 0:9] = [i, 0, -5, i + 1, 0, -5, i, 1, -5]
    # node 2k = inner (children 2k+1 leaf, 2k+2 inner/leaf)
    for k in range(depth):
        nd = nodes[2 * k]
        nd.bbox_min[:] = [0, 0, -6]; nd.bbox_max[:] = [depth + 2, 2, -4]
        nd.child_a, nd.child_b = 2 * k + 1, 2 * k + 2
        lf = nodes[2 * k + 1]
        lf.bbox_min[:] = [k, 0, -6]; lf.bbox_max[:] = [k + 1, 1, -4]
        lf.child_a = lf.child_b = -1; lf.tri_offs, lf.tri_len = 25 * k, 25
    last = nodes[2 * depth]
    last.bbox_min[:] = [depth, 0, -6]; last.bbox_max[:] = [depth + 1, 1, -4]
    last.child_a = last.child_b = -1; last.tri_offs, last.tri_len = 25 * depth, 25
    m = rt.Material(); m.type = 3; m.image_offset[:] = [-1, -1, -1]
    d = rt.SceneDesc()
    d.tris = tris.ctypes.data_as(C.POINTER(C.c_float)); d.n_tri_floats = tris.size
    d.nodes = nodes; d.n_nodes = n
    marr = (rt.Material * 1)(m); d.mats = marr; d.n_mats = 1
    L = rt.core_lib()
    h = C.c_void_p()
    assert L.rt_create(C.byref(h), 0) == 0
    try:
        assert L.rt_upload_scene(h, C.byref(d)) == -4 and b"exceeds the traversal stack" in L.rt_last_error(h)
    finally:
        L.rt_destroy(h)


# ---- BASELINE-size properties (size-independent checks at full resolution) ---------------------------------

def test_full_hd_properties(rt, gpu, scenes):
    """1920x1080 (BASELINE config 2 resolution), few spp: determinism, sample-slice additivity, band gather."""
    sc = scenes("scene.txt")
    gpu.upload(sc)
    W, H = 1920, 1080
    p = rt.default_params(sc.camera, spp=2, width=W, height=H, flags=rt.F_DEFAULT | rt.F_STATS)
    gpu.reset(); st1 = gpu.render(p); a = gpu.readback()
    gpu.reset(); st2 = gpu.render(p); b = gpu.readback()
    assert np.array_equal(a, b) and st1.rays == st2.rays                    # run-to-run bit reproducible
    pw = rt.RenderParams.from_buffer_copy(p); pw.flags |= rt.F_WAVEFRONT
    gpu.reset(); st3 = gpu.render(pw)
    assert np.array_equal(gpu.readback(), a) and st3.rays == st1.rays       # both pipelines agree bit for bit
    assert st1.paths == W * H * 2 and np.all(np.isfinite(a)) and a.min() >= 0
    q = rt.RenderParams.from_buffer_copy(p); q.sample_count = 1
    gpu.reset(); gpu.render(q); q.sample_begin = 1; gpu.render(q)
    assert np.allclose(gpu.readback(), a, rtol=1e-6, atol=1e-7)             # slices add up
    out = np.zeros_like(a)
    for r in range(8):
        q = rt.RenderParams.from_buffer_copy(p); q.band_rows, q.band_stride, q.band_index = 16, 8, r
        gpu.reset(); gpu.render(q)
        out[[y for y in range(H) if (y // 16) % 8 == r]] = gpu.readback()
    assert np.array_equal(out, a)                                           # 8-way tile-row gather == 1 GPU


@pytest.mark.parametrize("name,spp", [("scene.txt", 64), ("scene2.txt", 16), ("scene2_is.txt", 8)])
def test_shortcut_walk_at_full_hd(rt, gpu, scenes, name, spp):
    """BASELINE configs 2 and 3 at 1920x1080 (config 2 with all its 64 spp): the default walk (shortcut tree first) gives the
    frame of the reference's walk bit for bit, in fewer node visits; a few searches in a hundred million are not certified
    and walk the reference's tree after all."""
    sc = scenes(name)
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=spp, width=1920, height=1080, flags=rt.F_DEFAULT | rt.F_STATS)
    gpu.reset(); ref = gpu.render(p); a = gpu.readback()
    q = rt.RenderParams.from_buffer_copy(p); q.flags |= rt.F_SHORTCUT_STATS
    gpu.reset(); st = gpu.render(q)
    assert np.array_equal(gpu.readback(), a)
    assert (st.paths, st.rays, st.hits, st.texel_fetches, st.light_tests) == (ref.paths, ref.rays, ref.hits, ref.texel_fetches, ref.light_tests)
    assert st.dev_inner_visits < 0.6 * ref.dev_inner_visits and 0 < st.shortcut_rewalks < 1e-5 * st.rays and ref.shortcut_rewalks == 0
    q.flags = rt.F_DEFAULT
    gpu.reset(); gpu.render(q)
    assert np.array_equal(gpu.readback(), a)


def test_config4_field_at_full_size(ob, rt, gpu):
    """BASELINE config 4 at its real size: 628 dragons = 9 999 018 triangles in the reference's depth-16 tree (leaves of
    ~150 triangles, each with a sub-tree of its own on the device), 3840x2160.  Size-independent properties: the path
    kernel and the lock-step kernel (which walks the same leaves one triangle per lane-step) give the same image and
    the same counters; an 8-way tile-row gather equals the single image; and on a crop the oracle agrees bit for bit."""
    sc = rt.Scene.synth_field(628, env_path=os.path.join(rt.ASSET_DIR, "cape_hill_1k.hdr"), env_upsample=4)
    assert sc.n_tris == 9999018 and sc.desc.n_nodes > 100000
    gpu.upload(sc)
    W, H = 3840, 2160
    p = rt.default_params(sc.camera, spp=1, width=W, height=H, flags=rt.F_DEFAULT | rt.F_STATS | rt.F_MEGAKERNEL)
    gpu.reset(); st = gpu.render(p); img = gpu.readback()
    q = rt.RenderParams.from_buffer_copy(p); q.flags |= rt.F_LOCKSTEP
    gpu.reset(); st_l = gpu.render(q)
    assert np.array_equal(gpu.readback(), img)
    assert (st.rays, st.inner_visits, st.leaf_visits, st.tri_tests, st.hits) == \
           (st_l.rays, st_l.inner_visits, st_l.leaf_visits, st_l.tri_tests, st_l.hits)
    assert st.paths == W * H and st.tri_tests > 100 * st.rays and np.all(np.isfinite(img))
    out = np.zeros_like(img)
    for r in range(8):
        b = rt.RenderParams.from_buffer_copy(p); b.band_rows, b.band_stride, b.band_index = 16, 8, r
        gpu.reset(); gpu.render(b)
        out[[y for y in range(H) if (y // 16) % 8 == r]] = gpu.readback()
    assert np.array_equal(out, img)
    # the default walk: the shortcut tree (over the 65 k reference leaves here) first -- the same frame bit for bit, in
    # about half the node visits, and a handful of searches that had to walk the reference's tree after all
    s = rt.RenderParams.from_buffer_copy(p); s.flags |= rt.F_SHORTCUT_STATS
    gpu.reset(); st_s = gpu.render(s)
    assert np.array_equal(gpu.readback(), img)
    assert (st_s.rays, st_s.hits, st_s.texel_fetches) == (st.rays, st.hits, st.texel_fetches)
    assert st_s.dev_inner_visits < 0.7 * st.dev_inner_visits and st_s.shortcut_rewalks < 1e-5 * st_s.rays
    s.flags = rt.F_DEFAULT
    gpu.reset(); gpu.render(s)
    assert np.array_equal(gpu.readback(), img)
    # the oracle on the same 10M triangles at a sixth of the resolution (its plain walk tests ~600 triangles per ray)
    c = rt.default_params(sc.camera, spp=2, width=640, height=360, flags=rt.F_DEFAULT | rt.F_STATS | rt.F_MEGAKERNEL)
    gpu.reset(); st_c = gpu.render(c); crop = gpu.readback()
    o, ost = ob.render(sc, c, n_threads=16)
    assert np.array_equal(crop, o)
    c.flags = rt.F_DEFAULT
    gpu.reset(); gpu.render(c)
    assert np.array_equal(gpu.readback(), o)                                # (shortcut walk against the oracle)
    assert (st_c.rays, st_c.inner_visits, st_c.leaf_visits, st_c.tri_tests, st_c.hits) == \
           (ost.rays, ost.inner_visits, ost.leaf_visits, ost.tri_tests, ost.hits)


def test_sub_tree_ties_go_to_the_first_triangle(ob, rt, gpu, monkeypatch):
    """Coincident triangles (every triangle eight times, copies far apart in the array, each copy with another
    material) in leaves big enough for sub-trees: the reference keeps the FIRST of several triangles at the same
    distance, the sub-tree meets them in its own order and has to pick by id.  Hit ids, images and counters against
    the oracle's walk; with sub-trees forced on (tiny scene) and off."""
    rng = np.random.default_rng(5)
    base = rng.normal(0, 1.0, (24, 1, 3)) + np.array([0, 0, -6.0]) + rng.normal(0, 0.6, (24, 3, 3))
    order = rng.permutation(np.repeat(np.arange(24), 8))
    n = len(order)
    t = np.zeros((n, 25), np.float32)
    t[:, 0:9] = base[order].reshape(n, 9)
    nr = rng.normal(0, 1, (n, 3, 3)); nr /= np.linalg.norm(nr, axis=2, keepdims=True)
    t[:, 9:18] = nr.reshape(n, 9)
    t[:, 18:24] = rng.uniform(0, 0.999, (n, 6))
    mats = []
    for k in range(8):
        m = rt.Material(); m.type = [3, 1, 0, 4, 3, 5, 1, 3][k]; m.color = int(rng.integers(0, 1 << 24)); m.p1 = 0.8; m.p2 = 0.2
        m.image_offset[:] = [-1, -1, -1]
        mats.append(m)
    t[:, 24] = np.arange(n) % 8 + 0.001
    sc = rt.Scene.from_records(t, mats, bvh_max_depth=2, bvh_min_tris=1)
    rays = np.zeros((4000, 6), np.float32)
    rays[:, 0:3] = rng.normal(0, 0.3, (4000, 3))
    tgt = base[rng.integers(0, 24, 4000)].mean(1) + rng.normal(0, 0.3, (4000, 3))
    d = tgt - rays[:, 0:3]
    rays[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    oprim, ot = ob.trace_rays(sc, rays)[:2]
    assert (oprim >= 0).sum() > 1000
    p = rt.default_params(sc.camera, spp=2, width=64, height=48, max_bounces=4, flags=rt.F_BVH | rt.F_STATS)
    o, ost = ob.render(sc, p, n_threads=8)
    for min_tris in ("0", "1000000"):
        monkeypatch.setenv("RT_SUB_MIN_TRIS", min_tris)
        gpu.upload(sc)
        gprim, gt = gpu.trace_rays(rays, rt.F_BVH)[:2]
        assert np.array_equal(gprim, oprim) and np.array_equal(gt, ot), min_tris
        assert np.array_equal(gpu.trace_rays(rays, 0)[0], ob.trace_rays(sc, rays, 0)[0]), min_tris   # brute force: ties over the whole array
        for mode in ("megakernel", "lockstep", "wavefront"):
            q = rt.RenderParams.from_buffer_copy(p); q.flags |= rt.MODE_FLAGS[mode]
            gpu.reset(); st = gpu.render(q)
            assert np.array_equal(gpu.readback(), o), (min_tris, mode)
            assert (st.inner_visits, st.leaf_visits, st.tri_tests, st.hits) == (ost.inner_visits, ost.leaf_visits, ost.tri_tests, ost.hits)


@pytest.mark.parametrize("scale,dist", [(1e-3, 1.0), (1.0, 3e4), (1e3, 10.0), (50.0, 1.5e4), (1.0, 0.0)])
def test_sub_trees_at_extreme_scales(ob, rt, gpu, monkeypatch, scale, dist):
    """The sub-tree boxes are padded by 1e-4 max(1, |coordinate|) and pruned with a slack that grows with |origin|^2:
    tiny scenes (the padding swallows them), huge ones, origins very far from the geometry (squared distances lose
    their low bits) and origins inside it.  Closest hits as the oracle's walk over every triangle of the leaf."""
    rng = np.random.default_rng(int(scale * 7 + dist) % 1000)
    n = 1200
    c = rng.normal(0, 1.0, (n, 1, 3))
    v = (c + rng.normal(0, 0.08, (n, 3, 3))) * scale
    t = np.zeros((n, 25), np.float32)
    t[:, 0:9] = v.reshape(n, 9)
    t[:, 9:18] = np.tile([0, 0, 1.0], (n, 3))
    m = rt.Material(); m.type = 3; m.color = 0x808080; m.image_offset[:] = [-1, -1, -1]
    sc = rt.Scene.from_records(t, [m], bvh_max_depth=3, bvh_min_tris=1)     # 8 leaves of ~150 triangles
    nr = 6000
    d0 = rng.normal(0, 1, (nr, 3)); d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    org = d0 * dist * scale + rng.normal(0, 0.3 * scale, (nr, 3))
    tgt = rng.normal(0, 0.7 * scale, (nr, 3))
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[nr // 2:] *= rng.uniform(0.2, 4.0, (nr - nr // 2, 1))              # callers' rays need not be unit vectors
    rays = np.concatenate([org, d], axis=1).astype(np.float32)
    monkeypatch.setenv("RT_SUB_MIN_TRIS", "0")
    gpu.upload(sc)
    gp, gt, gpos = gpu.trace_rays(rays, rt.F_BVH)[:3]
    op, ot, opos = ob.trace_rays(sc, rays, rt.F_BVH)[:3]
    assert (op >= 0).sum() > nr // 20
    assert np.array_equal(gp, op) and np.array_equal(gt, ot) and np.array_equal(gpos, opos)


@pytest.mark.parametrize("seed", range(6))
def test_sub_tree_stress_against_the_full_walk(ob, rt, gpu, monkeypatch, seed):
    """Rays chosen to sit on the edges of the sub-tree arithmetic: origins ON triangles (bounce rays start there),
    at vertices and on box faces, directions along the axes and inside the planes of axis-aligned quads, grazing
    rays.  Geometry: a random soup (seeds 0-2) or a lattice of axis-aligned quads whose boxes are flat (3-5).
    Ids, distances and positions must equal the oracle's, which tests every triangle of every leaf it visits."""
    rng = np.random.default_rng(900 + seed)
    if seed < 3:
        n = 2000
        c = rng.uniform(-2, 2, (n, 1, 3))
        v = c + rng.normal(0, rng.choice([0.03, 0.15, 0.6]), (n, 3, 3))
    else:
        quads = []
        g = np.arange(-3, 4) * 0.5
        for axis in range(3):
            for a in g:
                for b in g[::2]:
                    for h in g[::3]:
                        p0 = np.zeros(3); p0[axis] = h; p0[(axis + 1) % 3] = a; p0[(axis + 2) % 3] = b
                        e1 = np.zeros(3); e1[(axis + 1) % 3] = 0.5
                        e2 = np.zeros(3); e2[(axis + 2) % 3] = 1.0
                        quads += [[p0, p0 + e1, p0 + e1 + e2], [p0, p0 + e1 + e2, p0 + e2]]
        v = np.array(quads)
        v = v[rng.permutation(len(v))]
        n = len(v)
    t = np.zeros((n, 25), np.float32)
    t[:, 0:9] = v.reshape(n, 9)
    t[:, 9:18] = np.tile([0, 0, 1.0], (n, 3))
    m = rt.Material(); m.type = 3; m.color = 0x808080; m.image_offset[:] = [-1, -1, -1]
    sc = rt.Scene.from_records(t, [m], bvh_max_depth=int(rng.choice([1, 3, 6])), bvh_min_tris=1)
    vf = t[:, 0:9].reshape(n, 3, 3).astype(np.float64)
    k = 5000
    pick = rng.integers(0, n, k)
    bary = rng.dirichlet([1, 1, 1], k)
    on_tri = np.einsum("kj,kjc->kc", bary, vf[pick])                       # points on triangles
    at_vertex = vf[pick, rng.integers(0, 3, k)]                             # triangle vertices
    dirs = rng.normal(0, 1, (k, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    axis_dirs = np.eye(3)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1))
    in_plane = dirs.copy(); in_plane[np.arange(k), rng.integers(0, 3, k)] = 0.0   # inside an axis-aligned plane
    in_plane /= np.maximum(np.linalg.norm(in_plane, axis=1, keepdims=True), 1e-9)
    target = np.einsum("kj,kjc->kc", rng.dirichlet([1, 1, 1], k), vf[rng.integers(0, n, k)])
    far = rng.normal(0, 6, (k, 3))
    toward = target - far; toward /= np.linalg.norm(toward, axis=1, keepdims=True)
    rays = np.concatenate([
        np.concatenate([on_tri + 1e-5 * dirs, dirs], 1),        # bounce-like
        np.concatenate([on_tri, axis_dirs], 1),                 # from a surface along an axis
        np.concatenate([at_vertex, dirs], 1),                   # from a vertex (on the corner of some box)
        np.concatenate([at_vertex, in_plane], 1),               # from a vertex inside an axis plane
        np.concatenate([far, toward], 1),                       # aimed at a point of a triangle from outside
        np.concatenate([np.round(far * 2) / 2, axis_dirs], 1),  # lattice origins along the axes (in the quads' planes)
    ]).astype(np.float32)
    op, ot, opos = ob.trace_rays(sc, rays, rt.F_BVH)[:3]
    assert (op >= 0).sum() > len(rays) // 10
    for min_tris in ("0", "1000000"):
        monkeypatch.setenv("RT_SUB_MIN_TRIS", min_tris)
        gpu.upload(sc)
        gp, gt, gpos = gpu.trace_rays(rays, rt.F_BVH)[:3]
        bad = np.flatnonzero(gp != op)
        assert bad.size == 0, (min_tris, bad[:10], gp[bad[:10]], op[bad[:10]])
        assert np.array_equal(gt, ot) and np.array_equal(gpos, opos), min_tris


def test_sub_tree_build_is_the_same_on_any_number_of_host_threads(ob, rt, gpu, monkeypatch):
    """The sub-trees of a big scene are built on several host threads (leaves are independent); the item stream, and
    with it every result and counter, must not depend on how many.  Two dragons = 31 844 triangles, > 4096 leaves
    with a sub-tree."""
    sc = rt.Scene.synth_field(2)
    rng = np.random.default_rng(11)
    org = rng.normal(0, 1, (20000, 3)) * np.array([6, 0.5, 6]) + np.array([0, 3.0, 0])
    d = rng.normal(0, 1, (20000, 3)) * np.array([1, 0.3, 1]) - np.array([0, 0.6, 0])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d], 1).astype(np.float32)
    ref = ob.trace_rays(sc, rays, rt.F_BVH, want_counts=True)
    assert (ref[0] >= 0).sum() > 2000
    for threads in ("1", "7"):
        monkeypatch.setenv("RT_HOST_THREADS", threads)
        gpu.upload(sc)
        got = gpu.trace_rays(rays, rt.F_BVH, want_counts=True)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), threads


def test_release_scratch_keeps_scene_and_sums(rt, gpu, scenes):
    """rt_release_scratch frees the buffers held between calls (per-path radiance, ray queues, readback scratch):
    device memory comes back, the accumulated sum is still there, and the next renders -- every schedule -- allocate
    again and give the same image."""
    import ctypes as C
    rt.core_lib()
    hip = C.CDLL("libamdhip64.so.7")   # (the HIP runtime the library itself is bound to, by SONAME)
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    sc = scenes("scene_cornellbox.txt")
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=64, width=512, height=512, max_bounces=4)
    gpu.reset(); gpu.render(p)
    first = gpu.readback(rt.READ_AVERAGE)
    held = free_bytes()
    gpu.release_scratch()
    assert free_bytes() >= held + 512 * 512 * 64 * 12                       # at least the radiance buffer came back
    assert np.array_equal(gpu.readback(rt.READ_AVERAGE), first)             # (the readback scratch is allocated again)
    for mode in ("megakernel", "wavefront", "lockstep", "pool"):
        q = rt.RenderParams.from_buffer_copy(p); q.flags |= rt.MODE_FLAGS[mode]
        gpu.reset(); gpu.render(q)
        assert np.array_equal(gpu.readback(rt.READ_AVERAGE), first), mode
        gpu.release_scratch()


def test_tiny_scene_with_its_triangles_in_lds(ob, rt, gpu, scenes, monkeypatch):
    """A scene whose whole item stream fits the LDS cache (Cornell box) runs the path kernel's TLDS variant: leaf steps
    read triangles from the LDS.  Same sums and counters as the oracle and as the variant that reads them from memory
    (RT_NO_LDS_TRIS)."""
    sc = scenes("scene_cornellbox.txt")
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=8, width=160, height=120, max_bounces=5, flags=rt.F_DEFAULT | rt.F_STATS)
    o, ost = ob.render(sc, p, n_threads=8)
    got = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("RT_NO_LDS_TRIS", "1")
        gpu.reset(); st = gpu.render(p)
        img = gpu.readback()
        assert np.array_equal(img, o), off
        assert (st.rays, st.inner_visits, st.leaf_visits, st.tri_tests, st.hits) == (ost.rays, ost.inner_visits, ost.leaf_visits, ost.tri_tests, ost.hits)
        got.append((st.dev_inner_visits, st.dev_lds_inner_visits, st.dev_tri_tests))
    assert got[0] == got[1] and got[0][0] == got[0][1] > 0      # every inner item from the LDS either way


def test_callers_of_an_older_header_are_not_written_past_their_struct(rt, gpu, scenes):
    """ABI versioning (include/rt.h): rt_render_stats grew at the end in round 2.  The symbol round 1 exported, rt_render,
    fills round 1's 88 bytes and nothing behind them; rt_render_v2 writes exactly the size the caller states."""
    L = rt.core_lib()
    assert L.rt_abi_version() == rt.ABI_VERSION
    sc = scenes("scene_cornellbox.txt")
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=2, width=32, height=32, flags=rt.F_DEFAULT | rt.F_STATS)
    full = gpu.render(p)                                               # the binding's own call: every field
    assert full.rays > 0 and full.dev_tri_tests > 0 and C.sizeof(full) >= 128
    ref = bytes(full)
    for fn, size in ((lambda buf: L.rt_render(gpu._h, C.byref(p), buf), 88),
                     (lambda buf: L.rt_render_v2(gpu._h, C.byref(p), buf, 96), 96),
                     (lambda buf: L.rt_render_v2(gpu._h, C.byref(p), buf, 4096), C.sizeof(full))):
        buf = (C.c_ubyte * 256)(*([0xA5] * 256))
        assert fn(buf) == 0
        got = bytes(buf)
        assert got[8:size] == ref[8:size]                               # (bytes 0-7: kernel_ms, differs from call to call)
        assert got[size:] == b"\xa5" * (256 - size)                     # nothing behind the caller's struct is touched
    info = (C.c_ubyte * 128)(*([0xA5] * 128))
    lay = C.c_void_p()
    assert L.rt_layout_create(C.byref(sc.desc), C.byref(lay), None, 0) == 0
    try:
        assert L.rt_layout_get(lay, info, None, None) == 0              # round 1's symbol: 28 bytes
        assert bytes(info)[28:] == b"\xa5" * 100 and bytes(info)[:4] != b"\xa5" * 4
    finally:
        L.rt_layout_destroy(lay)
