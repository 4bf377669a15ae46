"""Converged-image and physics-level checks THROUGH THE HIP PATH (north_star: "converged images must agree within a
stated RMSE or relative-luminance bound at matched spp"; SURVEY.md 8(c) tolerance (ii)).

Bit-identity with the oracle at the same seed (tests/test_gpu_parity.py) cannot see an error both sides share -- a
biased layout of the RNG dimensions, an estimator that loses energy.  These tests can:
  * independent seeds: GPU at seed A against the oracle at seeds B and C, 128 x 128, 1024 spp --
    |mean luminance(gpu_A) / mean luminance(orc_B) - 1| <= 1 % and RMSE(gpu_A, orc_B) <= 1.5 RMSE(orc_C, orc_B);
    on the BASELINE scenes as configured (Cornell box, scene.txt under the HDR sky), where one seed scatters by 3 %, the same
    1 % against the mean of 128 committed oracle seeds and 512 GPU seeds (tests/golden/conv_*.npz);
  * two GPU seeds approach each other as 1 / sqrt(spp) over 64 -> 256 -> 1024 spp;
  * known answers whatever the random numbers: an emissive wall, a white furnace (white Lambertian mesh in a
    constant environment: every pixel = the environment, geometry invisible), the cosine-weighted mean 2/3.
The tolerances are the ones SURVEY.md fixed, as constants: no bound here scales with a measured scatter."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LUMA = np.float64([0.2126, 0.7152, 0.0722])


def rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def gpu_average(rt, gpu, sc, seed, spp, size=128, bounces=6, flags=None):
    """Average image of one seed: (image, params)."""
    p = rt.default_params(sc.camera, spp=spp, width=size, height=size, max_bounces=bounces, seed=seed,
                          flags=rt.F_DEFAULT if flags is None else flags)
    gpu.reset()
    gpu.render(p)
    return gpu.readback(rt.READ_AVERAGE), p


def lum(img):
    return float((img.reshape(-1, 3).astype(np.float64) @ LUMA).mean())


@pytest.mark.parametrize("name,spp", [("scene.txt", 1024), ("scene2.txt", 256)])
def test_converged_image_against_independent_oracle_seeds(ob, rt, gpu, scenes, name, spp):
    """SURVEY 8(c)(ii) on the flat backdrop (seed-to-seed scatter of the mean luminance 0.006 % / 0.011 % at these sample
    counts, profiles/r02_lum_noise.txt): GPU at seed A against the oracle at seeds B and C, the oracle rendered here."""
    sc = scenes(name)
    gpu.upload(sc)
    flags = rt.F_DEFAULT & ~rt.F_SKYBOX
    g, p = gpu_average(rt, gpu, sc, seed=1001, spp=spp, flags=flags)
    ref = []
    for seed in (2002, 3003):
        q = rt.RenderParams.from_buffer_copy(p); q.seed = seed
        acc, _ = ob.render(sc, q, n_threads=16)
        ref.append(acc / np.float32(spp))
    assert abs(lum(g) / lum(ref[0]) - 1.0) <= 0.01, (lum(g), lum(ref[0]))          # the bound SURVEY fixed
    noise = rmse(ref[1], ref[0])                                                  # what two reference-code seeds differ by
    assert noise > 0 and rmse(g, ref[0]) <= 1.5 * noise, (rmse(g, ref[0]), noise)
    assert rmse(g, ref[1]) <= 1.5 * noise


@pytest.mark.parametrize("name", ["scene_cornellbox.txt", "scene.txt"])
def test_converged_luminance_of_the_baseline_scenes_as_configured(rt, gpu, scenes, name):
    """The SAME fixed bounds on the scenes BASELINE.json names as they are configured: the Cornell box, whose small light is
    found by chance (material.h:205-213), and scene.txt under the HDR environment (main.cu:374-383).  One seed of 1 024 spp
    scatters by 2.8 % / 3 % in mean luminance there (fireflies), so one pair of seeds cannot carry a 1 % bound; many can.
    tests/golden/conv_<scene>.npz holds K = 128 independent ORACLE seeds (tools/make_converged_golden.py, build container):
    per-seed mean luminance, the mean image, the first two seeds' images.  Here the GPU renders K' = 512 OTHER seeds:
        | mean_gpu / mean_oracle - 1 | <= 1 %           (standard error of the ratio ~ 0.3 %: a 3-sigma test, no scatter-scaled term)
        RMSE(gpu seed, oracle seed A) <= 1.5 RMSE(oracle seed A, oracle seed B)      for the first GPU seeds, as SURVEY states it
    and, beyond SURVEY's two criteria, the two MEAN images agree as far as seed noise lets means of K and K' seeds agree:
        RMSE(mean image gpu, mean image oracle) <= 2 x noise / sqrt(2) x sqrt(1 / K + 1 / K')."""
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "conv_" + name.replace(".txt", "") + ".npz"))
    size, spp, bounces, K = int(fx["size"]), int(fx["spp"]), int(fx["bounces"]), len(fx["lum"])
    assert K >= 64 and int(fx["flags"]) == rt.F_DEFAULT
    sc = scenes(name)
    gpu.upload(sc)
    KG = 512
    lums, mean = np.zeros(KG), np.zeros((size, size, 3), np.float64)
    first = []
    for k in range(KG):
        img, _ = gpu_average(rt, gpu, sc, seed=200000 + k, spp=spp, size=size, bounces=bounces)   # (oracle seeds: 100000 + k)
        lums[k] = lum(img)
        mean += img
        if k < 4:
            first.append(img)
    mean /= KG
    ratio = lums.mean() / fx["lum"].mean()
    se = np.hypot(lums.std(ddof=1) / np.sqrt(KG) / lums.mean(), fx["lum"].std(ddof=1) / np.sqrt(K) / fx["lum"].mean())
    print(f"{name}: mean luminance gpu {lums.mean():.6f} ({KG} seeds) / oracle {fx['lum'].mean():.6f} ({K} seeds) = {ratio:.5f}, "
          f"standard error of the ratio {100 * se:.2f} %, per-seed scatter gpu {100 * lums.std(ddof=1) / lums.mean():.2f} % / "
          f"oracle {100 * fx['lum'].std(ddof=1) / fx['lum'].mean():.2f} %")
    assert abs(ratio - 1.0) <= 0.01                                              # the bound SURVEY fixed, as a constant
    assert se < 0.004                                                             # ... and it means something: 1 % is > 2.5 sigma
    noise = rmse(fx["ref_b"], fx["ref_a"])                                        # what two reference-code seeds differ by
    for img in first:
        assert rmse(img, fx["ref_a"]) <= 1.5 * noise and rmse(img, fx["ref_b"]) <= 1.5 * noise
    # the two mean images differ by seed noise only: single-seed noise sigma_1 = noise / sqrt(2) per pixel, means of K and K' seeds
    predicted = noise / np.sqrt(2.0) * np.sqrt(1.0 / K + 1.0 / KG)
    print(f"{name}: RMSE of the mean images {rmse(mean, fx['mean_image']):.5f}, predicted from seed noise {predicted:.5f}")
    assert rmse(mean, fx["mean_image"]) <= 2.0 * predicted, (rmse(mean, fx["mean_image"]), predicted)


def test_two_gpu_seeds_converge_as_inverse_sqrt_spp(rt, gpu, scenes):
    sc = scenes("scene_cornellbox.txt")
    gpu.upload(sc)
    rel = []
    for spp in (64, 256, 1024):
        a, _ = gpu_average(rt, gpu, sc, seed=11, spp=spp)
        b, _ = gpu_average(rt, gpu, sc, seed=22, spp=spp)
        rel.append(rmse(a, b) / float(np.mean(0.5 * (a + b))))
    # 4 x the samples -> half the noise; the small light makes single steps ragged (fireflies), so each step is
    # held to 1.3 .. 3.0 and the whole 16-fold range to 2.5 .. 6 (ideal: 2 and 4)
    assert 1.3 < rel[0] / rel[1] < 3.0 and 1.3 < rel[1] / rel[2] < 3.0, rel
    assert 2.5 < rel[0] / rel[2] < 6.0, rel


@pytest.mark.parametrize("mode", ["wavefront", "megakernel", "lockstep"])
def test_emissive_wall_known_answer(rt, gpu, mode):
    """Every camera ray hits a one-sided light of colour 1 x p1 = 2: each sample adds exactly 2 (main.cu:357-359)."""
    rec = np.zeros((2, 25), np.float32)
    rec[0, 0:9] = [-50, -50, -2, 50, -50, -2, 50, 50, -2]
    rec[1, 0:9] = [-50, -50, -2, 50, 50, -2, -50, 50, -2]
    rec[:, 9:18] = np.tile([0, 0, 1], 3); rec[:, 24] = 0.001
    m = rt.Material(); m.type = 1; m.color = 0xFFFFFF; m.p1 = 2.0; m.image_offset[:] = [-1, -1, -1]
    sc = rt.Scene.from_records(rec, [m])
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=4, width=16, height=16, flags=rt.F_DEFAULT | rt.F_STATS | rt.MODE_FLAGS[mode])
    gpu.reset()
    st = gpu.render(p)
    assert np.array_equal(gpu.readback(), np.full((16, 16, 3), 8.0, np.float32)) and st.rays == st.paths


@pytest.mark.parametrize("mat_type", [3, 5])
def test_white_furnace(rt, gpu, mat_type):
    """A white (albedo 255/255) Lambertian monkey head and floor under the constant backdrop: the diffuse estimators
    (reflectType3: normal + unit sphere; reflectType5 without lights: cosine hemisphere, colour x pdf_cos / pdf) keep
    the throughput at 1, so every path that escapes adds exactly BACKGROUND_COLOR x BACKGROUND_BRIGHTNESS and the
    geometry is invisible.  Only paths that run out of bounces are missing (64 bounces: a fraction of a per cent)."""
    import os
    import tempfile
    txt = ("SET_OUTPUT_WIDTH 96\nSET_OUTPUT_HEIGHT 96\nSET_FOCAL_LENGTH 0.6\nSET_CAM_POS 0 0.3 0\n\n"
           "DEF_MODEL mo\nPATH monkey.obj\n\nDEF_MODEL pl\nPATH plane.obj\n\n"
           f"DEF_MATERIAL w\nTYPE {mat_type}\nCOLOR 1 1 1\n\nDEF_OBJ m\nMODEL mo\nMATERIAL w\n\nDEF_OBJ f\nMODEL pl\nMATERIAL w\n\n"
           "ADD_OBJECT\nOBJECT m\nPOSITION 0 0 -3\nSCALE 1.0\n\nADD_OBJECT\nOBJECT f\nPOSITION 0 -1 -3\nSCALE 6.0\n")
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "furnace.txt")
        open(f, "w").write(txt)
        sc = rt.Scene.load(f)
    gpu.upload(sc)
    spp = 256
    p = rt.default_params(sc.camera, spp=spp, max_bounces=64, flags=rt.F_ANTIALIAS | rt.F_BVH | rt.F_STATS)  # no skybox: flat backdrop
    gpu.reset()
    st = gpu.render(p)
    img = gpu.readback(rt.READ_AVERAGE).astype(np.float64)
    expect = np.float64([1.0, 1.0, 0.05]) * 0.5
    assert st.hits > st.paths // 4                                   # the head and the floor are in view
    assert np.all(img <= expect * (1 + 1e-5))                        # nothing can be brighter than the backdrop
    assert np.all(np.abs(img.mean(axis=(0, 1)) / expect - 1.0) < 0.01)   # energy is conserved to < 1 %
    assert np.percentile(img[..., 0], 1) > 0.45                      # and pixel by pixel the geometry is (nearly) invisible


def test_cosine_weighted_mean_on_the_gpu(rt, gpu):
    """normal + uniform unit-sphere point, normalised, is cosine-weighted: E[cos] = 2/3 (reflectType3, material.h:257-266)."""
    rec = np.zeros((1, 25), np.float32)
    rec[0, 0:9] = [-5, -5, -2, 5, -5, -2, 0, 5, -2]
    rec[0, 9:18] = np.tile([0, 0, 1], 3); rec[0, 24] = 0.001
    m = rt.Material(); m.type = 3; m.color = 0xFFFFFF; m.image_offset[:] = [-1, -1, -1]
    sc = rt.Scene.from_records(rec, [m])
    gpu.upload(sc)
    n = 8192
    r = np.zeros((n, 16), np.float32)
    r[:, 3:6] = [0, 0, -1]; r[:, 6:9] = 1; r[:, 9] = 1; r[:, 10:13] = [0, 0, -2]
    r[:, 14] = np.arange(n, dtype=np.uint32).view(np.float32)
    o = gpu.shade_hits(r, 7)
    d = o[:, 3:6].astype(np.float64)
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-5)
    assert abs(d[:, 2].mean() - 2.0 / 3.0) < 0.01 and np.all(d[:, 2] >= -1e-6)
    assert abs(d[:, 0].mean()) < 0.02 and abs(d[:, 1].mean()) < 0.02   # no preferred azimuth
