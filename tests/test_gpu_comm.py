"""rt_comm_* (include/rt.h: several GPUs from one process) through the C-ABI on ONE device: the LOCAL transport with
three ranks sharing the device runs the same partition, the same kernels and the same de-interleave / summation as
the RCCL transport on eight devices; the RCCL transport itself is exercised with the one rank a one-GPU box allows:
ncclReduce in place and a grouped ncclSend / ncclRecv of the rank to itself really run (rt_comm_rccl_calls counts them).
What no one-GPU box can show -- ncclReduce / ncclSend / ncclRecv BETWEEN devices -- stays unverified until an 8-GPU run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_ranks,band", [(3, 16), (8, 4), (2, 7)])
def test_rows_partition_gathered_image_is_bit_identical(rt, gpu, scenes, n_ranks, band):
    sc = scenes("scene.txt")
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=3, width=160, height=101, flags=rt.F_DEFAULT | rt.F_STATS)
    gpu.reset(); st1 = gpu.render(p); one = gpu.readback()
    comm = rt.Comm([0] * n_ranks, rt.COMM_LOCAL)
    try:
        assert comm.size == n_ranks and comm.transport == rt.COMM_LOCAL
        comm.upload(sc)
        st = comm.render(p, rt.PART_ROWS, band)
        img = comm.readback()
        assert np.array_equal(img, one)                               # keyed RNG: (pixel, sample) only
        assert (st.paths, st.rays, st.tri_tests, st.hits) == (st1.paths, st1.rays, st1.tri_tests, st1.hits)
        assert (st.dev_tri_tests, st.shortcut_rewalks, st.dev_light_tests) == (st1.dev_tri_tests, st1.shortcut_rewalks, st1.dev_light_tests)
        comm.render(p, rt.PART_ROWS, band)                            # accumulates like rt_render: twice the sums
        assert np.array_equal(comm.readback(), one + one)
        avg = comm.readback(rt.READ_AVERAGE)
        assert np.allclose(avg, one / np.float32(3), rtol=1e-6)
        comm.reset()
        comm.render(p, rt.PART_ROWS, band)
        assert np.array_equal(comm.readback(), one)
    finally:
        comm.close()


@pytest.mark.parametrize("name", ["scene.txt", "scene2_is.txt"])
def test_rows_and_sample_slices_on_the_default_walk(rt, gpu, scenes, name):
    """The same two partitions WITHOUT counters, i.e. through the kernel a plain call runs (the shortcut walk, its compact
    LDS window, the hit fetch of the vertex pass; with light queries on scene2_is.txt): rows bit-identical, sample slices
    equal to the rank-ordered sum of the slices rendered one by one."""
    sc = scenes(name)
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=6, width=136, height=77, flags=rt.F_DEFAULT)
    gpu.reset(); gpu.render(p, want_stats=False); one = gpu.readback()
    parts = []
    for r in range(3):
        q = rt.RenderParams.from_buffer_copy(p)
        q.sample_begin, q.sample_count = 2 * r, 2
        gpu.reset(); gpu.render(q, want_stats=False); parts.append(gpu.readback())
    comm = rt.Comm([0] * 3, rt.COMM_LOCAL)
    try:
        comm.upload(sc)
        comm.render(p, rt.PART_ROWS, 5)
        assert np.array_equal(comm.readback(), one)
        comm.reset()
        comm.render(p, rt.PART_SAMPLES)
        assert np.array_equal(comm.readback(), (parts[0] + parts[1]) + parts[2])
    finally:
        comm.close()


@pytest.mark.parametrize("n_ranks", [2, 4])
def test_sample_slices_summed_in_rank_order(rt, gpu, scenes, n_ranks):
    sc = scenes("scene_cornellbox.txt")
    gpu.upload(sc)
    spp = 10                                                          # does not divide by 4: slices of 2, 3, 2, 3
    p = rt.default_params(sc.camera, spp=spp, width=96, height=64, max_bounces=4, flags=rt.F_DEFAULT | rt.F_STATS)
    gpu.reset(); st1 = gpu.render(p); one = gpu.readback()
    parts = []
    for r in range(n_ranks):                                          # the slices one by one on a single context
        q = rt.RenderParams.from_buffer_copy(p)
        b, e = spp * r // n_ranks, spp * (r + 1) // n_ranks
        q.sample_begin, q.sample_count = b, e - b
        gpu.reset(); gpu.render(q); parts.append(gpu.readback())
    expect = parts[0].copy()
    for a in parts[1:]:
        expect = expect + a
    comm = rt.Comm([0] * n_ranks, rt.COMM_LOCAL)
    try:
        comm.upload(sc)
        st = comm.render(p, rt.PART_SAMPLES)
        img = comm.readback()
        assert np.array_equal(img, expect)                            # same slices, same order of additions
        assert np.allclose(img, one, rtol=1e-6, atol=1e-6)            # vs one GPU: float reassociation only
        assert (st.paths, st.rays) == (st1.paths, st1.rays)
        tm = comm.readback(rt.READ_TONEMAPPED)
        gpu.reset(); gpu.render(p)
        assert np.allclose(tm, gpu.readback(rt.READ_TONEMAPPED), rtol=1e-5, atol=1e-6)
    finally:
        comm.close()


def test_rccl_transport_with_the_one_rank_this_box_has(rt, gpu, scenes):
    """ncclCommInitAll, then with the single rank RCCL allows on one device: ncclReduce (in place, root 0) for the sample
    slices and a grouped ncclSend / ncclRecv of the rank to itself for the rows -- both calls are really issued (the
    comm counts them) and both combine steps give the one-GPU image.  Transfers between devices are NOT covered."""
    sc = scenes("scene_cornellbox.txt")
    gpu.upload(sc)
    p = rt.default_params(sc.camera, spp=2, width=64, height=48, max_bounces=4)
    gpu.reset(); gpu.render(p); one = gpu.readback()
    comm = rt.Comm([0], rt.COMM_RCCL)
    try:
        assert comm.transport == rt.COMM_RCCL, "librccl.so.1 did not load"
        comm.upload(sc)
        assert comm.rccl_calls() == (0, 0)
        comm.render(p, rt.PART_SAMPLES)
        assert np.array_equal(comm.readback(), one)
        assert comm.rccl_calls() == (1, 0)                            # ncclReduce ran
        comm.reset()
        comm.render(p, rt.PART_ROWS)
        assert np.array_equal(comm.readback(), one)
        assert comm.rccl_calls() == (1, 1)                            # ... and the grouped self send / receive
        comm.render(p, rt.PART_ROWS)                                  # accumulates: twice the sums, through the staging copy again
        assert np.array_equal(comm.readback(), one + one) and comm.rccl_calls() == (1, 2)
    finally:
        comm.close()
    with pytest.raises(RuntimeError):                                 # RCCL wants one rank per device
        rt.Comm([0, 0], rt.COMM_RCCL)
