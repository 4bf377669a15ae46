"""Host-side statements of the two multi-GPU combine steps (rt_band_rows_of_rank, rt_host_gather_rows,
rt_host_reduce_sum in include/rt.h): the index arithmetic of the band interleave / de-interleave and the summation
order that rt_comm_gather_rows / rt_comm_reduce_sum use on the devices, checked without a GPU against numpy and
against the Python partition of ray-tracer_amd/distributed.py (the torch.distributed path must cut the image the
same way)."""
import ctypes as C
import importlib

import numpy as np
import pytest

_fp = C.POINTER(C.c_float)


def rows_of(L, H, band, n, r):
    buf = np.zeros(max(H, 1), np.int32)
    cnt = L.rt_band_rows_of_rank(H, band, n, r, buf.ctypes.data_as(C.POINTER(C.c_int32)))
    assert cnt >= 0
    assert L.rt_band_rows_of_rank(H, band, n, r, None) == cnt      # counting only
    return buf[:cnt]


@pytest.mark.parametrize("H,band,n", [(1080, 16, 8), (2160, 16, 8), (33, 4, 3), (7, 16, 4), (1, 1, 1), (100, 7, 5)])
def test_band_interleave_is_a_partition_and_matches_the_python_side(rt, H, band, n):
    L = rt.core_lib()
    D = importlib.import_module("ray_tracer_amd.distributed")
    seen = np.zeros(H, np.int32)
    for r in range(n):
        rows = rows_of(L, H, band, n, r)
        assert list(rows) == D.rows_of_rank(H, r, n, band)
        assert np.all(np.diff(rows) > 0)
        seen[rows] += 1
        p = rt.RenderParams(); p.cam.height = H; p.band_rows, p.band_stride, p.band_index = band, n, r
        assert rt.band_rows_of(p) == len(rows) or n == 1            # what rt_render sizes the band-packed accumulator by
    assert np.all(seen == 1)                                        # every row exactly once
    assert L.rt_band_rows_of_rank(H, 0, n, 0, None) == -1 and L.rt_band_rows_of_rank(H, band, n, n, None) == -1


@pytest.mark.parametrize("W,H,band,n", [(13, 33, 4, 3), (64, 100, 16, 8), (5, 7, 16, 4), (3, 1, 1, 1)])
def test_gather_rows_de_interleaves_band_packed_buffers(rt, W, H, band, n):
    L = rt.core_lib()
    rng = np.random.default_rng(W * H + n)
    full = rng.normal(size=(H, W, 3)).astype(np.float32)
    packed = [np.ascontiguousarray(full[rows_of(L, H, band, n, r)]) for r in range(n)]   # what rank r's accumulator holds
    ptrs = (_fp * n)(*[p.ctypes.data_as(_fp) if p.size else None for p in packed])
    out = np.full((H, W, 3), np.nan, np.float32)
    assert L.rt_host_gather_rows(ptrs, n, W, H, band, out.ctypes.data_as(_fp)) == 0
    assert np.array_equal(out, full)


def test_reduce_sum_adds_in_rank_order(rt):
    L = rt.core_lib()
    rng = np.random.default_rng(5)
    n, m = 8, 4096
    parts = [(rng.normal(size=m) * 10.0 ** rng.integers(-3, 4, m)).astype(np.float32) for _ in range(n)]
    ptrs = (_fp * n)(*[p.ctypes.data_as(_fp) for p in parts])
    out = np.zeros(m, np.float32)
    assert L.rt_host_reduce_sum(ptrs, n, m, out.ctypes.data_as(_fp)) == 0
    expect = parts[0].copy()
    for p in parts[1:]:
        expect = (expect + p).astype(np.float32)                    # ((a0 + a1) + a2) + ... in float32
    assert np.array_equal(out, expect)
    exact = np.sum(np.stack(parts).astype(np.float64), 0)
    assert np.allclose(out, exact, rtol=1e-5, atol=1e-3)            # and any other order differs only by rounding
    assert L.rt_host_reduce_sum(None, n, m, out.ctypes.data_as(_fp)) == -1


def test_comm_create_without_a_gpu_fails_loudly(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    with pytest.raises(RuntimeError):
        rt.Comm([0, 1], rt.COMM_LOCAL)
