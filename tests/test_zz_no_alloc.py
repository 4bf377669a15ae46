"""rt_trace_rays_noalloc (include/rt.h): the closest-hit query for callers that cannot allocate.  Its queue, hit
records and control words live in a file-scope device array of the library; rays and results are device pointers.
Checked through the C-ABI against rt_trace_rays on the same rays, bit for bit, on two calls in a row -- the second
with a smaller input, so that whatever the first call left in the shared array would show."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Hip:
    """Device buffers for the test through the HIP runtime the library itself is bound to (by SONAME, so whichever
    copy is already mapped)."""

    def __init__(self, rt):
        rt.core_lib()
        self.L = C.CDLL("libamdhip64.so.7")
        self.L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.L.hipFree.argtypes = [C.c_void_p]
        self.L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.L.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.bufs = []

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.L.hipMalloc(C.byref(p), max(nbytes, 16)) == 0
        assert self.L.hipMemset(p, 0xCD, max(nbytes, 16)) == 0      # poison: every result must be written by the call
        self.bufs.append(p)
        return p

    def to_device(self, p, a):
        assert self.L.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0

    def to_host(self, a, p):
        assert self.L.hipMemcpy(a.ctypes.data_as(C.c_void_p), p, a.nbytes, 2) == 0
        return a

    def close(self):
        for p in self.bufs:
            self.L.hipFree(p)
        self.bufs = []


def rays_for(scene, n, seed):
    rng = np.random.default_rng(seed)
    nodes = scene.nodes()
    lo, hi = nodes["bbox_min"][0], nodes["bbox_max"][0]
    o = (lo + (hi - lo) * rng.uniform(-0.3, 1.3, (n, 3))).astype(np.float32)
    target = lo + (hi - lo) * rng.uniform(0.2, 0.8, (n, 3))          # aimed into the scene: most rays hit something
    d = (target - o + 1e-3 * rng.normal(size=(n, 3))).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1).astype(np.float32)


def test_no_alloc_entry_point_matches_trace_rays(rt, gpu, scenes):
    L = rt.core_lib()
    hip = Hip(rt)
    try:
        n_max = 50000
        d_rays = hip.alloc(n_max * 24)
        d_prim, d_t, d_pos, d_nrm = hip.alloc(n_max * 4), hip.alloc(n_max * 4), hip.alloc(n_max * 12), hip.alloc(n_max * 12)
        for name, n, seed in [("scene.txt", 50000, 1), ("scene.txt", 777, 2), ("scene_cornellbox.txt", 3000, 3)]:
            sc = scenes(name)
            gpu.upload(sc)
            rays = rays_for(sc, n, seed)
            prim, t, pos, nrm, _ = gpu.trace_rays(rays, rt.F_BVH)          # the existing entry point (host pointers)
            hip.to_device(d_rays, rays)
            for p, nb in ((d_prim, 4), (d_t, 4), (d_pos, 12), (d_nrm, 12)):
                assert hip.L.hipMemset(p, 0xCD, n_max * nb) == 0
            rc = L.rt_trace_rays_noalloc(gpu._h, d_rays, n, rt.F_BVH, d_prim, d_t, d_pos, d_nrm, 1)
            assert rc == 0, L.rt_last_error(gpu._h)
            g_prim = hip.to_host(np.empty(n, np.int32), d_prim)
            g_t = hip.to_host(np.empty(n, np.float32), d_t)
            g_pos = hip.to_host(np.empty((n, 3), np.float32), d_pos)
            g_nrm = hip.to_host(np.empty((n, 3), np.float32), d_nrm)
            assert (prim >= 0).sum() > n // 20                              # the rays do hit something
            assert np.array_equal(g_prim, prim), name                       # ids: bit-exact
            assert np.array_equal(g_t.view(np.uint32), t.view(np.uint32))   # same kernels, same bits
            assert np.array_equal(g_pos.view(np.uint32), pos.view(np.uint32))
            assert np.array_equal(g_nrm.view(np.uint32), nrm.view(np.uint32))
            # nothing beyond the n results of this call was touched
            tail = hip.to_host(np.empty(8, np.uint8), C.c_void_p(d_prim.value + n * 4)) if n < n_max else None
            assert tail is None or np.all(tail == 0xCD)
        # NULL outputs are allowed, sizes beyond the static scratch are refused with a clear error, n = 0 is a no-op
        assert L.rt_trace_rays_noalloc(gpu._h, d_rays, 100, rt.F_BVH, d_prim, None, None, None, 1) == 0
        assert L.rt_trace_rays_noalloc(gpu._h, d_rays, (1 << 18) + 1, rt.F_BVH, d_prim, d_t, d_pos, d_nrm, 1) == -4
        assert b"at most 262144 rays" in L.rt_last_error(gpu._h)
        assert L.rt_trace_rays_noalloc(gpu._h, d_rays, 0, rt.F_BVH, d_prim, d_t, d_pos, d_nrm, 1) == 0
        assert L.rt_trace_rays_noalloc(gpu._h, None, 5, rt.F_BVH, d_prim, d_t, d_pos, d_nrm, 1) == -1
    finally:
        hip.close()
