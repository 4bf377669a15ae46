"""ctypes binding of the CPU oracle (oracle/rt_oracle.cpp).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_package():
    """Import the hyphen-named package directory `ray-tracer_amd` as module `ray_tracer_amd`."""
    if "ray_tracer_amd" in sys.modules:
        return sys.modules["ray_tracer_amd"]
    spec = importlib.util.spec_from_file_location(
        "ray_tracer_amd", os.path.join(ROOT, "ray-tracer_amd", "__init__.py"),
        submodule_search_locations=[os.path.join(ROOT, "ray-tracer_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ray_tracer_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


rt = load_package()
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "lib", "librt_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make oracle`")
        L = C.CDLL(path)
        L.orc_render.argtypes = [C.POINTER(rt.SceneDesc), C.POINTER(rt.RenderParams), _fp,
                                 C.POINTER(rt.RenderStats), C.c_int]
        L.orc_rows.argtypes = [C.POINTER(rt.RenderParams)]
        L.orc_resolve.argtypes = [_fp, C.c_size_t, C.c_int, C.c_int, _fp]
        L.orc_resolve.restype = None
        L.orc_trace_rays.argtypes = [C.POINTER(rt.SceneDesc), _fp, C.c_size_t, C.c_uint32, _ip, _fp, _fp, _fp, _ip]
        L.orc_max_stack.argtypes = [C.POINTER(rt.SceneDesc), _fp, C.c_size_t]
        L.orc_debug_op.argtypes = [C.c_int, _fp, C.c_size_t, _fp, C.POINTER(rt.RenderParams)]
        L.orc_shade_hits.argtypes = [C.POINTER(rt.SceneDesc), _fp, C.c_size_t, C.c_uint32, _fp]
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_philox4x32_10.restype = None
        _lib = L
    return _lib


def _p(a, typ=_fp):
    return a.ctypes.data_as(typ)


def render(scene, params, n_threads=8, accum=None):
    """Returns (raw-sum accumulators [rows, W, 3], RenderStats)."""
    L = lib()
    rows = L.orc_rows(C.byref(params))
    if accum is None:
        accum = np.zeros((rows, params.cam.width, 3), np.float32)
    st = rt.RenderStats()
    rc = L.orc_render(C.byref(scene.desc), C.byref(params), _p(accum), C.byref(st), n_threads)
    assert rc == 0, rc
    return accum, st


def resolve(accum, samples, mode):
    a = np.ascontiguousarray(accum, np.float32)
    out = np.empty_like(a)
    lib().orc_resolve(_p(a), a.size // 3, samples, mode, _p(out))
    return out


def trace_rays(scene, org_dir, flags=rt.F_BVH, want_counts=False):
    rays = np.ascontiguousarray(org_dir, np.float32).reshape(-1, 6)
    n = rays.shape[0]
    prim = np.empty(n, np.int32)
    t = np.empty(n, np.float32)
    pos = np.empty((n, 3), np.float32)
    nrm = np.empty((n, 3), np.float32)
    cnt = np.empty((n, 3), np.int32) if want_counts else None
    rc = lib().orc_trace_rays(C.byref(scene.desc), _p(rays), n, flags, _p(prim, _ip), _p(t), _p(pos), _p(nrm),
                              _p(cnt, _ip) if want_counts else None)
    assert rc == 0, rc
    return prim, t, pos, nrm, cnt


def max_stack(scene, org_dir):
    rays = np.ascontiguousarray(org_dir, np.float32).reshape(-1, 6)
    return lib().orc_max_stack(C.byref(scene.desc), _p(rays), rays.shape[0])


def debug_op(op, records, params=None):
    si, so = rt.OP_STRIDES[op]
    a = np.ascontiguousarray(records).view(np.float32).reshape(-1, si)
    out = np.zeros((a.shape[0], so), np.float32)
    rc = lib().orc_debug_op(op, _p(a), a.shape[0], _p(out), C.byref(params) if params is not None else None)
    assert rc == 0, rc
    return out


def shade_hits(scene, records, seed):
    a = np.ascontiguousarray(records).view(np.float32).reshape(-1, 16)
    out = np.zeros_like(a)
    rc = lib().orc_shade_hits(C.byref(scene.desc), _p(a), a.shape[0], seed, _p(out))
    assert rc == 0, rc
    return out


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)
