"""ray-tracer_amd -- thin ctypes plumbing over the C-ABI of the MI355X path-tracing core.

The product is the two shared libraries built from ``csrc/``:

* ``lib/librt_scene.so`` -- host scene assembly (include/rt_scene.h; reference
  src/sceneParser.cpp, src/model.h, src/BVHcreator.cpp, src/noiseTexture.cpp, src/texture.cpp)
* ``lib/librt_amd.so``   -- the HIP render core (include/rt.h; reference src/main.cu:155-407,
  src/material.h and the headers they call)

This module only mirrors the structs of ``include/rt.h`` and forwards calls.  There is
no CPU fallback: if ``librt_amd.so`` is missing or no HIP device is usable, the
constructors raise.

(The directory name has a hyphen, so import it through ``importlib`` -- see
``__graft_entry__.load_package()``.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
LIB_DIR = os.path.join(_HERE, "lib")
ASSET_DIR = os.path.join(REPO_ROOT, "assets")
SCENE_DIR = os.path.join(REPO_ROOT, "scenes")

# ----------------------------------------------------------------------------- structs (include/rt.h)


class BvhNode(C.Structure):
    _fields_ = [("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3), ("child_a", C.c_int32),
                ("child_b", C.c_int32), ("tri_offs", C.c_int32), ("tri_len", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int8), ("pad0_", C.c_int8 * 3), ("color", C.c_int32), ("p1", C.c_float),
                ("p2", C.c_float), ("p3", C.c_float), ("image_offset", C.c_int32 * 3),
                ("image_width", C.c_uint16), ("image_height", C.c_uint16), ("pad1_", C.c_int32),
                ("textures_unused", C.c_uint64)]


class Camera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("num_samples", C.c_int32),
                ("focal_length", C.c_float), ("aperture", C.c_float), ("focus_distance", C.c_float),
                ("pos", C.c_float * 3), ("rot", C.c_float * 3), ("vel", C.c_float * 3),
                ("ang_vel", C.c_float * 3), ("env_width", C.c_int32), ("env_height", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("tris", C.POINTER(C.c_float)), ("n_tri_floats", C.c_size_t), ("n_spheres", C.c_int32),
                ("nodes", C.POINTER(BvhNode)), ("n_nodes", C.c_int32), ("mats", C.POINTER(Material)),
                ("n_mats", C.c_size_t), ("textures", C.POINTER(C.c_float)), ("n_tex_floats", C.c_size_t),
                ("light_tris", C.POINTER(C.c_int32)), ("n_light_tris", C.c_int32)]


class RenderParams(C.Structure):
    _fields_ = [("cam", Camera), ("max_bounces", C.c_int32), ("flags", C.c_uint32), ("seed", C.c_uint32),
                ("sample_begin", C.c_int32), ("sample_count", C.c_int32), ("band_rows", C.c_int32),
                ("band_stride", C.c_int32), ("band_index", C.c_int32), ("background", C.c_float * 3),
                ("background_brightness", C.c_float)]


class RenderStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("paths", C.c_uint64), ("rays", C.c_uint64),
                ("inner_visits", C.c_uint64), ("leaf_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("hits", C.c_uint64), ("texel_fetches", C.c_uint64), ("light_tests", C.c_uint64),
                ("trace_ms", C.c_double), ("iterations", C.c_uint32), ("trace_launches", C.c_uint32),
                ("dev_inner_visits", C.c_uint64), ("dev_lds_inner_visits", C.c_uint64), ("dev_tri_tests", C.c_uint64),
                ("dev_light_tests", C.c_uint64), ("shortcut_rewalks", C.c_uint64), ("dev_wide_visits", C.c_uint64), ("dev_light_loops", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


assert C.sizeof(BvhNode) == 40 and C.sizeof(Material) == 48 and C.sizeof(Camera) == 80

F_SKYBOX, F_ANTIALIAS, F_DEFOCUS, F_BVH, F_SPHERES, F_VOLUME, F_RAYMARCH = 1, 2, 4, 8, 16, 32, 64
F_STATS = 1 << 8
F_MEGAKERNEL = 1 << 9
F_PROFILE = 1 << 10
F_WAVEFRONT = 1 << 11
F_LOCKSTEP = 1 << 12
F_POOL = 1 << 13
F_REFWALK = 1 << 14
F_SHORTCUT_STATS = 1 << 15
# the three schedules of the same arithmetic (bit-identical results)
MODE_FLAGS = {"megakernel": F_MEGAKERNEL, "lockstep": F_MEGAKERNEL | F_LOCKSTEP, "wavefront": F_WAVEFRONT,
              "pool": F_MEGAKERNEL | F_POOL}
F_DEFAULT = F_SKYBOX | F_ANTIALIAS | F_DEFOCUS | F_BVH
READ_RAW_SUM, READ_AVERAGE, READ_TONEMAPPED = 0, 1, 2
(OP_TRI_INTERSECT, OP_AABB_DIST, OP_SPHERE, OP_FLOOR, OP_ITRI_PDF, OP_ITRI_SAMPLE, OP_ATTRACTOR_PDF,
 OP_ATTRACTOR_SAMPLE, OP_TONEMAP, OP_RNG, OP_ENV_INDEX, OP_CAMERA_RAY, OP_PMATH) = range(1, 14)
OP_STRIDES = {OP_TRI_INTERSECT: (15, 4), OP_AABB_DIST: (12, 1), OP_SPHERE: (10, 7), OP_FLOOR: (7, 7),
              OP_ITRI_PDF: (15, 1), OP_ITRI_SAMPLE: (14, 3), OP_ATTRACTOR_PDF: (15, 1),
              OP_ATTRACTOR_SAMPLE: (14, 3), OP_TONEMAP: (3, 3), OP_RNG: (4, 4), OP_ENV_INDEX: (5, 1),
              OP_CAMERA_RAY: (3, 6), OP_PMATH: (3, 2)}
TRI_FLOATS = 25

RT_SYMBOLS = ["rt_create", "rt_destroy", "rt_last_error", "rt_upload_scene", "rt_render", "rt_readback", "rt_readback_rgb8",
              "rt_reset_accum", "rt_release_scratch", "rt_bind_accum", "rt_set_sample_count", "rt_set_stream", "rt_resolve_device",
              "rt_trace_rays", "rt_trace_rays_noalloc", "rt_debug_op", "rt_shade_hits", "rt_layout_create", "rt_layout_get", "rt_layout_destroy",
              "rt_comm_create", "rt_comm_destroy", "rt_comm_last_error", "rt_comm_size", "rt_comm_transport", "rt_comm_context",
              "rt_comm_upload_scene", "rt_comm_render", "rt_comm_reduce_sum", "rt_comm_gather_rows", "rt_comm_reset",
              "rt_comm_readback", "rt_band_rows_of_rank", "rt_host_gather_rows", "rt_host_reduce_sum",
              "rt_abi_version", "rt_render_v2", "rt_comm_render_v2", "rt_layout_get_v2", "rt_comm_rccl_calls"]
ABI_VERSION = 3
SCENE_SYMBOLS = ["rt_scene_last_error", "rt_scene_load", "rt_scene_from_records", "rt_scene_synth_field",
                 "rt_scene_free", "rt_scene_get_desc", "rt_scene_get_camera", "rt_scene_bvh_depth",
                 "rt_bvh_build", "rt_obj_load", "rt_noise_texture", "rt_image_load", "rt_image_save",
                 "rt_image_save_rgb8", "rt_host_free", "rt_bvh_build_sah", "rt_scene_rebuild_bvh"]

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


def _load(name: str) -> C.CDLL:
    path = os.path.join(LIB_DIR, name)
    if name == "librt_amd.so" and os.environ.get("RT_AMD_LIB"):  # A/B builds of the HIP core (tools/ab_build.sh)
        path = os.environ["RT_AMD_LIB"]
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `make` (or __graft_entry__.build()) first; "
                           "there is no fallback path")
    return C.CDLL(path)


_scene_lib: Optional[C.CDLL] = None
_core_lib: Optional[C.CDLL] = None


def scene_lib() -> C.CDLL:
    global _scene_lib
    if _scene_lib is None:
        L = _load("librt_scene.so")
        L.rt_scene_last_error.restype = C.c_char_p
        L.rt_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_void_p)]
        L.rt_scene_from_records.argtypes = [_fp, C.c_size_t, C.POINTER(Material), C.c_size_t, _fp, C.c_size_t,
                                            C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.rt_scene_synth_field.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_char_p, C.c_int, C.c_int,
                                           C.c_int, C.POINTER(C.c_void_p)]
        L.rt_scene_free.argtypes = [C.c_void_p]
        L.rt_scene_free.restype = None
        L.rt_scene_get_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.rt_scene_get_camera.argtypes = [C.c_void_p, C.POINTER(Camera)]
        L.rt_scene_bvh_depth.argtypes = [C.c_void_p]
        L.rt_bvh_build.argtypes = [_fp, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.POINTER(BvhNode)), _ip]
        L.rt_obj_load.argtypes = [C.c_char_p, _fp, C.c_float, _fp, C.c_int, C.POINTER(_fp),
                                  C.POINTER(C.c_size_t)]
        L.rt_noise_texture.argtypes = [C.c_int, C.c_int, _fp]
        L.rt_image_load.argtypes = [C.c_char_p, C.POINTER(_fp), _ip, _ip]
        L.rt_image_save.argtypes = [C.c_char_p, _fp, C.c_int, C.c_int]
        L.rt_image_save_rgb8.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        L.rt_scene_rebuild_bvh.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.rt_host_free.argtypes = [C.c_void_p]
        L.rt_host_free.restype = None
        _scene_lib = L
    return _scene_lib


def core_lib() -> C.CDLL:
    global _core_lib
    if _core_lib is None:
        L = _load("librt_amd.so")
        L.rt_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.rt_destroy.argtypes = [C.c_void_p]
        L.rt_destroy.restype = None
        L.rt_last_error.argtypes = [C.c_void_p]
        L.rt_last_error.restype = C.c_char_p
        L.rt_upload_scene.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.rt_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p]          # round 1's symbol (88-byte stats)
        L.rt_render_v2.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_size_t]
        if L.rt_abi_version() != ABI_VERSION:
            raise RuntimeError(f"librt_amd.so speaks ABI {L.rt_abi_version()}, this binding {ABI_VERSION}: run `make`")
        L.rt_readback.argtypes = [C.c_void_p, _fp, C.c_size_t, C.c_int]
        L.rt_readback_rgb8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.rt_reset_accum.argtypes = [C.c_void_p]
        L.rt_release_scratch.argtypes = [C.c_void_p]
        L.rt_bind_accum.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.rt_set_sample_count.argtypes = [C.c_void_p, C.c_int64]
        L.rt_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_resolve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.rt_trace_rays.argtypes = [C.c_void_p, _fp, C.c_size_t, C.c_uint32, _ip, _fp, _fp, _fp, _ip]
        L.rt_trace_rays_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int]
        L.rt_debug_op.argtypes = [C.c_void_p, C.c_int, _fp, C.c_size_t, _fp, C.POINTER(RenderParams)]
        L.rt_shade_hits.argtypes = [C.c_void_p, _fp, C.c_size_t, C.c_uint32, _fp]
        L.rt_layout_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.rt_layout_get.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_fp), C.POINTER(_ip)]  # round 1's symbol (28-byte info)
        L.rt_layout_get_v2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(_fp), C.POINTER(_ip)]
        L.rt_layout_destroy.argtypes = [C.c_void_p]
        L.rt_layout_destroy.restype = None
        L.rt_comm_create.argtypes = [C.POINTER(C.c_void_p), _ip, C.c_int, C.c_int]
        L.rt_comm_destroy.argtypes = [C.c_void_p]
        L.rt_comm_destroy.restype = None
        L.rt_comm_last_error.argtypes = [C.c_void_p]
        L.rt_comm_last_error.restype = C.c_char_p
        L.rt_comm_size.argtypes = [C.c_void_p]
        L.rt_comm_transport.argtypes = [C.c_void_p]
        L.rt_comm_context.argtypes = [C.c_void_p, C.c_int]
        L.rt_comm_context.restype = C.c_void_p
        L.rt_comm_upload_scene.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.rt_comm_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_int, C.c_int, C.c_void_p]  # round 2's symbol
        L.rt_comm_render_v2.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.rt_comm_rccl_calls.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.rt_comm_reduce_sum.argtypes = [C.c_void_p]
        L.rt_comm_gather_rows.argtypes = [C.c_void_p]
        L.rt_comm_reset.argtypes = [C.c_void_p]
        L.rt_comm_readback.argtypes = [C.c_void_p, _fp, C.c_size_t, C.c_int]
        L.rt_band_rows_of_rank.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _ip]
        L.rt_host_gather_rows.argtypes = [C.POINTER(_fp), C.c_int, C.c_int, C.c_int, C.c_int, _fp]
        L.rt_host_reduce_sum.argtypes = [C.POINTER(_fp), C.c_int, C.c_size_t, _fp]
        _core_lib = L
    return _core_lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray, typ=_fp):
    return a.ctypes.data_as(typ)


# ----------------------------------------------------------------------------- host scene


class Scene:
    """Host scene = the arrays of render2's argument list (rt_scene_desc)."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)
        self.desc = SceneDesc()
        self.camera = Camera()
        L = scene_lib()
        if L.rt_scene_get_desc(self._h, C.byref(self.desc)) or L.rt_scene_get_camera(self._h, C.byref(self.camera)):
            raise RuntimeError(L.rt_scene_last_error().decode())

    @staticmethod
    def _check(rc: int, out: C.c_void_p) -> "Scene":
        if rc != 0:
            raise RuntimeError("scene: " + scene_lib().rt_scene_last_error().decode())
        return Scene(out.value)

    @classmethod
    def load(cls, scene_path: str, asset_root: Optional[str] = ASSET_DIR, env_override: Optional[str] = None,
             enable_spheres: bool = False, bvh_max_depth: int = 16, bvh_min_tris: int = 4) -> "Scene":
        if not os.path.isabs(scene_path) and not os.path.exists(scene_path):
            scene_path = os.path.join(SCENE_DIR, scene_path)
        out = C.c_void_p()
        rc = scene_lib().rt_scene_load(scene_path.encode(), asset_root.encode() if asset_root else None,
                                       env_override.encode() if env_override else None, int(enable_spheres),
                                       bvh_max_depth, bvh_min_tris, C.byref(out))
        return cls._check(rc, out)

    @classmethod
    def from_records(cls, tris, mats, textures=None, env_w=0, env_h=0, bvh_max_depth=16, bvh_min_tris=4):
        tris = _f32(tris).ravel()
        marr = (Material * len(mats))(*mats)
        tex = _f32(textures).ravel() if textures is not None else None
        out = C.c_void_p()
        rc = scene_lib().rt_scene_from_records(_ptr(tris), tris.size, marr, len(mats),
                                               _ptr(tex) if tex is not None else None,
                                               tex.size if tex is not None else 0, env_w, env_h,
                                               bvh_max_depth, bvh_min_tris, C.byref(out))
        return cls._check(rc, out)

    @classmethod
    def synth_field(cls, instances: int, seed: int = 1234, obj_path: Optional[str] = None,
                    env_path: Optional[str] = None, env_upsample: int = 1, bvh_max_depth=16, bvh_min_tris=4):
        obj_path = obj_path or os.path.join(ASSET_DIR, "dragon.obj")
        out = C.c_void_p()
        rc = scene_lib().rt_scene_synth_field(obj_path.encode(), instances, seed,
                                              env_path.encode() if env_path else None, env_upsample,
                                              bvh_max_depth, bvh_min_tris, C.byref(out))
        return cls._check(rc, out)

    def rebuild_bvh(self, mode: str = "sah", max_depth: int = 0, leaf_tris: int = 0) -> "Scene":
        """In place: "reference" = createBVH (depth 16, min 4), "sah" = the non-parity fast mode (depth 24, leaves <= 2)."""
        L = scene_lib()
        if L.rt_scene_rebuild_bvh(self._h, {"reference": 0, "sah": 1}[mode], max_depth, leaf_tris):
            raise RuntimeError("scene: " + L.rt_scene_last_error().decode())
        if L.rt_scene_get_desc(self._h, C.byref(self.desc)):
            raise RuntimeError(L.rt_scene_last_error().decode())
        return self

    def close(self):
        if self._h:
            scene_lib().rt_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # numpy views (valid while the scene lives)
    @property
    def n_tris(self) -> int:
        return self.desc.n_tri_floats // TRI_FLOATS

    @property
    def prefix(self) -> int:
        return 1 + 5 * self.desc.n_spheres if self.desc.n_spheres > 0 else 0

    def tris(self) -> np.ndarray:
        n = self.desc.n_tri_floats
        a = np.ctypeslib.as_array(self.desc.tris, shape=(self.prefix + n,))
        return a[self.prefix:].reshape(-1, TRI_FLOATS)

    def nodes(self) -> np.ndarray:
        dt = np.dtype([("bbox_min", "<f4", 3), ("bbox_max", "<f4", 3), ("child_a", "<i4"), ("child_b", "<i4"),
                       ("tri_offs", "<i4"), ("tri_len", "<i4")])
        n = self.desc.n_nodes
        buf = C.cast(self.desc.nodes, C.POINTER(C.c_char * (40 * n))).contents
        return np.frombuffer(buf, dtype=dt, count=n)

    def materials(self):
        return [self.desc.mats[i] for i in range(self.desc.n_mats)]

    def textures(self) -> np.ndarray:
        n = self.desc.n_tex_floats
        if n == 0:
            return np.zeros(0, np.float32)
        return np.ctypeslib.as_array(self.desc.textures, shape=(n,))

    def lights(self) -> np.ndarray:
        n = self.desc.n_light_tris
        if n == 0:
            return np.zeros(0, np.int32)
        return np.ctypeslib.as_array(self.desc.light_tris, shape=(n,))

    def bvh_depth(self) -> int:
        return scene_lib().rt_scene_bvh_depth(self._h)


def default_params(cam: Camera, spp: Optional[int] = None, max_bounces: int = 6, seed: int = 37811,
                   flags: int = F_DEFAULT, width: Optional[int] = None, height: Optional[int] = None) -> RenderParams:
    """Reference defaults: MAX_BOUNCES 6 (config.h:3), seed 37811 (main.cu:171), background
    (1,1,0.05)*0.5 (config.h:6-7)."""
    p = RenderParams()
    C.memmove(C.byref(p.cam), C.byref(cam), C.sizeof(Camera))
    if width:
        p.cam.width = width
    if height:
        p.cam.height = height
    p.max_bounces = max_bounces
    p.flags = flags
    p.seed = seed
    p.sample_begin = 0
    p.sample_count = spp if spp is not None else cam.num_samples
    p.band_rows, p.band_stride, p.band_index = 0, 1, 0
    p.background[:] = [1.0, 1.0, 0.05]
    p.background_brightness = 0.5
    return p


def band_rows_of(params: RenderParams) -> int:
    """Number of accumulator rows a call with these params renders (rt.h: tile-row partition)."""
    H = params.cam.height
    if params.band_stride <= 1 or params.band_rows <= 0:
        return H
    rows = 0
    for y0 in range(0, H, params.band_rows):
        if (y0 // params.band_rows) % params.band_stride == params.band_index:
            rows += min(params.band_rows, H - y0)
    return rows


class LayoutInfo(C.Structure):  # rt_layout_info, include/rt.h
    _fields_ = [("n_inner", C.c_int32), ("n_sub", C.c_int32), ("n_tris", C.c_int32), ("root_entry", C.c_int32),
                ("depth", C.c_int32), ("box_pad", C.c_float), ("cull_far", C.c_float), ("n_short", C.c_int32),
                ("short_root", C.c_int32), ("short_base", C.c_int32), ("short_depth", C.c_int32), ("short_single", C.c_int32), ("short_cut", C.c_int32),
                ("n_wide", C.c_int32), ("wide_need", C.c_int32), ("wide_sp_max", C.c_int32),
                ("light_root", C.c_int32), ("n_light_items", C.c_int32), ("light_need", C.c_int32)]


def device_layout(scene: "Scene", light_items: bool = False):
    """The layout rt_upload_scene would copy to the GPU, built on the host (rt_layout_create; no device needed):
    (LayoutInfo, items [n_items, 16] float32, places [n_tris] int32), copies owned by the caller.  light_items: the
    items include what stands behind the triangles (one item of padding, then info.n_light_items light items)."""
    L = core_lib()
    h = C.c_void_p()
    err = C.create_string_buffer(256)
    rc = L.rt_layout_create(C.byref(scene.desc), C.byref(h), err, 256)
    if rc != 0:
        raise RuntimeError(f"rt_layout_create: {rc} {err.value.decode()}")
    try:
        info = LayoutInfo()
        items_p, places_p = _fp(), _ip()
        L.rt_layout_get_v2(h, C.byref(info), C.sizeof(info), C.byref(items_p), C.byref(places_p))
        n_items = info.n_inner + info.n_tris + ((1 + info.n_light_items) if (light_items and info.n_light_items) else 0)
        items = np.ctypeslib.as_array(items_p, shape=(max(n_items, 1), 16))[:n_items].copy()
        places = np.ctypeslib.as_array(places_p, shape=(max(info.n_tris, 1),))[:info.n_tris].copy()
    finally:
        L.rt_layout_destroy(h)
    return info, items, places


# ----------------------------------------------------------------------------- device context


class Renderer:
    """One rt_context (one HIP device).  Mirrors the render call of main(): upload, launch, readback."""

    def __init__(self, device_id: int = -1):
        self._L = core_lib()
        self._h = C.c_void_p()
        rc = self._L.rt_create(C.byref(self._h), device_id)
        if rc != 0:
            raise RuntimeError(f"rt_create failed ({rc}): no usable HIP device; there is no CPU fallback")
        self._rows = 0
        self._width = 0

    def _ck(self, rc: int):
        if rc != 0:
            raise RuntimeError(f"rt error {rc}: {self._L.rt_last_error(self._h).decode()}")

    def close(self):
        if self._h:
            self._L.rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, scene: Scene):
        self._ck(self._L.rt_upload_scene(self._h, C.byref(scene.desc)))

    def render(self, params: RenderParams, want_stats: bool = True) -> Optional[RenderStats]:
        st = RenderStats()
        self._ck(self._L.rt_render_v2(self._h, C.byref(params), C.byref(st), C.sizeof(st)))
        self._rows = band_rows_of(params)
        self._width = params.cam.width
        return st if want_stats else None

    def readback(self, mode: int = READ_RAW_SUM) -> np.ndarray:
        out = np.empty((self._rows, self._width, 3), np.float32)
        self._ck(self._L.rt_readback(self._h, _ptr(out), out.size, mode))
        return out

    def readback_rgb8(self) -> np.ndarray:
        """Average + tone map + 8-bit conversion on the device (what the reference writes as PNG)."""
        out = np.empty((self._rows, self._width, 3), np.uint8)
        self._ck(self._L.rt_readback_rgb8(self._h, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def reset(self):
        self._ck(self._L.rt_reset_accum(self._h))

    def release_scratch(self):
        """Free the buffers the context keeps between calls (per-path radiance, ray queues, readback scratch)."""
        self._ck(self._L.rt_release_scratch(self._h))

    def bind_accum(self, dev_ptr: int, n_floats: int):
        self._ck(self._L.rt_bind_accum(self._h, C.c_void_p(dev_ptr), n_floats))

    def set_sample_count(self, n: int):
        self._ck(self._L.rt_set_sample_count(self._h, n))

    def set_stream(self, stream_ptr: int):
        self._ck(self._L.rt_set_stream(self._h, C.c_void_p(stream_ptr)))

    def resolve_device(self, dev_ptr: int, n_floats: int, mode: int):
        self._ck(self._L.rt_resolve_device(self._h, C.c_void_p(dev_ptr), n_floats, mode))

    def trace_rays(self, org_dir, flags: int = F_BVH, want_counts: bool = False):
        rays = _f32(org_dir).reshape(-1, 6)
        n = rays.shape[0]
        prim = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        pos = np.empty((n, 3), np.float32)
        nrm = np.empty((n, 3), np.float32)
        cnt = np.empty((n, 3), np.int32) if want_counts else None
        self._ck(self._L.rt_trace_rays(self._h, _ptr(rays), n, flags, _ptr(prim, _ip), _ptr(t), _ptr(pos),
                                       _ptr(nrm), _ptr(cnt, _ip) if want_counts else None))
        return prim, t, pos, nrm, cnt

    def debug_op(self, op: int, records, params: Optional[RenderParams] = None) -> np.ndarray:
        si, so = OP_STRIDES[op]
        a = np.ascontiguousarray(records).view(np.float32).reshape(-1, si)
        out = np.zeros((a.shape[0], so), np.float32)
        self._ck(self._L.rt_debug_op(self._h, op, _ptr(a), a.shape[0], _ptr(out),
                                     C.byref(params) if params is not None else None))
        return out

    def shade_hits(self, records, seed: int) -> np.ndarray:
        a = np.ascontiguousarray(records).view(np.float32).reshape(-1, 16)
        out = np.zeros_like(a)
        self._ck(self._L.rt_shade_hits(self._h, _ptr(a), a.shape[0], seed, _ptr(out)))
        return out


PART_ROWS, PART_SAMPLES = 0, 1
COMM_RCCL, COMM_LOCAL = 0, 1


class Comm:
    """Several GPUs from one process (rt_comm_*, include/rt.h): one context per device, one collective per frame."""

    def __init__(self, device_ids, transport: int = COMM_RCCL):
        self._L = core_lib()
        self._h = C.c_void_p()
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        rc = self._L.rt_comm_create(C.byref(self._h), ids, len(device_ids), transport)
        if rc != 0:
            raise RuntimeError(f"rt_comm_create failed ({rc})")
        self._shape = (0, 0)

    def _ck(self, rc: int):
        if rc != 0:
            raise RuntimeError(f"rt_comm error {rc}: {self._L.rt_comm_last_error(self._h).decode()}")

    @property
    def size(self) -> int:
        return self._L.rt_comm_size(self._h)

    @property
    def transport(self) -> int:
        return self._L.rt_comm_transport(self._h)

    def upload(self, scene: Scene):
        self._ck(self._L.rt_comm_upload_scene(self._h, C.byref(scene.desc)))

    def render(self, params: RenderParams, partition: int, band_rows: int = 16) -> RenderStats:
        st = RenderStats()
        self._ck(self._L.rt_comm_render_v2(self._h, C.byref(params), partition, band_rows, C.byref(st), C.sizeof(st)))
        self._shape = (params.cam.height, params.cam.width)
        return st

    def reset(self):
        self._ck(self._L.rt_comm_reset(self._h))

    def rccl_calls(self):
        """(sum-reduces, send / receive gathers) of this comm that went through RCCL."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._ck(self._L.rt_comm_rccl_calls(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def readback(self, mode: int = READ_RAW_SUM) -> np.ndarray:
        out = np.empty(self._shape + (3,), np.float32)
        self._ck(self._L.rt_comm_readback(self._h, _ptr(out), out.size, mode))
        return out

    def close(self):
        if self._h:
            self._L.rt_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
