// rt_core.hip -- kernels and C-ABI (include/rt.h) of the MI355X path-tracing core.
//
// Replaces the render call of the reference: uploads src/main.cu:418-460, kernel
// render2 src/main.cu:155-407 (launch :472-482), readback :500-501.
//
// Kernel design (gfx950):
//  * rt_render_kernel -- wavefront-persistent: a grid sized to the chip's residency,
//    256-thread blocks = 4 independent waves.  Each wave pulls 8x8-pixel tiles from a
//    global atomic counter; lane = pixel.  Inside a tile a lane runs a path state
//    machine with REGENERATION: when its path ends it starts its next sample in the
//    same iteration, so every iteration of the wave loop is "all lanes trace one ray,
//    all lanes shade one hit" instead of the reference's nested per-thread loops.
//    Traversal stack: per-lane short stack in LDS ([entry][lane] => conflict-free).
//    Node/triangle/material fetches are 16-byte vector loads from the 64/48-byte
//    records of rt_device.h.
//  * one lane owns one pixel for the whole call => accumulation is a plain
//    read-modify-write (no float atomics, bit-reproducible).
//  * raw-sum accumulators + separate resolve kernel (needed for RCCL sum-reduce and
//    progressive batches).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt.h"
#include "rt_device.h"
#include "rt_pmath.h"

using namespace rtd;

// =============================================================================== kernels

__device__ __forceinline__ int row_to_y(const Params& P, int r) {
    if (P.band_stride <= 1 || P.band_rows <= 0) return r;
    const int k = r / P.band_rows;  // ordinal of the owned band
    return (k * P.band_stride + P.band_index) * P.band_rows + (r - k * P.band_rows);
}

template <bool STATS>
__global__ __launch_bounds__(64 * RT_WAVES_PER_BLOCK) void rt_render_kernel(Scene sc, Params P, float* __restrict__ accum,
                                                                              unsigned int* __restrict__ tile_counter,
                                                                              unsigned long long* __restrict__ stats) {
    __shared__ int lds_stack[RT_WAVES_PER_BLOCK][RT_STACK_ENTRIES][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int* stk = &lds_stack[wave][0][lane];
    const bool use_bvh = (P.flags & RT_F_BVH) != 0;
    const bool spheres_on = (P.flags & RT_F_SPHERES) != 0 && sc.n_spheres > 0;
    Counters cn;
    cn.clear();

    for (;;) {
        unsigned int tile = 0;
        if (lane == 0) tile = atomicAdd(tile_counter, 1u);
        tile = __builtin_amdgcn_readfirstlane(tile);
        if (tile >= (unsigned)P.n_tiles) break;
        const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
        const int x = tx * 8 + (lane & 7);
        const int r = ty * 8 + (lane >> 3);
        const bool in_image = x < P.width && r < P.rows;
        const int y = row_to_y(P, r);
        const uint32_t pixel = (uint32_t)(x + y * P.width);  // main.cu:171

        f3 color = mk(0.f, 0.f, 0.f);
        int k = 0;                                  // samples started
        const int spp = in_image ? P.sample_count : 0;
        bool alive = false;
        int bounce = 0;
        PathState ray;
        Rng rng;
        rng.init(P.seed, pixel, 0);

        for (;;) {
            if (!alive && k < spp) {  // regenerate: camera sample k (main.cu:178-217)
                rng.init(P.seed, pixel, (uint32_t)(P.sample_begin + k));
                k++;
                camera_ray(P, x, y, rng, ray.o, ray.d);
                ray.thr = mk(1.f, 1.f, 1.f);
                ray.emission = mk(0.f, 0.f, 0.f);
                ray.ri = 1.0f;
                ray.volDensity = 0.0f;
                ray.volColor = mk(1.f, 1.f, 1.f);
                bounce = 0;
                alive = true;
                if (STATS) cn.paths++;
            }
            if (!__any(alive)) break;
            if (alive) {  // one iteration of the bounce loop, main.cu:220-389
                rng.seek(RT_RNG_DIM_BOUNCE0 + RT_RNG_DIMS_PER_BOUNCE * (uint32_t)bounce);
                int hitTri;
                float minD2;
                f3 hitPos;
                trace_closest<STATS>(sc, ray.o, ray.d, use_bvh, stk, hitTri, minD2, hitPos, cn);
                bool handled = false;
                if (spheres_on) {  // main.cu:305-346
                    int hitSphere = -1;
                    f3 sPos, sNrm;
                    for (int i = 0; i < sc.n_spheres; i++) {
                        const float4 s0 = sc.spheres[2 * i];
                        f3 hp, hn;
                        if (intersect_sphere(ray.o, ray.d, mk(s0.x, s0.y, s0.z), s0.w, hp, hn)) {
                            const float d2 = len2(hp - ray.o);
                            if (d2 >= minD2) continue;
                            minD2 = d2;
                            sPos = hp;
                            sNrm = hn;
                            hitSphere = i;
                        }
                    }
                    if (hitSphere >= 0) {
                        // material id from the sphere record (the reference reads materials[i+4], main.cu:341)
                        const int mat = __float_as_int(sc.spheres[2 * hitSphere + 1].x);
                        if (STATS) cn.hits++;
                        shade_sphere<STATS>(sc, ray, sNrm, sPos, mat, rng, cn);
                        handled = true;  // `continue` at main.cu:343: no renormalise
                    }
                }
                if (!handled) {
                    if (hitTri >= 0) {
                        if (STATS) cn.hits++;
                        const bool terminated = shade_triangle<STATS>(sc, ray, hitPos, hitTri, rng, cn);
                        if (terminated) {  // main.cu:357-360
                            const f3 e = ray.emission * ray.thr;
                            color.x += e.x; color.y += e.y; color.z += e.z;
                            alive = false;
                        } else {
                            ray.d = normalize_keep(ray.d);  // main.cu:361
                        }
                    } else {  // main.cu:374-387
                        f3 e;
                        if ((P.flags & RT_F_SKYBOX) && P.env_w > 0) {
                            const long long envI = env_index(ray.d, P.env_w, P.env_h);
                            if (STATS) cn.texels++;
                            e = fetch_texel(sc, envI) * ray.thr * P.bg_brightness;
                        } else {
                            e = ld3(P.bg) * ray.thr * P.bg_brightness;
                        }
                        color.x += e.x; color.y += e.y; color.z += e.z;
                        alive = false;
                    }
                }
                bounce++;
                if (bounce >= P.max_bounces) alive = false;  // path exhausted: contributes 0
            }
        }
        if (in_image) {
            float* a = accum + ((size_t)r * P.width + x) * 3;
            a[0] += color.x;
            a[1] += color.y;
            a[2] += color.z;
        }
    }

    if (STATS) {  // wave-reduce, one atomic per counter per wave
        uint32_t v[8] = {cn.paths, cn.rays, cn.inner, cn.leaf, cn.tri, cn.hits, cn.texels, cn.lights};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            unsigned long long s = v[i];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0 && s) atomicAdd(&stats[i], s);
        }
    }
}

// avg + tone map of main.cu:392-406 applied to raw sums
__global__ void rt_resolve_kernel(const float* __restrict__ accum, float* __restrict__ out, size_t n_pixels,
                                  float samples, int mode) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    f3 c = mk(accum[i * 3], accum[i * 3 + 1], accum[i * 3 + 2]);
    if (mode != RT_READ_RAW_SUM) {
        c.x /= samples; c.y /= samples; c.z /= samples;
        if (mode == RT_READ_TONEMAPPED) c = tone_map(c);
    }
    out[i * 3] = c.x; out[i * 3 + 1] = c.y; out[i * 3 + 2] = c.z;
}

__global__ __launch_bounds__(64 * RT_WAVES_PER_BLOCK) void rt_trace_kernel(Scene sc, const float* __restrict__ rays,
                                                                             size_t n, unsigned flags, int* prim,
                                                                             float* t, float* hit_pos, float* normal3,
                                                                             int* counts) {
    __shared__ int lds_stack[RT_WAVES_PER_BLOCK][RT_STACK_ENTRIES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int* stk = &lds_stack[wave][0][lane];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 o = ld3(rays + i * 6), d = ld3(rays + i * 6 + 3);
    Counters cn;
    cn.clear();
    int hitTri;
    float minD2;
    f3 hitPos = mk(0.f, 0.f, 0.f);
    trace_closest<true>(sc, o, d, (flags & RT_F_BVH) != 0, stk, hitTri, minD2, hitPos, cn);
    if (prim) prim[i] = hitTri;
    if (t) t[i] = hitTri >= 0 ? sqrtf(minD2) : -1.0f;
    if (hit_pos) {
        hit_pos[i * 3] = hitTri >= 0 ? hitPos.x : 0.f;
        hit_pos[i * 3 + 1] = hitTri >= 0 ? hitPos.y : 0.f;
        hit_pos[i * 3 + 2] = hitTri >= 0 ? hitPos.z : 0.f;
    }
    if (normal3) {
        f3 nn = mk(0.f, 0.f, 0.f);
        if (hitTri >= 0) nn = hit_frame(sc, hitTri, hitPos).normal;
        normal3[i * 3] = nn.x; normal3[i * 3 + 1] = nn.y; normal3[i * 3 + 2] = nn.z;
    }
    if (counts) {
        counts[i * 3] = (int)cn.inner;
        counts[i * 3 + 1] = (int)cn.leaf;
        counts[i * 3 + 2] = (int)cn.tri;
    }
}

__global__ void rt_shade_kernel(Scene sc, const float* __restrict__ in, size_t n, unsigned seed, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* q = in + i * 16;
    float* o = out + i * 16;
    PathState ray;
    ray.o = ld3(q); ray.d = ld3(q + 3); ray.thr = ld3(q + 6);
    ray.emission = mk(0.f, 0.f, 0.f);
    ray.ri = q[9];
    ray.volDensity = 0.f;
    ray.volColor = mk(1.f, 1.f, 1.f);
    const f3 hitPos = ld3(q + 10);
    const int tri = __float_as_int(q[13]);
    Rng rng;
    rng.init(seed, __float_as_uint(q[14]), __float_as_uint(q[15]));
    rng.seek(RT_RNG_DIM_BOUNCE0);
    Counters cn;
    cn.clear();
    bool term = true;
    if (tri >= 0 && tri < sc.n_tris) term = shade_triangle<false>(sc, ray, hitPos, tri, rng, cn);
    o[0] = ray.o.x; o[1] = ray.o.y; o[2] = ray.o.z;
    o[3] = ray.d.x; o[4] = ray.d.y; o[5] = ray.d.z;
    o[6] = ray.thr.x; o[7] = ray.thr.y; o[8] = ray.thr.z;
    o[9] = ray.emission.x; o[10] = ray.emission.y; o[11] = ray.emission.z;
    o[12] = ray.ri;
    o[13] = ray.volDensity;
    o[14] = __uint_as_float(term ? 1u : 0u);
    o[15] = 0.f;
}

__global__ void rt_debug_kernel(int op, const float* __restrict__ in, size_t n, float* __restrict__ out, Params P) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    switch (op) {
        case RT_OP_TRI_INTERSECT: {
            const float* q = in + r * 15;
            const f3 v0 = ld3(q + 6);
            f3 pos = mk(0.f, 0.f, 0.f);
            const bool hit = ray_triangle(ld3(q), ld3(q + 3), v0, ld3(q + 9) - v0, ld3(q + 12) - v0, pos);
            out[r * 4] = hit ? 1.f : 0.f; out[r * 4 + 1] = pos.x; out[r * 4 + 2] = pos.y; out[r * 4 + 3] = pos.z;
            break;
        }
        case RT_OP_AABB_DIST: {
            const float* q = in + r * 12;
            const f3 d = ld3(q + 3);
            out[r] = aabb_distance(ld3(q), mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), ld3(q + 6), ld3(q + 9));
            break;
        }
        case RT_OP_SPHERE: {
            const float* q = in + r * 10;
            f3 pos = mk(0.f, 0.f, 0.f), nrm = mk(0.f, 0.f, 0.f);
            const bool hit = intersect_sphere(ld3(q), ld3(q + 3), ld3(q + 6), q[9], pos, nrm);
            float* o = out + r * 7;
            o[0] = hit ? 1.f : 0.f; o[1] = pos.x; o[2] = pos.y; o[3] = pos.z; o[4] = nrm.x; o[5] = nrm.y; o[6] = nrm.z;
            break;
        }
        case RT_OP_FLOOR: {
            const float* q = in + r * 7;
            f3 pos = mk(0.f, 0.f, 0.f), nrm = mk(0.f, 0.f, 0.f);
            const bool hit = intersect_floor(ld3(q), ld3(q + 3), q[6], pos, nrm);
            float* o = out + r * 7;
            o[0] = hit ? 1.f : 0.f; o[1] = pos.x; o[2] = pos.y; o[3] = pos.z; o[4] = nrm.x; o[5] = nrm.y; o[6] = nrm.z;
            break;
        }
        case RT_OP_ITRI_PDF: {
            const float* q = in + r * 15;
            const f3 p1 = ld3(q + 6), p2 = ld3(q + 9), p3 = ld3(q + 12);
            const f3 c = cross(p2 - p1, p3 - p1);  // constructor, importanceTri.h:19-25
            out[r] = itri_pdf(ld3(q), ld3(q + 3), p1, p2 - p1, p3 - p1, normalize(c), 0.5f * sqrtf(len2(c)));
            break;
        }
        case RT_OP_ITRI_SAMPLE: {
            const float* q = in + r * 14;
            const f3 d = itri_sample(ld3(q), q[3], q[4], ld3(q + 5), ld3(q + 8), ld3(q + 11));
            out[r * 3] = d.x; out[r * 3 + 1] = d.y; out[r * 3 + 2] = d.z;
            break;
        }
        case RT_OP_ATTRACTOR_PDF: {
            const float* q = in + r * 15;
            Attractor a;
            a.init(ld3(q + 6), ld3(q + 9), ld3(q + 12));
            out[r] = a.pdf(ld3(q), ld3(q + 3));
            break;
        }
        case RT_OP_ATTRACTOR_SAMPLE: {
            const float* q = in + r * 14;
            Attractor a;
            a.init(ld3(q + 5), ld3(q + 8), ld3(q + 11));
            const f3 d = a.sample(ld3(q), q[3], q[4]);
            out[r * 3] = d.x; out[r * 3 + 1] = d.y; out[r * 3 + 2] = d.z;
            break;
        }
        case RT_OP_TONEMAP: {
            const f3 c = tone_map(ld3(in + r * 3));
            out[r * 3] = c.x; out[r * 3 + 1] = c.y; out[r * 3 + 2] = c.z;
            break;
        }
        case RT_OP_RNG: {
            const float* q = in + r * 4;
            Rng rng;
            rng.init(__float_as_uint(q[0]), __float_as_uint(q[1]), __float_as_uint(q[2]));
            rng.seek(__float_as_uint(q[3]) & ~3u);
            for (int k = 0; k < 4; k++) out[r * 4 + k] = __uint_as_float(rng.raw());
            break;
        }
        case RT_OP_ENV_INDEX: {
            const float* q = in + r * 5;
            out[r] = __int_as_float((int)env_index(ld3(q), (int)q[3], (int)q[4]));
            break;
        }
        case RT_OP_CAMERA_RAY: {
            const float* q = in + r * 3;
            const int x = __float_as_int(q[0]), y = __float_as_int(q[1]);
            Rng rng;
            rng.init(P.seed, (uint32_t)(x + y * P.width), __float_as_uint(q[2]));
            f3 o, d;
            camera_ray(P, x, y, rng, o, d);
            float* w = out + r * 6;
            w[0] = o.x; w[1] = o.y; w[2] = o.z; w[3] = d.x; w[4] = d.y; w[5] = d.z;
            break;
        }
        case RT_OP_PMATH: {
            const float* q = in + r * 3;
            float a = 0.f, b = 0.f;
            switch (__float_as_int(q[0])) {
                case 0: pm::sincos(q[1], &a, &b); break;
                case 1: a = pm::asin(q[1]); break;
                case 2: a = pm::acos(q[1]); break;
                case 3: a = pm::atan2(q[1], q[2]); break;
                case 4: a = pm::log(q[1]); break;
                default: break;
            }
            out[r * 2] = a; out[r * 2 + 1] = b;
            break;
        }
        default:
            break;
    }
}

// =============================================================================== host side

struct rt_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cus = 256;
    std::string err;

    // scene (device)
    bool has_scene = false;
    Scene sc{};
    void* d_wide = nullptr; void* d_leaves = nullptr; void* d_hot = nullptr; void* d_cold = nullptr;
    void* d_mats = nullptr; void* d_lights = nullptr; void* d_spheres = nullptr; void* d_tex = nullptr;
    int bvh_depth = 0;

    // accumulators
    float* d_accum_own = nullptr;
    size_t accum_own_floats = 0;
    float* d_accum_ext = nullptr;
    size_t accum_ext_floats = 0;
    int acc_rows = 0, acc_width = 0;
    int64_t samples_done = 0;

    unsigned int* d_tile_counter = nullptr;
    unsigned long long* d_stats = nullptr;
};

namespace {

int set_err(rt_context* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                            \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return set_err(ctx, RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

void free_scene(rt_context* c) {
    void** ptrs[] = {&c->d_wide, &c->d_leaves, &c->d_hot, &c->d_cold, &c->d_mats, &c->d_lights, &c->d_spheres, &c->d_tex};
    for (void** p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    c->has_scene = false;
}

int upload_buf(rt_context* c, void** dst, const void* src, size_t bytes) {
    const size_t alloc = bytes ? bytes : 16;
    HIP_TRY(c, hipMalloc(dst, alloc));
    if (bytes) HIP_TRY(c, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return RT_OK;
}

inline float i2f(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

struct H3 { float x, y, z; };
inline H3 hsub(H3 a, H3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline H3 hcross(H3 a, H3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float hlen(H3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
inline H3 hnormalize(H3 a) {  // vec3.h:89-95
    float invlen = 1.0f / hlen(a);
    if (invlen > 0) return {a.x * invlen, a.y * invlen, a.z * invlen};
    return {0.f, 0.f, 0.f};
}
inline H3 hnormalize_keep(H3 a) {  // vec3.h:97-105
    float invlen = 1.0f / hlen(a);
    if (invlen > 0) return {a.x * invlen, a.y * invlen, a.z * invlen};
    return a;
}

float* accum_ptr(rt_context* c) { return c->d_accum_ext ? c->d_accum_ext : c->d_accum_own; }
size_t accum_cap(rt_context* c) { return c->d_accum_ext ? c->accum_ext_floats : c->accum_own_floats; }

int rows_of(const rt_render_params* p) {
    const int H = p->cam.height;
    if (p->band_stride <= 1 || p->band_rows <= 0) return H;
    int rows = 0;
    for (int y0 = 0, b = 0; y0 < H; y0 += p->band_rows, b++)
        if (b % p->band_stride == p->band_index) rows += std::min(p->band_rows, H - y0);
    return rows;
}

int make_params(rt_context* c, const rt_render_params* p, Params& P) {
    if (!p) return set_err(c, RT_ERR_INVALID, "null params");
    if (p->cam.width <= 0 || p->cam.height <= 0) return set_err(c, RT_ERR_INVALID, "image size must be positive");
    if ((long long)p->cam.width * p->cam.height > 0x7fffffffLL / 4)
        return set_err(c, RT_ERR_INVALID, "image too large");
    if (p->sample_count < 0 || p->sample_begin < 0 || p->max_bounces < 0)
        return set_err(c, RT_ERR_INVALID, "negative sample/bounce count");
    if (p->flags & (RT_F_VOLUME | RT_F_RAYMARCH))
        return set_err(c, RT_ERR_INVALID, "volume scattering / ray marching are not implemented (off in reference config.h:9,14)");
    if (p->band_stride > 1 && (p->band_rows <= 0 || p->band_index < 0 || p->band_index >= p->band_stride))
        return set_err(c, RT_ERR_INVALID, "bad band partition");
    memset(&P, 0, sizeof(P));
    P.width = p->cam.width;
    P.height = p->cam.height;
    P.rows = rows_of(p);
    P.band_rows = p->band_rows; P.band_stride = p->band_stride; P.band_index = p->band_index;
    P.sample_begin = p->sample_begin; P.sample_count = p->sample_count;
    P.max_bounces = p->max_bounces;
    P.flags = p->flags;
    P.seed = p->seed;
    P.focal = p->cam.focal_length; P.aperture = p->cam.aperture; P.focus_dist = p->cam.focus_distance;
    for (int i = 0; i < 3; i++) {
        P.cam_pos[i] = p->cam.pos[i]; P.cam_rot[i] = p->cam.rot[i];
        P.cam_vel[i] = p->cam.vel[i]; P.cam_angvel[i] = p->cam.ang_vel[i];
        P.bg[i] = p->background[i];
    }
    P.bg_brightness = p->background_brightness;
    P.env_w = p->cam.env_width; P.env_h = p->cam.env_height;
    if (P.env_w < 0 || P.env_h < 0 || (P.env_w > 0) != (P.env_h > 0))
        return set_err(c, RT_ERR_INVALID, "bad environment map size");
    if (P.env_w > 0 && c->has_scene && (long long)P.env_w * P.env_h * 3 > c->sc.n_tex)
        return set_err(c, RT_ERR_INVALID, "environment map is larger than the texture pool");
    // Zero angular velocity: lerp(rot, rot + 0, t) == rot exactly, so the basis of
    // main.cu:190-195 is the same for every sample; computed once here with the same
    // float operations (Matrix::rotationMatrix, matrix.h:14-37).
    P.static_basis = (p->cam.ang_vel[0] == 0.f && p->cam.ang_vel[1] == 0.f && p->cam.ang_vel[2] == 0.f) ? 1 : 0;
    if (P.static_basis) {
        // e = rot + t * ((rot + 0) - rot) evaluated like the device does for t in (0,1]
        const float ex = p->cam.rot[0], ey = p->cam.rot[1], ez = p->cam.rot[2];
        float cX, sX, cY, sY, cZ, sZ;
        pm::sincos(ex, &sX, &cX);
        pm::sincos(ey, &sY, &cY);
        pm::sincos(ez, &sZ, &cZ);
        const float m0 = cY * cZ, m1 = -cY * sZ, m2 = sY;
        const float m3 = cX * sZ + sX * sY * cZ, m4 = cX * cZ - sX * sY * sZ, m5 = -sX * cY;
        const float m6 = sX * sZ - cX * sY * cZ, m7 = sX * cZ + cX * sY * sZ, m8 = cX * cY;
        H3 up = hnormalize_keep({m0 * 0.0f + m1 * 1.0f + m2 * 0.0f, m3 * 0.0f + m4 * 1.0f + m5 * 0.0f, m6 * 0.0f + m7 * 1.0f + m8 * 0.0f});
        H3 right = hnormalize_keep({m0 * 1.0f + m1 * 0.0f + m2 * 0.0f, m3 * 1.0f + m4 * 0.0f + m5 * 0.0f, m6 * 1.0f + m7 * 0.0f + m8 * 0.0f});
        H3 fwd = hnormalize(hcross(up, right));
        P.up[0] = up.x; P.up[1] = up.y; P.up[2] = up.z;
        P.right[0] = right.x; P.right[1] = right.y; P.right[2] = right.z;
        P.forward[0] = fwd.x; P.forward[1] = fwd.y; P.forward[2] = fwd.z;
    }
    P.tiles_x = (P.width + 7) / 8;
    P.n_tiles = P.tiles_x * ((P.rows + 7) / 8);
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_create(rt_context** out, int device_id) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RT_ERR_DEVICE;
    if (device_id < 0) {
        if (hipGetDevice(&device_id) != hipSuccess) return RT_ERR_DEVICE;
    }
    if (device_id >= count) return RT_ERR_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return RT_ERR_DEVICE;
    rt_context* c = new rt_context();
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    bool ok = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess &&
              hipMalloc((void**)&c->d_tile_counter, 256) == hipSuccess &&
              hipMalloc((void**)&c->d_stats, 8 * sizeof(unsigned long long)) == hipSuccess;
    if (!ok) {
        rt_destroy(c);
        return RT_ERR_DEVICE;
    }
    c->stream = c->own_stream;
    *out = c;
    return RT_OK;
}

void rt_destroy(rt_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    free_scene(c);
    if (c->d_accum_own) (void)hipFree(c->d_accum_own);
    if (c->d_tile_counter) (void)hipFree(c->d_tile_counter);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* rt_last_error(const rt_context* c) { return c ? c->err.c_str() : "null context"; }

int rt_set_stream(rt_context* c, void* hip_stream) {
    if (!c) return RT_ERR_INVALID;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return RT_OK;
}

int rt_upload_scene(rt_context* c, const rt_scene_desc* s) {
    if (!c) return RT_ERR_INVALID;
    if (!s) return set_err(c, RT_ERR_INVALID, "null scene");
    if (s->n_tri_floats % RT_TRI_FLOATS) return set_err(c, RT_ERR_INVALID, "n_tri_floats is not a multiple of 25");
    if ((s->n_tri_floats && !s->tris) || (s->n_nodes && !s->nodes) || (s->n_mats && !s->mats) ||
        (s->n_tex_floats && !s->textures) || (s->n_light_tris && !s->light_tris) || s->n_nodes < 0 ||
        s->n_spheres < 0 || s->n_light_tris < 0)
        return set_err(c, RT_ERR_INVALID, "inconsistent scene arrays");
    if (s->n_tri_floats / RT_TRI_FLOATS > 0x7fffffffu / RT_TRI_FLOATS)
        return set_err(c, RT_ERR_INVALID, "too many triangles for int32 float offsets");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t prefix = s->n_spheres > 0 ? 1 + 5 * (size_t)s->n_spheres : 0;
    const float* tris = s->tris + prefix;
    const int ntri = (int)(s->n_tri_floats / RT_TRI_FLOATS);

    // ---- BVH: reference AoS nodes -> wide inner records + leaf table (same topology, numbering, leaf order)
    const int nn = s->n_nodes;
    std::vector<int> entry(nn);
    int n_inner = 0, n_leaf = 0;
    for (int i = 0; i < nn; i++) {
        const rt_bvh_node& n = s->nodes[i];
        if (n.child_a == -1) {
            if (n.tri_offs < 0 || n.tri_len < 0 || n.tri_offs % RT_TRI_FLOATS || n.tri_len % RT_TRI_FLOATS ||
                (size_t)n.tri_offs + (size_t)n.tri_len > s->n_tri_floats)
                return set_err(c, RT_ERR_INVALID, "BVH leaf " + std::to_string(i) + " has a bad triangle range");
            entry[i] = ~(n_leaf++);
        } else {
            // children must come after the parent (BFS numbering): guarantees a finite, acyclic walk
            if (n.child_a <= i || n.child_a >= nn || n.child_b <= i || n.child_b >= nn)
                return set_err(c, RT_ERR_INVALID, "BVH node " + std::to_string(i) + " has bad child indices");
            entry[i] = n_inner++;
        }
    }
    std::vector<int> depth(nn, 0);
    int max_depth = 0;
    for (int i = 0; i < nn; i++) {
        const rt_bvh_node& n = s->nodes[i];
        if (n.child_a != -1) {
            depth[n.child_a] = std::max(depth[n.child_a], depth[i] + 1);
            depth[n.child_b] = std::max(depth[n.child_b], depth[i] + 1);
        }
        max_depth = std::max(max_depth, depth[i]);
    }
    if (max_depth > RT_STACK_ENTRIES)
        return set_err(c, RT_ERR_LIMIT, "BVH depth " + std::to_string(max_depth) + " exceeds the traversal stack (" +
                                           std::to_string(RT_STACK_ENTRIES) + ")");
    std::vector<float> wide((size_t)std::max(n_inner, 1) * 16, 0.f);
    std::vector<int32_t> leaves((size_t)std::max(n_leaf, 1) * 2, 0);
    for (int i = 0; i < nn; i++) {
        const rt_bvh_node& n = s->nodes[i];
        if (n.child_a == -1) {
            const int li = ~entry[i];
            leaves[2 * (size_t)li] = n.tri_offs / RT_TRI_FLOATS;
            leaves[2 * (size_t)li + 1] = n.tri_len / RT_TRI_FLOATS;
        } else {
            float* w = &wide[(size_t)entry[i] * 16];
            const rt_bvh_node& A = s->nodes[n.child_a];
            const rt_bvh_node& B = s->nodes[n.child_b];
            w[0] = A.bbox_min[0]; w[1] = A.bbox_min[1]; w[2] = A.bbox_min[2];
            w[3] = A.bbox_max[0]; w[4] = A.bbox_max[1]; w[5] = A.bbox_max[2];
            w[6] = B.bbox_min[0]; w[7] = B.bbox_min[1]; w[8] = B.bbox_min[2];
            w[9] = B.bbox_max[0]; w[10] = B.bbox_max[1]; w[11] = B.bbox_max[2];
            w[12] = i2f(entry[n.child_a]);
            w[13] = i2f(entry[n.child_b]);
            w[14] = i2f(i);
            w[15] = 0.f;
        }
    }

    // ---- triangles: hot (v0, e1, e2) / cold (normals, uvs, material id)
    std::vector<float> hot((size_t)std::max(ntri, 1) * 12, 0.f), cold((size_t)std::max(ntri, 1) * 16, 0.f);
    for (int i = 0; i < ntri; i++) {
        const float* t = tris + (size_t)i * RT_TRI_FLOATS;
        float* h = &hot[(size_t)i * 12];
        float* cd = &cold[(size_t)i * 16];
        const int mat = static_cast<int>(t[24]);  // main.cu:353
        h[0] = t[0]; h[1] = t[1]; h[2] = t[2];
        h[3] = t[3] - t[0]; h[4] = t[4] - t[1]; h[5] = t[5] - t[2];   // edge1 = v1 - v0, main.cu:43
        h[6] = t[6] - t[0]; h[7] = t[7] - t[1]; h[8] = t[8] - t[2];   // edge2 = v2 - v0, main.cu:44
        h[9] = 0.f; h[10] = 0.f; h[11] = i2f(mat);
        for (int k = 0; k < 15; k++) cd[k] = t[9 + k];
        cd[15] = i2f(mat);
    }

    // ---- materials: decoded colour (Vec3::fromColorInt, vec3.h:118-123)
    std::vector<float> mats(std::max<size_t>(s->n_mats, 1) * 12, 0.f);
    for (size_t i = 0; i < s->n_mats; i++) {
        const rt_material& m = s->mats[i];
        float* d = &mats[i * 12];
        d[0] = i2f((int)m.type);
        d[1] = static_cast<float>((m.color >> 16) & 0xFF) / 255.0f;
        d[2] = static_cast<float>((m.color >> 8) & 0xFF) / 255.0f;
        d[3] = static_cast<float>(m.color & 0xFF) / 255.0f;
        d[4] = m.p1; d[5] = m.p2; d[6] = m.p3;
        d[7] = i2f(m.image_offset[0]);
        d[8] = i2f(m.image_offset[1]);
        d[9] = i2f(m.image_offset[2]);
        d[10] = i2f((int)m.image_width);
        d[11] = i2f((int)m.image_height);
    }

    // ---- lights: ImportanceTri constructor (importanceTri.h:19-25) evaluated once
    std::vector<float> lights((size_t)std::max(s->n_light_tris, 1) * 20, 0.f);
    for (int i = 0; i < s->n_light_tris; i++) {
        const int off = s->light_tris[i];
        if (off < 0 || off % RT_TRI_FLOATS || (size_t)off + RT_TRI_FLOATS > s->n_tri_floats)
            return set_err(c, RT_ERR_INVALID, "light triangle offset out of range");
        const float* t = tris + off;
        const H3 p1{t[0], t[1], t[2]}, p2{t[3], t[4], t[5]}, p3{t[6], t[7], t[8]};
        const H3 e1 = hsub(p2, p1), e2 = hsub(p3, p1);
        const H3 cr = hcross(e1, e2);
        const H3 nrm = hnormalize(cr);
        const float area = 0.5f * hlen(cr);
        float* L = &lights[(size_t)i * 20];
        L[0] = p1.x; L[1] = p1.y; L[2] = p1.z; L[3] = e1.x; L[4] = e1.y; L[5] = e1.z;
        L[6] = e2.x; L[7] = e2.y; L[8] = e2.z; L[9] = nrm.x; L[10] = nrm.y; L[11] = nrm.z;
        L[12] = area; L[13] = p2.x; L[14] = p2.y; L[15] = p2.z; L[16] = p3.x; L[17] = p3.y; L[18] = p3.z; L[19] = 0.f;
    }

    // ---- spheres
    std::vector<float> spheres((size_t)std::max(s->n_spheres, 1) * 8, 0.f);
    for (int i = 0; i < s->n_spheres; i++) {
        const float* q = s->tris + 1 + 5 * (size_t)i;
        float* d = &spheres[(size_t)i * 8];
        d[0] = q[0]; d[1] = q[1]; d[2] = q[2]; d[3] = q[3];
        d[4] = i2f(static_cast<int>(q[4]));
    }

    free_scene(c);
    int rc;
    if ((rc = upload_buf(c, &c->d_wide, wide.data(), wide.size() * 4)) != RT_OK) return rc;
    if ((rc = upload_buf(c, &c->d_leaves, leaves.data(), leaves.size() * 4)) != RT_OK) return rc;
    if ((rc = upload_buf(c, &c->d_hot, hot.data(), hot.size() * 4)) != RT_OK) return rc;
    if ((rc = upload_buf(c, &c->d_cold, cold.data(), cold.size() * 4)) != RT_OK) return rc;
    if ((rc = upload_buf(c, &c->d_mats, mats.data(), mats.size() * 4)) != RT_OK) return rc;
    if ((rc = upload_buf(c, &c->d_lights, lights.data(), lights.size() * 4)) != RT_OK) return rc;
    if ((rc = upload_buf(c, &c->d_spheres, spheres.data(), spheres.size() * 4)) != RT_OK) return rc;
    if ((rc = upload_buf(c, &c->d_tex, s->textures, s->n_tex_floats * 4)) != RT_OK) return rc;

    Scene& sc = c->sc;
    sc.wide = (const float4*)c->d_wide;
    sc.leaves = (const int2*)c->d_leaves;
    sc.tri_hot = (const float4*)c->d_hot;
    sc.tri_cold = (const float4*)c->d_cold;
    sc.mats = (const float4*)c->d_mats;
    sc.lights = (const float4*)c->d_lights;
    sc.spheres = (const float4*)c->d_spheres;
    sc.tex = (const float*)c->d_tex;
    sc.n_tex = (long long)s->n_tex_floats;
    sc.n_tris = ntri;
    sc.n_mats = (int)s->n_mats;
    sc.n_lights = s->n_light_tris;
    sc.n_spheres = s->n_spheres;
    sc.has_root = nn > 0;
    sc.root_entry = nn > 0 ? entry[0] : 0;
    for (int k = 0; k < 3; k++) {
        sc.root_min[k] = nn > 0 ? s->nodes[0].bbox_min[k] : 0.f;
        sc.root_max[k] = nn > 0 ? s->nodes[0].bbox_max[k] : 0.f;
    }
    c->bvh_depth = max_depth;
    c->has_scene = true;
    return RT_OK;
}

int rt_reset_accum(rt_context* c) {
    if (!c) return RT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    float* a = accum_ptr(c);
    if (a && c->acc_rows > 0)
        HIP_TRY(c, hipMemsetAsync(a, 0, (size_t)c->acc_rows * c->acc_width * 3 * sizeof(float), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->samples_done = 0;
    return RT_OK;
}

int rt_bind_accum(rt_context* c, float* dev_ptr, size_t n_floats) {
    if (!c) return RT_ERR_INVALID;
    c->d_accum_ext = dev_ptr;
    c->accum_ext_floats = dev_ptr ? n_floats : 0;
    c->acc_rows = 0;
    c->acc_width = 0;
    c->samples_done = 0;
    return RT_OK;
}

int rt_set_sample_count(rt_context* c, int64_t samples) {
    if (!c || samples < 0) return RT_ERR_INVALID;
    c->samples_done = samples;
    return RT_OK;
}

int rt_render(rt_context* c, const rt_render_params* p, rt_render_stats* stats) {
    if (!c) return RT_ERR_INVALID;
    if (!c->has_scene) return set_err(c, RT_ERR_NO_SCENE, "rt_render before rt_upload_scene");
    Params P;
    int rc = make_params(c, p, P);
    if (rc != RT_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t need = (size_t)P.rows * P.width * 3;
    const bool shape_changed = (P.rows != c->acc_rows || P.width != c->acc_width);
    if (c->d_accum_ext) {
        if (need > c->accum_ext_floats) return set_err(c, RT_ERR_INVALID, "bound accumulator is too small");
    } else if (need > c->accum_own_floats) {
        if (c->d_accum_own) (void)hipFree(c->d_accum_own);
        c->d_accum_own = nullptr;
        c->accum_own_floats = 0;
        HIP_TRY(c, hipMalloc((void**)&c->d_accum_own, std::max<size_t>(need, 4) * sizeof(float)));
        c->accum_own_floats = need;
    }
    float* accum = accum_ptr(c);
    if (shape_changed) {  // a new image shape starts a new accumulation
        if (need) HIP_TRY(c, hipMemsetAsync(accum, 0, need * sizeof(float), c->stream));
        c->acc_rows = P.rows;
        c->acc_width = P.width;
        c->samples_done = 0;
    }
    const bool want_stats = (P.flags & RT_F_STATS) != 0;
    HIP_TRY(c, hipMemsetAsync(c->d_tile_counter, 0, sizeof(unsigned int), c->stream));
    if (want_stats) HIP_TRY(c, hipMemsetAsync(c->d_stats, 0, 8 * sizeof(unsigned long long), c->stream));
    if (P.n_tiles > 0 && P.sample_count > 0) {
        // persistent grid: enough blocks to fill every CU at the kernel's occupancy, never more than the work
        int occ = 0;
        const void* kfn = want_stats ? (const void*)rt_render_kernel<true> : (const void*)rt_render_kernel<false>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, 64 * RT_WAVES_PER_BLOCK, 0) != hipSuccess || occ < 1) occ = 2;
        const int max_blocks = c->num_cus * occ;
        const int want_blocks = (P.n_tiles + RT_WAVES_PER_BLOCK - 1) / RT_WAVES_PER_BLOCK;
        const int blocks = std::max(1, std::min(max_blocks, want_blocks));
        HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
        if (want_stats)
            hipLaunchKernelGGL(rt_render_kernel<true>, dim3(blocks), dim3(64 * RT_WAVES_PER_BLOCK), 0, c->stream, c->sc, P, accum,
                               c->d_tile_counter, c->d_stats);
        else
            hipLaunchKernelGGL(rt_render_kernel<false>, dim3(blocks), dim3(64 * RT_WAVES_PER_BLOCK), 0, c->stream, c->sc, P, accum,
                               c->d_tile_counter, c->d_stats);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->samples_done += P.sample_count;
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        if (P.n_tiles > 0 && P.sample_count > 0) {
            float ms = 0.f;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
            stats->kernel_ms = ms;
        }
        if (want_stats) {
            unsigned long long h[8];
            HIP_TRY(c, hipMemcpy(h, c->d_stats, sizeof(h), hipMemcpyDeviceToHost));
            stats->paths = h[0]; stats->rays = h[1]; stats->inner_visits = h[2]; stats->leaf_visits = h[3];
            stats->tri_tests = h[4]; stats->hits = h[5]; stats->texel_fetches = h[6]; stats->light_tests = h[7];
        }
    }
    return RT_OK;
}

int rt_resolve_device(rt_context* c, float* dev_out, size_t n_floats, int mode) {
    if (!c || !dev_out) return RT_ERR_INVALID;
    if (mode < RT_READ_RAW_SUM || mode > RT_READ_TONEMAPPED) return set_err(c, RT_ERR_INVALID, "bad readback mode");
    const size_t n = (size_t)c->acc_rows * c->acc_width * 3;
    if (n == 0 || !accum_ptr(c)) return set_err(c, RT_ERR_INVALID, "nothing rendered yet");
    if (n_floats != n) return set_err(c, RT_ERR_INVALID, "n_floats does not match rows*width*3 of the last render");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t npix = n / 3;
    const float samples = static_cast<float>(c->samples_done > 0 ? c->samples_done : 1);
    hipLaunchKernelGGL(rt_resolve_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, c->stream, accum_ptr(c), dev_out,
                       npix, samples, mode);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_readback(rt_context* c, float* rgb, size_t n_floats, int mode) {
    if (!c || !rgb) return RT_ERR_INVALID;
    const size_t n = (size_t)c->acc_rows * c->acc_width * 3;
    if (n == 0 || !accum_ptr(c)) return set_err(c, RT_ERR_INVALID, "nothing rendered yet");
    if (n_floats != n) return set_err(c, RT_ERR_INVALID, "n_floats does not match rows*width*3 of the last render");
    HIP_TRY(c, hipSetDevice(c->device));
    if (mode == RT_READ_RAW_SUM) {
        HIP_TRY(c, hipMemcpyAsync(rgb, accum_ptr(c), n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return RT_OK;
    }
    float* tmp = nullptr;
    HIP_TRY(c, hipMalloc((void**)&tmp, n * sizeof(float)));
    int rc = rt_resolve_device(c, tmp, n, mode);
    if (rc == RT_OK) {
        hipError_t e = hipMemcpy(rgb, tmp, n * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = set_err(c, RT_ERR_DEVICE, std::string("readback copy: ") + hipGetErrorString(e));
    }
    (void)hipFree(tmp);
    return rc;
}

namespace {
struct DevTmp {  // scoped device scratch for the test hooks
    std::vector<void*> ptrs;
    ~DevTmp() { for (void* p : ptrs) (void)hipFree(p); }
    void* get(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return p;
    }
};
}  // namespace

int rt_trace_rays(rt_context* c, const float* org_dir, size_t n, uint32_t flags, int32_t* prim_id, float* t,
                  float* hit_pos, float* normal3, int32_t* counts) {
    if (!c) return RT_ERR_INVALID;
    if (!c->has_scene) return set_err(c, RT_ERR_NO_SCENE, "rt_trace_rays before rt_upload_scene");
    if (n == 0) return RT_OK;
    if (!org_dir) return set_err(c, RT_ERR_INVALID, "null rays");
    HIP_TRY(c, hipSetDevice(c->device));
    DevTmp tmp;
    float* d_rays = (float*)tmp.get(n * 6 * 4);
    int* d_prim = (int*)tmp.get(n * 4);
    float* d_t = (float*)tmp.get(n * 4);
    float* d_pos = (float*)tmp.get(n * 12);
    float* d_nrm = (float*)tmp.get(n * 12);
    int* d_cnt = (int*)tmp.get(n * 12);
    if (!d_rays || !d_prim || !d_t || !d_pos || !d_nrm || !d_cnt) return set_err(c, RT_ERR_NOMEM, "device scratch allocation failed");
    HIP_TRY(c, hipMemcpyAsync(d_rays, org_dir, n * 24, hipMemcpyHostToDevice, c->stream));
    const int bs = 64 * RT_WAVES_PER_BLOCK;
    hipLaunchKernelGGL(rt_trace_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, c->stream, c->sc, d_rays, n, flags, d_prim,
                       d_t, d_pos, d_nrm, d_cnt);
    HIP_TRY(c, hipGetLastError());
    if (prim_id) HIP_TRY(c, hipMemcpyAsync(prim_id, d_prim, n * 4, hipMemcpyDeviceToHost, c->stream));
    if (t) HIP_TRY(c, hipMemcpyAsync(t, d_t, n * 4, hipMemcpyDeviceToHost, c->stream));
    if (hit_pos) HIP_TRY(c, hipMemcpyAsync(hit_pos, d_pos, n * 12, hipMemcpyDeviceToHost, c->stream));
    if (normal3) HIP_TRY(c, hipMemcpyAsync(normal3, d_nrm, n * 12, hipMemcpyDeviceToHost, c->stream));
    if (counts) HIP_TRY(c, hipMemcpyAsync(counts, d_cnt, n * 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_debug_op(rt_context* c, int op, const float* in, size_t n, float* out, const rt_render_params* params) {
    if (!c) return RT_ERR_INVALID;
    static const int strides[14][2] = {{0, 0}, {15, 4}, {12, 1}, {10, 7}, {7, 7}, {15, 1}, {14, 3}, {15, 1}, {14, 3},
                                       {3, 3}, {4, 4}, {5, 1}, {3, 6}, {3, 2}};
    if (op < 1 || op > 13) return set_err(c, RT_ERR_INVALID, "unknown debug op");
    if (n == 0) return RT_OK;
    if (!in || !out) return set_err(c, RT_ERR_INVALID, "null buffers");
    Params P;
    memset(&P, 0, sizeof(P));
    if (op == RT_OP_CAMERA_RAY) {
        int rc = make_params(c, params, P);
        if (rc != RT_OK) return rc;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    DevTmp tmp;
    const size_t ib = n * strides[op][0] * 4, ob = n * strides[op][1] * 4;
    float* d_in = (float*)tmp.get(ib);
    float* d_out = (float*)tmp.get(ob);
    if (!d_in || !d_out) return set_err(c, RT_ERR_NOMEM, "device scratch allocation failed");
    HIP_TRY(c, hipMemcpyAsync(d_in, in, ib, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(d_out, 0, ob, c->stream));
    hipLaunchKernelGGL(rt_debug_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, op, d_in, n, d_out, P);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, d_out, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_shade_hits(rt_context* c, const float* in, size_t n, uint32_t seed, float* out) {
    if (!c) return RT_ERR_INVALID;
    if (!c->has_scene) return set_err(c, RT_ERR_NO_SCENE, "rt_shade_hits before rt_upload_scene");
    if (n == 0) return RT_OK;
    if (!in || !out) return set_err(c, RT_ERR_INVALID, "null buffers");
    HIP_TRY(c, hipSetDevice(c->device));
    DevTmp tmp;
    float* d_in = (float*)tmp.get(n * 64);
    float* d_out = (float*)tmp.get(n * 64);
    if (!d_in || !d_out) return set_err(c, RT_ERR_NOMEM, "device scratch allocation failed");
    HIP_TRY(c, hipMemcpyAsync(d_in, in, n * 64, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(rt_shade_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->sc, d_in, n, seed, d_out);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, d_out, n * 64, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

}  // extern "C"
