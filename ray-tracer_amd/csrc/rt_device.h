// rt_device.h -- gfx950 device code of the path-tracing core: layouts, math,
// counter RNG, intersection, BVH traversal and shading.
//
// Written for CDNA4 (wave64, LDS short stack, 16-byte vector loads); not a
// translation of the reference's per-thread CUDA loop.  What each function
// computes follows the reference (file:line cited), evaluated in the same
// float operation order with FMA contraction OFF (-ffp-contract=off) so that
// closest-hit ids agree bit for bit with the oracle.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt.h"
#include "rt_pmath.h"

namespace rtd {

// ------------------------------------------------------------------ layouts in HBM
//
// ITEM STREAM (64 B = 4 x float4 per item): inner nodes and triangles live in ONE array with ONE index
// space, so a traversal cursor is a single int and a step -- an inner-node visit or one triangle test
// -- is a few 16-byte loads at 64 * cursor.  Items [0, n_top) are the INNER nodes of the reference BVH in its BFS order,
// items [n_top, n_inner) the inner nodes of LEAF SUB-TREES (below), items [n_inner, n_inner + n_tris) the triangles
// in the reference's (leaf-contiguous) order, except inside a leaf with a sub-tree.
//   inner item: both child boxes + both child entries (one 64-B fetch per inner visit instead
//               of the reference's two 40-B child fetches, main.cu:268-269)
//     q0 = (A.min.x, A.min.y, A.max.x, A.max.y)      x/y pairs and z pairs: the slab differences and
//     q1 = (B.min.x, B.min.y, B.max.x, B.max.y)      products run as packed two-float operations
//     q2 = (A.min.z, A.max.z, B.min.z, B.max.z)
//     q3 = (entryA, entryB, refNodeId, 0)            as int bits
//          sub-tree node: (entryA, entryB, triangles of the leaf [root of the sub-tree] or 0, 1)
//     entry = item index of the child (inner node, or FIRST triangle of a leaf child),
//             RT_TRAV_EMPTY for a leaf without triangles
//   triangle item: v0, e1 = v1 - v0, e2 = v2 - v0 (the single subtraction of main.cu:43-44 done
//               at upload, bit-identical)
//     q0 = (v0.x, v0.y, v0.z, e1.x)  q1 = (e1.y, e1.z, e2.x, e2.y)
//     q2 = (e2.z, flags, leafEnd, id)  flags bit0 = last triangle of its (sub-)leaf, bit1 = first of a reference
//                                      leaf, bit2 = inside a sub-tree (its test is counted at the sub-tree's root);
//                                      leafEnd = one past the last triangle of its reference leaf; id = the
//                                      triangle's place in the reference's array (= its place here without sub-trees)
//     q3 = (refLeafNode, 0, 0, 0)      the reference's leaf (node index) that holds the triangle
// SHORTCUT TREE.  The reference's tree is what it is (mean splits, depth cap 16); its RESULT per ray is the triangle with
// the smallest |fl(o + d t) - o|^2 among those it tests.  The fused path-parallel kernel first searches a surface-area tree
// of this library's own, built at upload over the reference's non-empty leaves (each with its triangles or its sub-tree
// below it; item q3.w = 2, padded boxes, the sub-trees' slack rule), remembering besides the best hit m (in leaf L) the
// RIVAL: the smallest squared distance of a hit in another reference leaf (exact ties included).  When the search is over:
//     bd = AABBIntersectDistance(ray, the reference's box of L), exactly as the reference evaluates it;
//     certified  <=>  bd^2 < rival  and  m < rival  and  bd^2 <= m (1 + 1e-5) + slack / 2.
// Reference boxes are nested (checked at upload) and the slab arithmetic is monotonic in the box, so every ancestor A of
// L has bd(A)^2 <= bd^2; the reference prunes A only if bd(A)^2 >= its minDistSqr at that moment, which is 1e12 or the
// squared distance of a hit from another leaf, i.e. >= rival > bd^2: the reference reaches L, tests all its triangles,
// and nothing it finds elsewhere is as close as m.  Same triangle, same t.  The third condition keeps bd^2 inside what
// the slack rule guarantees the shortcut walk has tested.  A ray that is not certified (a hit on an edge shared by two
// leaves, coincident geometry) walks the reference's tree from the root; rays with a non-finite 1/d or from beyond
// cull_far never try the shortcut.  RT_F_STATS counts the reference's walk and therefore walks it (RT_F_REFWALK).
// What hangs below the shortcut tree (rt_core.hip, build_layout): the RUNS of the item stream (triangles tested one after
// the other up to a 'last' flag); in a scene with more than 2^20 runs what stands a few levels below each reference leaf
// in its sub-tree (the walk then continues into the sub-tree); in a scene too small for sub-trees its SINGLE triangles
// (short_single: a shortcut search leaves a leaf after one test, whatever the 'last' flags say).
// WIDE SHORTCUT NODES (round 3).  The shortcut tree is this library's own, so its arity is free too: the binary tree the
// surface-area builder makes is collapsed into nodes of up to FOUR children (the child with the biggest box is replaced by
// its own two children until there are four), stored as TWO consecutive items = 128 bytes:
//     item 0: q0 = min.x of children 0..3, q1 = min.y, q2 = min.z, q3 = entries 0..3 (int bits)
//     item 1: q0 = max.x,                  q1 = max.y, q2 = max.z, q3 = (children, 0, 0, 3) (int bits; not read by the kernels)
// A missing child has the box min = max = 3e38 (no ray reaches it) and the entry RT_TRAV_EMPTY.  A visit tests the four
// boxes in one go -- half as many dependent node fetches per ray, four independent slab tests per lane to overlap -- with
// t = fma(bound, 1/d, -o/d) (one instruction per plane; off by 6e-8 (|o| + |bound|) in space, well inside the boxes'
// padding, and a shortcut search only has to be CONSERVATIVE: the certificate restores exactness), sorts the children
// that are hit by entry distance, pushes up to three and goes on with the nearest.  Wide nodes stand in ONE index range,
// [wide_lo, wide_lo + wide_n), biggest boxes first: that range's head is the LDS window of the shortcut walk.
// LEAF SUB-TREES.  The reference tests EVERY triangle of a leaf, in array order, and a later triangle replaces the
// hit only if strictly closer (main.cu:249-264): the leaf's result is the closest hit below the distance the ray
// arrived with and, among equals, the one first in the array -- which does not depend on the order of the tests, and
// a triangle the ray misses contributes nothing.  A leaf with many triangles (the reference caps its tree at
// depth 16: ~150 per leaf in a 10M-triangle scene) therefore gets a small tree of its own here: triangles
// reordered inside the leaf, median splits, PADDED boxes and a pruning test with slack, so that a sub-tree node is
// only skipped when no triangle in it can be a hit that counts.  Counters: the root of a sub-tree counts one
// leaf visit and all the leaf's triangle tests, like the walk it replaces.
// tri cold   (64 B = 4 x float4): n0 n1 n2 (9), uv0 uv1 uv2 (6), matId bits (1)
// material   (48 B = 3 x float4): (type bits, r, g, b) (p1, p2, p3, off0 bits) (off1, off2, w, h bits)
// light      (96 B = 6 x float4): p1(3) e1(3) e2(3) normal(3) area(1) p2(3) p3(3) reference index (int bits), then (centre, padded radius) of
//            the triangle's bounding sphere; stored plane by plane per group of RT_LIGHT_GROUP lights (RT_LIGHT_F4 below)
// light box  (32 B = 2 x float4): padded bounds (min, 0) (max, 0) of lights [16 g, 16 g + 16), behind the light records
// sphere     (32 B = 2 x float4): (cx, cy, cz, r) (matId bits, 0, 0, 0)
struct Scene {
    const float4* items;
    const int* tri_place;    // reference triangle id -> place in the item stream (callers' ids, test hooks)
    const float4* ref_boxes; // the reference's node boxes as uploaded, (min, 0) (max, 0) per node: the shortcut certificate tests the hit's leaf
    int short_root;          // item of the shortcut tree's root, RT_TRAV_DONE without one
    int short_base;          // first item of the window [short_base, short_base + n_cached) the shortcut walk caches in LDS
    int short_single;        // ... over single triangles (tiny scenes): a shortcut search leaves a leaf after ONE test, whatever its 'last' flag
    int short_runs;          // the shortcut tree stands over the runs of the item stream (short ones: two triangle tests per leaf iteration pay)
    int light_root;          // root (wide node) of the LIGHT TREE over the light items, RT_TRAV_DONE without one
    int n_light_items;       // light items behind the triangles: items [n_inner + n_tris + 1, + n_light_items)
    int wide_lo, wide_n;     // WIDE shortcut nodes (four children, two items each): items [wide_lo, wide_lo + wide_n); wide_n = 0: binary shortcut nodes
    int wide_sp_max;         // a wide visit needs room for three pushes (and the sub-trees below the shortcut tree for theirs): with more than this
                             // many entries on its stack the search gives up and the ray walks the reference's tree
    const float4* tri_cold;
    const float4* mats;
    const float4* lights;
    const float4* light_boxes;
    const int* light_place;  // reference light index -> record (records are in spatial order when the wave-cooperative loop is used)
    const float4* spheres;
    const float* tex;
    long long n_tex;
    int n_inner;      // inner items: the reference's nodes, then the sub-tree nodes
    int n_sub;        // sub-tree nodes among them (0: every triangle is at its reference place)
    float cull_far;   // rays that start farther than this from the origin (any axis) are not culled by padded boxes
    int n_tris;
    int n_mats;
    int n_lights;
    int n_spheres;
    int root_entry;  // entry of node 0
    int has_root;    // n_nodes > 0
    float root_min[3];
    float root_max[3];
};

// Per-call parameters (kernel argument, lives in SGPRs / kernarg segment).
struct Params {
    int width, height;       // full image
    int rows;                // accumulator rows of this call (band-packed)
    int band_rows, band_stride, band_index;
    int sample_begin, sample_count;
    int max_bounces;
    unsigned flags;
    unsigned seed;
    float focal, aperture, focus_dist;
    float cam_pos[3], cam_rot[3], cam_vel[3], cam_angvel[3];
    int static_basis;        // angular velocity == 0: basis below is valid for every sample
    float up[3], right[3], forward[3];
    int env_w, env_h;
    float bg[3];
    float bg_brightness;
    int tiles_x, n_tiles;
};

// Light records are stored plane by plane per group of RT_LIGHT_GROUP lights: float4 p of light j stands at
// ((j / G) * RT_LIGHT_PLANES + p) * G + j % G, so that the G lanes of a slot which test the G lights of a group (wave_light_pdf) read
// 16 G contiguous bytes per request instead of 16 bytes every 80 (the texture path serves a request in proportion to
// the lines it touches: it was the limit of the light loop).
#ifndef RT_LIGHT_GROUP
#define RT_LIGHT_GROUP 16  // (8 / 16 / 32 measured: profiles/r02_ab.txt)
#endif
#define RT_LIGHT_PLANES 6  // five float4 of the record + (centre, padded radius) of the triangle's bounding sphere
#define RT_LIGHT_F4(j, p) ((((size_t)(j) / RT_LIGHT_GROUP) * RT_LIGHT_PLANES + (p)) * RT_LIGHT_GROUP + (size_t)(j) % RT_LIGHT_GROUP)
#define RT_STACK_ENTRIES 24   // per-lane LDS stack depth (a depth-D tree needs D entries here)
#define RT_WAVES_PER_BLOCK 4

// ------------------------------------------------------------------ float3 helpers (vec3.h:14-197)
struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float len2(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// Vec3::normalize, vec3.h:89-105: multiply by 1/length, zero vector if that is not > 0.
__device__ __forceinline__ f3 normalize(f3 a) {
    float invlen = 1.0f / sqrtf(len2(a));
    if (invlen > 0) return mk(a.x * invlen, a.y * invlen, a.z * invlen);
    return mk(0.f, 0.f, 0.f);
}
// static Vec3::normalize(Vec3&), vec3.h:97-105: leaves the vector untouched if 1/len is not > 0.
__device__ __forceinline__ f3 normalize_keep(f3 a) {
    float invlen = 1.0f / sqrtf(len2(a));
    if (invlen > 0) return mk(a.x * invlen, a.y * invlen, a.z * invlen);
    return a;
}
// Vec3::reflect, vec3.h:42-47
__device__ __forceinline__ f3 reflect(f3 v, f3 n) {
    float d = 2.0f * dot(v, n);
    return mk(v.x - d * n.x, v.y - d * n.y, v.z - d * n.z);
}
// Vec3::refract, vec3.h:49-62
__device__ __forceinline__ f3 refract(f3 v, f3 n, float eta) {
    float cos_theta = fminf(-dot(v, n), 1.0f);
    f3 perp = mk((v.x + n.x * cos_theta) * eta, (v.y + n.y * cos_theta) * eta, (v.z + n.z * cos_theta) * eta);
    float k = -sqrtf(fabsf(1.0f - len2(perp)));
    return mk(perp.x + n.x * k, perp.y + n.y * k, perp.z + n.z * k);
}
__device__ __forceinline__ f3 lerp3(f3 a, f3 b, float t) {  // vec3.h:125-131
    return mk(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z));
}
__device__ __forceinline__ f3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

// ------------------------------------------------------------------ counter RNG (rt.h)
#ifndef RT_PHILOX_MAD64
#define RT_PHILOX_MAD64 1
#endif
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
#if RT_PHILOX_MAD64
        // both halves of a product from ONE v_mad_u64_u32 (32-bit integer multiplies issue at a quarter of the rate: 40 of them per block otherwise)
        const unsigned long long p0 = (unsigned long long)c0 * 0xD2511F53u, p1 = (unsigned long long)c2 * 0xCD9E8D57u;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#else
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
#endif
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One (pixel, sample) stream.  Draws are addressed by DIMENSION (rt.h); a Philox block yields the
// four dimensions 4b .. 4b+3.  Every call site below names its dimensions statically, so Philox is
// generated at a handful of places (camera, lens, one or two blocks per bounce) and nothing but the
// three key words stays live across the traversal.
struct Rng {
    uint32_t seed, pixel, sample;
    __device__ __forceinline__ void init(uint32_t s, uint32_t p, uint32_t k) { seed = s; pixel = p; sample = k; }
    __device__ __forceinline__ uint4 block(uint32_t b) const {
        uint32_t o[4];
        philox4x32_10(pixel, sample, b, RT_RNG_TAG, seed, RT_RNG_KEY1, o);
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
};
__device__ __forceinline__ float u01(uint32_t x) { return (float)((x >> 8) + 1u) * (1.0f / 16777216.0f); }  // (0,1]
__device__ __forceinline__ float gauss(uint32_t a, uint32_t b) {  // normal from two consecutive dimensions
    return sqrtf(-2.0f * pm::log(u01(a))) * pm::cos(6.2831855f * u01(b));
}

// ------------------------------------------------------------------ counters
struct Counters {
    uint32_t paths, rays, inner, leaf, tri, hits, texels, lights;
    // what the DEVICE really does (the eight above count the reference's walk, which a sub-tree root stands in for):
    // inner items visited (reference nodes and sub-tree nodes), of those served by the LDS node cache, triangle tests run
    uint32_t rewalks;  // shortcut searches that were not certified and walked the reference's tree again
    uint32_t dev_inner, dev_cached, dev_tri, dev_lights;  // (dev_lights: light-triangle pdf tests really run, after the group culling)
    uint32_t dev_wide;  // visits of wide shortcut nodes (each also counts two items in dev_inner / dev_cached)
    uint32_t dev_light_loops;  // light queries the light tree did not serve: summed by the loop over every light
#ifdef RT_CYCLE_PROF  // instrumented experiment builds only (tools/ab_build.sh): where do a wave's cycles go?
    unsigned long long prof[16];  // (8..15: lane census per iteration kind, rt_render_paths_kernel)   0 trav wave-iterations, 1 trav lane-steps, 2 trav cycles, 3 vertex passes, 4 vertex lanes, 5 vertex cycles, 6 loop cycles
#endif
    __device__ __forceinline__ void clear() {
        paths = rays = inner = leaf = tri = hits = texels = lights = 0;
        dev_inner = dev_cached = dev_tri = dev_lights = rewalks = dev_wide = dev_light_loops = 0;
#ifdef RT_CYCLE_PROF
        for (int i = 0; i < 16; i++) prof[i] = 0;
#endif
    }
};
#ifdef RT_CYCLE_PROF
__device__ __forceinline__ bool prof_leader() { return (threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1; }
// -DRT_CYCLE_PROF=2: the clocks alone, in the kernel WITHOUT counters (the counters of RT_F_STATS slow the traversal down by
// a factor of two and more, which distorts the split); the lane census is left out then
#if RT_CYCLE_PROF == 2
#define RT_PROF_ON true
#else
#define RT_PROF_ON STATS
#endif
#define RT_PROF_COUNT(cn, i, v) do { if (RT_PROF_ON && prof_leader()) (cn).prof[i] += (v); } while (0)
#define RT_PROF_LANE(cn, i) do { if (RT_PROF_ON) (cn).prof[i] += 1; } while (0)
#define RT_PROF_CLOCK() (RT_PROF_ON ? (unsigned long long)clock64() : 0ull)
#else
#define RT_PROF_ON false
#define RT_PROF_COUNT(cn, i, v) do {} while (0)
#define RT_PROF_LANE(cn, i) do {} while (0)
#define RT_PROF_CLOCK() 0ull
#endif

// ------------------------------------------------------------------ intersection

// rayTriangleIntersect, main.cu:41-70, on the pre-subtracted edges.  Yields the ray parameter; the hit
// position is o + d * t (main.cu:66).
__device__ __forceinline__ bool ray_triangle_t(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float& outT) {
    const float epsilon = 1e-9f;
    const f3 p = cross(d, e2);
    const float det = dot(e1, p);
    if (det > -epsilon && det < epsilon) return false;
    const float inv_det = 1.0f / det;
    const f3 s = o - v0;
    const float u = inv_det * dot(s, p);
    if (u < 0.0f || u > 1.0f) return false;
    const f3 q = cross(s, e1);
    const float v = inv_det * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = inv_det * dot(e2, q);
    if (t > epsilon) {
        outT = t;
        return true;
    }
    return false;
}
__device__ __forceinline__ bool ray_triangle(f3 o, f3 d, f3 v0, f3 e1, f3 e2, f3& outPos) {
    float t;
    if (!ray_triangle_t(o, d, v0, e1, e2, t)) return false;
    outPos = o + d * t;
    return true;
}

// AABBIntersectDistance, main.cu:73-117, with 1/dir hoisted out (same value).
__device__ __forceinline__ float aabb_distance(f3 o, f3 inv, f3 mn, f3 mx) {
    if (o.x >= mn.x && o.x <= mx.x && o.y >= mn.y && o.y <= mx.y && o.z >= mn.z && o.z <= mx.z) return 0.0f;
    float tmin = 0.0f;
    float tmax = 1e13f;
    float t1 = (mn.x - o.x) * inv.x;
    float t2 = (mx.x - o.x) * inv.x;
    tmin = fmaxf(tmin, fminf(t1, t2));
    tmax = fminf(tmax, fmaxf(t1, t2));
    t1 = (mn.y - o.y) * inv.y;
    t2 = (mx.y - o.y) * inv.y;
    tmin = fmaxf(tmin, fminf(t1, t2));
    tmax = fminf(tmax, fmaxf(t1, t2));
    t1 = (mn.z - o.z) * inv.z;
    t2 = (mx.z - o.z) * inv.z;
    tmin = fmaxf(tmin, fminf(t1, t2));
    tmax = fminf(tmax, fmaxf(t1, t2));
    if (tmax < tmin) return 1e13f;
    if (tmin < 0.0f) return 1e13f;
    return tmin;
}

// Sphere::intersectSphere, sphere.h:30-46
__device__ __forceinline__ bool intersect_sphere(f3 o, f3 d, f3 c, float radius, f3& out_pos, f3& out_normal) {
    const f3 fromSphere = o - c;
    const float b = 2.0f * dot(fromSphere, d);
    if (b > 0) return false;
    const float cc = len2(fromSphere) - radius * radius;
    const float disc = b * b - 4.0f * cc;
    if (disc < 0) return false;
    const float t = (-b - sqrtf(disc)) / 2.0f;
    out_pos = o + d * t;
    out_normal = normalize(out_pos - c);
    return true;
}

// Floor::intersection, floor.h:8-21
__device__ __forceinline__ bool intersect_floor(f3 o, f3 d, float elevation, f3& pos, f3& normal) {
    if (d.y >= 0) return false;
    const float t = (elevation - o.y) / d.y;
    if (t < 0) return false;
    pos = o + d * t;
    normal = mk(0.f, 1.f, 0.f);
    return true;
}

// ImportanceTri::pdf_value, importanceTri.h:67-76 (normal, area precomputed at upload
// with the constructor's operations, importanceTri.h:19-25)
__device__ __forceinline__ float itri_pdf(f3 origin, f3 direction, f3 p1, f3 e1, f3 e2, f3 normal, float area) {
    f3 hit;
    if (!ray_triangle(origin, direction, p1, e1, e2, hit)) return 0.0f;
    const f3 to = hit - origin;
    const float distance_squared = len2(to);
    const float cos_theta = fabsf(dot(normalize(to), normal));
    return distance_squared / (cos_theta * area);
}

// ImportanceTri::generate_random, importanceTri.h:79-90
__device__ __forceinline__ f3 itri_sample(f3 origin, float r1, float r2, f3 p1, f3 p2, f3 p3) {
    const float su = sqrtf(r1);
    const float b0 = 1.0f - su;
    const float b1 = r2 * su;
    const float b2 = 1.0f - b0 - b1;
    const f3 random_point = p1 * b0 + p2 * b1 + p3 * b2;
    return normalize(random_point - origin);
}

// RayAttractor, rayAttractor.h:11-61 (vestigial in the reference; library function only)
struct Attractor {
    f3 pos, normal, u, v;
    float area;
    __device__ __forceinline__ void init(f3 p, f3 uu, f3 vv) {
        pos = p;
        area = sqrtf(len2(cross(uu, vv))) * 0.25f;
        u = normalize(uu);
        v = normalize(vv);
        normal = normalize(cross(u, v));
    }
    __device__ __forceinline__ bool intersect(f3 origin, f3 direction, f3& hit) const {
        const float denom = dot(normal, direction);
        if (fabs((double)denom) < 1e-12) return false;
        const float t = dot(pos - origin, normal) / denom;
        if (t <= 0) return false;
        hit = origin + direction * t;
        const float u_pos = dot(hit - pos, u);
        const float v_pos = dot(hit - pos, v);
        return u_pos >= -0.5f && u_pos <= 0.5f && v_pos >= -0.5f && v_pos <= 0.5f;
    }
    __device__ __forceinline__ float pdf(f3 origin, f3 direction) const {
        f3 hit;
        if (!intersect(origin, direction, hit)) return 0.0f;
        const f3 to = hit - origin;
        const float distance_squared = len2(to);
        const float cos_theta = fabsf(dot(normalize(to), normal));
        return distance_squared / (cos_theta * area);
    }
    __device__ __forceinline__ f3 sample(f3 origin, float r1, float r2) const {
        return pos + u * (r1 - 0.5f) + v * (r2 - 0.5f) - origin;
    }
};

// ------------------------------------------------------------------ traversal (main.cu:221-301)
//
// Visit order contract of the reference: root tested once (< 1e12); at an inner node
// both child boxes are tested, the far child is pushed first and the near child last,
// each only if dist^2 < minDistSqr AT PUSH TIME; ties (dA^2 == dB^2) pop B first; no
// re-test at pop; leaf triangles in array order; strict '<' keeps the first-tested
// triangle on equal distance.  The near child is kept in a register instead of being
// pushed and popped again (same order, half the LDS traffic).
//
// stk: this lane's column of the wave's LDS stack; entry i lives at stk[i * 64].
extern __shared__ float4 rt_lds[];  // dynamic LDS: [4 * n_cached node pieces][per-lane stacks]

#define RT_TRAV_DONE (-1)   // query finished
#define RT_TRAV_EMPTY 0x3ffffff0  // a leaf without triangles (visited, nothing to test); sorts after every item index
typedef float v2f __attribute__((ext_vector_type(2)));

// Per-lane traversal state of one closest-hit query.  Origin and 1/direction are kept as an (x, y) pair
// and z, the shape the packed box test consumes.
struct Trav {
    v2f oxy, ixy;
    float oz, iz;
    f3 d;
    bool axis_parallel;  // some 1/d component is not finite: the box test needs its inside shortcut (aabb_distance_nb)
    __device__ __forceinline__ f3 o() const { return mk(oxy.x, oxy.y, oz); }
    __device__ __forceinline__ f3 inv() const { return mk(ixy.x, ixy.y, iz); }
    int cur;      // item to visit next (inner node or triangle), RT_TRAV_EMPTY, or RT_TRAV_DONE
    int sp;       // entries on this lane's LDS stack
    int hitTri;   // place of the hit triangle in the item stream (Scene::tri_id gives hitTriIndex / 25), or -1
    float slack;  // sub-tree pruning: bound on the rounding error of a squared distance from this origin
    float subLim; // minD2 with that slack: what a sub-tree node's squared distance is compared with
    float invD2;  // 1 / |d|^2, a little up: the reference compares t^2 of a box with |d t|^2 of a hit (equal for unit directions)
    int sgx, sgy, sgz;  // wide shortcut nodes: 80 (bytes from a node's min.x piece to its max.x piece in the LDS window; & 64 = the same in device
                        // memory) where the ray runs DOWN that axis, else 0: a visit then fetches the near and far planes themselves
    float limT;   // wide shortcut nodes: sqrt(subLim), a little up -- what a child box's entry PARAMETER is compared with
    __device__ __forceinline__ void set_min(float m) {
        minD2 = m;
        subLim = (m * 1.00002f + slack) * invD2;
        limT = __builtin_amdgcn_sqrtf(subLim) * 1.00001f;  // (v_sqrt_f32, 1 ulp: the factor covers it; only ever makes the walk visit more)
    }
    float minD2;  // minDistSqr, main.cu:223
    float rival;  // shortcut walk: smallest squared distance of a hit in another reference leaf than the best one's; < 0 = this search walks the reference's tree
    int bestLeaf; // shortcut walk: leafEnd of the best hit (names its reference leaf), -1 before the first hit
    float hitT;   // ray parameter of the closest hit: its position o + d * hitT is formed again when the query is
                  // over (same operands, same operations as main.cu:66 -> same bits) instead of being carried along
    __device__ __forceinline__ f3 hit_pos() const { return o() + d * hitT; }
    __device__ __forceinline__ void set_hit(float ht, f3) { hitT = ht; }
};

// main.cu:221-236: reset the query and test the root box once.
__device__ __forceinline__ void trav_begin(const Scene& sc, Trav& t, f3 o, f3 d, float minD2_0 = 1e12f) {
    t.oxy = (v2f){o.x, o.y};
    t.oz = o.z;
    t.d = d;
    t.ixy = (v2f){1.0f / d.x, 1.0f / d.y};
    t.iz = 1.0f / d.z;
    t.axis_parallel = !(fabsf(t.ixy.x) < INFINITY && fabsf(t.ixy.y) < INFINITY && fabsf(t.iz) < INFINITY);
    t.sp = 0;
    t.hitTri = -1;
    {   // |fl(o + d t) - o|^2 is off by ~2 dist E, E = a few ulp of (|o| + dist); 2 dist E <= 4e-7 (|o|_1^2 + 3 dist^2)
        const float osum = fabsf(o.x) + fabsf(o.y) + fabsf(o.z);
        // from very far away the slab test itself is coarser than the padding: infinite slack = every node of a
        // sub-tree is visited (its boxes still order the children), i.e. the reference's walk over the whole leaf
        t.slack = fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))) <= sc.cull_far ? 1e-5f * osum * osum : INFINITY;
        t.invD2 = 1.00001f / len2(d);
    }
    t.set_min(minD2_0);  // 1e12, or |marchVelocity|^2 under ENABLE_RAY_MARCHING (main.cu:223-227)
    t.set_hit(0.0f, mk(0.f, 0.f, 0.f));
    // `AABBIntersectDistance(...) < 1e12` compares float with the DOUBLE 1e12 (main.cu:234):
    // true up to and including the float nearest below 1e12, i.e. f <= 1e12f.
    t.cur = (sc.has_root && aabb_distance(o, t.inv(), ld3(sc.root_min), ld3(sc.root_max)) <= 1e12f) ? sc.root_entry : RT_TRAV_DONE;
}

__device__ __forceinline__ void trav_pop(Trav& t, const int* stk) {
    if (t.sp > 0) { t.sp--; t.cur = stk[t.sp * 64]; }
    else t.cur = RT_TRAV_DONE;
}

// ONE traversal step of this lane (t.cur != RT_TRAV_DONE on entry): an inner-node visit
// (main.cu:265-279) or one triangle test of the current leaf (main.cu:249-264).  All lanes of
// the wave share the four item loads, whatever kind of item each of them holds.
//
// n_cached: the first n_cached inner items (the top of the tree in BFS order) are held in the
// dynamic LDS array as four piece arrays rt_lds[k * n_cached + node]: a divergent 16-byte request costs the
// vector-memory pipeline a full cycle each, the LDS serves them several times faster.
// The node cache: the first n_cached inner items (the top of the tree in BFS order) live in the dynamic
// LDS array as four piece arrays rt_lds[k * n_cached + node] -- a divergent 16-byte request costs the
// vector-memory pipeline a full cycle per lane, the LDS serves them several times faster.
// volatile + explicit address space: keeps these four ds_read_b128 apart from global loads (without it
// the compiler merges both sides into FLAT loads through a selected generic pointer, split into eleven
// narrower loads per step).
#ifndef RT_INNER_LOAD_PIN
#define RT_INNER_LOAD_PIN 1
#endif
#ifndef RT_TRI_LOAD_PIN
#define RT_TRI_LOAD_PIN 1
#endif
#ifndef RT_LDS_REC
#define RT_LDS_REC 5  // float4s per cached item: 4 pieces + 1 of padding (0 = four piece arrays, n_cached float4s apart)
#endif
#if RT_LDS_REC
#define RT_LDS_SLOT(item, piece, n_cached) ((int)__umul24((unsigned)(item), RT_LDS_REC) + (piece))  // (v_mul_u32_u24: full rate)
#define RT_LDS_CACHE_FLOAT4S(n_cached) ((size_t)RT_LDS_REC * (size_t)(n_cached))
#else
#define RT_LDS_SLOT(item, piece, n_cached) ((piece) * (n_cached) + (item))
#define RT_LDS_CACHE_FLOAT4S(n_cached) ((size_t)4 * (size_t)(n_cached))
#endif
__device__ __forceinline__ void load_cached_item(int item, int n_cached, float4& q0, float4& q1, float4& q2, float4& q3) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef const volatile __attribute__((address_space(3))) v4f* lds_v4f;
    const lds_v4f c = (lds_v4f)rt_lds;
    // records of 80 bytes: ONE address (item * 80) and four ds_read_b128 with immediate offsets; the odd stride in
    // 16-byte units spreads the items over all 16 bank groups like separate piece arrays would, without their three
    // extra uniform strides (which the compiler had no scalar registers left for: v_readlane from a spill VGPR per step)
    const v4f a0 = c[RT_LDS_SLOT(item, 0, n_cached)], a1 = c[RT_LDS_SLOT(item, 1, n_cached)], a2 = c[RT_LDS_SLOT(item, 2, n_cached)],
              a3 = c[RT_LDS_SLOT(item, 3, n_cached)];
    q0 = make_float4(a0.x, a0.y, a0.z, a0.w);
    q1 = make_float4(a1.x, a1.y, a1.z, a1.w);
    q2 = make_float4(a2.x, a2.y, a2.z, a2.w);
    q3 = make_float4(a3.x, a3.y, a3.z, a3.w);
}

// AABBIntersectDistance (main.cu:73-117) once more, for the traversal loop: no branches, and the x/y and
// z components of one child box as two-float vectors (v_pk_add_f32 / v_pk_mul_f32).  Same values as
// aabb_distance: a = mn - o and b = mx - o are the differences the slab test forms anyway, and "o inside
// the box" (mn <= o <= mx on every axis) is "every a <= 0 and every b >= 0" because an f32 difference
// has the sign of the exact one (denormals are kept); box bounds and ray origins are never NaN.  tmin
// cannot be negative (it starts at 0 and only grows), so the reference's `tmin < 0` exit is dead.
__device__ __forceinline__ float aabb_distance_pk(v2f oxy, v2f ozz, v2f ixy, v2f izz, v2f mn_xy, v2f mx_xy, v2f mnmx_z) {
    const v2f a = mn_xy - oxy, b = mx_xy - oxy, z = mnmx_z - ozz;  // z = (a.z, b.z)
    const bool inside = fmaxf(fmaxf(a.x, a.y), z.x) <= 0.0f && fminf(fminf(b.x, b.y), z.y) >= 0.0f;
    const v2f t1 = a * ixy, t2 = b * ixy, tz = z * izz;
    float tmin = fmaxf(0.0f, fminf(t1.x, t2.x));
    float tmax = fminf(1e13f, fmaxf(t1.x, t2.x));
    tmin = fmaxf(tmin, fminf(t1.y, t2.y));
    tmax = fminf(tmax, fmaxf(t1.y, t2.y));
    tmin = fmaxf(tmin, fminf(tz.x, tz.y));
    tmax = fminf(tmax, fmaxf(tz.x, tz.y));
    const float r = (tmax < tmin) ? 1e13f : tmin;
    return inside ? 0.0f : r;
}

#ifndef RT_INNER_PK
#define RT_INNER_PK 0  // 1 = the packed two-float form below: fewer instructions, slower (v_pk_*_f32 issues at half rate)
#endif
// INSIDE = false drops the inside test.  Allowed when 1/d is finite on every axis: an origin inside the box then has
// a <= 0 <= b per axis, so every slab gives min(t1, t2) <= 0 <= max(t1, t2) (no 0 * inf), tmin stays (+-)0 and
// tmax >= 0: the slab test itself returns (+-)0, whose square is the +0 of the shortcut.  Only a ray with a zero (or
// denormal) direction component can differ, and carries Trav::axis_parallel.
template <bool INSIDE = true>
__device__ __forceinline__ float aabb_distance_nb(f3 o, f3 inv, f3 mn, f3 mx) {  // scalar form of aabb_distance_pk
    const f3 a = mn - o, b = mx - o;
    const bool inside = INSIDE && fmaxf(fmaxf(a.x, a.y), a.z) <= 0.0f && fminf(fminf(b.x, b.y), b.z) >= 0.0f;
    const f3 t1 = a * inv, t2 = b * inv;
    float tmin = fmaxf(0.0f, fminf(t1.x, t2.x));
    float tmax = fminf(1e13f, fmaxf(t1.x, t2.x));
    tmin = fmaxf(tmin, fminf(t1.y, t2.y));
    tmax = fminf(tmax, fmaxf(t1.y, t2.y));
    tmin = fmaxf(tmin, fminf(t1.z, t2.z));
    tmax = fminf(tmax, fmaxf(t1.z, t2.z));
    const float r = (tmax < tmin) ? 1e13f : tmin;
    return inside ? 0.0f : r;
}

// Inner-node visit (main.cu:265-279) on the fetched item.  SUB: the scene has leaf sub-trees.
template <bool STATS, bool SUB>
__device__ __forceinline__ void visit_inner(Trav& t, int* stk, Counters& cn, float4 q0, float4 q1, float4 q2, float4 q3) {
    const bool sub = SUB && __float_as_int(q3.w) != 0;
    if (STATS) {
        cn.dev_inner++;
        if (sub) {  // the root of a sub-tree stands for the visit of its leaf and the tests of all its triangles
            cn.tri += (uint32_t)__float_as_int(q3.z);
            cn.leaf += __float_as_int(q3.z) != 0;
        } else {
            cn.inner++;
        }
    }
#if RT_INNER_PK
    const v2f oxy = t.oxy, ozz = {t.oz, t.oz}, ixy = t.ixy, izz = {t.iz, t.iz};
    float dA = aabb_distance_pk(oxy, ozz, ixy, izz, (v2f){q0.x, q0.y}, (v2f){q0.z, q0.w}, (v2f){q2.x, q2.y});
    float dB = aabb_distance_pk(oxy, ozz, ixy, izz, (v2f){q1.x, q1.y}, (v2f){q1.z, q1.w}, (v2f){q2.z, q2.w});
#else
    float dA, dB;
    if (__ballot(t.axis_parallel) == 0) {  // (wave-uniform: no lane of this step needs the inside test)
        dA = aabb_distance_nb<false>(t.o(), t.inv(), mk(q0.x, q0.y, q2.x), mk(q0.z, q0.w, q2.y));
        dB = aabb_distance_nb<false>(t.o(), t.inv(), mk(q1.x, q1.y, q2.z), mk(q1.z, q1.w, q2.w));
    } else {
        dA = aabb_distance_nb<true>(t.o(), t.inv(), mk(q0.x, q0.y, q2.x), mk(q0.z, q0.w, q2.y));
        dB = aabb_distance_nb<true>(t.o(), t.inv(), mk(q1.x, q1.y, q2.z), mk(q1.z, q1.w, q2.w));
    }
#endif
    dA = dA * dA;
    dB = dB * dB;
    const int eA = __float_as_int(q3.x), eB = __float_as_int(q3.y);
    const bool aNear = dA < dB;
    const int eNear = aNear ? eA : eB, eFar = aNear ? eB : eA;
    const float dNear = aNear ? dA : dB, dFar = aNear ? dB : dA;
    // the reference's rule for its own nodes; inside a sub-tree a child is skipped only if no triangle in its
    // (padded) box can come out at or below minD2, whatever the rounding of either squared distance
    const float lim = sub ? t.subLim : t.minD2;
    if (dFar < lim) { stk[t.sp * 64] = eFar; t.sp++; }
    if (dNear < lim) t.cur = eNear;
    else trav_pop(t, stk);
}

// Equal squared distances inside one reference leaf: the reference keeps the triangle it met first, i.e. the one
// with the smaller id (rare: the held hit's record is fetched again).
__device__ __forceinline__ bool tie_replaces(const Scene& sc, int heldPlace, float4 q2, bool whole_array) {
    if (heldPlace < 0) return false;
    const float4 h2 = sc.items[4 * ((size_t)sc.n_inner + heldPlace) + 2];
    return (whole_array || __float_as_int(h2.z) == __float_as_int(q2.z)) && __float_as_int(q2.w) < __float_as_int(h2.w);
}

// One triangle test of the current leaf (main.cu:249-264) on the fetched item.
// LQ: the lane may be running a LIGHT QUERY (Trav::bestLeaf <= RT_LQ_RUNNING, see light_query_begin): the item is then a light
// item, and a hit adds the light's pdf term (ImportanceTri::pdf_value, importanceTri.h:67-76, over the number of lights) to
// the lane's sum instead of competing for the closest hit.
#define RT_LQ_RUNNING (-3)
#define RT_LQ_GAVE_UP (-4)
template <bool STATS, bool SUB, bool FAST = false, bool LQ = false>
__device__ __forceinline__ void visit_triangle(const Scene& sc, Trav& t, const int* stk, Counters& cn, int item, float4 q0, float4 q1,
                                               float4 q2, float4 q3 = make_float4(0.f, 0.f, 0.f, 0.f)) {
    const int flags = __float_as_int(q2.y);
    const bool lq = LQ && t.bestLeaf <= RT_LQ_RUNNING;
    if (STATS) {
        if (lq) cn.dev_lights++;
        else { cn.tri += !(flags & 4); cn.leaf += (flags >> 1) & 1; cn.dev_tri++; }
    }
    float ht;
    const f3 o = t.o();
    if (ray_triangle_t(o, t.d, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x), ht)) {
        const f3 pos = o + t.d * ht;
        if (lq) {  // itri_pdf behind its ray_triangle: q2.z = the light's area, the fourth piece of the item (q3, fetched with the others
                   // where light queries run: a fetch of its own here would be one more memory latency for the whole wave) its normal.
                   // (Noting the hit and forming the term in the finishing pass was measured: slower, 58.1 vs 54.5 ms on scene2_is.txt.)
            const f3 to = pos - o;
            const float distance_squared = len2(to);
            const float cos_theta = fabsf(dot(normalize(to), mk(q3.x, q3.y, q3.z)));
            const float term = distance_squared / (cos_theta * q2.z) / (float)sc.n_lights;
            if (term != 0.0f) {  // (0 + a) + b: up to two nonzero terms commute; the third sends the query to the loop over all lights
                t.minD2 += term;
                t.hitTri++;
            }
            trav_pop(t, stk);
            return;
        }
        const float nd = len2(pos - o);
        if (nd < t.minD2 || (SUB && nd == t.minD2 && tie_replaces(sc, t.hitTri, q2, false))) {
            if (FAST && __float_as_int(q2.z) != t.bestLeaf) {  // the hit it replaces (1e12 before the first) is from another leaf: a rival
                t.rival = fminf(t.rival, t.minD2);
                t.bestLeaf = __float_as_int(q2.z);
            }
            t.set_min(nd);
            t.set_hit(ht, pos);
            t.hitTri = item - sc.n_inner;
        } else if (FAST && __float_as_int(q2.z) != t.bestLeaf) {
            t.rival = fminf(t.rival, nd);
        }
    }
    if ((flags & 1) || (FAST && sc.short_single && t.rival >= 0.0f)) trav_pop(t, stk);
    else t.cur++;
}

__device__ __forceinline__ bool trav_at_wide(const Scene& sc, const Trav& t);
template <bool STATS, bool ALL_CACHED = false, bool WIDE7 = false>
__device__ __forceinline__ void trav_step_wide(const Scene& sc, Trav& t, int* stk, Counters& cn, int n_cached, int cache_base);

// ONE traversal step of this lane (t.cur != RT_TRAV_DONE on entry): an inner-node visit or one
// triangle test of the current leaf.  All lanes of the wave share the four item loads, whatever kind
// of item each of them holds.
template <bool STATS, bool SUB, bool FAST = false>
__device__ __forceinline__ void trav_step(const Scene& sc, Trav& t, int* stk, Counters& cn, int n_cached = 0, int cache_base = 0) {
    if (t.cur == RT_TRAV_EMPTY) {  // empty leaf: counted as a visit, nothing to test
        if (STATS) cn.leaf++;
        trav_pop(t, stk);
        return;
    }
    if (FAST && trav_at_wide(sc, t)) {
        trav_step_wide<STATS>(sc, t, stk, cn, n_cached, cache_base);
        return;
    }
    float4 q0, q1, q2, q3;
    if ((unsigned)(t.cur - cache_base) < (unsigned)n_cached) {
        if (STATS) cn.dev_cached++;
        load_cached_item(t.cur - cache_base, n_cached, q0, q1, q2, q3);
    } else {
        const float4* w = sc.items + 4 * (size_t)t.cur;
        q0 = w[0]; q1 = w[1]; q2 = w[2]; q3 = w[3];
    }
    if (t.cur < sc.n_inner) visit_inner<STATS, SUB>(t, stk, cn, q0, q1, q2, q3);
    else visit_triangle<STATS, SUB, FAST>(sc, t, stk, cn, t.cur, q0, q1, q2);
}

// The two kinds of step on their own, for schedulers that run them in separate wave-uniform phases
// (rt_render_paths_kernel): an iteration then executes the code of one kind only.
__device__ __forceinline__ bool trav_at_inner(const Scene& sc, const Trav& t) { return (unsigned)t.cur < (unsigned)sc.n_inner; }

// cache_base: the LDS holds items [cache_base, cache_base + n_cached) (0 except in the shortcut walk's kernel)
template <bool STATS, bool SUB, bool ALL_CACHED = false>  // ALL_CACHED: every inner item is in the LDS (no other path compiled in)
__device__ __forceinline__ void trav_step_inner(const Scene& sc, Trav& t, int* stk, Counters& cn, int n_cached, int cache_base = 0) {  // trav_at_inner(t)
    float4 q0, q1, q2, q3;
    if (ALL_CACHED || (unsigned)(t.cur - cache_base) < (unsigned)n_cached) {
        if (STATS) cn.dev_cached++;
        load_cached_item(t.cur - cache_base, n_cached, q0, q1, q2, q3);
    } else {
        const float4* w = sc.items + 4 * (size_t)t.cur;
        q0 = w[0]; q1 = w[1]; q2 = w[2]; q3 = w[3];
#if RT_INNER_LOAD_PIN
        asm volatile("" : "+v"(q3.x), "+v"(q3.y), "+v"(q3.z), "+v"(q3.w));  // (one 16-byte load, not a 16-byte and a 4-byte one)
#endif
    }
    visit_inner<STATS, SUB>(t, stk, cn, q0, q1, q2, q3);
}

// ---- wide shortcut nodes (see WIDE SHORTCUT NODES above)
#ifndef RT_WIDE_SORT
#define RT_WIDE_SORT 1  // 1 = the children that are hit go on the stack in order of their entry distance (five compare-exchanges);
#endif                  // 0 = only the nearest is picked out, the others are pushed as they come (fewer instructions, a few more visits)
__device__ __forceinline__ bool trav_at_wide(const Scene& sc, const Trav& t) { return (unsigned)(t.cur - sc.wide_lo) < (unsigned)sc.wide_n; }

// entry parameter of one child box, +inf when the ray misses it or cannot find anything closer than its best hit there.
// n* / f*: the box's near and far plane per axis for this ray (its min or max bound, by the sign of the direction).
__device__ __forceinline__ float wide_child(f3 inv, f3 noi, float limT, float nx, float ny, float nz, float fx, float fy, float fz) {
    const float tmin = fmaxf(fmaxf(__builtin_fmaf(nx, inv.x, noi.x), __builtin_fmaf(ny, inv.y, noi.y)), fmaxf(__builtin_fmaf(nz, inv.z, noi.z), 0.0f));
    const float tmax = fminf(fminf(__builtin_fmaf(fx, inv.x, noi.x), __builtin_fmaf(fy, inv.y, noi.y)), fminf(__builtin_fmaf(fz, inv.z, noi.z), limT));
    return tmin <= tmax ? tmin : INFINITY;
}

#define RT_WIDE_CSWAP(da, ea, db, eb)            \
    do {                                         \
        const bool sw_ = db < da;                \
        const float dt_ = sw_ ? db : da;         \
        const int et_ = sw_ ? eb : ea;           \
        db = sw_ ? da : db;                      \
        eb = sw_ ? ea : eb;                      \
        da = dt_;                                \
        ea = et_;                                \
    } while (0)

// One visit of a wide shortcut node (trav_at_wide(t)).  cache_base / n_cached as in trav_step_inner; both items of the
// node must be inside the window for the LDS path.
// WIDE7 (the path kernel of a scene that does not sit in the LDS entirely): the LDS window holds WIDE nodes only, as records of the seven
// float4s a visit reads (112 bytes a node instead of the 160 of two padded items: 43 % more nodes in the window); n_cached still counts
// items (two per node) and Trav::sgx.. are 64 (short_begin / light_query_begin with sg = 64), as in device memory.
#ifndef RT_WIDE7_F4
#define RT_WIDE7_F4 7
#endif
template <bool STATS, bool ALL_CACHED, bool WIDE7>
__device__ __forceinline__ void trav_step_wide(const Scene& sc, Trav& t, int* stk, Counters& cn, int n_cached, int cache_base) {
    // a0..a2: the NEAR planes of the four children (x, y, z), b0..b2 the FAR ones, a3 the entries.  Which of a node's min / max
    // pieces that is follows from the sign of the ray's direction (Trav::sgx..): the addresses differ, not the arithmetic.
    float4 a0, a1, a2, a3, b0, b1, b2;
    const int rel = t.cur - cache_base;
    // (both items of the node inside the window; an EMPTY window -- n_cached = 0: RT_LDS_NODES=0, or an LDS that the stacks fill -- must
    // not wrap the bound around: every node then comes from device memory)
    if (ALL_CACHED || (unsigned)rel < (unsigned)max(n_cached - 1, 0)) {
        if (STATS) cn.dev_cached += 2;
        typedef float v4f __attribute__((ext_vector_type(4)));
        typedef const volatile __attribute__((address_space(3))) v4f* lds_v4f;
        static_assert(RT_LDS_REC == 5, "Trav::sgx.. assume records of five float4s per item in the LDS window");
        typedef const volatile __attribute__((address_space(3))) char* lds_bytes;
        if (WIDE7) {
            const lds_bytes c = (lds_bytes)rt_lds + (int)__umul24((unsigned)rel, 8 * RT_WIDE7_F4);  // (rel / 2) nodes of 16 * 7 bytes
            const v4f p0 = *(lds_v4f)(c + t.sgx), p1 = *(lds_v4f)(c + t.sgy + 16), p2 = *(lds_v4f)(c + t.sgz + 32), p3 = *(lds_v4f)(c + 48),
                      p4 = *(lds_v4f)(c - t.sgx + 64), p5 = *(lds_v4f)(c - t.sgy + 80), p6 = *(lds_v4f)(c - t.sgz + 96);
            a0 = make_float4(p0.x, p0.y, p0.z, p0.w); a1 = make_float4(p1.x, p1.y, p1.z, p1.w); a2 = make_float4(p2.x, p2.y, p2.z, p2.w);
            a3 = make_float4(p3.x, p3.y, p3.z, p3.w); b0 = make_float4(p4.x, p4.y, p4.z, p4.w); b1 = make_float4(p5.x, p5.y, p5.z, p5.w);
            b2 = make_float4(p6.x, p6.y, p6.z, p6.w);
        } else {
            const lds_bytes c = (lds_bytes)rt_lds + 16 * RT_LDS_SLOT(rel, 0, n_cached);
            const v4f p0 = *(lds_v4f)(c + t.sgx), p1 = *(lds_v4f)(c + t.sgy + 16), p2 = *(lds_v4f)(c + t.sgz + 32), p3 = *(lds_v4f)(c + 48),
                      p4 = *(lds_v4f)(c - t.sgx + 80), p5 = *(lds_v4f)(c - t.sgy + 96), p6 = *(lds_v4f)(c - t.sgz + 112);
            a0 = make_float4(p0.x, p0.y, p0.z, p0.w); a1 = make_float4(p1.x, p1.y, p1.z, p1.w); a2 = make_float4(p2.x, p2.y, p2.z, p2.w);
            a3 = make_float4(p3.x, p3.y, p3.z, p3.w); b0 = make_float4(p4.x, p4.y, p4.z, p4.w); b1 = make_float4(p5.x, p5.y, p5.z, p5.w);
            b2 = make_float4(p6.x, p6.y, p6.z, p6.w);
        }
    } else {
        // (32-bit byte offsets from the first wide node: the range of wide nodes is far below 4 GB; uniform base + lane offset)
        const char* wb = (const char*)(sc.items + 4 * (size_t)sc.wide_lo);
        const unsigned o = (unsigned)(t.cur - sc.wide_lo) * 64u;
        const unsigned gx = (unsigned)t.sgx & 64u, gy = (unsigned)t.sgy & 64u, gz = (unsigned)t.sgz & 64u;
        a0 = *(const float4*)(wb + (o + gx)); a1 = *(const float4*)(wb + (o + gy + 16u)); a2 = *(const float4*)(wb + (o + gz + 32u));
        a3 = *(const float4*)(wb + (o + 48u));
        b0 = *(const float4*)(wb + (o - gx + 64u)); b1 = *(const float4*)(wb + (o - gy + 80u)); b2 = *(const float4*)(wb + (o - gz + 96u));
        asm volatile("" : "+v"(a3.x), "+v"(a3.y), "+v"(a3.z), "+v"(a3.w));  // (16-byte loads, all issued before the first use)
    }
    if (STATS) { cn.dev_inner += 2; cn.dev_wide++; }
    if (t.sp > sc.wide_sp_max) {  // no room for three more entries: give up, the reference's walk decides (short_certified)
        t.bestLeaf = t.bestLeaf <= RT_LQ_RUNNING ? RT_LQ_GAVE_UP : -2;  // (a light query: its sum is then formed by the loop over all lights)
        t.cur = RT_TRAV_DONE;
        return;
    }
    const f3 inv = t.inv();
    const f3 noi = mk(-(t.oxy.x * inv.x), -(t.oxy.y * inv.y), -(t.oz * inv.z));
    float d0 = wide_child(inv, noi, t.limT, a0.x, a1.x, a2.x, b0.x, b1.x, b2.x);
    float d1 = wide_child(inv, noi, t.limT, a0.y, a1.y, a2.y, b0.y, b1.y, b2.y);
    float d2 = wide_child(inv, noi, t.limT, a0.z, a1.z, a2.z, b0.z, b1.z, b2.z);
    float d3 = wide_child(inv, noi, t.limT, a0.w, a1.w, a2.w, b0.w, b1.w, b2.w);
    int e0 = __float_as_int(a3.x), e1 = __float_as_int(a3.y), e2 = __float_as_int(a3.z), e3 = __float_as_int(a3.w);
#if RT_WIDE_SORT
    RT_WIDE_CSWAP(d0, e0, d1, e1);
    RT_WIDE_CSWAP(d2, e2, d3, e3);
    RT_WIDE_CSWAP(d0, e0, d2, e2);
    RT_WIDE_CSWAP(d1, e1, d3, e3);
    RT_WIDE_CSWAP(d1, e1, d2, e2);
#else
    RT_WIDE_CSWAP(d0, e0, d1, e1);  // the nearest to the front, the rest as they are
    RT_WIDE_CSWAP(d0, e0, d2, e2);
    RT_WIDE_CSWAP(d0, e0, d3, e3);
#endif
    // far ones first; an entry that is not hit is written too and left above the stack pointer (no branches)
    stk[t.sp * 64] = e3; t.sp += d3 < INFINITY;
    stk[t.sp * 64] = e2; t.sp += d2 < INFINITY;
    stk[t.sp * 64] = e1; t.sp += d1 < INFINITY;
    if (d0 < INFINITY) t.cur = e0;
    else trav_pop(t, stk);
}

// TLDS: a scene so small that its WHOLE item stream sits in the LDS cache (n_cached = inner items + triangles).
template <bool STATS, bool SUB, bool TLDS = false, bool FAST = false, bool LQ = false>
__device__ __forceinline__ void trav_step_leaf(const Scene& sc, Trav& t, const int* stk, Counters& cn, int n_cached = 0) {  // a triangle or RT_TRAV_EMPTY
    if (t.cur == RT_TRAV_EMPTY) {
        if (STATS) cn.leaf++;
        trav_pop(t, stk);
        return;
    }
    if (TLDS) {
        float4 q0, q1, q2, q3;
        load_cached_item(t.cur, n_cached, q0, q1, q2, q3);  // (the fourth piece is padding unless light queries run: its read is dropped)
        visit_triangle<STATS, SUB, FAST, LQ>(sc, t, stk, cn, t.cur, q0, q1, q2, q3);
        return;
    }
    const float4* w = sc.items + 4 * (size_t)t.cur;
    float4 q0 = w[0], q1 = w[1], q2 = w[2], q3 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (LQ) {
        q3 = w[3];
        asm volatile("" : "+v"(q3.x), "+v"(q3.y), "+v"(q3.z), "+v"(q3.w));
    }
#if RT_TRI_LOAD_PIN
    // three aligned 16-byte loads issued together: left alone, the compiler sinks the load of v0 behind the
    // determinant test (a second memory latency in most steps) and fetches the rest as unaligned pieces
    asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w), "+v"(q2.x), "+v"(q2.y),
                 "+v"(q2.z), "+v"(q2.w));
#endif
    visit_triangle<STATS, SUB, FAST, LQ>(sc, t, stk, cn, t.cur, q0, q1, q2, q3);
}

// Shortcut walk (see SHORTCUT TREE above).  short_begin: after trav_begin has tested the reference's root box, send the
// search down the shortcut tree if this ray may use it.  short_certified: the search is over (t.cur == RT_TRAV_DONE);
// true = the result stands (also: no hit at all -- the shortcut walk tests every triangle whose padded boxes the ray
// reaches, so the reference finds none either --, or the search already was the reference's walk).
__device__ __forceinline__ void short_begin(const Scene& sc, Trav& t, int sg = 80) {  // sg: see Trav::sgx (64 in a kernel with the WIDE7 window)
    t.bestLeaf = -1;
    // (wide nodes form bound / d - o / d: o / d must be finite, or a box the ray reaches could test as missed)
    const bool finite_od = sc.wide_n == 0 || (fabsf(t.oxy.x * t.ixy.x) < INFINITY && fabsf(t.oxy.y * t.ixy.y) < INFINITY && fabsf(t.oz * t.iz) < INFINITY);
    t.sgx = t.ixy.x < 0.0f ? sg : 0;
    t.sgy = t.ixy.y < 0.0f ? sg : 0;
    t.sgz = t.iz < 0.0f ? sg : 0;
    if (t.cur != RT_TRAV_DONE && !t.axis_parallel && t.slack < INFINITY && finite_od) {
        t.cur = sc.short_root;
        t.rival = 1e12f;
    } else {
        t.rival = -1.0f;
    }
}
template <bool HAVE_BOX = false>  // HAVE_BOX: (bmn, bmx) is the box of the hit's leaf, fetched already (hit_fetch) wherever this gets as far as reading it
__device__ __forceinline__ bool short_certified(const Scene& sc, const Trav& t, float4 bmn = make_float4(0.f, 0.f, 0.f, 0.f),
                                                float4 bmx = make_float4(0.f, 0.f, 0.f, 0.f)) {
    if (t.rival < 0.0f) return true;
    if (t.bestLeaf == -2) return false;  // the search gave up (stack room, trav_step_wide)
    if (t.hitTri < 0) return true;
    float4 mn, mx;
    if (HAVE_BOX) {
        mn = bmn; mx = bmx;
    } else {
        const int leaf = __float_as_int(sc.items[4 * ((size_t)sc.n_inner + t.hitTri) + 3].x);
        mn = sc.ref_boxes[2 * (size_t)leaf]; mx = sc.ref_boxes[2 * (size_t)leaf + 1];
    }
    const float bd = aabb_distance(t.o(), t.inv(), mk(mn.x, mn.y, mn.z), mk(mx.x, mx.y, mx.z));
    const float bd2 = bd * bd;
    return bd2 < t.rival && t.minD2 < t.rival && bd2 <= t.minD2 * 1.00001f + 0.5f * t.slack;
}

// LIGHT QUERY.  reflectType5 sums ImportanceTri::pdf_value over EVERY importance-sampled triangle for the sampled direction
// (material.h:315-323); the terms that are not +0 belong to the lights the ray (hitPos, outgoing) HITS.  With a light tree
// (Scene::light_root) the lane finds them as an all-hits search through its ordinary traversal steps: nothing is pruned by
// distance, a leaf is one light item, a hit adds its term (visit_triangle).  The query's state rides in fields a closest-hit
// search does not need meanwhile: minD2 = the sum so far, hitTri = nonzero terms so far, hitT = pdf_cos, bestLeaf = the mode.
// False: this ray cannot use the tree (same conditions as short_begin) -- the caller runs the loop over all lights instead.
__device__ __forceinline__ bool light_query_begin(const Scene& sc, Trav& t, f3 o, f3 d, float pdf_cos, int sg = 80) {
    const f3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const bool ok = fabsf(inv.x) < INFINITY && fabsf(inv.y) < INFINITY && fabsf(inv.z) < INFINITY && fabsf(o.x * inv.x) < INFINITY &&
                    fabsf(o.y * inv.y) < INFINITY && fabsf(o.z * inv.z) < INFINITY &&
                    fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))) <= sc.cull_far;
    if (!ok) return false;
    t.oxy = (v2f){o.x, o.y};
    t.oz = o.z;
    t.d = d;
    t.ixy = (v2f){inv.x, inv.y};
    t.iz = inv.z;
    t.axis_parallel = false;
    t.sgx = inv.x < 0.0f ? sg : 0;
    t.sgy = inv.y < 0.0f ? sg : 0;
    t.sgz = inv.z < 0.0f ? sg : 0;
    t.sp = 0;
    t.limT = 3e38f;   // all hits: no child is too far
    t.rival = -1.0f;  // (not a shortcut search: nothing to certify)
    t.bestLeaf = RT_LQ_RUNNING;
    t.minD2 = 0.0f;
    t.hitTri = 0;
    t.hitT = pdf_cos;
    t.cur = sc.light_root;
    return true;
}

__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// ENABLE_BVH false, main.cu:285-301: every triangle, in array order there; here in item order with the tie rule
template <bool STATS>
__device__ __forceinline__ void trav_brute_force(const Scene& sc, Trav& t, Counters& cn) {
    for (int i = 0; i < sc.n_tris; i++) {
        const float4* h = sc.items + 4 * ((size_t)sc.n_inner + i);
        const float4 h0 = h[0], h1 = h[1], h2 = h[2];
        if (STATS) { cn.tri++; cn.dev_tri++; }
        float ht;
        const f3 o = t.o();
        if (ray_triangle_t(o, t.d, mk(h0.x, h0.y, h0.z), mk(h0.w, h1.x, h1.y), mk(h1.z, h1.w, h2.x), ht)) {
            const f3 pos = o + t.d * ht;
            const float nd = len2(pos - o);
            if (nd < t.minD2 || (sc.n_sub > 0 && nd == t.minD2 && tie_replaces(sc, t.hitTri, h2, true))) {
                t.set_min(nd);
                t.set_hit(ht, pos);
                t.hitTri = i;
            }
        }
    }
    t.cur = RT_TRAV_DONE;
}

// Can the ray o + t d, t >= 0, reach the box?  Errs on the side of yes (the boxes are padded at upload and the
// comparison has slack), since a no skips triangle tests whose outcome must not change.
struct BoxRay {
    f3 o, inv;
    __device__ __forceinline__ void set(f3 origin, f3 d) {
        o = origin;  // an axis the ray does not move along: a huge finite 1/d puts the slab's interval at -+1e26 and beyond
        inv = mk(d.x == 0.0f ? 1e30f : 1.0f / d.x, d.y == 0.0f ? 1e30f : 1.0f / d.y, d.z == 0.0f ? 1e30f : 1.0f / d.z);
    }
    __device__ __forceinline__ bool may_reach(float4 mn, float4 mx) const {
        const float ax = (mn.x - o.x) * inv.x, bx = (mx.x - o.x) * inv.x;
        const float ay = (mn.y - o.y) * inv.y, by = (mx.y - o.y) * inv.y;
        const float az = (mn.z - o.z) * inv.z, bz = (mx.z - o.z) * inv.z;
        const float tmin = fmaxf(fmaxf(0.0f, fminf(ax, bx)), fmaxf(fminf(ay, by), fminf(az, bz)));
        const float tmax = fminf(fmaxf(ax, bx), fminf(fmaxf(ay, by), fmaxf(az, bz)));
        return tmin <= tmax * 1.00001f;
    }
};

// Analytic spheres after the triangles, main.cu:305-334.  Returns the sphere index or -1.
__device__ __forceinline__ int trav_spheres(const Scene& sc, Trav& t, f3& sPos) {
    int hitSphere = -1;
    for (int i = 0; i < sc.n_spheres; i++) {
        const float4 s0 = sc.spheres[2 * i];
        f3 hp, hn;
        if (intersect_sphere(t.o(), t.d, mk(s0.x, s0.y, s0.z), s0.w, hp, hn)) {
            const float d2 = len2(hp - t.o());
            if (d2 >= t.minD2) continue;
            t.minD2 = d2;
            sPos = hp;
            hitSphere = i;
        }
    }
    return hitSphere;
}

// Whole query for one lane (lock-step fused kernel): steps until done.
template <bool STATS>
__device__ __forceinline__ void trace_closest(const Scene& sc, f3 o, f3 d, bool use_bvh, int* stk, int& hitTri,
                                              float& minD2, f3& hitPos, Counters& cn, int n_cached = 0) {
    Trav t;
    if (STATS) cn.rays++;
    trav_begin(sc, t, o, d);
    if (!use_bvh) trav_brute_force<STATS>(sc, t, cn);
    // one mixed step per iteration (a "while-while" split of inner and triangle steps measured slower in lock step,
    // profiles/r01_ab_megakernel.txt)
    while (t.cur != RT_TRAV_DONE) {
        RT_PROF_COUNT(cn, 0, 1);
        RT_PROF_LANE(cn, 1);
        trav_step<STATS, true>(sc, t, stk, cn, n_cached);
    }
    hitTri = t.hitTri;
    minD2 = t.minD2;
    hitPos = t.hitTri >= 0 ? t.hit_pos() : mk(0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------ shading (material.h)

struct PathState {
    f3 o, d;        // Ray::position, direction        ray.h:13-14
    f3 thr;         // Ray::diffuseMultiplier          ray.h:15
    f3 emission;    // Ray::emission                   ray.h:16
    float ri;       // Ray::refractiveIndex            ray.h:17
    float volDensity;  // Ray::volumeDensity           ray.h:18
    f3 volColor;    // Ray::volumeColor                ray.h:19
};

__device__ __forceinline__ f3 fetch_texel(const Scene& sc, long long i) {
    if (sc.n_tex < 3) return mk(0.f, 0.f, 0.f);
    const long long hi = sc.n_tex - 3;
    i = i < 0 ? 0 : (i > hi ? hi : i);  // only the final index is clamped
    return mk(sc.tex[i], sc.tex[i + 1], sc.tex[i + 2]);
}

// Material::randSphereVec, material.h:130-140 (same body: Sphere::randSphereVec, sphere.h:48-58)
__device__ __forceinline__ f3 rand_sphere_vec(float r1, float r2) {
    const float theta = r1 * 2.0f * 3.14159f;
    const float phi = pm::acos(2.0f * r2 - 1.0f);
    float sp, cp, st, ct;
    pm::sincos(phi, &sp, &cp);
    pm::sincos(theta, &st, &ct);
    return mk(sp * ct, sp * st, cp);
}

// Vec3::randSphere, vec3.h:133-142: the same but for theta, whose pi is the DOUBLE literal 3.14159 (vec3.h:136):
// (r1 * 2.0f) is promoted, multiplied in double and rounded to float.  Only checkScatteredVolume calls it (main.cu:137).
__device__ __forceinline__ f3 rand_sphere_vec3(float r1, float r2) {
    const float theta = (float)((double)(r1 * 2.0f) * 3.14159);
    const float phi = pm::acos(2.0f * r2 - 1.0f);
    float sp, cp, st, ct;
    pm::sincos(phi, &sp, &cp);
    pm::sincos(theta, &st, &ct);
    return mk(sp * ct, sp * st, cp);
}

// Material::randCosineHemisphere, material.h:152-161
__device__ __forceinline__ f3 rand_cosine_hemisphere(f3 normal, float r1, float r2) {
    const f3 s = rand_sphere_vec(r1, r2);
    const float cos_theta = dot(s, normal);
    if (cos_theta > 0.0f) return s * sqrtf(cos_theta);
    return s * -1.0f * sqrtf(-cos_theta);
}

// barycentric + interpolateNormal, material.h:110-128, on the stored edges
__device__ __forceinline__ f3 barycentric(f3 a, f3 e1, f3 e2, f3 pos) {
    const f3 v2 = pos - a;
    const float d00 = dot(e1, e1);
    const float d01 = dot(e1, e2);
    const float d11 = dot(e2, e2);
    const float d20 = dot(v2, e1);
    const float d21 = dot(v2, e2);
    const float denom = d00 * d11 - d01 * d01;
    const float v = (d11 * d20 - d01 * d21) / denom;
    const float w = (d00 * d21 - d01 * d20) / denom;
    const float u = 1.0f - v - w;
    return mk(u, v, w);
}

struct HitFrame {
    f3 normal;
    float u, v;
    int mat;
};

// The reference's id (hitTriIndex / 25) of the triangle at a place of the item stream.
__device__ __forceinline__ int tri_id(const Scene& sc, int place) {
    return __float_as_int(sc.items[4 * ((size_t)sc.n_inner + place) + 2].w);
}

#ifndef RT_HIT_FETCH_LDS
#define RT_HIT_FETCH_LDS 1
#endif
#ifndef RT_HIT_PREFETCH_RNG
#define RT_HIT_PREFETCH_RNG 1
#endif
// What a vertex pass reads about a hit triangle at its top (hit_fetch): the triangle's item -- all four pieces, once, for the certificate
// (the id of its reference leaf) AND the shading (vertex, edges, id) -- and the leaf's box behind it: a chain of four dependent
// requests (item -> leaf box | cold data -> texel) instead of five (leaf id -> leaf box -> item -> cold data -> texel).  (The cold
// data fetched up front as well was measured: 36 more live registers, spills.)
struct HitFetch {
    float4 h0, h1, h2, mn, mx;
    uint4 A;  // the bounce's first Philox block (every material that draws numbers starts with it), formed while the item is on its way
};
template <bool TLDS = false>  // TLDS: the whole item stream sits in the LDS (n_cached items): the triangle's record is read there
__device__ __forceinline__ void hit_fetch(const Scene& sc, int place, HitFetch& f, const Rng& rng, uint32_t bb, int n_cached = 0) {
    float4 h3;
    if (TLDS && RT_HIT_FETCH_LDS) {
        load_cached_item(sc.n_inner + place, n_cached, f.h0, f.h1, f.h2, h3);
    } else {
        const float4* t = sc.items + 4 * ((size_t)sc.n_inner + place);
        f.h0 = t[0]; f.h1 = t[1]; f.h2 = t[2];
        h3 = t[3];
        asm volatile("" : "+v"(h3.x), "+v"(f.h0.x), "+v"(f.h1.x), "+v"(f.h2.x));  // (the four pieces of the item in one go)
    }
#if RT_HIT_PREFETCH_RNG
    f.A = rng.block(bb);
    asm volatile("" : "+v"(f.A.x), "+v"(f.A.y), "+v"(f.A.z), "+v"(f.A.w), "+v"(h3.x));  // (... before the item is looked at)
#endif
    const int leaf = __float_as_int(h3.x);
    f.mn = sc.ref_boxes[2 * (size_t)leaf]; f.mx = sc.ref_boxes[2 * (size_t)leaf + 1];
}

template <bool PF = false>  // PF: pf holds the records (hit_fetch)
__device__ __forceinline__ HitFrame hit_frame(const Scene& sc, int place, f3 hitPos, const HitFetch& pf) {  // material.h:44-62
    const float4* t = sc.items + 4 * ((size_t)sc.n_inner + place);
    float4 h0, h1, h2, c0, c1, c2, c3;
    if (PF) {
        h0 = pf.h0; h1 = pf.h1; h2 = pf.h2;
    } else {
        h0 = t[0]; h1 = t[1]; h2 = t[2];
    }
    const float4* c = sc.tri_cold + 4 * (size_t)__float_as_int(h2.w);  // (cold data stays in the reference's order)
    c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3];
    const f3 bary = barycentric(mk(h0.x, h0.y, h0.z), mk(h0.w, h1.x, h1.y), mk(h1.z, h1.w, h2.x), hitPos);
    const f3 n0 = mk(c0.x, c0.y, c0.z), n1 = mk(c0.w, c1.x, c1.y), n2 = mk(c1.z, c1.w, c2.x);
    HitFrame f;
    f.normal = normalize(n0 * bary.x + n1 * bary.y + n2 * bary.z);
    // uv0 = (c2.y, c2.z) uv1 = (c2.w, c3.x) uv2 = (c3.y, c3.z)
    f.u = bary.x * c2.y + bary.y * c2.w + bary.z * c3.y;
    f.v = bary.x * c2.z + bary.y * c3.x + bary.z * c3.z;
    f.mat = __float_as_int(c3.w);
    return f;
}

__device__ __forceinline__ HitFrame hit_frame(const Scene& sc, int place, f3 hitPos) {
    HitFetch none;
    return hit_frame<false>(sc, place, hitPos, none);
}

// reflectType5 (material.h:280-336) sums the pdf of EVERY importance-sampled triangle for the sampled direction
// (:315-323), an O(numLights) loop of ray / triangle tests per hit.  A kernel whose waves are converged around the
// shading call can ask shade_triangle to stop in front of that loop (DEFER), run it with all 64 lanes for one
// waiting lane after the other (wave_light_pdf) and finish with finish_type5.
#ifndef RT_LIGHTS_COOP_MIN
#define RT_LIGHTS_COOP_MIN 32  // fewer lights: the lane's own loop is cheaper
#endif
struct LightQuery {
    bool pending;
    f3 hitPos, outgoing, clr;
    float pdf_cos;
};

// Material::reflect, material.h:41-106.  Returns true when the path terminates.
template <bool STATS, bool PF = false>
// The bounce's dimensions 8+8j .. 15+8j are Philox blocks bb and bb+1 (bb = 2 + 2j): A = dims k0..k3,
// B = dims k4..k7, consumed in the reference's call order.
__device__ __forceinline__ bool shade_triangle(const Scene& sc, PathState& ray, f3 hitPos, int tri, const Rng& rng,
                                               uint32_t bb, Counters& cn, LightQuery* lq, int mats_lds, const HitFetch& pf) {
    const HitFrame hf = hit_frame<PF>(sc, tri, hitPos, pf);
    const f3 normal = hf.normal;
    int type = 127;  // out-of-range material id -> "the color of error"
    f3 clr = mk(0.f, 0.f, 0.f);
    float p1 = 0.f, p2 = 0.f;
    int off0 = -1, off1 = -1, off2 = -1, tw = 0, th = 0;
    if (hf.mat >= 0 && hf.mat < sc.n_mats) {
        float4 m0, m1, m2;
        if (mats_lds >= 0) {  // the material table sits in LDS behind the stacks (rt_render_paths_kernel): three ds_read_b128
            typedef float v4f __attribute__((ext_vector_type(4)));  // instead of four or five narrow global requests per hit
            typedef const volatile __attribute__((address_space(3))) v4f* lds_v4f;
            const lds_v4f c = (lds_v4f)rt_lds + mats_lds + 3 * hf.mat;
            const v4f a0 = c[0], a1 = c[1], a2 = c[2];
            m0 = make_float4(a0.x, a0.y, a0.z, a0.w);
            m1 = make_float4(a1.x, a1.y, a1.z, a1.w);
            m2 = make_float4(a2.x, a2.y, a2.z, a2.w);
        } else {
            const float4* m = sc.mats + 3 * (size_t)hf.mat;
            m0 = m[0]; m1 = m[1]; m2 = m[2];
        }
        type = __float_as_int(m0.x);
        clr = mk(m0.y, m0.z, m0.w);
        p1 = m1.x;
        p2 = m1.y;
        off0 = __float_as_int(m1.w);
        off1 = __float_as_int(m2.x);
        off2 = __float_as_int(m2.y);
        tw = __float_as_int(m2.z);
        th = __float_as_int(m2.w);
    }
    f3 clr0 = clr;
    if (off0 >= 0) {  // getTextureColor, material.h:163-175
        const float px = hf.u * (float)tw;
        const float py = hf.v * (float)th;
        const long long i = (long long)off0 + 3LL * (int)px + 3LL * tw * (int)py;
        clr0 = fetch_texel(sc, i);
        if (STATS) cn.texels++;
    }
    // images 2 and 3 are fetched and never used by the reference (material.h:70-75): only counted.
    if (STATS) cn.texels += (off1 >= 0) + (off2 >= 0);

    switch (type) {
        case 0: {  // reflectType0, material.h:182-202
            ray.o = hitPos;
            const uint4 A = (PF && RT_HIT_PREFETCH_RNG) ? pf.A : rng.block(bb);
            if (u01(A.x) < p1) {
                ray.thr = ray.thr * mk(0.95f, 0.95f, 0.95f);
                ray.d = reflect(ray.d, normal);
            } else {
                ray.thr = ray.thr * clr;
                const uint4 B = rng.block(bb + 1);
                const float g0 = gauss(A.y, A.z);
                const float g1 = gauss(A.w, B.x);
                const float g2 = gauss(B.y, B.z);
                ray.d = reflect(ray.d, normalize(normal + mk(g0, g1, g2) * p2));
            }
            ray.o = ray.o + ray.d * 1e-5f;
            return false;
        }
        case 1: {  // reflectType1, material.h:205-213; always terminates (:82-84)
            if (dot(ray.d, normal) > 0.0f) ray.emission = mk(0.f, 0.f, 0.f);
            else ray.emission = clr * p1;
            return true;
        }
        case 2: {  // reflectType2, material.h:216-254
            ray.o = hitPos;
            const bool fromOutside = dot(ray.d, normal) < 0.0f;
            const float oldRI = ray.ri;
            const float newRI = fromOutside ? oldRI + p1 : oldRI - p1;
            const f3 normal2 = fromOutside ? normal : normal * -1.0f;
            const float ri = oldRI / newRI;
            const float cos_theta = fminf(-dot(ray.d, normal2), 1.0f);
            const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
            bool will_refract = ri * sin_theta <= 1.0f;
            if (will_refract) {
                float r0 = (1 - ri) / (1 + ri);
                r0 = r0 * r0;
                // std::pow(float,int) is a double pow in the reference; x^5 by double products.
                const double x = (double)(1 - cos_theta);
                const double x2 = x * x;
                r0 = (float)((double)r0 + (double)(1 - r0) * (x2 * x2 * x));
                if (r0 > u01(((PF && RT_HIT_PREFETCH_RNG) ? pf.A : rng.block(bb)).x)) will_refract = false;
            }
            if (will_refract) {
                ray.d = refract(ray.d, normal2, ri);
                ray.ri = newRI;
            } else {
                ray.d = reflect(ray.d, normal2);  // (three discarded normals: dims left unused)
            }
            ray.o = ray.o + ray.d * 1e-5f;
            return false;
        }
        case 3: {  // reflectType3, material.h:257-266
            const f3 normal2 = (dot(normal, ray.d) < 0.0f) ? normal : (normal * -1.0f);
            ray.thr = ray.thr * clr0;
            ray.o = hitPos;
            const uint4 A = (PF && RT_HIT_PREFETCH_RNG) ? pf.A : rng.block(bb);
            ray.d = normalize(normal2 + rand_sphere_vec(u01(A.x), u01(A.y)));
            ray.o = ray.o + ray.d * 1e-5f;
            return false;
        }
        case 4: {  // reflectType4, material.h:269-277
            ray.thr = ray.thr * clr;
            ray.o = hitPos;
            ray.d = reflect(ray.d, normal);
            const uint4 A = (PF && RT_HIT_PREFETCH_RNG) ? pf.A : rng.block(bb);
            ray.d = ray.d + rand_sphere_vec(u01(A.x), u01(A.y)) * p1;
            ray.o = ray.o + ray.d * 1e-5f;
            return false;
        }
        case 5: {  // reflectType5, material.h:280-336
            const f3 normal2 = (dot(normal, ray.d) < 0.0f) ? normal : (normal * -1.0f);
            ray.o = hitPos;
            f3 outgoing;
            bool sampledLight = false;
            const uint4 A = (PF && RT_HIT_PREFETCH_RNG) ? pf.A : rng.block(bb);
            if (sc.n_lights > 0) {
                if (u01(A.x) < 0.5f) {  // dims k0 (choice), k1 (raw pick), k2, k3
                    const uint32_t pick = A.y % (uint32_t)sc.n_lights;
                    const int pj = sc.light_place[pick];
                    const float4 l0 = sc.lights[RT_LIGHT_F4(pj, 0)], l3 = sc.lights[RT_LIGHT_F4(pj, 3)], l4 = sc.lights[RT_LIGHT_F4(pj, 4)];
                    outgoing = itri_sample(hitPos, u01(A.z), u01(A.w), mk(l0.x, l0.y, l0.z), mk(l3.y, l3.z, l3.w), mk(l4.x, l4.y, l4.z));
                    sampledLight = true;
                } else {
                    outgoing = rand_cosine_hemisphere(normal2, u01(A.y), u01(A.z));  // k1, k2
                }
            } else {
                outgoing = rand_cosine_hemisphere(normal2, u01(A.x), u01(A.y));  // no choice draw without lights
            }
            float cos_theta = dot(normal2, outgoing);
            if (cos_theta <= 0.0f && sampledLight) {  // redraw: k4, k5
                const uint4 B = rng.block(bb + 1);
                outgoing = rand_cosine_hemisphere(normal2, u01(B.x), u01(B.y));
                cos_theta = dot(normal2, outgoing);
            }
            if (cos_theta < 0.0f) cos_theta = 0.0f;
            const float inv_pi = 1.0f / 3.14159f;
            const float pdf_cos = cos_theta * inv_pi;
            if (lq && sc.n_lights >= RT_LIGHTS_COOP_MIN) {  // the caller runs the light loop with the whole wave
                lq->pending = true;
                lq->hitPos = hitPos;
                lq->outgoing = outgoing;
                lq->clr = clr;
                lq->pdf_cos = pdf_cos;
                return false;
            }
            float pdf_light = 0.0f;
            for (int i = 0; i < sc.n_lights; i++) {  // O(numLights), material.h:315-323, in the reference's order
                const int pj = sc.light_place[i];
                const float4 l0 = sc.lights[RT_LIGHT_F4(pj, 0)], l1 = sc.lights[RT_LIGHT_F4(pj, 1)], l2 = sc.lights[RT_LIGHT_F4(pj, 2)],
                             l3 = sc.lights[RT_LIGHT_F4(pj, 3)];
                if (STATS) { cn.lights++; cn.dev_lights++; }
                pdf_light += itri_pdf(hitPos, outgoing, mk(l0.x, l0.y, l0.z), mk(l0.w, l1.x, l1.y), mk(l1.z, l1.w, l2.x),
                                      mk(l2.y, l2.z, l2.w), l3.x) / (float)sc.n_lights;
            }
            const float pdf = (sc.n_lights > 0) ? (0.5f * pdf_cos + 0.5f * pdf_light) : pdf_cos;
            if (pdf <= 1e-12f) {
                ray.thr = mk(0.f, 0.f, 0.f);
                return false;
            }
            ray.thr = ray.thr * clr * (pdf_cos / pdf);
            ray.d = outgoing;
            ray.o = ray.o + ray.d * 1e-5f;
            return false;
        }
        case 6: {  // reflectType6, material.h:339-352 (state only; scattering is compiled out, config.h:9)
            if (dot(normal, ray.d) < 0.0f) {
                ray.volDensity += p1;
                ray.volColor = clr;
            } else {
                ray.volDensity -= p1;
            }
            ray.o = hitPos;
            ray.o = ray.o + ray.d * 1e-5f;
            return false;
        }
        default:
            ray.emission = mk(1.0f, 0.0f, 1.0f);
            return true;
    }
}

template <bool STATS>
__device__ __forceinline__ bool shade_triangle(const Scene& sc, PathState& ray, f3 hitPos, int tri, const Rng& rng, uint32_t bb, Counters& cn,
                                               LightQuery* lq = nullptr, int mats_lds = -1) {
    HitFetch none;
    return shade_triangle<STATS, false>(sc, ray, hitPos, tri, rng, bb, cn, lq, mats_lds, none);
}

// The loop of material.h:315-323 as the reference runs it: every light, in list order.
template <bool STATS>
__device__ __forceinline__ float light_pdf_loop(const Scene& sc, f3 hitPos, f3 outgoing, Counters& cn) {
    float pdf_light = 0.0f;
    for (int i = 0; i < sc.n_lights; i++) {
        const int pj = sc.light_place[i];
        const float4 l0 = sc.lights[RT_LIGHT_F4(pj, 0)], l1 = sc.lights[RT_LIGHT_F4(pj, 1)], l2 = sc.lights[RT_LIGHT_F4(pj, 2)],
                     l3 = sc.lights[RT_LIGHT_F4(pj, 3)];
        if (STATS) { cn.lights++; cn.dev_lights++; }
        pdf_light += itri_pdf(hitPos, outgoing, mk(l0.x, l0.y, l0.z), mk(l0.w, l1.x, l1.y), mk(l1.z, l1.w, l2.x), mk(l2.y, l2.z, l2.w), l3.x) /
                     (float)sc.n_lights;
    }
    return pdf_light;
}

// The tail of reflectType5 behind the light loop (material.h:324-335).
__device__ __forceinline__ void finish_type5(const Scene& sc, PathState& ray, const LightQuery& q, float pdf_light) {
    const float pdf = (sc.n_lights > 0) ? (0.5f * q.pdf_cos + 0.5f * pdf_light) : q.pdf_cos;
    if (pdf <= 1e-12f) {
        ray.thr = mk(0.f, 0.f, 0.f);
        return;
    }
    ray.thr = ray.thr * q.clr * (q.pdf_cos / pdf);
    ray.d = q.outgoing;
    ray.o = ray.o + ray.d * 1e-5f;
}

// The light loop of reflectType5 for every lane with a pending query.  Call with the wave converged.  Same sum as
// the lane's own loop: the terms left out are the +0 of triangles the direction misses (x + 0 = x), and the nonzero
// terms are added in the REFERENCE's light order.  The records are stored in spatial order (upload), so the order
// of the tests is not the order of the sum: up to two nonzero terms commute ((0 + a) + b == (0 + b) + a bit for bit),
// which is every query but the rare one whose direction crosses three or more lights; those are summed again by
// selection (wave_light_pdf_ordered).  Per pass over RT_LIGHT_CHUNK lights: every waiting lane marks the groups of
// RT_LIGHT_GROUP lights whose bounds its ray can reach (all lanes at once, bounds through the scalar cache), then
// lane after lane the wave tests the marked groups' triangles, four groups per step.
#define RT_LIGHT_CHUNK (64 * RT_LIGHT_GROUP)

// Exact sum for ONE query (wave-uniform hp, og) whatever the number of nonzero terms: pass after pass over all the
// lights, each pass adds the term of the smallest reference index not added yet.
template <bool STATS>
__device__ __forceinline__ float wave_light_pdf_ordered(const Scene& sc, f3 hp, f3 og, Counters& cn) {
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    int last = -1;
    for (;;) {
        int best = 0x7fffffff;
        float bestTerm = 0.0f;
        for (int i0 = 0; i0 < sc.n_lights; i0 += 64) {
            const int i = i0 + lane;
            float term = 0.0f;
            int oi = 0x7fffffff;
            if (i < sc.n_lights) {
                if (STATS) cn.dev_lights++;
                const float4 l0 = sc.lights[RT_LIGHT_F4(i, 0)], l1 = sc.lights[RT_LIGHT_F4(i, 1)], l2 = sc.lights[RT_LIGHT_F4(i, 2)],
                             l3 = sc.lights[RT_LIGHT_F4(i, 3)], l4 = sc.lights[RT_LIGHT_F4(i, 4)];
                term = itri_pdf(hp, og, mk(l0.x, l0.y, l0.z), mk(l0.w, l1.x, l1.y), mk(l1.z, l1.w, l2.x), mk(l2.y, l2.z, l2.w), l3.x) /
                       (float)sc.n_lights;
                oi = __float_as_int(l4.w);
            }
            for (unsigned long long h = __ballot(term != 0.0f && oi > last); h; h &= h - 1) {
                const int l = __ffsll((long long)h) - 1;
                const int o = __builtin_amdgcn_readlane(oi, l);
                if (o < best) { best = o; bestTerm = lane_f(term, l); }
            }
        }
        if (best == 0x7fffffff) break;
        acc += bestTerm;
        last = best;
    }
    return acc;
}

template <bool STATS>
__device__ __forceinline__ float wave_light_pdf(const Scene& sc, const LightQuery& q, Counters& cn) {
    const int lane = threadIdx.x & 63;
    const int slot = lane / RT_LIGHT_GROUP, sub = lane % RT_LIGHT_GROUP;
    float result = 0.0f;
    int nonzero = 0;  // nonzero terms of this lane's query so far (over all chunks)
    BoxRay br;
    br.set(q.hitPos, q.outgoing);
    for (int base = 0; base < sc.n_lights; base += RT_LIGHT_CHUNK) {
        const int n_groups = min(64, (sc.n_lights - base + RT_LIGHT_GROUP - 1) / RT_LIGHT_GROUP);
        const float4* B = sc.light_boxes + 2 * (base / RT_LIGHT_GROUP);
        unsigned long long mine = 0;
        if (q.pending) {
            if (fmaxf(fabsf(q.hitPos.x), fmaxf(fabsf(q.hitPos.y), fabsf(q.hitPos.z))) <= sc.cull_far) {
                for (int g = 0; g < n_groups; g++)
                    if (br.may_reach(B[2 * g], B[2 * g + 1])) mine |= 1ull << g;
            } else {  // too far out for the padding of the boxes to cover the rounding of the slab test: every group
                mine = n_groups >= 64 ? ~0ull : (1ull << n_groups) - 1;
            }
        }
        for (unsigned long long m = __ballot(mine != 0); m; m &= m - 1) {
            const int ow = __ffsll((long long)m) - 1;
            const f3 hp = mk(lane_f(q.hitPos.x, ow), lane_f(q.hitPos.y, ow), lane_f(q.hitPos.z, ow));
            const f3 og = mk(lane_f(q.outgoing.x, ow), lane_f(q.outgoing.y, ow), lane_f(q.outgoing.z, ow));
            unsigned long long gm = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), ow) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane((int)mine, ow);
            float acc = lane_f(result, ow);
            int cnt = __builtin_amdgcn_readlane(nonzero, ow);
            while (gm) {
                int g[64 / RT_LIGHT_GROUP];
#pragma unroll
                for (int k = 0; k < 64 / RT_LIGHT_GROUP; k++) {
                    g[k] = gm ? __ffsll((long long)gm) - 1 : -1;
                    gm &= gm - 1;
                }
                int gs = g[0];
#pragma unroll
                for (int k = 1; k < 64 / RT_LIGHT_GROUP; k++) gs = slot == k ? g[k] : gs;
                const int i = base + gs * RT_LIGHT_GROUP + sub;
                float term = 0.0f;
                if (gs >= 0 && i < sc.n_lights) {
                    if (STATS) cn.dev_lights++;
                    // (group gs, light sub: plane p of the group is 16 RT_LIGHT_GROUP contiguous bytes)
                    const float4* L = sc.lights + ((size_t)(base / RT_LIGHT_GROUP + gs) * RT_LIGHT_PLANES) * RT_LIGHT_GROUP + sub;
                    const float4 l0 = L[0], l1 = L[RT_LIGHT_GROUP], l2 = L[2 * RT_LIGHT_GROUP], l3 = L[3 * RT_LIGHT_GROUP];
                    term = itri_pdf(hp, og, mk(l0.x, l0.y, l0.z), mk(l0.w, l1.x, l1.y), mk(l1.z, l1.w, l2.x), mk(l2.y, l2.z, l2.w),
                                    l3.x) / (float)sc.n_lights;
                }
                for (unsigned long long h = __ballot(term != 0.0f); h; h &= h - 1) {
                    cnt++;
                    acc += lane_f(term, __ffsll((long long)h) - 1);  // (meaningful while cnt <= 2; beyond, the query is summed again below)
                }
            }
            if (lane == ow) { result = acc; nonzero = cnt; }
        }
    }
    // three or more nonzero terms: their sum depends on the order; form it in the reference's
    for (unsigned long long m = __ballot(q.pending && nonzero > 2); m; m &= m - 1) {
        const int ow = __ffsll((long long)m) - 1;
        const f3 hp = mk(lane_f(q.hitPos.x, ow), lane_f(q.hitPos.y, ow), lane_f(q.hitPos.z, ow));
        const f3 og = mk(lane_f(q.outgoing.x, ow), lane_f(q.outgoing.y, ow), lane_f(q.outgoing.z, ow));
        const float exact = wave_light_pdf_ordered<STATS>(sc, hp, og, cn);
        if (lane == ow) result = exact;
    }
    if (STATS && q.pending) cn.lights += (uint32_t)sc.n_lights;
    return result;
}

// wave_light_pdf in two phases (rt_render_paths_kernel).  Most lights of a marked group are missed by the query's ray, and
// the triangle test with its correctly rounded divisions is what the loop spends its instructions on.  Phase one: the 64
// lanes test the ray against the BOUNDING SPHERES of 64 lights (one 16-byte load, a dozen instructions; the radius is
// padded and the comparison has slack for its own cancellation, so a light the ray can hit is never dropped) and the
// lights that pass are queued, (owner lane, light), in `queue` (RT_LQ_CAP words of LDS of this wave).  Phase two: whenever
// 64 are queued, and at the end, each lane runs the triangle test of one queued pair -- whatever query it belongs to --
// and a nonzero term goes to the registers of the lane that owns the query.  Same terms, same sums as wave_light_pdf.
#define RT_LQ_CAP 128
template <bool STATS>
__device__ __forceinline__ float wave_light_pdf_q(const Scene& sc, const LightQuery& q, Counters& cn, volatile unsigned* queue) {
    const int lane = threadIdx.x & 63;
    const int slot = lane / RT_LIGHT_GROUP, sub = lane % RT_LIGHT_GROUP;
    float result = 0.0f;
    int nonzero = 0;  // nonzero terms of this lane's query so far
    int count = 0;    // queued pairs (wave-uniform), < 64 between appends
    BoxRay br;
    br.set(q.hitPos, q.outgoing);
    auto run_queued = [&](int n) {  // the first n <= 64 queued pairs, one per lane
        const unsigned c = lane < n ? queue[lane] : ((unsigned)lane << 24);
        const int ow = (int)(c >> 24), i = (int)(c & 0xffffffu);
        const f3 hp = mk(__shfl(q.hitPos.x, ow, 64), __shfl(q.hitPos.y, ow, 64), __shfl(q.hitPos.z, ow, 64));
        const f3 og = mk(__shfl(q.outgoing.x, ow, 64), __shfl(q.outgoing.y, ow, 64), __shfl(q.outgoing.z, ow, 64));
        float term = 0.0f;
        if (lane < n) {
            if (STATS) cn.dev_lights++;
            const float4 l0 = sc.lights[RT_LIGHT_F4(i, 0)], l1 = sc.lights[RT_LIGHT_F4(i, 1)], l2 = sc.lights[RT_LIGHT_F4(i, 2)],
                         l3 = sc.lights[RT_LIGHT_F4(i, 3)];
            term = itri_pdf(hp, og, mk(l0.x, l0.y, l0.z), mk(l0.w, l1.x, l1.y), mk(l1.z, l1.w, l2.x), mk(l2.y, l2.z, l2.w), l3.x) /
                   (float)sc.n_lights;
        }
        for (unsigned long long h = __ballot(term != 0.0f); h; h &= h - 1) {
            const int l = __ffsll((long long)h) - 1;
            const int o = __builtin_amdgcn_readlane(ow, l);
            const float tv = lane_f(term, l);
            if (lane == o) {  // (meaningful while nonzero <= 2: those commute; beyond, the query is summed again below)
                result += tv;
                nonzero++;
            }
        }
    };
    for (int base = 0; base < sc.n_lights; base += RT_LIGHT_CHUNK) {
        const int n_groups = min(64, (sc.n_lights - base + RT_LIGHT_GROUP - 1) / RT_LIGHT_GROUP);
        const float4* B = sc.light_boxes + 2 * (base / RT_LIGHT_GROUP);
        unsigned long long mine = 0;
        if (q.pending) {
            if (fmaxf(fabsf(q.hitPos.x), fmaxf(fabsf(q.hitPos.y), fabsf(q.hitPos.z))) <= sc.cull_far) {
                for (int g = 0; g < n_groups; g++)
                    if (br.may_reach(B[2 * g], B[2 * g + 1])) mine |= 1ull << g;
            } else {  // too far out for the padding of the boxes to cover the rounding of the slab test: every group
                mine = n_groups >= 64 ? ~0ull : (1ull << n_groups) - 1;
            }
        }
        for (unsigned long long m = __ballot(mine != 0); m; m &= m - 1) {
            const int ow = __ffsll((long long)m) - 1;
            const f3 hp = mk(lane_f(q.hitPos.x, ow), lane_f(q.hitPos.y, ow), lane_f(q.hitPos.z, ow));
            const f3 og = mk(lane_f(q.outgoing.x, ow), lane_f(q.outgoing.y, ow), lane_f(q.outgoing.z, ow));
            const float og2 = len2(og);
            // (from beyond cull_far the spheres' padding does not cover the rounding either: every light of the group passes)
            const bool far = !(fmaxf(fabsf(hp.x), fmaxf(fabsf(hp.y), fabsf(hp.z))) <= sc.cull_far);
            unsigned long long gm = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), ow) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane((int)mine, ow);
            while (gm) {
                int g[64 / RT_LIGHT_GROUP];
#pragma unroll
                for (int k = 0; k < 64 / RT_LIGHT_GROUP; k++) {
                    g[k] = gm ? __ffsll((long long)gm) - 1 : -1;
                    gm &= gm - 1;
                }
                int gs = g[0];
#pragma unroll
                for (int k = 1; k < 64 / RT_LIGHT_GROUP; k++) gs = slot == k ? g[k] : gs;
                const int i = base + gs * RT_LIGHT_GROUP + sub;
                bool pass = false;
                if (gs >= 0 && i < sc.n_lights) {
                    const float4 sp = sc.lights[((size_t)(base / RT_LIGHT_GROUP + gs) * RT_LIGHT_PLANES + 5) * RT_LIGHT_GROUP + sub];
                    const f3 v = mk(sp.x, sp.y, sp.z) - hp;
                    const float b = dot(v, og), v2 = len2(v), r2 = sp.w * sp.w;
                    // squared distance of the centre from the ray's LINE, times |og|^2, against the radius.  The slack covers the
                    // rounding of v2 og2 - b^2 itself: v2 and og2 are off by 1.8e-7 each and their product by 6e-8 more, b by
                    // 1.8e-7 |v| |og| and so b^2 by 3.6e-7 v2 og2 -- 8.4e-7 v2 og2 at worst, which for a small light hundreds of
                    // radii away is more than the radius' 0.1 % of padding.  Behind the origin (b < 0) only a sphere that holds
                    // the origin can be hit.
                    pass = far || ((v2 * og2 - b * b <= r2 * og2 + 2e-6f * v2 * og2) && (b >= 0.0f || v2 <= r2 * 1.000001f));
                }
                const unsigned long long pm = __ballot(pass);
                if (pm) {
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pm, 0u));
                    if (pass) queue[count + (int)rank] = ((unsigned)ow << 24) | (unsigned)i;
                    count += __popcll(pm);
                    if (count >= 64) {
                        run_queued(64);
                        const unsigned moved = lane < count - 64 ? queue[64 + lane] : 0u;
                        if (lane < count - 64) queue[lane] = moved;
                        count -= 64;
                    }
                }
            }
        }
    }
    if (count) run_queued(count);
    // three or more nonzero terms: their sum depends on the order; form it in the reference's
    for (unsigned long long m = __ballot(q.pending && nonzero > 2); m; m &= m - 1) {
        const int ow = __ffsll((long long)m) - 1;
        const f3 hp = mk(lane_f(q.hitPos.x, ow), lane_f(q.hitPos.y, ow), lane_f(q.hitPos.z, ow));
        const f3 og = mk(lane_f(q.outgoing.x, ow), lane_f(q.outgoing.y, ow), lane_f(q.outgoing.z, ow));
        const float exact = wave_light_pdf_ordered<STATS>(sc, hp, og, cn);
        if (lane == ow) result = exact;
    }
    if (STATS && q.pending) cn.lights += (uint32_t)sc.n_lights;
    return result;
}

// Sphere::hitSphere, sphere.h:77-88
template <bool STATS>
__device__ __forceinline__ void shade_sphere(const Scene& sc, PathState& ray, f3 normal, f3 hitPos, int mat, const Rng& rng,
                                             uint32_t bb, Counters& cn) {
    const float u = pm::atan2(ray.d.z, ray.d.x) / (2.0f * 3.14159f) + 0.5f;
    const float v = -pm::asin(ray.d.y) / (2.0f * 3.14159f) + 0.5f;
    f3 clr = mk(1.0f, 0.0f, 1.0f);
    if (mat >= 0 && mat < sc.n_mats) {
        const float4* m = sc.mats + 3 * (size_t)mat;
        const float4 m1 = m[1], m2 = m[2];
        const int off0 = __float_as_int(m1.w), tw = __float_as_int(m2.z), th = __float_as_int(m2.w);
        if (off0 >= 0) {
            const float px = u * (float)tw;
            const float py = v * (float)th;
            clr = fetch_texel(sc, (long long)off0 + 3LL * (int)px + 3LL * tw * (int)py);
            if (STATS) cn.texels++;
        }
    }
    ray.thr = ray.thr * clr;
    ray.o = hitPos;
    const uint4 A = rng.block(bb);
    ray.d = normalize(normal + rand_sphere_vec(u01(A.x), u01(A.y)));
    ray.o = ray.o + ray.d * 1e-5f;
}

// checkScatteredVolume, main.cu:127-141 (ENABLE_VOLUME_SCATTERING).  Draws: Philox block
// (RT_RNG_DIM_VOLUME0 + 4 * bounce) >> 2 of the path's stream: x free flight, y z Vec3::randSphere.
__device__ __forceinline__ bool check_scattered_volume(PathState& ray, f3 hitPos, const Rng& rng, uint32_t bounce) {
    if (ray.volDensity <= 0.0f) return false;
    const uint4 V = rng.block((RT_RNG_DIM_VOLUME0 + RT_RNG_DIMS_PER_VOLUME * bounce) >> 2);
    const float travel_dist = -pm::log(u01(V.x)) / ray.volDensity;
    const float dist_to_target = sqrtf(len2(hitPos - ray.o));
    if (travel_dist > dist_to_target) return false;  // no scattering
    ray.o = ray.o + ray.d * travel_dist;
    ray.d = rand_sphere_vec3(u01(V.y), u01(V.z));  // Vec3::randSphere, not Material::randSphereVec
    ray.thr = ray.thr * ray.volColor;
    return true;
}

// marchRay, main.cu:143-153 (ENABLE_RAY_MARCHING)
__device__ __forceinline__ void march_ray(f3& pos, f3& dir, f3& vel) {
    const float dt = 1.0f;
    const float G = 10000000.0f;
    const float r2 = len2(pos) + 1e-5f;
    const float s = r2 * sqrtf(r2);
    const f3 acc = mk(pos.x / s, pos.y / s, pos.z / s) * -1.0f;
    vel = vel + acc * dt * G;
    const f3 last = pos;
    pos = pos + vel * dt;
    dir = normalize(pos - last);
}

// env lookup, main.cu:376-382 -> float index of the texel
__device__ __forceinline__ long long env_index(f3 d, int W, int H) {
    float bx = pm::atan2(d.z, d.x) / (2.0f * 3.14159f) + 0.5f;
    float by = -pm::asin(d.y) / (2.0f * 3.14159f) + 0.5f;
    bx *= (float)W;
    by *= (float)H;
    int ix = (int)bx, iy = (int)by;
    // `% W`, `% H` (main.cu:380-381) change nothing for 0 <= i < W: the two integer divisions run only if some lane
    // of the wave is outside (atan2 = pi exactly, NaN directions)
    if (__any((unsigned)ix >= (unsigned)W || (unsigned)iy >= (unsigned)H)) {
        ix %= W;
        iy %= H;
    }
    return ((long long)iy * W + ix) * 3;
}

// toneMap, main.cu:23-38
__device__ __forceinline__ f3 tone_map(f3 c) {
    const float exposure_factor = powf(2.0f, -0.8f);
    c.x = c.x * (c.x * 2.51f + 0.03f) / (c.x * (c.x * 2.43f + 0.59f) + 0.14f) * exposure_factor;
    c.y = c.y * (c.y * 2.51f + 0.03f) / (c.y * (c.y * 2.43f + 0.59f) + 0.14f) * exposure_factor;
    c.z = c.z * (c.z * 2.51f + 0.03f) / (c.z * (c.z * 2.43f + 0.59f) + 0.14f) * exposure_factor;
    const float inv_gamma = 1.0f / 2.2f;
    return mk(powf(c.x, inv_gamma), powf(c.y, inv_gamma), powf(c.z, inv_gamma));
}

// ------------------------------------------------------------------ camera (main.cu:173-217)
__device__ __forceinline__ void camera_ray(const Params& P, int x, int y, const Rng& rng, f3& origin, f3& dir) {
    const float sx = (float)x / (float)P.width - 0.5f;
    const float sy = -(float)y / (float)P.height + 0.5f;
    const uint4 C = rng.block(RT_RNG_DIM_CAMERA >> 2);  // dims 0 time, 1-2 anti-alias jitter
    const float t = u01(C.x);
    const f3 pos0 = ld3(P.cam_pos);
    const f3 camPos = lerp3(pos0, pos0 + ld3(P.cam_vel), t);
    f3 up, right, forward;
    if (P.static_basis) {
        up = ld3(P.up);
        right = ld3(P.right);
        forward = ld3(P.forward);
    } else {  // Matrix::rotationMatrix, matrix.h:14-37 at the lerped Euler angles
        const f3 rot0 = ld3(P.cam_rot);
        const f3 e = lerp3(rot0, rot0 + ld3(P.cam_angvel), t);
        float cX, sX, cY, sY, cZ, sZ;
        pm::sincos(e.x, &sX, &cX);
        pm::sincos(e.y, &sY, &cY);
        pm::sincos(e.z, &sZ, &cZ);
        const float m0 = cY * cZ, m1 = -cY * sZ;
        const float m3 = cX * sZ + sX * sY * cZ, m4 = cX * cZ - sX * sY * sZ;
        const float m6 = sX * sZ - cX * sY * cZ, m7 = sX * cZ + cX * sY * sZ;
        const float m2 = sY, m5 = -sX * cY, m8 = cX * cY;
        // rot * (0,1,0) and rot * (1,0,0) with the full three-term sums of matrix.h:48-54
        up = normalize_keep(mk(m0 * 0.0f + m1 * 1.0f + m2 * 0.0f, m3 * 0.0f + m4 * 1.0f + m5 * 0.0f, m6 * 0.0f + m7 * 1.0f + m8 * 0.0f));
        right = normalize_keep(mk(m0 * 1.0f + m1 * 0.0f + m2 * 0.0f, m3 * 1.0f + m4 * 0.0f + m5 * 0.0f, m6 * 1.0f + m7 * 0.0f + m8 * 0.0f));
        forward = normalize(cross(up, right));
    }
    float aa1 = 0.0f, aa2 = 0.0f;
    if (P.flags & RT_F_ANTIALIAS) {
        aa1 = u01(C.y) - 0.5f;
        aa2 = u01(C.z) - 0.5f;
    }
    const f3 s = forward * P.focal + right * (sx + aa1 / (float)P.width) + up * (sy + aa2 / (float)P.height);
    dir = normalize(s);
    origin = camPos;
    if (P.aperture > 1e-5f && (P.flags & RT_F_DEFOCUS)) {  // randomPointInDisk, main.cu:119-125
        const uint4 Ln = rng.block(RT_RNG_DIM_LENS >> 2);  // dims 4 theta, 5 radius
        const float theta = 2.0f * 3.14159f * u01(Ln.x);
        const float r = sqrtf(u01(Ln.y)) * (P.aperture * 0.5f);
        float st, ct;
        pm::sincos(theta, &st, &ct);
        const float lx = r * ct;
        const float ly = r * st;
        const f3 lensOffset = up * ly + right * lx;
        const f3 focusPoint = camPos + dir * P.focus_dist;
        origin = camPos + lensOffset;
        dir = normalize(focusPoint - origin);
    }
}

}  // namespace rtd
