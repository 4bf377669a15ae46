// rt_oracle.cpp -- CPU restatement of the reference's trace/shade path.
//
// TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may build, load or call this file.  The product
// (ray-tracer_amd/) never links or calls it and has no CPU fallback.
//
// PARITY UNPINNED.  The reference (EricWittrock/Ray-Tracer) has no tests, no
// golden vectors and no published outputs, and none of its sources compile in
// this image: every header pulls <curand_kernel.h> (src/vec3.h:5) and main.cu
// needs nvcc; writing stand-in headers is not allowed.  So this file is a
// line-by-line restatement of the algorithm from the reference text, checked
// against hand-derived known answers (tests/test_oracle_known_answers.py) and
// against its own brute-force path (ENABLE_BVH=false, main.cu:285-301), not
// against reference-produced vectors.
//
// What is restated (reference file:line at each function):
//   render2                   src/main.cu:155-407
//   rayTriangleIntersect      src/main.cu:41-70
//   AABBIntersectDistance     src/main.cu:73-117
//   randomPointInDisk         src/main.cu:119-125
//   toneMap                   src/main.cu:23-38
//   Material::reflect & co.   src/material.h:41-352
//   ImportanceTri             src/importanceTri.h:7-91
//   RayAttractor              src/rayAttractor.h:11-61
//   Sphere                    src/sphere.h:30-88
//   Floor                     src/floor.h:8-21
//   Vec3 / Matrix / Ray       src/vec3.h, src/matrix.h, src/ray.h
// The RNG is NOT cuRAND (third party, not in the tree): it is the
// counter-based generator specified in include/rt.h, as north_star asks.
//
// Build: g++ -O2 -ffp-contract=off (no fast-math), see oracle/Makefile.

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rt.h"
#include "rt_pmath_oracle.h"

namespace {

// ------------------------------------------------------------------ Vec3 (vec3.h:14-197)
struct Vec3 {
    float x, y, z;
    Vec3() : x(0.f), y(0.f), z(0.f) {}
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float dot(const Vec3& v) const { return x * v.x + y * v.y + z * v.z; }
    Vec3 cross(const Vec3& v) const { return Vec3(y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x); }
    float lengthSqr() const { return x * x + y * y + z * z; }
    float length() const { return sqrtf(lengthSqr()); }
    void reflect(const Vec3& n) {  // vec3.h:42-47
        float d = 2.0f * dot(n);
        x = x - d * n.x;
        y = y - d * n.y;
        z = z - d * n.z;
    }
    void refract(const Vec3& n, float eta) {  // vec3.h:49-62
        float cos_theta = fminf(-this->dot(n), 1.0f);
        Vec3 perp((x + n.x * cos_theta) * eta, (y + n.y * cos_theta) * eta, (z + n.z * cos_theta) * eta);
        float k = -sqrtf(fabsf(1.0f - perp.lengthSqr()));
        Vec3 par(n.x * k, n.y * k, n.z * k);
        x = perp.x + par.x;
        y = perp.y + par.y;
        z = perp.z + par.z;
    }
    Vec3 normalize() const {  // vec3.h:89-95
        float invlen = 1.0f / length();
        if (invlen > 0) return Vec3(x * invlen, y * invlen, z * invlen);
        return Vec3(0, 0, 0);
    }
};
inline void normalize_inplace(Vec3& v) {  // vec3.h:97-105
    float invlen = 1.0f / v.length();
    if (invlen > 0) {
        v.x *= invlen;
        v.y *= invlen;
        v.z *= invlen;
    }
}
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(const Vec3& a, const Vec3& b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline Vec3 operator*(const Vec3& a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator/(const Vec3& a, float s) { return Vec3(a.x / s, a.y / s, a.z / s); }  // vec3.h:195-197
inline Vec3 lerp(const Vec3& a, const Vec3& b, float t) {  // vec3.h:125-131
    return Vec3(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z));
}
inline Vec3 from_color_int(int color) {  // vec3.h:118-123
    float r = static_cast<float>((color >> 16) & 0xFF) / 255.0f;
    float g = static_cast<float>((color >> 8) & 0xFF) / 255.0f;
    float b = static_cast<float>(color & 0xFF) / 255.0f;
    return Vec3(r, g, b);
}

// ------------------------------------------------------------------ Matrix (matrix.h:14-54)
struct Matrix {
    float d[9];
};
inline Matrix rotation_matrix(const Vec3& v) {
    Matrix m;
    float cosX, sinX, cosY, sinY, cosZ, sinZ;
    pm::sincos(v.x, &sinX, &cosX);
    pm::sincos(v.y, &sinY, &cosY);
    pm::sincos(v.z, &sinZ, &cosZ);
    m.d[0] = cosY * cosZ;
    m.d[1] = -cosY * sinZ;
    m.d[2] = sinY;
    m.d[3] = cosX * sinZ + sinX * sinY * cosZ;
    m.d[4] = cosX * cosZ - sinX * sinY * sinZ;
    m.d[5] = -sinX * cosY;
    m.d[6] = sinX * sinZ - cosX * sinY * cosZ;
    m.d[7] = sinX * cosZ + cosX * sinY * sinZ;
    m.d[8] = cosX * cosY;
    return m;
}
inline Vec3 operator*(const Matrix& m, const Vec3& v) {
    Vec3 r;
    r.x = m.d[0] * v.x + m.d[1] * v.y + m.d[2] * v.z;
    r.y = m.d[3] * v.x + m.d[4] * v.y + m.d[5] * v.z;
    r.z = m.d[6] * v.x + m.d[7] * v.y + m.d[8] * v.z;
    return r;
}

// ------------------------------------------------------------------ Ray (ray.h:11-48)
struct Ray {
    Vec3 position, direction, diffuseMultiplier, emission;
    float refractiveIndex, volumeDensity;
    Vec3 volumeColor;
    Vec3 marchVelocity;
    Ray(const Vec3& o, const Vec3& d)
        : position(o), direction(d), diffuseMultiplier(1.f, 1.f, 1.f), emission(0.f, 0.f, 0.f),
          refractiveIndex(1.0f), volumeDensity(0.0f), volumeColor(1.f, 1.f, 1.f) {
        marchVelocity = direction * 0.1f;  // ray.h:30
    }
    void marchForward(float dist) {
        Vec3 s = direction * dist;
        position.x += s.x;
        position.y += s.y;
        position.z += s.z;
    }
};

// ------------------------------------------------------------------ RNG (include/rt.h)
inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct Rng {
    uint32_t seed, pixel, sample;
    uint32_t dim;           // next dimension
    uint32_t cached_block;  // block index held in `w`
    uint32_t w[4];
    Rng(uint32_t s, uint32_t p, uint32_t k) : seed(s), pixel(p), sample(k), dim(0), cached_block(0xffffffffu) {}
    void seek(uint32_t d) { dim = d; }
    uint32_t raw() {
        uint32_t b = dim >> 2;
        if (b != cached_block) {
            uint32_t ctr[4] = {pixel, sample, b, RT_RNG_TAG};
            uint32_t key[2] = {seed, RT_RNG_KEY1};
            philox4x32_10(ctr, key, w);
            cached_block = b;
        }
        return w[(dim++) & 3];
    }
    float uniform() { return (float)((raw() >> 8) + 1u) * (1.0f / 16777216.0f); }  // (0,1]
    float normal() {
        float u1 = uniform();
        float u2 = uniform();
        return sqrtf(-2.0f * pm::log(u1)) * pm::cos(6.2831855f * u2);
    }
};

// ------------------------------------------------------------------ counters
struct Counters {
    uint64_t paths = 0, rays = 0, inner = 0, leaf = 0, tri = 0, hits = 0, texels = 0, lights = 0;
    void add(const Counters& o) {
        paths += o.paths; rays += o.rays; inner += o.inner; leaf += o.leaf;
        tri += o.tri; hits += o.hits; texels += o.texels; lights += o.lights;
    }
};

// ------------------------------------------------------------------ intersection

// rayTriangleIntersect, main.cu:41-70 (twin: ImportanceTri::ray_intersect, importanceTri.h:36-65)
bool ray_triangle(const Vec3& org, const Vec3& dir, const Vec3& v0, const Vec3& v1, const Vec3& v2, Vec3& outPos) {
    const float epsilon = 1e-9f;
    Vec3 edge1 = v1 - v0;
    Vec3 edge2 = v2 - v0;
    Vec3 ray_cross_e2 = dir.cross(edge2);
    float det = edge1.dot(ray_cross_e2);
    if (det > -epsilon && det < epsilon) return false;
    float inv_det = 1.0f / det;
    Vec3 s = org - v0;
    float u = inv_det * s.dot(ray_cross_e2);
    if (u < 0.0f || u > 1.0f) return false;
    Vec3 q = s.cross(edge1);
    float v = inv_det * dir.dot(q);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = inv_det * edge2.dot(q);
    if (t > epsilon) {
        outPos = org + dir * t;
        return true;
    }
    return false;
}

// AABBIntersectDistance, main.cu:73-117
float aabb_distance(const Vec3& o, const Vec3& d, const float* mn, const float* mx) {
    if (o.x >= mn[0] && o.x <= mx[0] && o.y >= mn[1] && o.y <= mx[1] && o.z >= mn[2] && o.z <= mx[2]) return 0.0f;
    float tmin = 0.0f;
    float tmax = 1e13f;
    float invDirX = 1.0f / d.x;
    float invDirY = 1.0f / d.y;
    float invDirZ = 1.0f / d.z;
    float t1, t2, tNear, tFar;
    t1 = (mn[0] - o.x) * invDirX;
    t2 = (mx[0] - o.x) * invDirX;
    tNear = fminf(t1, t2);
    tFar = fmaxf(t1, t2);
    tmin = fmaxf(tmin, tNear);
    tmax = fminf(tmax, tFar);
    t1 = (mn[1] - o.y) * invDirY;
    t2 = (mx[1] - o.y) * invDirY;
    tNear = fminf(t1, t2);
    tFar = fmaxf(t1, t2);
    tmin = fmaxf(tmin, tNear);
    tmax = fminf(tmax, tFar);
    t1 = (mn[2] - o.z) * invDirZ;
    t2 = (mx[2] - o.z) * invDirZ;
    tNear = fminf(t1, t2);
    tFar = fmaxf(t1, t2);
    tmin = fmaxf(tmin, tNear);
    tmax = fminf(tmax, tFar);
    if (tmax < tmin) return 1e13f;
    if (tmin < 0.0f) return 1e13f;
    return tmin;
}

// Sphere::intersectSphere, sphere.h:30-46
bool intersect_sphere(const Vec3& org, const Vec3& dir, const Vec3& c, float radius, Vec3& out_pos, Vec3& out_normal) {
    Vec3 fromSphere = org - c;
    float b = 2.0f * fromSphere.dot(dir);
    if (b > 0) return false;
    float cc = fromSphere.lengthSqr() - radius * radius;
    float d = b * b - 4.0f * cc;
    if (d < 0) return false;
    // `sqrt(d)` on a float: float overload under nvcc; kept in float.
    float t = (-b - sqrtf(d)) / 2.0f;
    out_pos = org + dir * t;
    out_normal = (out_pos - c).normalize();
    return true;
}

// Floor::intersection, floor.h:8-21
bool intersect_floor(const Vec3& org, const Vec3& dir, float elevation, Vec3& pos, Vec3& normal) {
    if (dir.y >= 0) return false;
    float t = (elevation - org.y) / dir.y;
    if (t < 0) return false;
    pos = org + dir * t;
    normal = Vec3(0, 1, 0);
    return true;
}

// ------------------------------------------------------------------ ImportanceTri (importanceTri.h)
struct ImportanceTri {
    Vec3 p1, p2, p3, normal;
    float area;
    ImportanceTri(const Vec3& a, const Vec3& b, const Vec3& c) : p1(a), p2(b), p3(c) {
        normal = (p2 - p1).cross(p3 - p1).normalize();
        area = 0.5f * (p2 - p1).cross(p3 - p1).length();
    }
    float pdf_value(const Vec3& origin, const Vec3& direction) const {  // :67-76
        Vec3 hit;
        if (!ray_triangle(origin, direction, p1, p2, p3, hit)) return 0.0f;
        Vec3 to = hit - origin;
        float distance_squared = to.lengthSqr();
        float cos_theta = fabsf(to.normalize().dot(normal));
        return distance_squared / (cos_theta * area);
    }
    Vec3 generate_random(const Vec3& origin, float r1, float r2) const {  // :79-90
        float su = sqrtf(r1);
        float b0 = 1.0f - su;
        float b1 = r2 * su;
        float b2 = 1.0f - b0 - b1;
        Vec3 random_point = p1 * b0 + p2 * b1 + p3 * b2;
        return (random_point - origin).normalize();
    }
};

// ------------------------------------------------------------------ RayAttractor (rayAttractor.h)
struct RayAttractor {
    Vec3 pos, normal, u, v;
    float area;
    RayAttractor(const Vec3& p, const Vec3& uu, const Vec3& vv) {  // :13-19
        pos = p;
        area = uu.cross(vv).length() * 0.25f;
        u = uu.normalize();
        v = vv.normalize();
        normal = u.cross(v).normalize();
    }
    bool ray_intersect(const Vec3& origin, const Vec3& direction, Vec3& hit) const {  // :22-37
        float denom = normal.dot(direction);
        if (fabs((double)denom) < 1e-12) return false;
        float t = (pos - origin).dot(normal) / denom;
        if (t <= 0) return false;
        hit = origin + direction * t;
        float u_pos = (hit - pos).dot(u);
        float v_pos = (hit - pos).dot(v);
        return u_pos >= -0.5f && u_pos <= 0.5f && v_pos >= -0.5f && v_pos <= 0.5f;
    }
    float pdf_value(const Vec3& origin, const Vec3& direction) const {  // :39-48
        Vec3 hit;
        if (!ray_intersect(origin, direction, hit)) return 0.0f;
        Vec3 to = hit - origin;
        float distance_squared = to.lengthSqr();
        float cos_theta = fabsf(to.normalize().dot(normal));
        return distance_squared / (cos_theta * area);
    }
    Vec3 generate_random(const Vec3& origin, float r1, float r2) const {  // :50-54 (not normalised)
        return pos + u * (r1 - 0.5f) + v * (r2 - 0.5f) - origin;
    }
};

// ------------------------------------------------------------------ scene view
struct Scene {
    const float* spheres = nullptr;  // 5 floats each
    int numSpheres = 0;
    const float* tris = nullptr;  // first triangle record
    size_t nTriFloats = 0;
    const rt_bvh_node* nodes = nullptr;
    int nNodes = 0;
    const rt_material* mats = nullptr;
    size_t nMats = 0;
    const float* tex = nullptr;
    size_t nTex = 0;
    const int32_t* lights = nullptr;
    int nLights = 0;
};

Scene view_of(const rt_scene_desc* d) {
    Scene s;
    const size_t prefix = d->n_spheres > 0 ? 1 + 5 * (size_t)d->n_spheres : 0;
    s.spheres = d->n_spheres > 0 ? d->tris + 1 : nullptr;
    s.numSpheres = d->n_spheres;
    s.tris = d->tris + prefix;
    s.nTriFloats = d->n_tri_floats;
    s.nodes = d->nodes;
    s.nNodes = d->n_nodes;
    s.mats = d->mats;
    s.nMats = d->n_mats;
    s.tex = d->textures;
    s.nTex = d->n_tex_floats;
    s.lights = d->light_tris;
    s.nLights = d->n_light_tris;
    return s;
}

// Texel fetch with the reference's unclamped index arithmetic; only the final
// index is clamped into the pool so that out-of-range uv cannot fault
// (deviation documented in DESIGN.md; the product clamps identically).
inline Vec3 fetch_texel(const Scene& sc, long long i) {
    if (sc.nTex < 3) return Vec3(0.f, 0.f, 0.f);
    long long hi = (long long)sc.nTex - 3;
    if (i < 0) i = 0;
    if (i > hi) i = hi;
    return Vec3(sc.tex[i], sc.tex[i + 1], sc.tex[i + 2]);
}

// ------------------------------------------------------------------ closest hit (main.cu:221-301)
struct Hit {
    int tri = -1;  // float offset of the record, main.cu:261
    float minDistSqr = 1e12f;
    Vec3 pos;
};

void closest_hit(const Scene& sc, const Vec3& org, const Vec3& dir, bool use_bvh, Hit& h, Counters& c,
                 int* max_stack = nullptr, float minDistSqr0 = 1e12f) {
    h.tri = -1;
    h.minDistSqr = minDistSqr0;  // 1e12, or |marchVelocity|^2 under ENABLE_RAY_MARCHING (main.cu:223-227)
    c.rays++;
    if (use_bvh) {
        // The reference's stack has BVH_DEPTH (16) slots although a depth-16 tree
        // needs 17 (latent overflow); sized generously here, results identical
        // wherever the reference is defined.
        int stack[64];
        int sp = 0;
        if (sc.nNodes > 0 &&
            (double)aabb_distance(org, dir, sc.nodes[0].bbox_min, sc.nodes[0].bbox_max) < 1e12)  // main.cu:234
            stack[sp++] = 0;
        while (sp > 0) {
            const rt_bvh_node& node = sc.nodes[stack[--sp]];
            if (node.child_a == -1) {
                c.leaf++;
                int start = node.tri_offs;
                int end = start + node.tri_len;
                for (int i = start; i < end; i += 25) {
                    const float* t = sc.tris + i;
                    Vec3 v0(t[0], t[1], t[2]), v1(t[3], t[4], t[5]), v2(t[6], t[7], t[8]);
                    c.tri++;
                    Vec3 pos;
                    if (ray_triangle(org, dir, v0, v1, v2, pos)) {
                        float newDistSqr = (pos - org).lengthSqr();
                        if (newDistSqr < h.minDistSqr) {
                            h.minDistSqr = newDistSqr;
                            h.pos = pos;
                            h.tri = i;
                        }
                    }
                }
            } else {
                c.inner++;
                const rt_bvh_node& A = sc.nodes[node.child_a];
                const rt_bvh_node& B = sc.nodes[node.child_b];
                float distAsqr = aabb_distance(org, dir, A.bbox_min, A.bbox_max);
                float distBsqr = aabb_distance(org, dir, B.bbox_min, B.bbox_max);
                distAsqr = distAsqr * distAsqr;
                distBsqr = distBsqr * distBsqr;
                if (distAsqr < distBsqr) {
                    if (distBsqr < h.minDistSqr) stack[sp++] = node.child_b;
                    if (distAsqr < h.minDistSqr) stack[sp++] = node.child_a;
                } else {
                    if (distAsqr < h.minDistSqr) stack[sp++] = node.child_a;
                    if (distBsqr < h.minDistSqr) stack[sp++] = node.child_b;
                }
                if (max_stack && sp > *max_stack) *max_stack = sp;
            }
        }
    } else {  // main.cu:285-301
        for (size_t i = 0; i < sc.nTriFloats; i += 25) {
            const float* t = sc.tris + i;
            Vec3 v0(t[0], t[1], t[2]), v1(t[3], t[4], t[5]), v2(t[6], t[7], t[8]);
            c.tri++;
            Vec3 pos;
            if (ray_triangle(org, dir, v0, v1, v2, pos)) {
                float newDistSqr = (pos - org).lengthSqr();
                if (newDistSqr < h.minDistSqr) {
                    h.minDistSqr = newDistSqr;
                    h.pos = pos;
                    h.tri = (int)i;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ Material (material.h)

Vec3 barycentric(const Vec3& a, const Vec3& b, const Vec3& c, const Vec3& pos) {  // :110-124
    Vec3 v0 = b - a;
    Vec3 v1 = c - a;
    Vec3 v2 = pos - a;
    float d00 = v0.dot(v0);
    float d01 = v0.dot(v1);
    float d11 = v1.dot(v1);
    float d20 = v2.dot(v0);
    float d21 = v2.dot(v1);
    float denom = d00 * d11 - d01 * d01;
    float v = (d11 * d20 - d01 * d21) / denom;
    float w = (d00 * d21 - d01 * d20) / denom;
    float u = 1.0f - v - w;
    return Vec3(u, v, w);
}

Vec3 shading_normal(const float* t, const Vec3& hitPos, Vec3* bary_out = nullptr) {  // :44-52, :126-128
    Vec3 v0(t[0], t[1], t[2]), v1(t[3], t[4], t[5]), v2(t[6], t[7], t[8]);
    Vec3 n0(t[9], t[10], t[11]), n1(t[12], t[13], t[14]), n2(t[15], t[16], t[17]);
    Vec3 bary = barycentric(v0, v1, v2, hitPos);
    if (bary_out) *bary_out = bary;
    return (n0 * bary.x + n1 * bary.y + n2 * bary.z).normalize();
}

Vec3 rand_sphere_vec(Rng& rng) {  // :130-140
    float r1 = rng.uniform();
    float r2 = rng.uniform();
    float theta = r1 * 2.0f * 3.14159f;
    float phi = pm::acos(2.0f * r2 - 1.0f);
    float x = pm::sin(phi) * pm::cos(theta);
    float y = pm::sin(phi) * pm::sin(theta);
    float z = pm::cos(phi);
    return Vec3(x, y, z);
}

// Vec3::randSphere, vec3.h:133-142: as above except that theta's pi is the double literal 3.14159 (vec3.h:136), so
// the product is formed in double and rounded to float.  Called by checkScatteredVolume only (main.cu:137).
Vec3 vec3_rand_sphere(Rng& rng) {
    float r1 = rng.uniform();
    float r2 = rng.uniform();
    float theta = r1 * 2.0f * 3.14159;
    float phi = pm::acos(2.0f * r2 - 1.0f);
    float x = pm::sin(phi) * pm::cos(theta);
    float y = pm::sin(phi) * pm::sin(theta);
    float z = pm::cos(phi);
    return Vec3(x, y, z);
}

Vec3 rand_cosine_hemisphere(const Vec3& normal, Rng& rng) {  // :152-161
    Vec3 s = rand_sphere_vec(rng);
    float cos_theta = s.dot(normal);
    if (cos_theta > 0.0f) return s * sqrtf(cos_theta);
    return s * -1.0f * sqrtf(-cos_theta);
}

Vec3 texture_color(const Scene& sc, const rt_material& m, float u, float v, int image_id, Counters& c) {  // :163-175
    float px = u * m.image_width;
    float py = v * m.image_height;
    int offset = m.image_offset[image_id];
    if (offset < 0) return Vec3(1.0f, 0.0f, 1.0f);
    long long i = (long long)offset + 3LL * (int)px + 3LL * m.image_width * (int)py;
    c.texels++;
    return fetch_texel(sc, i);
}

// Material::reflect, material.h:41-106.  Returns true if the path terminates.
bool material_reflect(const Scene& sc, const rt_material& m, Ray& ray, const Vec3& hitPos, int hitTri, Rng& rng,
                      Counters& c) {
    const float* t = sc.tris + hitTri;
    Vec3 bary;
    Vec3 normal = shading_normal(t, hitPos, &bary);
    float u = bary.x * t[18] + bary.y * t[20] + bary.z * t[22];
    float v = bary.x * t[19] + bary.y * t[21] + bary.z * t[23];
    Vec3 clr = from_color_int(m.color);
    Vec3 clr0 = clr;
    if (m.image_offset[0] >= 0) clr0 = texture_color(sc, m, u, v, 0, c);
    if (m.image_offset[1] >= 0) (void)texture_color(sc, m, u, v, 1, c);  // fetched and unused, :70-75
    if (m.image_offset[2] >= 0) (void)texture_color(sc, m, u, v, 2, c);

    switch (m.type) {
        case 0: {  // reflectType0, :182-202
            ray.position = hitPos;
            if (rng.uniform() < m.p1) {
                ray.diffuseMultiplier = ray.diffuseMultiplier * Vec3(0.95f, 0.95f, 0.95f);
                ray.direction.reflect(normal);
            } else {
                ray.diffuseMultiplier = ray.diffuseMultiplier * clr;
                float n0 = rng.normal();
                float n1 = rng.normal();
                float n2 = rng.normal();
                Vec3 randVec(n0, n1, n2);
                ray.direction.reflect((normal + randVec * m.p2).normalize());
            }
            ray.marchForward(1e-5f);
            return false;
        }
        case 1: {  // reflectType1, :205-213 ; terminates, :82-84
            if (ray.direction.dot(normal) > 0.0f) ray.emission = Vec3(0.f, 0.f, 0.f);
            else ray.emission = clr * m.p1;
            return true;
        }
        case 2: {  // reflectType2, :216-254
            ray.position = hitPos;
            float dotNorm = ray.direction.dot(normal);
            bool fromOutside = (dotNorm < 0.0f);
            float oldRI = ray.refractiveIndex;
            float newRI = fromOutside ? oldRI + m.p1 : oldRI - m.p1;
            Vec3 normal2 = fromOutside ? normal : normal * -1.0f;
            float ri = oldRI / newRI;
            float cos_theta = fminf(-ray.direction.dot(normal2), 1.0f);
            float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
            bool will_refract = ri * sin_theta <= 1.0f;
            if (will_refract) {
                float r0 = (1 - ri) / (1 + ri);
                r0 = r0 * r0;
                // std::pow(float, int) promotes to double (C++11); the 5th power is
                // formed by multiplication in double so CPU and GPU agree bit for bit.
                double x = (double)(1 - cos_theta);
                double x2 = x * x;
                r0 = (float)((double)r0 + (double)(1 - r0) * (x2 * x2 * x));
                if (r0 > rng.uniform()) will_refract = false;
            }
            if (will_refract) {
                ray.direction.refract(normal2, ri);
                ray.refractiveIndex = newRI;
            } else {
                // three curand_normal draws are made and discarded here (:244-248);
                // with a counter RNG the bounce's dimensions are simply left unused.
                ray.direction.reflect(normal2);
            }
            ray.marchForward(1e-5f);
            return false;
        }
        case 3: {  // reflectType3, :257-266
            Vec3 normal2 = (normal.dot(ray.direction) < 0.0f) ? normal : (normal * -1.0f);
            ray.diffuseMultiplier = ray.diffuseMultiplier * clr0;
            ray.position = hitPos;
            Vec3 randomDir = rand_sphere_vec(rng);
            ray.direction = (normal2 + randomDir).normalize();
            ray.marchForward(1e-5f);
            return false;
        }
        case 4: {  // reflectType4, :269-277
            ray.diffuseMultiplier = ray.diffuseMultiplier * clr;
            ray.position = hitPos;
            ray.direction.reflect(normal);
            Vec3 p = rand_sphere_vec(rng) * m.p1;
            ray.direction.x += p.x;
            ray.direction.y += p.y;
            ray.direction.z += p.z;
            ray.marchForward(1e-5f);
            return false;
        }
        case 5: {  // reflectType5, :280-336
            Vec3 normal2 = (normal.dot(ray.direction) < 0.0f) ? normal : (normal * -1.0f);
            ray.position = hitPos;
            Vec3 outgoing;
            bool sampledLight = false;
            if (sc.nLights > 0 && rng.uniform() < 0.5f) {
                int it = sc.lights[rng.raw() % (uint32_t)sc.nLights];
                const float* L = sc.tris + it;
                ImportanceTri itri(Vec3(L[0], L[1], L[2]), Vec3(L[3], L[4], L[5]), Vec3(L[6], L[7], L[8]));
                float r1 = rng.uniform();
                float r2 = rng.uniform();
                outgoing = itri.generate_random(hitPos, r1, r2);
                sampledLight = true;
            } else {
                outgoing = rand_cosine_hemisphere(normal2, rng);
            }
            float cos_theta = normal2.dot(outgoing);
            if (cos_theta <= 0.0f) {
                if (sampledLight) {
                    outgoing = rand_cosine_hemisphere(normal2, rng);
                    cos_theta = normal2.dot(outgoing);
                }
            }
            if (cos_theta < 0.0f) cos_theta = 0.0f;
            const float inv_pi = 1.0f / 3.14159f;
            float pdf_cos = cos_theta * inv_pi;
            float pdf_light = 0.0f;
            if (sc.nLights > 0) {
                for (int i = 0; i < sc.nLights; i++) {
                    const float* L = sc.tris + sc.lights[i];
                    ImportanceTri tri(Vec3(L[0], L[1], L[2]), Vec3(L[3], L[4], L[5]), Vec3(L[6], L[7], L[8]));
                    c.lights++;
                    pdf_light += tri.pdf_value(hitPos, outgoing) / static_cast<float>(sc.nLights);
                }
            }
            float pdf = (sc.nLights > 0) ? (0.5f * pdf_cos + 0.5f * pdf_light) : pdf_cos;
            if (pdf <= 1e-12f) {
                ray.diffuseMultiplier = Vec3(0.f, 0.f, 0.f);
                return false;
            }
            ray.diffuseMultiplier = ray.diffuseMultiplier * clr * (pdf_cos / pdf);
            ray.direction = outgoing;
            ray.marchForward(1e-5f);
            return false;
        }
        case 6: {  // reflectType6, :339-352
            bool entering = (normal.dot(ray.direction) < 0.0f);
            if (entering) {
                ray.volumeDensity += m.p1;
                ray.volumeColor = clr;
            } else {
                ray.volumeDensity -= m.p1;
            }
            ray.position = hitPos;
            ray.marchForward(1e-5f);
            return false;
        }
        default:
            ray.emission = Vec3(1.0f, 0.0f, 1.0f);  // "the color of error", :100-102
            return true;
    }
}

// Sphere::hitSphere, sphere.h:77-88 (declared bool, returns nothing: treated as "not terminated")
void hit_sphere(const Scene& sc, const rt_material& m, Ray& ray, const Vec3& normal, const Vec3& hitPos, Rng& rng,
                Counters& c) {
    float u = pm::atan2(ray.direction.z, ray.direction.x) / (2.0f * 3.14159f) + 0.5f;
    float v = -pm::asin(ray.direction.y) / (2.0f * 3.14159f) + 0.5f;
    Vec3 clr = texture_color(sc, m, u, v, 0, c);
    ray.diffuseMultiplier = ray.diffuseMultiplier * clr;
    ray.position = hitPos;
    Vec3 randomDir = rand_sphere_vec(rng);
    ray.direction = (normal + randomDir).normalize();
    ray.marchForward(1e-5f);
}

// env lookup index, main.cu:376-382
inline long long env_index(const Vec3& d, int W, int H) {
    float bx = pm::atan2(d.z, d.x) / (2.0f * 3.14159f) + 0.5f;
    float by = -pm::asin(d.y) / (2.0f * 3.14159f) + 0.5f;
    bx *= W;
    by *= H;
    int ix = static_cast<int>(bx) % W;
    int iy = static_cast<int>(by) % H;
    return ((long long)iy * W + ix) * 3;
}

// toneMap, main.cu:23-38
inline void tone_map(Vec3& color) {
    const float exposure = -0.8f;
    const float exposure_factor = powf(2.0f, exposure);
    color.x = color.x * (color.x * 2.51f + 0.03f) / (color.x * (color.x * 2.43f + 0.59f) + 0.14f) * exposure_factor;
    color.y = color.y * (color.y * 2.51f + 0.03f) / (color.y * (color.y * 2.43f + 0.59f) + 0.14f) * exposure_factor;
    color.z = color.z * (color.z * 2.51f + 0.03f) / (color.z * (color.z * 2.43f + 0.59f) + 0.14f) * exposure_factor;
    const float inv_gamma = 1.0f / 2.2f;
    color.x = powf(color.x, inv_gamma);
    color.y = powf(color.y, inv_gamma);
    color.z = powf(color.z, inv_gamma);
}

// randomPointInDisk, main.cu:119-125
inline Vec3 random_point_in_disk(const Vec3& up, const Vec3& right, float radius, Rng& rng) {
    float theta = 2.0f * 3.14159f * rng.uniform();
    float r = sqrtf(rng.uniform()) * radius;
    float x = r * pm::cos(theta);
    float y = r * pm::sin(theta);
    return up * y + right * x;
}

// camera ray for sample k of pixel (x,y): main.cu:173-217
Ray camera_ray(const rt_render_params& p, int x, int y, Rng& rng) {
    const rt_camera& cam = p.cam;
    const float sx = static_cast<float>(x) / cam.width - 0.5f;
    const float sy = -static_cast<float>(y) / cam.height + 0.5f;
    rng.seek(RT_RNG_DIM_CAMERA);
    float animation_time = rng.uniform();
    Vec3 rot0(cam.rot[0], cam.rot[1], cam.rot[2]), av(cam.ang_vel[0], cam.ang_vel[1], cam.ang_vel[2]);
    Vec3 pos0(cam.pos[0], cam.pos[1], cam.pos[2]), vel(cam.vel[0], cam.vel[1], cam.vel[2]);
    Vec3 eulerRot = lerp(rot0, rot0 + av, animation_time);
    Vec3 camPos = lerp(pos0, pos0 + vel, animation_time);
    Matrix rot = rotation_matrix(eulerRot);
    Vec3 up = rot * Vec3(0.0f, 1.0f, 0.0f);
    Vec3 right = rot * Vec3(1.0f, 0.0f, 0.0f);
    normalize_inplace(up);
    normalize_inplace(right);
    Vec3 forward = up.cross(right).normalize();
    float aa1 = 0.0f, aa2 = 0.0f;
    if (p.flags & RT_F_ANTIALIAS) {
        aa1 = rng.uniform() - 0.5f;
        aa2 = rng.uniform() - 0.5f;
    }
    Vec3 s = forward * cam.focal_length + right * (sx + aa1 / cam.width) + up * (sy + aa2 / cam.height);
    Vec3 dir = s.normalize();
    Vec3 rayOrigin = camPos;
    if (cam.aperture > 1e-5f && (p.flags & RT_F_DEFOCUS)) {
        rng.seek(RT_RNG_DIM_LENS);
        Vec3 lensOffset = random_point_in_disk(up, right, cam.aperture * 0.5f, rng);
        Vec3 focusPoint = camPos + dir * cam.focus_distance;
        rayOrigin = camPos + lensOffset;
        dir = (focusPoint - rayOrigin).normalize();
    }
    return Ray(rayOrigin, dir);
}

// checkScatteredVolume, main.cu:127-141.  Its three draws (free-flight distance, Vec3::randSphere's two) come
// from dimensions RT_RNG_DIM_VOLUME0 + 4 j of the path's stream (rt.h): a bounce's own eight are all taken.
bool check_scattered_volume(Ray& ray, const Vec3& hitPos, Rng& rng, uint32_t bounce) {
    if (ray.volumeDensity <= 0.0f) return false;
    rng.seek(RT_RNG_DIM_VOLUME0 + RT_RNG_DIMS_PER_VOLUME * bounce);
    float r = rng.uniform();
    float travel_dist = -pm::log(r) / ray.volumeDensity;
    float dist_to_target = (hitPos - ray.position).length();
    if (travel_dist > dist_to_target) return false;  // no scattering
    Vec3 step = ray.direction * travel_dist;
    ray.position.x += step.x;
    ray.position.y += step.y;
    ray.position.z += step.z;
    ray.direction = vec3_rand_sphere(rng);  // Vec3::randSphere (double pi), main.cu:137
    ray.diffuseMultiplier = ray.diffuseMultiplier * ray.volumeColor;
    return true;
}

// marchRay, main.cu:143-153
void march_ray(Ray& ray) {
    const float dt = 1.0f;
    const float G = 10000000.0f;
    float r2 = ray.position.lengthSqr() + 1e-5f;
    Vec3 acceleration = ray.position / (r2 * sqrtf(r2)) * -1.0f;
    Vec3 dv = acceleration * dt * G;
    ray.marchVelocity.x += dv.x;
    ray.marchVelocity.y += dv.y;
    ray.marchVelocity.z += dv.z;
    Vec3 lastPos = ray.position;
    Vec3 dp = ray.marchVelocity * dt;
    ray.position.x += dp.x;
    ray.position.y += dp.y;
    ray.position.z += dp.z;
    ray.direction = (ray.position - lastPos).normalize();
}

// One pixel's sum over samples [sample_begin, sample_begin+sample_count): main.cu:176-390
Vec3 render_pixel(const Scene& sc, const rt_render_params& p, int x, int y, Counters& c) {
    Vec3 color(0.f, 0.f, 0.f);
    const bool use_bvh = (p.flags & RT_F_BVH) != 0;
    const bool spheres_on = (p.flags & RT_F_SPHERES) != 0 && sc.numSpheres > 0;
    const bool volume = (p.flags & RT_F_VOLUME) != 0;    // ENABLE_VOLUME_SCATTERING, config.h:9
    const bool marching = (p.flags & RT_F_RAYMARCH) != 0;  // ENABLE_RAY_MARCHING, config.h:14 (max_bounces = RAY_MARCHING_MAX_STEPS)
    const Vec3 bg(p.background[0], p.background[1], p.background[2]);
    const uint32_t pixel = (uint32_t)(x + y * p.cam.width);
    for (int k = 0; k < p.sample_count; k++) {
        Rng rng(p.seed, pixel, (uint32_t)(p.sample_begin + k));
        Ray ray = camera_ray(p, x, y, rng);
        c.paths++;
        for (int j = 0; j < p.max_bounces; j++) {
            rng.seek(RT_RNG_DIM_BOUNCE0 + RT_RNG_DIMS_PER_BOUNCE * (uint32_t)j);
            Hit h;  // (h.pos = 0: `Vec3 hitPos;` of main.cu:221)
            closest_hit(sc, ray.position, ray.direction, use_bvh, h, c, nullptr,
                        marching ? ray.marchVelocity.lengthSqr() : 1e12f);
            if (spheres_on) {  // main.cu:305-346
                bool closestHitIsSphere = false;
                int hitSphereIndex = 0;
                Vec3 sphereHitPos, sphereHitNormal;
                for (int i = 0; i < sc.numSpheres * 5; i += 5) {
                    Vec3 sp(sc.spheres[i], sc.spheres[i + 1], sc.spheres[i + 2]);
                    Vec3 hp, hn;
                    if (intersect_sphere(ray.position, ray.direction, sp, sc.spheres[i + 3], hp, hn)) {
                        float distSqr = (hp - ray.position).lengthSqr();
                        if (distSqr >= h.minDistSqr) continue;
                        closestHitIsSphere = true;
                        h.minDistSqr = distSqr;
                        sphereHitPos = hp;
                        sphereHitNormal = hn;
                        hitSphereIndex = i;
                    }
                }
                if (closestHitIsSphere) {
                    // main.cu:338 hands checkScatteredVolume the TRIANGLE hit position (zero without one), as here
                    if (volume && check_scattered_volume(ray, h.pos, rng, (uint32_t)j)) continue;
                    rng.seek(RT_RNG_DIM_BOUNCE0 + RT_RNG_DIMS_PER_BOUNCE * (uint32_t)j);
                    // Reference indexes materials[hitSphereIndex + 4] (main.cu:341), which is
                    // the wrong array; the material id stored in the sphere record is used.
                    // An id outside the material array (the reference would read out of bounds) shades like a
                    // material without a texture: "the color of error" of Sphere::getTextureColor, sphere.h:64.
                    int mi = static_cast<int>(sc.spheres[hitSphereIndex + 4]);
                    rt_material m;
                    if (mi >= 0 && (size_t)mi < sc.nMats) m = sc.mats[mi];
                    else { memset(&m, 0, sizeof(m)); m.image_offset[0] = m.image_offset[1] = m.image_offset[2] = -1; }
                    c.hits++;
                    hit_sphere(sc, m, ray, sphereHitNormal, sphereHitPos, rng, c);
                    continue;
                }
            }
            if (h.tri >= 0) {
                if (volume && check_scattered_volume(ray, h.pos, rng, (uint32_t)j)) continue;  // main.cu:350-352
                rng.seek(RT_RNG_DIM_BOUNCE0 + RT_RNG_DIMS_PER_BOUNCE * (uint32_t)j);
                c.hits++;
                int hitIndex = static_cast<int>(sc.tris[h.tri + 24]);
                rt_material m;
                if (hitIndex >= 0 && (size_t)hitIndex < sc.nMats) m = sc.mats[hitIndex];
                else { memset(&m, 0, sizeof(m)); m.type = 127; }  // out-of-range id -> error colour
                bool terminated = material_reflect(sc, m, ray, h.pos, h.tri, rng, c);
                if (terminated) {
                    Vec3 e = ray.emission * ray.diffuseMultiplier;
                    color.x += e.x; color.y += e.y; color.z += e.z;
                    break;
                }
                normalize_inplace(ray.direction);
            } else {
                Vec3 e;
                if (marching) {  // main.cu:364-372: no hit within this step, so march; the backdrop only far from the origin
                    march_ray(ray);
                    if (ray.position.lengthSqr() > 20.0f * 20.0f) {
                        e = bg * ray.diffuseMultiplier * p.background_brightness;
                        color.x += e.x; color.y += e.y; color.z += e.z;
                        break;
                    }
                    continue;
                }
                if ((p.flags & RT_F_SKYBOX) && p.cam.env_width > 0) {
                    long long envI = env_index(ray.direction, p.cam.env_width, p.cam.env_height);
                    c.texels++;
                    e = fetch_texel(sc, envI) * ray.diffuseMultiplier * p.background_brightness;
                } else {
                    e = bg * ray.diffuseMultiplier * p.background_brightness;
                }
                color.x += e.x; color.y += e.y; color.z += e.z;
                break;
            }
        }
    }
    return color;
}

inline bool row_selected(const rt_render_params& p, int y, int* packed_row) {
    if (p.band_stride <= 1 || p.band_rows <= 0) {
        *packed_row = y;
        return true;
    }
    const int band = y / p.band_rows;
    if (band % p.band_stride != p.band_index) return false;
    *packed_row = (band / p.band_stride) * p.band_rows + (y % p.band_rows);
    return true;
}

}  // namespace

extern "C" {

// Known-answer hook for the RNG block function.
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }

// Number of accumulator rows a render with these params touches.
int orc_rows(const rt_render_params* p) {
    int rows = 0, pr;
    for (int y = 0; y < p->cam.height; y++)
        if (row_selected(*p, y, &pr)) rows++;
    return rows;
}

// render2 restated: adds sample_count samples to accum (rows*W*3 raw sums).
int orc_render(const rt_scene_desc* scene, const rt_render_params* p, float* accum, rt_render_stats* stats,
               int n_threads) {
    if (!scene || !p || !accum) return RT_ERR_INVALID;
    const Scene sc = view_of(scene);
    const int W = p->cam.width, H = p->cam.height;
    if (n_threads < 1) n_threads = 1;
    std::vector<Counters> counters(n_threads);
    auto work = [&](int tid) {
        Counters& c = counters[tid];
        for (int y = tid; y < H; y += n_threads) {
            int pr;
            if (!row_selected(*p, y, &pr)) continue;
            for (int x = 0; x < W; x++) {
                Vec3 col = render_pixel(sc, *p, x, y, c);
                float* a = accum + ((size_t)pr * W + x) * 3;
                a[0] += col.x;
                a[1] += col.y;
                a[2] += col.z;
            }
        }
    };
    if (n_threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    if (stats) {
        Counters tot;
        for (auto& c : counters) tot.add(c);
        stats->kernel_ms = 0.0;
        stats->paths = tot.paths;
        stats->rays = tot.rays;
        stats->inner_visits = tot.inner;
        stats->leaf_visits = tot.leaf;
        stats->tri_tests = tot.tri;
        stats->hits = tot.hits;
        stats->texel_fetches = tot.texels;
        stats->light_tests = tot.lights;
    }
    return RT_OK;
}

// `color /= numSamples; toneMap(color)` of main.cu:392-395 applied to raw sums.
void orc_resolve(const float* accum, size_t n_pixels, int samples, int mode, float* out) {
    for (size_t i = 0; i < n_pixels; i++) {
        Vec3 c(accum[i * 3], accum[i * 3 + 1], accum[i * 3 + 2]);
        if (mode != RT_READ_RAW_SUM) {
            float s = static_cast<float>(samples);
            c.x /= s; c.y /= s; c.z /= s;
            if (mode == RT_READ_TONEMAPPED) tone_map(c);
        }
        out[i * 3] = c.x; out[i * 3 + 1] = c.y; out[i * 3 + 2] = c.z;
    }
}

// Closest hit of explicit rays (same outputs as rt_trace_rays in rt.h).
int orc_trace_rays(const rt_scene_desc* scene, const float* org_dir, size_t n, uint32_t flags, int32_t* prim_id,
                   float* t, float* hit_pos, float* normal3, int32_t* counts) {
    if (!scene || !org_dir) return RT_ERR_INVALID;
    const Scene sc = view_of(scene);
    for (size_t r = 0; r < n; r++) {
        const float* q = org_dir + r * 6;
        Vec3 o(q[0], q[1], q[2]), d(q[3], q[4], q[5]);
        Hit h;
        Counters c;
        closest_hit(sc, o, d, (flags & RT_F_BVH) != 0, h, c);
        if (prim_id) prim_id[r] = h.tri >= 0 ? h.tri / 25 : -1;
        if (t) t[r] = h.tri >= 0 ? sqrtf(h.minDistSqr) : -1.0f;
        if (hit_pos) {
            hit_pos[r * 3] = h.tri >= 0 ? h.pos.x : 0.f;
            hit_pos[r * 3 + 1] = h.tri >= 0 ? h.pos.y : 0.f;
            hit_pos[r * 3 + 2] = h.tri >= 0 ? h.pos.z : 0.f;
        }
        if (normal3) {
            Vec3 nn(0.f, 0.f, 0.f);
            if (h.tri >= 0) nn = shading_normal(sc.tris + h.tri, h.pos);
            normal3[r * 3] = nn.x; normal3[r * 3 + 1] = nn.y; normal3[r * 3 + 2] = nn.z;
        }
        if (counts) {
            counts[r * 3] = (int32_t)c.inner;
            counts[r * 3 + 1] = (int32_t)c.leaf;
            counts[r * 3 + 2] = (int32_t)c.tri;
        }
    }
    return RT_OK;
}

// Deepest traversal stack any of the rays needs (for sizing the LDS stack).
int orc_max_stack(const rt_scene_desc* scene, const float* org_dir, size_t n) {
    const Scene sc = view_of(scene);
    int mx = 0;
    for (size_t r = 0; r < n; r++) {
        const float* q = org_dir + r * 6;
        Hit h;
        Counters c;
        closest_hit(sc, Vec3(q[0], q[1], q[2]), Vec3(q[3], q[4], q[5]), true, h, c, &mx);
    }
    return mx;
}

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// Unit primitives (record layouts: RT_OP_* in include/rt.h).
int orc_debug_op(int op, const float* in, size_t n, float* out, const rt_render_params* params) {
    for (size_t r = 0; r < n; r++) {
        switch (op) {
            case RT_OP_TRI_INTERSECT: {
                const float* q = in + r * 15;
                float* o = out + r * 4;
                Vec3 pos;
                bool hit = ray_triangle(Vec3(q[0], q[1], q[2]), Vec3(q[3], q[4], q[5]), Vec3(q[6], q[7], q[8]),
                                        Vec3(q[9], q[10], q[11]), Vec3(q[12], q[13], q[14]), pos);
                o[0] = hit ? 1.f : 0.f; o[1] = pos.x; o[2] = pos.y; o[3] = pos.z;
                break;
            }
            case RT_OP_AABB_DIST: {
                const float* q = in + r * 12;
                out[r] = aabb_distance(Vec3(q[0], q[1], q[2]), Vec3(q[3], q[4], q[5]), q + 6, q + 9);
                break;
            }
            case RT_OP_SPHERE: {
                const float* q = in + r * 10;
                float* o = out + r * 7;
                Vec3 pos, nrm;
                bool hit = intersect_sphere(Vec3(q[0], q[1], q[2]), Vec3(q[3], q[4], q[5]), Vec3(q[6], q[7], q[8]), q[9], pos, nrm);
                o[0] = hit ? 1.f : 0.f; o[1] = pos.x; o[2] = pos.y; o[3] = pos.z; o[4] = nrm.x; o[5] = nrm.y; o[6] = nrm.z;
                break;
            }
            case RT_OP_FLOOR: {
                const float* q = in + r * 7;
                float* o = out + r * 7;
                Vec3 pos, nrm;
                bool hit = intersect_floor(Vec3(q[0], q[1], q[2]), Vec3(q[3], q[4], q[5]), q[6], pos, nrm);
                o[0] = hit ? 1.f : 0.f; o[1] = pos.x; o[2] = pos.y; o[3] = pos.z; o[4] = nrm.x; o[5] = nrm.y; o[6] = nrm.z;
                break;
            }
            case RT_OP_ITRI_PDF: {
                const float* q = in + r * 15;
                ImportanceTri tri(Vec3(q[6], q[7], q[8]), Vec3(q[9], q[10], q[11]), Vec3(q[12], q[13], q[14]));
                out[r] = tri.pdf_value(Vec3(q[0], q[1], q[2]), Vec3(q[3], q[4], q[5]));
                break;
            }
            case RT_OP_ITRI_SAMPLE: {
                const float* q = in + r * 14;
                ImportanceTri tri(Vec3(q[5], q[6], q[7]), Vec3(q[8], q[9], q[10]), Vec3(q[11], q[12], q[13]));
                Vec3 d = tri.generate_random(Vec3(q[0], q[1], q[2]), q[3], q[4]);
                out[r * 3] = d.x; out[r * 3 + 1] = d.y; out[r * 3 + 2] = d.z;
                break;
            }
            case RT_OP_ATTRACTOR_PDF: {
                const float* q = in + r * 15;
                RayAttractor a(Vec3(q[6], q[7], q[8]), Vec3(q[9], q[10], q[11]), Vec3(q[12], q[13], q[14]));
                out[r] = a.pdf_value(Vec3(q[0], q[1], q[2]), Vec3(q[3], q[4], q[5]));
                break;
            }
            case RT_OP_ATTRACTOR_SAMPLE: {
                const float* q = in + r * 14;
                RayAttractor a(Vec3(q[5], q[6], q[7]), Vec3(q[8], q[9], q[10]), Vec3(q[11], q[12], q[13]));
                Vec3 d = a.generate_random(Vec3(q[0], q[1], q[2]), q[3], q[4]);
                out[r * 3] = d.x; out[r * 3 + 1] = d.y; out[r * 3 + 2] = d.z;
                break;
            }
            case RT_OP_TONEMAP: {
                Vec3 c(in[r * 3], in[r * 3 + 1], in[r * 3 + 2]);
                tone_map(c);
                out[r * 3] = c.x; out[r * 3 + 1] = c.y; out[r * 3 + 2] = c.z;
                break;
            }
            case RT_OP_RNG: {
                const float* q = in + r * 4;
                uint32_t dim = f2u(q[3]);
                Rng rng(f2u(q[0]), f2u(q[1]), f2u(q[2]));
                rng.seek(dim & ~3u);
                for (int k = 0; k < 4; k++) out[r * 4 + k] = u2f(rng.raw());
                break;
            }
            case RT_OP_ENV_INDEX: {
                const float* q = in + r * 5;
                out[r] = u2f((uint32_t)(int32_t)env_index(Vec3(q[0], q[1], q[2]), (int)q[3], (int)q[4]));
                break;
            }
            case RT_OP_CAMERA_RAY: {
                if (!params) return RT_ERR_INVALID;
                const float* q = in + r * 3;
                int x = (int)f2u(q[0]), y = (int)f2u(q[1]);
                Rng rng(params->seed, (uint32_t)(x + y * params->cam.width), f2u(q[2]));
                Ray ray = camera_ray(*params, x, y, rng);
                float* o = out + r * 6;
                o[0] = ray.position.x; o[1] = ray.position.y; o[2] = ray.position.z;
                o[3] = ray.direction.x; o[4] = ray.direction.y; o[5] = ray.direction.z;
                break;
            }
            case RT_OP_PMATH: {
                const float* q = in + r * 3;
                float a = 0.f, b = 0.f;
                switch ((int)f2u(q[0])) {
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
                return RT_ERR_INVALID;
        }
    }
    return RT_OK;
}

// One Material::reflect per record (layout: rt_shade_hits in include/rt.h).
int orc_shade_hits(const rt_scene_desc* scene, const float* in, size_t n, uint32_t seed, float* out) {
    const Scene sc = view_of(scene);
    for (size_t r = 0; r < n; r++) {
        const float* q = in + r * 16;
        float* o = out + r * 16;
        Ray ray(Vec3(q[0], q[1], q[2]), Vec3(q[3], q[4], q[5]));
        ray.diffuseMultiplier = Vec3(q[6], q[7], q[8]);
        ray.refractiveIndex = q[9];
        Vec3 hitPos(q[10], q[11], q[12]);
        int tri = (int)f2u(q[13]) * 25;
        Rng rng(seed, f2u(q[14]), f2u(q[15]));
        rng.seek(RT_RNG_DIM_BOUNCE0);
        Counters c;
        int hitIndex = static_cast<int>(sc.tris[tri + 24]);
        rt_material m;
        if (hitIndex >= 0 && (size_t)hitIndex < sc.nMats) m = sc.mats[hitIndex];
        else { memset(&m, 0, sizeof(m)); m.type = 127; }
        bool term = material_reflect(sc, m, ray, hitPos, tri, rng, c);
        o[0] = ray.position.x; o[1] = ray.position.y; o[2] = ray.position.z;
        o[3] = ray.direction.x; o[4] = ray.direction.y; o[5] = ray.direction.z;
        o[6] = ray.diffuseMultiplier.x; o[7] = ray.diffuseMultiplier.y; o[8] = ray.diffuseMultiplier.z;
        o[9] = ray.emission.x; o[10] = ray.emission.y; o[11] = ray.emission.z;
        o[12] = ray.refractiveIndex;
        o[13] = ray.volumeDensity;
        o[14] = u2f(term ? 1u : 0u);
        o[15] = 0.f;
    }
    return RT_OK;
}

}  // extern "C"
