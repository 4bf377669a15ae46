// rt_scene.cpp -- host-side scene assembly (see include/rt_scene.h).
//
// Own implementation of the reference's host producers; behaviour follows
//   src/sceneParser.cpp, src/model.h, src/BVHcreator.cpp, src/noiseTexture.cpp,
//   src/texture.cpp (citations at each function).
// Plain C++17, no GPU, no third-party code.  Build with -ffp-contract=off so
// the float op order below is what is executed.

#include "rt_scene.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#include <sys/mman.h>

// std::vector<float> whose resize() does not clear: record arrays of a 10M-triangle scene are 1 GB, and every element is
// written right after the array is sized (value-initialising allocator calls only).
// Big arrays are 2 MB-aligned and advised to use huge pages (advice only): the 10 M-triangle scene touches some 3 GB of
// fresh memory while it is assembled, a fault per 4 KB page otherwise.  RT_NO_HUGE_PAGES=1: plain malloc.
static void* big_alloc(size_t bytes) {
    static const bool plain = getenv("RT_NO_HUGE_PAGES") != nullptr;
    const size_t huge = (size_t)2 << 20;
    bytes = std::max<size_t>(bytes, 1);
    if (bytes < 4 * huge || plain) return malloc(bytes);
    const size_t rounded = (bytes + huge - 1) / huge * huge;
    void* q = aligned_alloc(huge, rounded);
    if (q) madvise(q, rounded, MADV_HUGEPAGE);
    return q;
}
template <class T> struct BigFree { void operator()(T* q) const { free(q); } };
template <class T> using BigBuf = std::unique_ptr<T[], BigFree<T>>;  // n elements of a trivial T, not filled
template <class T> BigBuf<T> big_buf(size_t n) {
    void* q = big_alloc(n * sizeof(T));
    if (!q) throw std::bad_alloc();
    return BigBuf<T>((T*)q);
}
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAlloc<U>; };
    using std::allocator<T>::allocator;
    T* allocate(size_t n) {
        void* q = big_alloc(n * sizeof(T));
        if (!q) throw std::bad_alloc();
        return (T*)q;
    }
    void deallocate(T* q, size_t) { free(q); }
    template <class U> void construct(U* ptr) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(ptr)) U; }
    template <class U, class... A> void construct(U* ptr, A&&... a) { ::new (static_cast<void*>(ptr)) U(std::forward<A>(a)...); }
};
using FloatVec = std::vector<float, DefaultInitAlloc<float>>;

namespace {

thread_local std::string g_err;

int fail(const std::string& msg) {
    g_err = msg;
    return RT_ERR_INVALID;
}

struct V3 {
    float x = 0.f, y = 0.f, z = 0.f;
};

// Matrix::rotationMatrix, matrix.h:14-37 (XYZ Euler).  Float sin/cos.
struct Mat3 {
    float m[9];
};

Mat3 rotation_matrix(const V3& v) {
    Mat3 r;
    const float cx = cosf(v.x), sx = sinf(v.x);
    const float cy = cosf(v.y), sy = sinf(v.y);
    const float cz = cosf(v.z), sz = sinf(v.z);
    r.m[0] = cy * cz;
    r.m[1] = -cy * sz;
    r.m[2] = sy;
    r.m[3] = cx * sz + sx * sy * cz;
    r.m[4] = cx * cz - sx * sy * sz;
    r.m[5] = -sx * cy;
    r.m[6] = sx * sz - cx * sy * cz;
    r.m[7] = sx * cz + cx * sy * sz;
    r.m[8] = cx * cy;
    return r;
}

// operator*(Matrix, Vec3), matrix.h:48-54: left-to-right sums.
V3 mul(const Mat3& a, const V3& v) {
    V3 r;
    r.x = a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z;
    r.y = a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z;
    r.z = a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z;
    return r;
}

// ------------------------------------------------------------------ images

struct Image {
    int w = 0, h = 0;
    std::vector<float> rgb;
};

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; i++)
        if (tolower((unsigned char)s[s.size() - n + i]) != suf[i]) return false;
    return true;
}

bool read_file(const std::string& path, std::vector<unsigned char>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
    fclose(f);
    return got == out.size();
}

// Radiance RGBE -> float as stb_image does it (value = mantissa * 2^(e-136),
// no half-step offset; e == 0 -> black).
inline void rgbe_to_float(const unsigned char* p, float* out) {
    if (p[3] != 0) {
        const float f1 = (float)ldexp(1.0f, (int)p[3] - (128 + 8));
        out[0] = p[0] * f1;
        out[1] = p[1] * f1;
        out[2] = p[2] * f1;
    } else {
        out[0] = out[1] = out[2] = 0.f;
    }
}

bool load_hdr(const std::string& path, Image& img, std::string& err) {
    std::vector<unsigned char> buf;
    if (!read_file(path, buf)) {
        err = "could not open image: " + path;
        return false;
    }
    size_t pos = 0;
    auto next_line = [&](std::string& line) {
        line.clear();
        while (pos < buf.size() && buf[pos] != '\n') line.push_back((char)buf[pos++]);
        if (pos < buf.size()) pos++;
    };
    std::string line;
    next_line(line);
    if (line != "#?RADIANCE" && line != "#?RGBE") {
        err = "not a Radiance HDR file: " + path;
        return false;
    }
    bool fmt_ok = false;
    for (;;) {
        next_line(line);
        if (line.empty()) break;
        if (line == "FORMAT=32-bit_rle_rgbe") fmt_ok = true;
        if (pos >= buf.size()) break;
    }
    if (!fmt_ok) {
        err = "unsupported HDR format: " + path;
        return false;
    }
    next_line(line);
    int h = 0, w = 0;
    if (sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) {
        err = "unsupported HDR orientation: " + path;
        return false;
    }
    img.w = w;
    img.h = h;
    img.rgb.assign((size_t)w * h * 3, 0.f);
    std::vector<unsigned char> scan((size_t)w * 4);
    auto need = [&](size_t n) { return pos + n <= buf.size(); };
    if (w < 8 || w >= 32768) {  // flat
        for (size_t i = 0; i < (size_t)w * h; i++) {
            if (!need(4)) { err = "truncated HDR: " + path; return false; }
            rgbe_to_float(&buf[pos], &img.rgb[i * 3]);
            pos += 4;
        }
        return true;
    }
    for (int y = 0; y < h; y++) {
        if (!need(4)) { err = "truncated HDR: " + path; return false; }
        const unsigned char c1 = buf[pos], c2 = buf[pos + 1], len = buf[pos + 2];
        if (c1 != 2 || c2 != 2 || (len & 0x80)) {
            // not run-length encoded: whole file is flat from here (first scanline only case)
            if (y != 0) { err = "mixed HDR encodings: " + path; return false; }
            for (size_t i = 0; i < (size_t)w * h; i++) {
                if (!need(4)) { err = "truncated HDR: " + path; return false; }
                rgbe_to_float(&buf[pos], &img.rgb[i * 3]);
                pos += 4;
            }
            return true;
        }
        const int sl = ((int)len << 8) | buf[pos + 3];
        pos += 4;
        if (sl != w) { err = "corrupt HDR scanline width: " + path; return false; }
        for (int k = 0; k < 4; k++) {
            int i = 0;
            while (i < w) {
                if (!need(1)) { err = "truncated HDR: " + path; return false; }
                unsigned char count = buf[pos++];
                if (count > 128) {  // run
                    count -= 128;
                    if (!need(1) || count == 0 || i + count > w) { err = "corrupt HDR run: " + path; return false; }
                    const unsigned char v = buf[pos++];
                    for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = v;
                } else {  // dump
                    if (count == 0 || !need(count) || i + count > w) { err = "corrupt HDR dump: " + path; return false; }
                    for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = buf[pos++];
                }
            }
        }
        for (int x = 0; x < w; x++) rgbe_to_float(&scan[(size_t)x * 4], &img.rgb[((size_t)y * w + x) * 3]);
    }
    return true;
}

bool ppm_token(const std::vector<unsigned char>& buf, size_t& pos, std::string& tok) {
    tok.clear();
    for (;;) {
        while (pos < buf.size() && isspace(buf[pos])) pos++;
        if (pos < buf.size() && buf[pos] == '#') {
            while (pos < buf.size() && buf[pos] != '\n') pos++;
            continue;
        }
        break;
    }
    while (pos < buf.size() && !isspace(buf[pos])) tok.push_back((char)buf[pos++]);
    return !tok.empty();
}

bool load_ppm(const std::string& path, Image& img, std::string& err) {
    std::vector<unsigned char> buf;
    if (!read_file(path, buf)) { err = "could not open image: " + path; return false; }
    size_t pos = 0;
    std::string t;
    if (!ppm_token(buf, pos, t) || t != "P6") { err = "not a binary PPM: " + path; return false; }
    int w = 0, h = 0, mx = 0;
    if (!ppm_token(buf, pos, t)) { err = "bad PPM header"; return false; }
    w = atoi(t.c_str());
    if (!ppm_token(buf, pos, t)) { err = "bad PPM header"; return false; }
    h = atoi(t.c_str());
    if (!ppm_token(buf, pos, t)) { err = "bad PPM header"; return false; }
    mx = atoi(t.c_str());
    pos++;  // single whitespace after maxval
    if (w <= 0 || h <= 0 || mx != 255 || pos + (size_t)w * h * 3 > buf.size()) {
        err = "unsupported or truncated PPM: " + path;
        return false;
    }
    img.w = w;
    img.h = h;
    img.rgb.resize((size_t)w * h * 3);
    // stbi_loadf on LDR input: pow(v/255, 2.2) evaluated in double (stb's ldr_to_hdr).
    float lut[256];
    for (int i = 0; i < 256; i++) lut[i] = (float)(pow((double)(i / 255.0f), (double)2.2f) * (double)1.0f);
    for (size_t i = 0; i < img.rgb.size(); i++) img.rgb[i] = lut[buf[pos + i]];
    return true;
}

bool load_pfm(const std::string& path, Image& img, std::string& err) {
    std::vector<unsigned char> buf;
    if (!read_file(path, buf)) { err = "could not open image: " + path; return false; }
    size_t pos = 0;
    std::string t;
    if (!ppm_token(buf, pos, t) || t != "PF") { err = "not a colour PFM: " + path; return false; }
    if (!ppm_token(buf, pos, t)) { err = "bad PFM header"; return false; }
    const int w = atoi(t.c_str());
    if (!ppm_token(buf, pos, t)) { err = "bad PFM header"; return false; }
    const int h = atoi(t.c_str());
    if (!ppm_token(buf, pos, t)) { err = "bad PFM header"; return false; }
    const double scale = atof(t.c_str());
    pos++;
    if (w <= 0 || h <= 0 || scale >= 0 || pos + (size_t)w * h * 12 > buf.size()) {
        err = "unsupported (big-endian?) or truncated PFM: " + path;
        return false;
    }
    img.w = w;
    img.h = h;
    img.rgb.resize((size_t)w * h * 3);
    // PFM rows are bottom-to-top
    for (int y = 0; y < h; y++)
        memcpy(&img.rgb[(size_t)y * w * 3], &buf[pos + (size_t)(h - 1 - y) * w * 12], (size_t)w * 12);
    return true;
}

bool load_image(const std::string& path, Image& img, std::string& err) {
    if (ends_with(path, ".hdr")) return load_hdr(path, img, err);
    if (ends_with(path, ".ppm")) return load_ppm(path, img, err);
    if (ends_with(path, ".pfm")) return load_pfm(path, img, err);
    err = "unsupported image format (need .hdr/.ppm/.pfm): " + path;
    return false;
}

// ------------------------------------------------------------------ noise

// NoiseTexture::fourier_series, noiseTexture.cpp:13-20
float fourier_series(float x, float y) {
    float h = 0.0f;
    h += sinf(x * 1.52f + y * 23.0f);
    h += sinf(x * 9.5f + y * 10.0f) * 0.9f;
    h += sinf(x * 15.5f + y * 23.0f) * 0.7f;
    h += sinf(x * 35.0f + y * 2.0f) * 0.5f;
    return h;
}

// NoiseTexture::generate, noiseTexture.cpp:22-40.  Double literals in the
// reference promote those products to double; kept.
void noise_generate(int width, int height, float* data) {
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            const float u = static_cast<float>(x) / width;
            const float v = static_cast<float>(y) / height;
            float ux = u, uy = v;
            ux = (float)((double)ux + (double)fourier_series(ux + 0.2f, uy + 0.9f) * 0.1);
            uy = (float)((double)uy + (double)fourier_series(ux + 0.1f, uy + 0.3f) * 0.1);
            float h = fourier_series(ux, uy);
            h = (float)(1.0 / (1.0 + std::exp((double)(-h) * 2.0)));
            const size_t idx = ((size_t)y * width + x) * 3;
            data[idx] = h;
            data[idx + 1] = h;
            data[idx + 2] = h;
        }
    }
}

// ------------------------------------------------------------------ OBJ

struct ParsedObj {
    std::vector<float> v, vn, vt;  // raw (untransformed)
    std::vector<int> idx;          // 9 per face: v0 v1 v2 n0 n1 n2 t0 t1 t2 (1-based, 0 = absent)
};

bool parse_obj(const std::string& path, ParsedObj& o, std::string& err) {
    std::ifstream file(path);
    if (!file.is_open()) {
        err = "could not open model file: " + path;
        return false;
    }
    std::string line;
    while (std::getline(file, line)) {
        if (line.size() < 2) continue;
        const char c0 = line[0], c1 = line[1];
        if (c0 == 'v' && c1 == ' ') {
            float x = 0, y = 0, z = 0;
            sscanf(line.c_str(), "v %f %f %f", &x, &y, &z);
            o.v.push_back(x); o.v.push_back(y); o.v.push_back(z);
        } else if (c0 == 'v' && c1 == 'n') {
            float x = 0, y = 0, z = 0;
            sscanf(line.c_str(), "vn %f %f %f", &x, &y, &z);
            o.vn.push_back(x); o.vn.push_back(y); o.vn.push_back(z);
        } else if (c0 == 'v' && c1 == 't') {
            float u = 0, v = 0;
            sscanf(line.c_str(), "vt %f %f", &u, &v);
            o.vt.push_back(u); o.vt.push_back(v);
        } else if (c0 == 'f' && c1 == ' ') {
            int v[3] = {0, 0, 0}, t[3] = {0, 0, 0}, n[3] = {0, 0, 0};
            const char* s = line.c_str();
            if (sscanf(s, "f %d/%d/%d %d/%d/%d %d/%d/%d", &v[0], &t[0], &n[0], &v[1], &t[1], &n[1], &v[2], &t[2], &n[2]) == 9) {
            } else if (sscanf(s, "f %d//%d %d//%d %d//%d", &v[0], &n[0], &v[1], &n[1], &v[2], &n[2]) == 6) {
                t[0] = t[1] = t[2] = 0;
            } else if (sscanf(s, "f %d/%d %d/%d %d/%d", &v[0], &t[0], &v[1], &t[1], &v[2], &t[2]) == 6) {
                n[0] = n[1] = n[2] = 0;
            } else if (sscanf(s, "f %d %d %d", &v[0], &v[1], &v[2]) == 3) {
                t[0] = t[1] = t[2] = 0;
                n[0] = n[1] = n[2] = 0;
            } else {
                err = "unsupported face line in " + path + ": " + line;
                return false;
            }
            const int nv = (int)(o.v.size() / 3), nn = (int)(o.vn.size() / 3), nt = (int)(o.vt.size() / 2);
            for (int k = 0; k < 3; k++) {
                if (v[k] < 1 || v[k] > nv || n[k] < 0 || n[k] > nn || t[k] < 0 || t[k] > nt) {
                    err = "face index out of range in " + path + ": " + line;
                    return false;
                }
            }
            for (int k = 0; k < 3; k++) o.idx.push_back(v[k]);
            for (int k = 0; k < 3; k++) o.idx.push_back(n[k]);
            for (int k = 0; k < 3; k++) o.idx.push_back(t[k]);
        }
    }
    return true;
}

// Model::loadMesh, model.h:62-163: v' = (R*v)*scale + T ; n' = R*n ; records in face order.
// Records of one instance into w[0 .. 25 * triangles).
void emit_records_to(const ParsedObj& o, const V3& T, float scale, const V3& rot, int material_index, float* w) {
    const Mat3 R = rotation_matrix(rot);
    std::vector<float> verts(o.v.size()), norms(o.vn.size());
    for (size_t i = 0; i < o.v.size(); i += 3) {
        V3 p{o.v[i], o.v[i + 1], o.v[i + 2]};
        V3 r = mul(R, p);
        verts[i] = r.x * scale + T.x;
        verts[i + 1] = r.y * scale + T.y;
        verts[i + 2] = r.z * scale + T.z;
    }
    for (size_t i = 0; i < o.vn.size(); i += 3) {
        V3 p{o.vn[i], o.vn[i + 1], o.vn[i + 2]};
        V3 r = mul(R, p);
        norms[i] = r.x; norms[i + 1] = r.y; norms[i + 2] = r.z;
    }
    const float matf = (float)((double)static_cast<float>(material_index) + 0.001);  // model.h:162
    for (size_t i = 0; i < o.idx.size(); i += 9, w += RT_TRI_FLOATS) {
        for (int k = 0; k < 3; k++) {
            const int vi = (o.idx[i + k] - 1) * 3;
            w[k * 3 + 0] = verts[vi]; w[k * 3 + 1] = verts[vi + 1]; w[k * 3 + 2] = verts[vi + 2];
        }
        bool have_n = o.idx[i + 3] > 0 && o.idx[i + 4] > 0 && o.idx[i + 5] > 0;
        if (have_n) {
            for (int k = 0; k < 3; k++) {
                const int ni = (o.idx[i + 3 + k] - 1) * 3;
                w[9 + k * 3 + 0] = norms[ni]; w[9 + k * 3 + 1] = norms[ni + 1]; w[9 + k * 3 + 2] = norms[ni + 2];
            }
        } else {  // extension: geometric normal when the file has none
            const float e1x = w[3] - w[0], e1y = w[4] - w[1], e1z = w[5] - w[2];
            const float e2x = w[6] - w[0], e2y = w[7] - w[1], e2z = w[8] - w[2];
            float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
            const float len = sqrtf(nx * nx + ny * ny + nz * nz);
            if (len > 0) { nx /= len; ny /= len; nz /= len; }
            for (int k = 0; k < 3; k++) { w[9 + k * 3] = nx; w[10 + k * 3] = ny; w[11 + k * 3] = nz; }
        }
        for (int k = 0; k < 3; k++) {
            const int ti = o.idx[i + 6 + k];
            w[18 + k * 2] = ti > 0 ? o.vt[(ti - 1) * 2] : 0.f;
            w[19 + k * 2] = ti > 0 ? o.vt[(ti - 1) * 2 + 1] : 0.f;
        }
        w[24] = matf;
    }
}

template <class Vec>
void emit_records(const ParsedObj& o, const V3& T, float scale, const V3& rot, int material_index, Vec& out) {
    const size_t base = out.size();
    out.resize(base + o.idx.size() / 9 * RT_TRI_FLOATS);
    emit_records_to(o, T, scale, rot, material_index, out.data() + base);
}

// Stage timer: RT_SCENE_TIMING=1 prints the wall time of the scene-assembly stages to stderr.
struct StageTimer {
    const char* name;
    std::chrono::steady_clock::time_point t0;
    explicit StageTimer(const char* n) : name(n), t0(std::chrono::steady_clock::now()) {}
    ~StageTimer() {
        static const bool on = getenv("RT_SCENE_TIMING") != nullptr;
        if (on) fprintf(stderr, "[rt_scene] %-28s %8.1f ms\n", name,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

// ------------------------------------------------------------------ BVH

// Host worker threads for the data-parallel parts of scene assembly (RT_HOST_THREADS overrides; at most 16).
int host_threads() {
    static const int n = [] {
        int t = (int)std::thread::hardware_concurrency();
        if (const char* e = getenv("RT_HOST_THREADS")) t = atoi(e);
        return std::max(1, std::min(t, 16));
    }();
    return n;
}

// fn(i) for i in [0, n), dynamically scheduled over the host threads.
template <class F>
void parallel_for(size_t n, size_t grain, F fn) {
    const int nt = (int)std::min<size_t>((size_t)host_threads(), (n + grain - 1) / std::max<size_t>(grain, 1));
    if (nt <= 1) {
        for (size_t i = 0; i < n; i++) fn(i);
        return;
    }
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (;;) {
            const size_t b = next.fetch_add(grain);
            if (b >= n) return;
            const size_t e = std::min(n, b + grain);
            for (size_t i = b; i < e; i++) fn(i);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}

struct TmpNode {
    float mn[3], mx[3];
    uint32_t begin = 0, end = 0;  // its triangles: ids[begin, end), input order preserved
    int a = -1, b = -1;
    int depth = 0;
    float sum3[3] = {0.f, 0.f, 0.f};  // per axis, the serial float sum of its triangles' vertex sums in triangle order ...
    bool has_sum = false;             // ... when the pass that made the node formed it on the way
};

// Output stage shared by the builders: nodes are in creation order = BFS order, A before B (BFSIndex,
// BVHcreator.cpp:41-61); triangles are rewritten leaf by leaf in that order (populateBVHNodes :209-237).
// owner (optional): the vector whose tail `tris` is (records behind a prefix of owner->size() - n_floats floats).  The
// reordered records are then written into a fresh vector that takes the owner's place, instead of into a scratch array
// that is copied back (1 GB each way for 10 M triangles).
int finish_bvh(const std::vector<TmpNode>& nodes, const uint32_t* ids, float* tris, size_t n_floats,
               std::vector<rt_bvh_node>& out, int* out_depth, FloatVec* owner = nullptr) {
    StageTimer tm_out("bvh: numbering + reorder");
    out.resize(nodes.size());
    int offs = 0, deepest_leaf = 0;
    for (size_t k = 0; k < nodes.size(); k++) {
        const TmpNode& n = nodes[k];
        rt_bvh_node& o = out[k];
        for (int c = 0; c < 3; c++) { o.bbox_min[c] = n.mn[c]; o.bbox_max[c] = n.mx[c]; }
        o.child_a = n.a;
        o.child_b = n.b;
        o.tri_offs = offs;
        o.tri_len = n.a < 0 ? (int)(n.end - n.begin) * RT_TRI_FLOATS : 0;
        offs += o.tri_len;
        if (n.a < 0) deepest_leaf = std::max(deepest_leaf, n.depth);
    }
    if (owner && owner->size() >= n_floats && owner->data() + (owner->size() - n_floats) == tris) {
        const size_t prefix = owner->size() - n_floats;
        FloatVec fresh(owner->size());  // (not zero-filled: every float is written below)
        if (prefix) memcpy(fresh.data(), owner->data(), prefix * sizeof(float));
        float* base = fresh.data() + prefix;
        parallel_for(nodes.size(), 64, [&](size_t k) {
            const TmpNode& n = nodes[k];
            if (n.a >= 0) return;
            float* dst = base + (size_t)out[k].tri_offs;
            for (uint32_t i = n.begin; i < n.end; i++, dst += RT_TRI_FLOATS)
                memcpy(dst, tris + (size_t)ids[i] * RT_TRI_FLOATS, RT_TRI_FLOATS * sizeof(float));
        });
        owner->swap(fresh);
        if (out_depth) *out_depth = deepest_leaf;
        return RT_OK;
    }
    BigBuf<float> newtris = big_buf<float>(std::max<size_t>(n_floats, 1));  // (not zero-filled: every float is written below)
    parallel_for(nodes.size(), 64, [&](size_t k) {
        const TmpNode& n = nodes[k];
        if (n.a >= 0) return;
        float* dst = &newtris[(size_t)out[k].tri_offs];
        for (uint32_t i = n.begin; i < n.end; i++, dst += RT_TRI_FLOATS)
            memcpy(dst, tris + (size_t)ids[i] * RT_TRI_FLOATS, RT_TRI_FLOATS * sizeof(float));
    });
    if (n_floats) {
        const size_t cp = (size_t)1 << 22, nch = (n_floats + cp - 1) / cp;
        parallel_for(nch, 1, [&](size_t c) { memcpy(tris + c * cp, &newtris[c * cp], std::min(cp, n_floats - c * cp) * sizeof(float)); });
    }
    if (out_depth) *out_depth = deepest_leaf;
    return RT_OK;
}

// createBVH, BVHcreator.cpp:240-256 (createInitialNode :187-205, splitRecursive :173-184, splitNode :86-170,
// BFSIndex :41-61, populateBVHNodes :209-237) -- same tree, same numbering, same triangle order, built level by
// level instead of recursively:
//  * per triangle, the three vertex sums a+b+c per axis (what both the mean and the centroid test of splitNode add
//    up, :113-135) and its box are formed once (12 + 24 bytes instead of gathers from 100-byte records);
//  * a node owns a range of ONE id array; splitting is a stable partition of that range, so both children keep the
//    input order the reference's push_back loops produce;
//  * the mean stays a serial float accumulation in triangle order (its rounding is part of the result); nodes of a
//    level are independent and run on the host threads; big nodes partition and bound their ranges in parallel
//    chunks (min/max are exact, so chunking cannot change a box);
//  * children are created A then B in parent order, which IS the reference's BFS numbering.
int build_bvh(float* tris, size_t n_floats, int max_depth, int min_tris, std::vector<rt_bvh_node>& out,
              int* out_depth, FloatVec* owner = nullptr) {
    if (n_floats % RT_TRI_FLOATS != 0) return fail("triangle array length is not a multiple of 25");
    const size_t ntri = n_floats / RT_TRI_FLOATS;
    if (ntri > 0x7fffffffu / RT_TRI_FLOATS) return fail("too many triangles for int32 float offsets");
    size_t chunk_max = (size_t)1 << 18;  // triangles per parallel chunk of a big node, at most (RT_BVH_CHUNK: tests force small chunks)
    if (const char* e = getenv("RT_BVH_CHUNK")) chunk_max = (size_t)std::max(1, atoi(e));
    const size_t small_node = std::min<size_t>(chunk_max, (size_t)1 << 16);  // levels of nodes this small never use the chunked path
    const bool force_single_pass = getenv("RT_BVH_SINGLE_PASS") != nullptr;  // (tests: both paths on the same input)
    const size_t chunk = chunk_max;        // (range_box below: chunk of the fold)
    // (none of these is zero-filled: every element is written before it is read)
    BigBuf<float> csum_mem = big_buf<float>(std::max<size_t>(ntri, 1) * 3), tbox_mem = big_buf<float>(std::max<size_t>(ntri, 1) * 6);
    float* csum = csum_mem.get();
    float* tbox = tbox_mem.get();
    BigBuf<uint32_t> ids_mem = big_buf<uint32_t>(std::max<size_t>(ntri, 1)), tmp_mem = big_buf<uint32_t>(std::max<size_t>(ntri, 1));
    uint32_t* ids = ids_mem.get();
    uint32_t* tmp = tmp_mem.get();
    std::vector<TmpNode> nodes(1);
    {
        StageTimer tm("bvh: per-triangle sums, boxes");
        parallel_for(ntri, 1 << 16, [&](size_t i) {
            const float* t = tris + i * RT_TRI_FLOATS;
            float* bx = &tbox[i * 6];
            for (int c = 0; c < 3; c++) {
                csum[i * 3 + c] = t[c] + t[3 + c] + t[6 + c];
                bx[c] = fminf(fminf(fminf(1e12f, t[c]), t[3 + c]), t[6 + c]);
                bx[3 + c] = fmaxf(fmaxf(fmaxf(-1e12f, t[c]), t[3 + c]), t[6 + c]);
            }
            ids[i] = (uint32_t)i;
        });
    }
    // box of ids[b, e): fold of the per-triangle boxes (starts at +-1e12 like createInitialNode / splitNode)
    auto range_box = [&](uint32_t b, uint32_t e, float* mn, float* mx) {
        const size_t n = e - b, nchunks = (n + chunk - 1) / chunk;
        std::vector<float> part(std::max<size_t>(nchunks, 1) * 6);
        auto fold = [&](size_t k) {
            float p[6] = {1e12f, 1e12f, 1e12f, -1e12f, -1e12f, -1e12f};  // (local: the chunks' results share cache lines)
            const size_t lo = b + k * chunk, hi = std::min<size_t>(e, lo + chunk);
            for (size_t i = lo; i < hi; i++) {
                const float* bx = &tbox[(size_t)ids[i] * 6];
                for (int c = 0; c < 3; c++) { p[c] = fminf(p[c], bx[c]); p[3 + c] = fmaxf(p[3 + c], bx[3 + c]); }
            }
            memcpy(&part[k * 6], p, sizeof(p));
        };
        if (nchunks > 1) parallel_for(nchunks, 1, fold);
        else fold(0);
        for (int c = 0; c < 3; c++) { mn[c] = 1e12f; mx[c] = -1e12f; }
        for (size_t k = 0; k < std::max<size_t>(nchunks, 1); k++)
            for (int c = 0; c < 3; c++) { mn[c] = fminf(mn[c], part[k * 6 + c]); mx[c] = fmaxf(mx[c], part[k * 6 + 3 + c]); }
    };
    nodes[0].begin = 0;
    nodes[0].end = (uint32_t)ntri;
    range_box(0, (uint32_t)ntri, nodes[0].mn, nodes[0].mx);

    bool regrouped = false;
    BigBuf<uint32_t> orig_mem;
    size_t regroup_below = (size_t)1 << 17;
    if (const char* e = getenv("RT_BVH_REGROUP")) regroup_below = (size_t)std::max(0, atoi(e));  // (0: never; tests: other thresholds)
    StageTimer* tm_split = new StageTimer("bvh: splits");
    std::vector<int> level{0}, splitting;
    int deepest_leaf = 0;
    while (!level.empty()) {
        splitting.clear();
        for (int ni : level) {  // splitRecursive :173-184, splitNode :97-99
            const TmpNode& n = nodes[ni];
            const uint32_t cnt = n.end - n.begin;
            if (n.depth >= max_depth || (int)cnt <= min_tris || cnt <= 2) { deepest_leaf = std::max(deepest_leaf, n.depth); continue; }
            splitting.push_back(ni);
        }
        const auto lvl_t0 = std::chrono::steady_clock::now();
        const size_t first_child = nodes.size();
        nodes.resize(first_child + 2 * splitting.size());
        for (size_t k = 0; k < splitting.size(); k++) {
            TmpNode& n = nodes[splitting[k]];
            n.a = (int)(first_child + 2 * k);
            n.b = n.a + 1;
            nodes[n.a].depth = nodes[n.b].depth = n.depth + 1;
        }
        auto axis_of = [&](const TmpNode& n) {
            const float xw = n.mx[0] - n.mn[0], yw = n.mx[1] - n.mn[1], zw = n.mx[2] - n.mn[2];
            if (xw >= yw && xw >= zw) return 0;
            if (yw >= xw && yw >= zw) return 1;
            return 2;
        };
        // the means of the level's nodes first, one node per thread: a mean is a SERIAL float sum in triangle order (its
        // rounding is part of the result, :113-123), so the few huge nodes at the top of the tree would otherwise
        // walk their millions of triangles one node after the other
        std::vector<float> means(splitting.size());
        // (tasks of several nodes once there are thousands: neighbours in `nodes` share cache lines, and one counter hands out the tasks)
        const size_t node_grain = std::min<size_t>(64, std::max<size_t>(1, splitting.size() / ((size_t)host_threads() * 16)));
        parallel_for(splitting.size(), node_grain, [&](size_t k) {
            const TmpNode& n = nodes[splitting[k]];
            const int axis = axis_of(n);
            float mean = 0.0f;
            if (n.has_sum) mean = n.sum3[axis];  // (the same additions in the same order, made while the node was filled)
            else for (uint32_t i = n.begin; i < n.end; i++) mean += csum[(size_t)ids[i] * 3 + axis];
            means[k] = mean / static_cast<float>((size_t)(n.end - n.begin) * 3);
        });
        const double lvl_means_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - lvl_t0).count();
        // few nodes (the top of the tree): one after the other, each shared by the threads in chunks; enough nodes to go
        // round (or nothing big among them): one node per task, split by its thread in a single pass
        size_t biggest = 0;
        for (int ni : splitting) biggest = std::max<size_t>(biggest, nodes[ni].end - nodes[ni].begin);
        const bool node_per_task = splitting.size() >= (size_t)host_threads() || biggest <= small_node || force_single_pass;
        if (!regrouped && biggest <= regroup_below && !splitting.empty()) {
            // once the nodes are small: the per-triangle sums and boxes rewritten in the order of the id array, so that a
            // node's data is one contiguous piece (a few MB at most) instead of a gather over its whole instance; from here
            // on the id array holds POSITIONS in that copy and `orig` leads back to the triangles
            regrouped = true;
            BigBuf<float> csum2 = big_buf<float>(ntri * 3), tbox2 = big_buf<float>(ntri * 6);
            orig_mem = big_buf<uint32_t>(ntri);
            uint32_t* orig = orig_mem.get();
            float* c2 = csum2.get();
            float* b2 = tbox2.get();
            parallel_for(ntri, 1 << 16, [&](size_t j) {
                const uint32_t t = ids[j];
                memcpy(&c2[j * 3], &csum[(size_t)t * 3], 3 * sizeof(float));
                memcpy(&b2[j * 6], &tbox[(size_t)t * 6], 6 * sizeof(float));
                orig[j] = t;
                ids[j] = (uint32_t)j;
            });
            csum_mem.swap(csum2);
            tbox_mem.swap(tbox2);
            csum = csum_mem.get();
            tbox = tbox_mem.get();
        }
        auto split_one = [&](size_t k) {
            TmpNode& n = nodes[splitting[k]];
            TmpNode& A = nodes[n.a];
            TmpNode& B = nodes[n.b];
            const int axis = axis_of(n);
            const uint32_t b = n.begin, e = n.end, cnt = e - b;
            const float mean = means[k];
            if (node_per_task) {
                // one pass: A's ids move forward in place (never past the read position), B's wait in tmp; the children's
                // boxes and their three serial sums (the next level's means, whichever axis it picks) are formed on the way
                float amn[3] = {1e12f, 1e12f, 1e12f}, amx[3] = {-1e12f, -1e12f, -1e12f};
                float bmn[3] = {1e12f, 1e12f, 1e12f}, bmx[3] = {-1e12f, -1e12f, -1e12f};
                float sa[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};
                uint32_t pa = b, pb = b;
                for (uint32_t i = b; i < e; i++) {
                    const uint32_t t = ids[i];
                    const float* cs = &csum[(size_t)t * 3];
                    const float* bx = &tbox[(size_t)t * 6];
                    if (cs[axis] / 3.0f < mean) {
                        ids[pa++] = t;
                        for (int c = 0; c < 3; c++) { sa[c] += cs[c]; amn[c] = fminf(amn[c], bx[c]); amx[c] = fmaxf(amx[c], bx[3 + c]); }
                    } else {
                        tmp[pb++] = t;
                        for (int c = 0; c < 3; c++) { sb[c] += cs[c]; bmn[c] = fminf(bmn[c], bx[c]); bmx[c] = fmaxf(bmx[c], bx[3 + c]); }
                    }
                }
                memcpy(&ids[pa], &tmp[b], (size_t)(pb - b) * sizeof(uint32_t));
                A.begin = b; A.end = pa;
                B.begin = pa; B.end = e;
                for (int c = 0; c < 3; c++) {
                    A.mn[c] = amn[c]; A.mx[c] = amx[c]; A.sum3[c] = sa[c];
                    B.mn[c] = bmn[c]; B.mx[c] = bmx[c]; B.sum3[c] = sb[c];
                }
                A.has_sum = B.has_sum = true;
                return;
            }
            // a big node: the threads share it.  Stable partition of ids[b, e): centroid < mean goes to A (:125-135)
            const size_t chunk = std::min(chunk_max, std::max<size_t>((size_t)1 << 14, (size_t)cnt / ((size_t)host_threads() * 4)));
            const size_t nchunks = ((size_t)cnt + chunk - 1) / chunk;
            std::vector<uint32_t> na(nchunks + 1, 0);
            auto count = [&](size_t c) {
                const size_t lo = b + c * chunk, hi = std::min<size_t>(e, lo + chunk);
                uint32_t m = 0;
                for (size_t i = lo; i < hi; i++) m += (csum[(size_t)ids[i] * 3 + axis] / 3.0f < mean) ? 1u : 0u;
                na[c + 1] = m;
            };
            if (nchunks > 1) parallel_for(nchunks, 1, count);
            else count(0);
            for (size_t c = 0; c < nchunks; c++) na[c + 1] += na[c];
            const uint32_t total_a = na[nchunks];
            std::vector<float> part(nchunks * 12);  // per chunk: box of its A part, box of its B part (min / max are exact: any grouping)
            auto scatter = [&](size_t c) {
                const size_t lo = b + c * chunk, hi = std::min<size_t>(e, lo + chunk);
                uint32_t pa = b + na[c], pb = b + total_a + (uint32_t)(lo - b) - na[c];
                float bb[12] = {1e12f, 1e12f, 1e12f, -1e12f, -1e12f, -1e12f, 1e12f, 1e12f, 1e12f, -1e12f, -1e12f, -1e12f};
                for (size_t i = lo; i < hi; i++) {
                    const uint32_t t = ids[i];
                    const float* bx = &tbox[(size_t)t * 6];
                    float* q;
                    if (csum[(size_t)t * 3 + axis] / 3.0f < mean) { tmp[pa++] = t; q = bb; }
                    else { tmp[pb++] = t; q = bb + 6; }
                    for (int d = 0; d < 3; d++) { q[d] = fminf(q[d], bx[d]); q[3 + d] = fmaxf(q[3 + d], bx[3 + d]); }
                }
                memcpy(&part[c * 12], bb, sizeof(bb));
            };
            if (nchunks > 1) parallel_for(nchunks, 1, scatter);
            else scatter(0);
            auto copy_back = [&](size_t c) {
                const size_t lo = b + c * chunk, hi = std::min<size_t>(e, lo + chunk);
                memcpy(&ids[lo], &tmp[lo], (hi - lo) * sizeof(uint32_t));
            };
            if (nchunks > 1) parallel_for(nchunks, 1, copy_back);
            else copy_back(0);
            A.begin = b; A.end = b + total_a;
            B.begin = b + total_a; B.end = e;
            for (int d = 0; d < 3; d++) { A.mn[d] = B.mn[d] = 1e12f; A.mx[d] = B.mx[d] = -1e12f; }
            for (size_t c = 0; c < nchunks; c++)
                for (int d = 0; d < 3; d++) {
                    A.mn[d] = fminf(A.mn[d], part[c * 12 + d]);     A.mx[d] = fmaxf(A.mx[d], part[c * 12 + 3 + d]);
                    B.mn[d] = fminf(B.mn[d], part[c * 12 + 6 + d]); B.mx[d] = fmaxf(B.mx[d], part[c * 12 + 9 + d]);
                }
        };
        if (!node_per_task) for (size_t k = 0; k < splitting.size(); k++) split_one(k);
        else parallel_for(splitting.size(), node_grain, split_one);
        level.clear();
        for (int ni : splitting) { level.push_back(nodes[ni].a); level.push_back(nodes[ni].b); }
        if (getenv("RT_SCENE_TIMING") && atoi(getenv("RT_SCENE_TIMING")) >= 2)
            fprintf(stderr, "[rt_scene]   level of %zu splitting nodes: %.1f ms (means %.1f)\n", splitting.size(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - lvl_t0).count(), lvl_means_ms);
    }
    delete tm_split;
    if (regrouped) {
        const uint32_t* orig = orig_mem.get();
        parallel_for(ntri, 1 << 16, [&](size_t j) { ids[j] = orig[ids[j]]; });
    }
    return finish_bvh(nodes, ids, tris, n_floats, out, out_depth, owner);
}

// NON-PARITY fast mode: a binned-SAH BVH2 in the same node / triangle layout, leaves of at most
// `leaf_tris` triangles, depth capped at `max_depth` (the traversal stack holds RT_STACK_ENTRIES).  The kernels and the
// oracle walk whatever tree the scene carries, so they still agree with each other bit for bit; what changes against
// the reference is the tree (visit counts, and which of two equally distant triangles wins).
int build_bvh_sah(float* tris, size_t n_floats, int max_depth, int leaf_tris, std::vector<rt_bvh_node>& out, int* out_depth) {
    if (n_floats % RT_TRI_FLOATS != 0) return fail("triangle array length is not a multiple of 25");
    const size_t ntri = n_floats / RT_TRI_FLOATS;
    if (ntri > 0x7fffffffu / RT_TRI_FLOATS) return fail("too many triangles for int32 float offsets");
    leaf_tris = std::max(leaf_tris, 1);
    std::vector<float> cen(ntri * 3), tbox(ntri * 6);
    std::vector<uint32_t> ids(ntri);
    parallel_for(ntri, 1 << 16, [&](size_t i) {
        const float* t = tris + i * RT_TRI_FLOATS;
        float* bx = &tbox[i * 6];
        for (int c = 0; c < 3; c++) {
            bx[c] = fminf(fminf(t[c], t[3 + c]), t[6 + c]);
            bx[3 + c] = fmaxf(fmaxf(t[c], t[3 + c]), t[6 + c]);
            cen[i * 3 + c] = 0.5f * (bx[c] + bx[3 + c]);
        }
        ids[i] = (uint32_t)i;
    });
    auto range_box = [&](uint32_t b, uint32_t e, float* mn, float* mx) {
        for (int c = 0; c < 3; c++) { mn[c] = 1e12f; mx[c] = -1e12f; }  // the reference's empty box
        for (uint32_t i = b; i < e; i++) {
            const float* bx = &tbox[(size_t)ids[i] * 6];
            for (int c = 0; c < 3; c++) { mn[c] = fminf(mn[c], bx[c]); mx[c] = fmaxf(mx[c], bx[3 + c]); }
        }
    };
    auto half_area = [](const float* mn, const float* mx) {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<TmpNode> nodes(1);
    nodes[0].begin = 0;
    nodes[0].end = (uint32_t)ntri;
    range_box(0, (uint32_t)ntri, nodes[0].mn, nodes[0].mx);
    StageTimer tm("bvh (sah): splits");
    std::vector<int> level{0}, splitting;
    std::vector<uint32_t> mids;
    while (!level.empty()) {
        splitting.clear();
        for (int ni : level) {
            const TmpNode& n = nodes[ni];
            if (n.depth >= max_depth || (int)(n.end - n.begin) <= leaf_tris) continue;
            splitting.push_back(ni);
        }
        mids.assign(splitting.size(), 0);
        auto choose = [&](size_t k) {  // partitions ids[begin, end) and records the split position
            const TmpNode& n = nodes[splitting[k]];
            const uint32_t b = n.begin, e = n.end, cnt = e - b;
            float cmn[3] = {1e30f, 1e30f, 1e30f}, cmx[3] = {-1e30f, -1e30f, -1e30f};
            for (uint32_t i = b; i < e; i++)
                for (int c = 0; c < 3; c++) {
                    cmn[c] = fminf(cmn[c], cen[(size_t)ids[i] * 3 + c]);
                    cmx[c] = fmaxf(cmx[c], cen[(size_t)ids[i] * 3 + c]);
                }
            // depth budget: once the remaining levels only just suffice for a balanced tree, split at the median
            int need = 0;
            while (((size_t)leaf_tris << need) < cnt) need++;
            const bool forced_median = need >= max_depth - n.depth;
            int best_axis = -1, best_bin = -1;
            constexpr int NB = 16;
            if (!forced_median) {
                float best_cost = (float)cnt * half_area(n.mn, n.mx);  // cost of not splitting (x traversal weight 1)
                for (int axis = 0; axis < 3; axis++) {
                    const float w = cmx[axis] - cmn[axis];
                    if (!(w > 0.0f)) continue;
                    uint32_t bc[NB] = {0};
                    float bmn[NB][3], bmx[NB][3];
                    for (int j = 0; j < NB; j++)
                        for (int c = 0; c < 3; c++) { bmn[j][c] = 1e30f; bmx[j][c] = -1e30f; }
                    const float scale = (float)NB / w;
                    for (uint32_t i = b; i < e; i++) {
                        const uint32_t t = ids[i];
                        int j = (int)((cen[(size_t)t * 3 + axis] - cmn[axis]) * scale);
                        j = std::min(std::max(j, 0), NB - 1);
                        bc[j]++;
                        const float* bx = &tbox[(size_t)t * 6];
                        for (int c = 0; c < 3; c++) { bmn[j][c] = fminf(bmn[j][c], bx[c]); bmx[j][c] = fmaxf(bmx[j][c], bx[3 + c]); }
                    }
                    float ra[NB];  // half area of bins j..NB-1
                    uint32_t rc[NB];
                    float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
                    uint32_t cc = 0;
                    for (int j = NB - 1; j >= 0; j--) {
                        for (int c = 0; c < 3; c++) { mn[c] = fminf(mn[c], bmn[j][c]); mx[c] = fmaxf(mx[c], bmx[j][c]); }
                        cc += bc[j];
                        ra[j] = cc ? half_area(mn, mx) : 0.0f;
                        rc[j] = cc;
                    }
                    for (int c = 0; c < 3; c++) { mn[c] = 1e30f; mx[c] = -1e30f; }
                    cc = 0;
                    for (int j = 0; j < NB - 1; j++) {  // split after bin j
                        for (int c = 0; c < 3; c++) { mn[c] = fminf(mn[c], bmn[j][c]); mx[c] = fmaxf(mx[c], bmx[j][c]); }
                        cc += bc[j];
                        if (cc == 0 || rc[j + 1] == 0) continue;
                        const float cost = half_area(n.mn, n.mx) * 1.0f + (float)cc * half_area(mn, mx) + (float)rc[j + 1] * ra[j + 1];
                        if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = j; }
                    }
                }
            }
            uint32_t mid;
            if (best_axis >= 0) {
                const float w = cmx[best_axis] - cmn[best_axis], scale = (float)NB / w;
                const float lo = cmn[best_axis];
                uint32_t* first = &ids[b];
                uint32_t* m = std::partition(first, first + cnt, [&](uint32_t t) {
                    int j = (int)((cen[(size_t)t * 3 + best_axis] - lo) * scale);
                    j = std::min(std::max(j, 0), NB - 1);
                    return j <= best_bin;
                });
                mid = b + (uint32_t)(m - first);
            } else {  // median along the widest centroid axis (or no SAH gain on a node that must still shrink)
                int axis = 0;
                if (cmx[1] - cmn[1] > cmx[axis] - cmn[axis]) axis = 1;
                if (cmx[2] - cmn[2] > cmx[axis] - cmn[axis]) axis = 2;
                mid = b + cnt / 2;
                std::nth_element(&ids[b], &ids[mid], &ids[b] + cnt, [&](uint32_t x, uint32_t y) {
                    const float cx = cen[(size_t)x * 3 + axis], cy = cen[(size_t)y * 3 + axis];
                    return cx < cy || (cx == cy && x < y);
                });
            }
            mids[k] = mid;
        };
        if (splitting.size() < (size_t)host_threads() * 2) for (size_t k = 0; k < splitting.size(); k++) choose(k);
        else parallel_for(splitting.size(), 1, choose);
        const size_t first_child = nodes.size();
        nodes.resize(first_child + 2 * splitting.size());
        parallel_for(splitting.size(), 4, [&](size_t k) {
            TmpNode& n = nodes[splitting[k]];
            n.a = (int)(first_child + 2 * k);
            n.b = n.a + 1;
            TmpNode& A = nodes[n.a];
            TmpNode& B = nodes[n.b];
            A.depth = B.depth = n.depth + 1;
            A.begin = n.begin; A.end = mids[k];
            B.begin = mids[k]; B.end = n.end;
            range_box(A.begin, A.end, A.mn, A.mx);
            range_box(B.begin, B.end, B.mn, B.mx);
        });
        level.clear();
        for (int ni : splitting) { level.push_back(nodes[ni].a); level.push_back(nodes[ni].b); }
    }
    return finish_bvh(nodes, ids.data(), tris, n_floats, out, out_depth);
}

// ------------------------------------------------------------------ scene DSL

// A scene line cut into words once (the DSL of sceneParser.cpp:22-171 separates words by ONE space, so two spaces
// in a row hold an empty word between them, as the reference's per-call word lookup sees it too).
struct Words {
    std::vector<std::string> w;
    explicit Words(const std::string& line) {
        std::string cur;
        for (char ch : line) {
            if (ch == ' ') { w.push_back(cur); cur.clear(); }
            else cur.push_back(ch);
        }
        w.push_back(cur);
    }
    const std::string& operator[](size_t n) const {
        static const std::string none;
        return n < w.size() ? w[n] : none;
    }
};

struct ATexture { std::string name, path; bool loaded = false; int offset = 0, w = 0, h = 0; };
struct AModel { std::string name, path; };
// Members the reference leaves uninitialised (sceneParser.h:25-33, 39-46)
// default to 0 / scale 1 here.
struct AMaterial { std::string name; int type = 0; V3 color; float p1 = 0, p2 = 0, p3 = 0; std::vector<std::string> images; };
struct AObject { std::string name, model, material; };
struct AInstance { std::string object; V3 pos, rot; float scale = 1.0f; bool importance = false; };
struct ASphere { V3 pos; float radius = 0.f; std::string material; };

}  // namespace

struct rt_host_scene {
    rt_camera cam{};
    FloatVec tris;  // sphere prefix + records
    int n_spheres = 0;
    std::vector<rt_bvh_node> nodes;
    std::vector<rt_material> mats;
    FloatVec textures;  // (every float is written right after the array is sized)
    std::vector<int32_t> lights;
    int bvh_depth = 0;
};

namespace {

bool stof_ok(const std::string& s, float& out) {
    try {
        out = std::stof(s);
        return true;
    } catch (...) {
        return false;
    }
}
bool stoi_ok(const std::string& s, int& out) {
    try {
        out = std::stoi(s);
        return true;
    } catch (...) {
        return false;
    }
}

struct Parser {
    rt_camera cam{};
    std::vector<ATexture> textures;
    std::vector<AModel> models;
    std::vector<AMaterial> materials;
    std::vector<AObject> objects;
    std::vector<AInstance> instances;
    std::vector<ASphere> spheres;
    std::string background;
    std::string err;
    bool enable_spheres = false;

    Parser() {
        // SceneConfigs defaults, sceneConfigs.h:6-17
        cam.width = 512; cam.height = 512; cam.num_samples = 1000;
        cam.focal_length = 1.0f; cam.aperture = 0.0f; cam.focus_distance = 4.0f;
        cam.env_width = 0; cam.env_height = 0;  // (reference: 4096x2048 even without an env map; forced to 0 = none)
    }

    bool vec3(const Words& line, V3& v) { return stof_ok(line[1], v.x) && stof_ok(line[2], v.y) && stof_ok(line[3], v.z); }
    bool vec3(const Words& line, float* v) {
        V3 t;
        if (!vec3(line, t)) return false;
        v[0] = t.x; v[1] = t.y; v[2] = t.z;
        return true;
    }

    // sceneParser.cpp:22-171
    bool parse(const std::string& path) {
        std::ifstream file(path);
        if (!file.is_open()) { err = "could not open scene file: " + path; return false; }
        enum Scope { GLOBAL, IMAGE, MODEL, MATERIAL, OBJECT, INSTANCE, SPHERE } scope = GLOBAL;
        std::string line;
        int lineno = 0;
        auto bad = [&](const std::string& what) {
            err = path + ":" + std::to_string(lineno) + ": " + what;
            return false;
        };
        while (std::getline(file, line)) {
            lineno++;
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) { scope = GLOBAL; continue; }
            if (line[0] == '#') continue;
            const Words words(line);
            const std::string& w0 = words[0];
            bool ok = true;
            switch (scope) {
                case IMAGE:
                    if (w0 == "PATH") textures.back().path = words[1];
                    else return bad("unknown IMAGE definition command: " + w0);
                    continue;
                case MODEL:
                    if (w0 == "PATH") models.back().path = words[1];
                    else return bad("unknown MODEL definition command: " + w0);
                    continue;
                case MATERIAL: {
                    AMaterial& m = materials.back();
                    if (w0 == "TYPE") ok = stoi_ok(words[1], m.type);
                    else if (w0 == "COLOR") ok = vec3(words, m.color);
                    else if (w0 == "P1") ok = stof_ok(words[1], m.p1);
                    else if (w0 == "P2") ok = stof_ok(words[1], m.p2);
                    else if (w0 == "P3") ok = stof_ok(words[1], m.p3);
                    else if (w0 == "TEXTURE") {
                        if (m.images.size() >= 3) return bad("too many textures for material: " + m.name);
                        m.images.push_back(words[1]);
                    } else return bad("unknown MATERIAL definition command: " + w0);
                    if (!ok) return bad("bad number");
                    continue;
                }
                case OBJECT:
                    if (w0 == "MODEL") objects.back().model = words[1];
                    else if (w0 == "MATERIAL") objects.back().material = words[1];
                    else return bad("unknown OBJECT definition command: " + w0);
                    continue;
                case INSTANCE: {
                    AInstance& in = instances.back();
                    if (w0 == "OBJECT") in.object = words[1];
                    else if (w0 == "POSITION") ok = vec3(words, in.pos);
                    else if (w0 == "SCALE") ok = stof_ok(words[1], in.scale);
                    else if (w0 == "ROTATION") ok = vec3(words, in.rot);
                    else if (w0 == "IMPORTANCE_SAMPLE") in.importance = true;
                    else return bad("unknown OBJECT INSTANCE definition command: " + w0);
                    if (!ok) return bad("bad number");
                    continue;
                }
                case SPHERE: {
                    if (!enable_spheres) continue;  // sceneParser.cpp:285
                    ASphere& s = spheres.back();
                    if (w0 == "POSITION") ok = vec3(words, s.pos);
                    else if (w0 == "RADIUS") ok = stof_ok(words[1], s.radius);
                    else if (w0 == "MATERIAL") s.material = words[1];
                    else return bad("unknown SPHERE definition command: " + w0);
                    if (!ok) return bad("bad number");
                    continue;
                }
                case GLOBAL: break;
            }
            if (w0 == "SET_OUTPUT_WIDTH") ok = stoi_ok(words[1], cam.width);
            else if (w0 == "SET_OUTPUT_HEIGHT") ok = stoi_ok(words[1], cam.height);
            else if (w0 == "SET_NUM_SAMPLES") ok = stoi_ok(words[1], cam.num_samples);
            else if (w0 == "SET_CAM_POS") ok = vec3(words, cam.pos);
            else if (w0 == "SET_CAM_ROTATION") ok = vec3(words, cam.rot);
            else if (w0 == "SET_CAM_VEL") ok = vec3(words, cam.vel);
            else if (w0 == "SET_CAM_ANGULAR_VEL") ok = vec3(words, cam.ang_vel);
            else if (w0 == "SET_FOCAL_LENGTH") ok = stof_ok(words[1], cam.focal_length);
            else if (w0 == "SET_APERTURE_RADIUS") ok = stof_ok(words[1], cam.aperture);
            else if (w0 == "SET_FOCUS_DISTANCE") ok = stof_ok(words[1], cam.focus_distance);
            else if (w0 == "SET_BACKGROUND") background = words[1];
            else if (w0 == "ADD_OBJECT") { scope = INSTANCE; instances.emplace_back(); }
            else if (w0 == "DEF_IMAGE") { scope = IMAGE; textures.emplace_back(); textures.back().name = words[1]; }
            else if (w0 == "DEF_MODEL") { scope = MODEL; models.emplace_back(); models.back().name = words[1]; }
            else if (w0 == "DEF_MATERIAL") { scope = MATERIAL; materials.emplace_back(); materials.back().name = words[1]; }
            else if (w0 == "DEF_OBJ") { scope = OBJECT; objects.emplace_back(); objects.back().name = words[1]; }
            else if (w0 == "ADD_SPHERE") { scope = SPHERE; spheres.emplace_back(); }
            else return bad("unknown command: " + w0);
            if (!ok) return bad("bad number");
        }
        return true;
    }

    template <class T>
    T* by_name(std::vector<T>& v, const std::string& name) {
        for (auto& e : v)
            if (e.name == name) return &e;
        return nullptr;
    }
    int material_index(const std::string& name) {
        for (size_t i = 0; i < materials.size(); i++)
            if (materials[i].name == name) return (int)i;
        return -1;
    }
};

std::string resolve_path(const std::string& p, const char* asset_root) {
    if (!asset_root) return p;
    size_t cut = p.find_last_of("/\\");
    std::string base = cut == std::string::npos ? p : p.substr(cut + 1);
    std::string root(asset_root);
    if (!root.empty() && root.back() != '/') root.push_back('/');
    return root + base;
}

// Vec3::colorToInt, vec3.h:111-116
int32_t color_to_int(const V3& c) {
    const int r = static_cast<int>(fminf(fmaxf(c.x * 255.0f, 0.0f), 255.0f));
    const int g = static_cast<int>(fminf(fmaxf(c.y * 255.0f, 0.0f), 255.0f));
    const int b = static_cast<int>(fminf(fmaxf(c.z * 255.0f, 0.0f), 255.0f));
    return (r << 16) | (g << 8) | b;
}

rt_material blank_material() {
    rt_material m;
    memset(&m, 0, sizeof(m));
    m.color = 0xFFFFFF;
    m.image_offset[0] = m.image_offset[1] = m.image_offset[2] = -1;
    return m;
}

// light list + marker removal, sceneParser.cpp:455-470
void extract_lights(float* tris, size_t n_floats, std::vector<int32_t>& lights) {
    lights.clear();
    const size_t ntri = n_floats / RT_TRI_FLOATS, chunk = (size_t)1 << 18, nch = (ntri + chunk - 1) / chunk;
    std::vector<std::vector<int32_t>> part(nch);
    parallel_for(nch, 1, [&](size_t c) {
        for (size_t t = c * chunk; t < std::min(ntri, (c + 1) * chunk); t++) {
            const size_t i = t * RT_TRI_FLOATS;
            if (tris[i + 24] > 1000.0f) {
                tris[i + 24] -= 1000.0f;
                part[c].push_back(static_cast<int32_t>(i));
            }
        }
    });
    for (auto& p : part) lights.insert(lights.end(), p.begin(), p.end());  // ascending offsets, like the serial scan
}

int finish_geometry(rt_host_scene* sc, FloatVec& records, int max_depth, int min_tris) {
    // prefix (sphere block) already in sc->tris
    const size_t prefix = sc->tris.size();
    if (prefix == 0) {
        sc->tris.swap(records);  // no sphere block in front: take the records as they are
    } else {
        sc->tris.resize(prefix + records.size());
        if (!records.empty()) memcpy(sc->tris.data() + prefix, records.data(), records.size() * sizeof(float));
    }
    FloatVec().swap(records);
    int rc = build_bvh(sc->tris.data() + prefix, sc->tris.size() - prefix, max_depth, min_tris, sc->nodes, &sc->bvh_depth, &sc->tris);
    if (rc != RT_OK) return rc;
    extract_lights(sc->tris.data() + prefix, sc->tris.size() - prefix, sc->lights);
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_scene_last_error(void) { return g_err.c_str(); }

void rt_host_free(void* p) { free(p); }

void rt_scene_free(rt_host_scene* s) { delete s; }

int rt_scene_get_desc(const rt_host_scene* s, rt_scene_desc* out) {
    if (!s || !out) return fail("null argument");
    const size_t prefix = s->n_spheres > 0 ? 1 + 5 * (size_t)s->n_spheres : 0;
    out->tris = s->tris.data();
    out->n_tri_floats = s->tris.size() - prefix;
    out->n_spheres = s->n_spheres;
    out->nodes = s->nodes.data();
    out->n_nodes = (int32_t)s->nodes.size();
    out->mats = s->mats.data();
    out->n_mats = s->mats.size();
    out->textures = s->textures.data();
    out->n_tex_floats = s->textures.size();
    out->light_tris = s->lights.data();
    out->n_light_tris = (int32_t)s->lights.size();
    return RT_OK;
}

int rt_scene_get_camera(const rt_host_scene* s, rt_camera* out) {
    if (!s || !out) return fail("null argument");
    *out = s->cam;
    return RT_OK;
}

int rt_scene_bvh_depth(const rt_host_scene* s) { return s ? s->bvh_depth : -1; }

int rt_bvh_build(float* tris, size_t n_floats, int max_depth, int min_tris, rt_bvh_node** out_nodes,
                 int32_t* out_n_nodes) {
    if (!tris && n_floats) return fail("null triangle array");
    if (!out_nodes || !out_n_nodes) return fail("null output");
    if (max_depth <= 0) max_depth = 16;
    if (min_tris <= 0) min_tris = 4;
    std::vector<rt_bvh_node> nodes;
    int rc = build_bvh(tris, n_floats, max_depth, min_tris, nodes, nullptr);
    if (rc != RT_OK) return rc;
    rt_bvh_node* p = (rt_bvh_node*)malloc(nodes.size() * sizeof(rt_bvh_node));
    if (!p) return fail("out of memory");
    memcpy(p, nodes.data(), nodes.size() * sizeof(rt_bvh_node));
    *out_nodes = p;
    *out_n_nodes = (int32_t)nodes.size();
    return RT_OK;
}

int rt_bvh_build_sah(float* tris, size_t n_floats, int max_depth, int leaf_tris, rt_bvh_node** out_nodes,
                     int32_t* out_n_nodes) {
    if (!tris && n_floats) return fail("null triangle array");
    if (!out_nodes || !out_n_nodes) return fail("null output");
    if (max_depth <= 0) max_depth = 24;
    if (leaf_tris <= 0) leaf_tris = 2;
    std::vector<rt_bvh_node> nodes;
    int rc = build_bvh_sah(tris, n_floats, max_depth, leaf_tris, nodes, nullptr);
    if (rc != RT_OK) return rc;
    rt_bvh_node* p = (rt_bvh_node*)malloc(nodes.size() * sizeof(rt_bvh_node));
    if (!p) return fail("out of memory");
    memcpy(p, nodes.data(), nodes.size() * sizeof(rt_bvh_node));
    *out_nodes = p;
    *out_n_nodes = (int32_t)nodes.size();
    return RT_OK;
}

int rt_scene_rebuild_bvh(rt_host_scene* s, int mode, int max_depth, int leaf_tris) {
    if (!s) return fail("null scene");
    if (mode != RT_BVH_REFERENCE && mode != RT_BVH_SAH) return fail("unknown BVH mode");
    const size_t prefix = s->n_spheres > 0 ? 1 + 5 * (size_t)s->n_spheres : 0;
    float* rec = s->tris.data() + prefix;
    const size_t n = s->tris.size() - prefix;
    for (int32_t off : s->lights) rec[(size_t)off + 24] += 1000.0f;  // the marker extract_lights looks for (sceneParser.cpp:408-410)
    int rc;
    if (mode == RT_BVH_SAH) rc = build_bvh_sah(rec, n, max_depth > 0 ? max_depth : 24, leaf_tris > 0 ? leaf_tris : 2, s->nodes, &s->bvh_depth);
    else rc = build_bvh(rec, n, max_depth > 0 ? max_depth : 16, leaf_tris > 0 ? leaf_tris : 4, s->nodes, &s->bvh_depth);
    if (rc != RT_OK) return rc;
    extract_lights(rec, n, s->lights);
    return RT_OK;
}

int rt_obj_load(const char* path, const float translation[3], float scale, const float rotation[3],
                int material_index, float** out_tris, size_t* out_n_floats) {
    if (!path || !translation || !rotation || !out_tris || !out_n_floats) return fail("null argument");
    ParsedObj o;
    std::string err;
    if (!parse_obj(path, o, err)) return fail(err);
    std::vector<float> rec;
    emit_records(o, V3{translation[0], translation[1], translation[2]}, scale,
                 V3{rotation[0], rotation[1], rotation[2]}, material_index, rec);
    float* p = (float*)malloc(std::max<size_t>(rec.size(), 1) * sizeof(float));
    if (!p) return fail("out of memory");
    if (!rec.empty()) memcpy(p, rec.data(), rec.size() * sizeof(float));
    *out_tris = p;
    *out_n_floats = rec.size();
    return RT_OK;
}

int rt_noise_texture(int w, int h, float* out_rgb) {
    if (w <= 0 || h <= 0 || !out_rgb) return fail("bad noise texture arguments");
    noise_generate(w, h, out_rgb);
    return RT_OK;
}

int rt_image_load(const char* path, float** out_rgb, int* w, int* h) {
    if (!path || !out_rgb || !w || !h) return fail("null argument");
    Image img;
    std::string err;
    if (!load_image(path, img, err)) return fail(err);
    float* p = (float*)malloc(img.rgb.size() * sizeof(float));
    if (!p) return fail("out of memory");
    memcpy(p, img.rgb.data(), img.rgb.size() * sizeof(float));
    *out_rgb = p;
    *w = img.w;
    *h = img.h;
    return RT_OK;
}

namespace {
// Minimal PNG encoder (8-bit RGB, zlib "stored" blocks): enough for the reference's stbi_write_png output stage.
uint32_t crc32_update(uint32_t crc, const unsigned char* d, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ d[i]) & 0xff] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<unsigned char>& v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
void png_chunk(FILE* f, const char* type, const std::vector<unsigned char>& data) {
    std::vector<unsigned char> hdr;
    put_be32(hdr, (uint32_t)data.size());
    fwrite(hdr.data(), 1, 4, f);
    fwrite(type, 1, 4, f);
    if (!data.empty()) fwrite(data.data(), 1, data.size(), f);
    uint32_t crc = crc32_update(0xffffffffu, (const unsigned char*)type, 4);
    if (!data.empty()) crc = crc32_update(crc, data.data(), data.size());
    crc ^= 0xffffffffu;
    std::vector<unsigned char> tail;
    put_be32(tail, crc);
    fwrite(tail.data(), 1, 4, f);
}
void write_png(FILE* f, const unsigned char* rgb8, int w, int h) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    fwrite(sig, 1, 8, f);
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, (uint32_t)w);
    put_be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(f, "IHDR", ihdr);
    std::vector<unsigned char> raw;  // filter byte 0 + row
    raw.reserve((size_t)h * (3 * (size_t)w + 1));
    for (int y = 0; y < h; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb8 + (size_t)y * w * 3, rgb8 + (size_t)(y + 1) * w * 3);
    }
    std::vector<unsigned char> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    size_t pos = 0;
    do {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((unsigned char)(n & 0xff)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xff)); z.push_back((unsigned char)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
    } while (pos < raw.size());
    put_be32(z, (b << 16) | a);
    png_chunk(f, "IDAT", z);
    png_chunk(f, "IEND", std::vector<unsigned char>());
}
}  // namespace

int rt_image_save(const char* path, const float* rgb, int w, int h) {
    if (!path || !rgb || w <= 0 || h <= 0) return fail("bad image arguments");
    const std::string p(path);
    FILE* f = fopen(path, "wb");
    if (!f) return fail("could not open for writing: " + p);
    const size_t n = (size_t)w * h * 3;
    if (ends_with(p, ".png")) {
        std::vector<unsigned char> img(n);
        // texture.cpp:55: static_cast<unsigned char>(data[i] * 255), no clamp
        for (size_t i = 0; i < n; i++) img[i] = (unsigned char)(int)(rgb[i] * 255);
        write_png(f, img.data(), w, h);
    } else if (ends_with(p, ".ppm")) {
        fprintf(f, "P6\n%d %d\n255\n", w, h);
        std::vector<unsigned char> img(n);
        // texture.cpp:55: static_cast<unsigned char>(data[i] * 255), no clamp.
        // (float -> int -> low 8 bits keeps that defined for out-of-range input.)
        for (size_t i = 0; i < n; i++) img[i] = (unsigned char)(int)(rgb[i] * 255);
        fwrite(img.data(), 1, n, f);
    } else if (ends_with(p, ".pfm")) {
        fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
        for (int y = h - 1; y >= 0; y--) fwrite(rgb + (size_t)y * w * 3, sizeof(float), (size_t)w * 3, f);
    } else if (ends_with(p, ".hdr")) {
        fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", h, w);
        for (size_t i = 0; i < (size_t)w * h; i++) {
            const float r = rgb[i * 3], g = rgb[i * 3 + 1], b = rgb[i * 3 + 2];
            const float m = fmaxf(r, fmaxf(g, b));
            unsigned char px[4] = {0, 0, 0, 0};
            if (m >= 1e-32f) {
                int e;
                const float norm = frexpf(m, &e) * 256.0f / m;
                px[0] = (unsigned char)(r * norm);
                px[1] = (unsigned char)(g * norm);
                px[2] = (unsigned char)(b * norm);
                px[3] = (unsigned char)(e + 128);
            }
            fwrite(px, 1, 4, f);
        }
    } else {
        fclose(f);
        return fail("unsupported output format (need .png/.ppm/.pfm/.hdr): " + p);
    }
    fclose(f);
    return RT_OK;
}

int rt_image_save_rgb8(const char* path, const unsigned char* rgb8, int w, int h) {
    if (!path || !rgb8 || w <= 0 || h <= 0) return fail("bad image arguments");
    const std::string p(path);
    const bool png = ends_with(p, ".png");
    if (!png && !ends_with(p, ".ppm")) return fail("unsupported 8-bit output format (need .png/.ppm): " + p);
    FILE* f = fopen(path, "wb");
    if (!f) return fail("could not open for writing: " + p);
    if (png) {
        write_png(f, rgb8, w, h);
    } else {
        fprintf(f, "P6\n%d %d\n255\n", w, h);
        fwrite(rgb8, 1, (size_t)w * h * 3, f);
    }
    fclose(f);
    return RT_OK;
}

int rt_scene_from_records(const float* tris, size_t n_floats, const rt_material* mats, size_t n_mats,
                          const float* textures, size_t n_tex_floats, int env_w, int env_h,
                          int bvh_max_depth, int bvh_min_tris, rt_host_scene** out) {
    if (!out || (!tris && n_floats) || (!mats && n_mats)) return fail("null argument");
    if (n_floats % RT_TRI_FLOATS) return fail("triangle array length is not a multiple of 25");
    if (bvh_max_depth <= 0) bvh_max_depth = 16;
    if (bvh_min_tris <= 0) bvh_min_tris = 4;
    rt_host_scene* sc = new rt_host_scene();
    sc->cam.width = 512; sc->cam.height = 512; sc->cam.num_samples = 16;
    sc->cam.focal_length = 1.0f; sc->cam.focus_distance = 4.0f;
    sc->cam.env_width = env_w; sc->cam.env_height = env_h;
    FloatVec rec(tris, tris + n_floats);
    int rc = finish_geometry(sc, rec, bvh_max_depth, bvh_min_tris);
    if (rc != RT_OK) { delete sc; return rc; }
    sc->mats.assign(mats, mats + n_mats);
    if (textures && n_tex_floats) sc->textures.assign(textures, textures + n_tex_floats);
    *out = sc;
    return RT_OK;
}

int rt_scene_load(const char* scene_path, const char* asset_root, const char* env_override,
                  int enable_spheres, int bvh_max_depth, int bvh_min_tris, rt_host_scene** out) {
    if (!scene_path || !out) return fail("null argument");
    if (bvh_max_depth <= 0) bvh_max_depth = 16;
    if (bvh_min_tris <= 0) bvh_min_tris = 4;
    Parser P;
    P.enable_spheres = enable_spheres != 0;
    if (!P.parse(scene_path)) return fail(P.err);

    rt_host_scene* sc = new rt_host_scene();
    auto die = [&](const std::string& m) { delete sc; return fail(m); };

    // ---- getTriangleData, sceneParser.cpp:398-475 (parsed meshes are cached per model) ----
    std::vector<std::pair<std::string, ParsedObj>> cache;
    FloatVec records;
    for (const AInstance& in : P.instances) {
        AObject* obj = P.by_name(P.objects, in.object);
        if (!obj) return die("object not found: " + in.object);
        int mi = P.material_index(obj->material);
        if (mi < 0) return die("material not found: " + obj->material);
        if (in.importance) mi += 1000;  // :408-410
        AModel* model = P.by_name(P.models, obj->model);
        if (!model) return die("model not found: " + obj->model);
        const std::string path = resolve_path(model->path, asset_root);
        const ParsedObj* po = nullptr;
        for (auto& c : cache)
            if (c.first == path) po = &c.second;
        if (!po) {
            cache.emplace_back(path, ParsedObj());
            std::string err;
            if (!parse_obj(path, cache.back().second, err)) return die(err);
            po = &cache.back().second;
        }
        emit_records(*po, in.pos, in.scale, in.rot, mi, records);
    }
    // sphere prefix, sceneParser.cpp:423-453
    if (P.enable_spheres) {
        sc->n_spheres = (int)P.spheres.size();
        if (sc->n_spheres > 0) {
            sc->tris.push_back(static_cast<float>(P.spheres.size()) + 0.0001f);
            for (const ASphere& s : P.spheres) {
                const int mi = P.material_index(s.material);
                if (mi < 0) return die("material not found: " + s.material);
                sc->tris.push_back(s.pos.x);
                sc->tris.push_back(s.pos.y);
                sc->tris.push_back(s.pos.z);
                sc->tris.push_back(s.radius);
                sc->tris.push_back(static_cast<float>(mi) + 0.0001f);
            }
        }
    }
    int rc = finish_geometry(sc, records, bvh_max_depth, bvh_min_tris);
    if (rc != RT_OK) { delete sc; return rc; }

    // ---- getMaterialData, sceneParser.cpp:477-517 ----
    sc->cam = P.cam;
    FloatVec& pixels = sc->textures;
    auto load_into_pool = [&](ATexture* t, const std::string& path, std::string& err) {
        Image img;
        if (!load_image(path, img, err)) return false;
        t->offset = (int)pixels.size();
        pixels.insert(pixels.end(), img.rgb.begin(), img.rgb.end());
        t->loaded = true;
        t->w = img.w;
        t->h = img.h;
        return true;
    };
    if (!P.background.empty()) {
        ATexture* t = P.by_name(P.textures, P.background);
        if (!t) return die("texture not found: " + P.background);
        std::string err;
        const std::string path = env_override ? std::string(env_override) : resolve_path(t->path, asset_root);
        if (!load_into_pool(t, path, err)) return die(err);
        sc->cam.env_width = t->w;
        sc->cam.env_height = t->h;
    }
    for (const AMaterial& am : P.materials) {
        rt_material m = blank_material();
        m.type = (int8_t)am.type;
        m.color = color_to_int(am.color);
        m.p1 = am.p1; m.p2 = am.p2; m.p3 = am.p3;
        for (size_t j = 0; j < am.images.size(); j++) {
            const std::string& name = am.images[j];
            if (name == "NOISE") {  // tryLoadTexture, sceneParser.cpp:520-534: always slot 0, 512x512
                m.image_offset[0] = (int32_t)pixels.size();
                m.image_width = 512;
                m.image_height = 512;
                const size_t base = pixels.size();
                pixels.resize(base + (size_t)512 * 512 * 3);
                noise_generate(512, 512, pixels.data() + base);
                continue;
            }
            ATexture* t = P.by_name(P.textures, name);
            if (!t) return die("texture not found: " + name);
            if (!t->loaded) {
                std::string err;
                if (!load_into_pool(t, resolve_path(t->path, asset_root), err)) return die(err);
            }
            m.image_offset[j] = t->offset;
            if (m.image_height != 0 && (m.image_height != t->h || m.image_width != t->w))
                return die("textures must have the same dimensions for a material: " + am.name);
            if (t->w > 65535 || t->h > 65535) return die("texture too large for 16-bit dimensions: " + name);
            m.image_width = (uint16_t)t->w;
            m.image_height = (uint16_t)t->h;
        }
        sc->mats.push_back(m);
    }
    *out = sc;
    return RT_OK;
}

int rt_scene_synth_field(const char* obj_path, int instances, uint32_t seed, const char* env_path,
                         int env_upsample, int bvh_max_depth, int bvh_min_tris, rt_host_scene** out) {
    if (!obj_path || !out || instances <= 0) return fail("bad synth arguments");
    if (bvh_max_depth <= 0) bvh_max_depth = 16;
    if (bvh_min_tris <= 0) bvh_min_tris = 4;
    ParsedObj po;
    std::string err;
    if (!parse_obj(obj_path, po, err)) return fail(err);
    // object-space bounds to normalise the mesh to a unit footprint
    float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
    for (size_t i = 0; i < po.v.size(); i += 3)
        for (int c = 0; c < 3; c++) {
            mn[c] = fminf(mn[c], po.v[i + c]);
            mx[c] = fmaxf(mx[c], po.v[i + c]);
        }
    const float extent = fmaxf(mx[0] - mn[0], fmaxf(mx[1] - mn[1], mx[2] - mn[2]));
    if (!(extent > 0)) return fail("degenerate mesh: " + std::string(obj_path));
    // recentre the raw vertices so rotation happens about the mesh centre, base at y = 0
    for (size_t i = 0; i < po.v.size(); i += 3) {
        po.v[i] -= 0.5f * (mn[0] + mx[0]);
        po.v[i + 1] -= mn[1];
        po.v[i + 2] -= 0.5f * (mn[2] + mx[2]);
    }
    rt_host_scene* sc = new rt_host_scene();
    const int side = (int)ceil(sqrt((double)instances));
    const float cell = 1.6f;
    std::mt19937 rng(seed);
    auto uni = [&]() { return (float)(rng() >> 8) * (1.0f / 16777216.0f); };
    // the random draws in instance order (one stream), then the instances in parallel
    struct Inst { V3 pos, rot; float scale; int mat; };
    std::vector<Inst> inst((size_t)instances);
    const float scale = 1.0f / extent;
    for (int k = 0; k < instances; k++) {
        const int gx = k % side, gz = k / side;
        Inst& in = inst[(size_t)k];
        in.pos.x = (gx - 0.5f * (side - 1)) * cell + (uni() - 0.5f) * 0.4f;
        in.pos.y = 0.0f;
        in.pos.z = -3.0f - gz * cell + (uni() - 0.5f) * 0.4f;
        in.rot = V3{0.f, uni() * 6.2831853f, 0.f};
        in.mat = 1 + (k % 3);
        in.scale = scale * (0.8f + 0.6f * uni());
    }
    const size_t per_inst = po.idx.size() / 9 * RT_TRI_FLOATS;
    FloatVec records;
    records.reserve((size_t)instances * per_inst + 2 * RT_TRI_FLOATS);
    records.resize((size_t)instances * per_inst);
    StageTimer* tm_emit = new StageTimer("synth: instance records");
    parallel_for((size_t)instances, 1, [&](size_t k) {
        const Inst& in = inst[k];
        emit_records_to(po, in.pos, in.scale, in.rot, in.mat, records.data() + k * per_inst);
    });
    {   // ground quad (two triangles), material 0
        const float hx = 0.5f * side * cell + 4.0f, z0 = 2.0f, z1 = -3.0f - side * cell - 4.0f;
        const float q[4][3] = {{-hx, 0.f, z0}, {hx, 0.f, z0}, {hx, 0.f, z1}, {-hx, 0.f, z1}};
        const int tri[2][3] = {{0, 1, 2}, {0, 2, 3}};
        for (int t = 0; t < 2; t++) {
            float r[RT_TRI_FLOATS] = {0};
            for (int v = 0; v < 3; v++) {
                for (int c = 0; c < 3; c++) r[v * 3 + c] = q[tri[t][v]][c];
                r[9 + v * 3 + 1] = 1.0f;
                r[18 + v * 2] = (q[tri[t][v]][0] + hx) / (2 * hx);
                r[19 + v * 2] = (q[tri[t][v]][2] - z1) / (z0 - z1);
            }
            r[24] = 0.001f;
            records.insert(records.end(), r, r + RT_TRI_FLOATS);
        }
    }
    delete tm_emit;
    int rc;
    {
        StageTimer tm("synth: finish_geometry");
        rc = finish_geometry(sc, records, bvh_max_depth, bvh_min_tris);
    }
    if (rc != RT_OK) { delete sc; return rc; }
    StageTimer tm_env("synth: materials + env");
    // materials: 0 ground Lambert, 1 Lambert, 2 glossy (type 0), 3 metal (type 4)
    rt_material m = blank_material();
    m.type = 3; m.color = color_to_int(V3{0.5f, 0.6f, 0.55f}); sc->mats.push_back(m);
    m = blank_material(); m.type = 3; m.color = color_to_int(V3{1.0f, 0.52f, 0.12f}); sc->mats.push_back(m);
    m = blank_material(); m.type = 0; m.color = color_to_int(V3{0.2f, 0.4f, 0.9f}); m.p1 = 0.3f; m.p2 = 0.1f; sc->mats.push_back(m);
    m = blank_material(); m.type = 4; m.color = color_to_int(V3{0.9f, 0.9f, 0.9f}); m.p1 = 0.1f; sc->mats.push_back(m);
    // camera looking down the field from above
    sc->cam.width = 3840; sc->cam.height = 2160; sc->cam.num_samples = 128;
    sc->cam.focal_length = 0.6f; sc->cam.aperture = 0.f; sc->cam.focus_distance = 4.f;
    sc->cam.pos[0] = 0.f; sc->cam.pos[1] = 2.5f; sc->cam.pos[2] = 1.0f;
    sc->cam.rot[0] = -0.35f; sc->cam.rot[1] = 0.f; sc->cam.rot[2] = 0.f;
    if (env_path) {
        Image img;
        if (!load_image(env_path, img, err)) { delete sc; return fail(err); }
        const int f = std::max(env_upsample, 1);
        const int W = img.w * f, H = img.h * f;
        sc->textures.resize((size_t)W * H * 3);
        parallel_for((size_t)H, 16, [&](size_t y) {
            for (int x = 0; x < W; x++)
                memcpy(&sc->textures[(y * (size_t)W + (size_t)x) * 3], &img.rgb[((y / (size_t)f) * (size_t)img.w + (size_t)(x / f)) * 3], 12);
        });
        sc->cam.env_width = W;
        sc->cam.env_height = H;
    }
    *out = sc;
    return RT_OK;
}

}  // extern "C"
