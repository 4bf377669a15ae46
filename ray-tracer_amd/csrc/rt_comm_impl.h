// rt_comm_impl.h -- several GPUs of one node from ONE process: included at the end of rt_core.hip (same translation
// unit: it uses rt_context and the resolve kernels).  C-ABI in include/rt.h ("multi-GPU").
//
// No counterpart in the reference (single GPU, single launch, src/main.cu:472-482).  BASELINE.json asks for the two
// natural partitions of the image, combined by one RCCL collective per frame on the float accumulators:
//   * tile rows  -- bands of `band_rows` image rows dealt round-robin to the ranks (band b -> rank b % N: sky rows are
//                   cheap, mesh rows dear), each rank renders into a band-packed accumulator; the root receives every
//                   peer's bands (grouped ncclSend / ncclRecv: each peer's 1/N of the image over its own xGMI link) and
//                   de-interleaves them into the full image.  Bit-identical to one GPU: the RNG is keyed by (pixel, sample).
//   * sample slices -- rank r renders samples [begin_r, begin_r + count_r) of every pixel; one ncclReduce(sum) to the root.
// The scene is replicated; nothing is exchanged inside the render loop.  One host thread per device drives rt_render
// (blocking), the collective is issued for all devices from the calling thread inside ncclGroupStart / End.
//
// RCCL is loaded at run time (dlopen of librccl.so.1) so that librt_amd.so itself does not depend on it; the LOCAL
// transport does the same data movement with hipMemcpyPeerAsync and a fixed summation order ((a0 + a1) + a2 ...): it
// is what the tests use on a one-GPU box (several contexts on one device, which RCCL refuses) and the fallback when
// librccl is absent.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err) {
        if (handle) return true;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
#define RT_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(handle, sym)); if (!field) { err = std::string("librccl lacks ") + sym; return false; }
        RT_SYM(CommInitAll, "ncclCommInitAll") RT_SYM(CommDestroy, "ncclCommDestroy") RT_SYM(Reduce, "ncclReduce")
        RT_SYM(Send, "ncclSend") RT_SYM(Recv, "ncclRecv") RT_SYM(GroupStart, "ncclGroupStart") RT_SYM(GroupEnd, "ncclGroupEnd")
        RT_SYM(GetErrorString, "ncclGetErrorString")
#undef RT_SYM
        return true;
    }
};
RcclApi g_rccl;

// rows (ascending) of the image that `rank` owns under the band interleave; returns their number
int band_rows_of_rank(int height, int band_rows, int n_ranks, int rank, int32_t* rows) {
    int n = 0;
    for (int y = 0; y < height; y++)
        if ((y / band_rows) % n_ranks == rank) {
            if (rows) rows[n] = y;
            n++;
        }
    return n;
}

}  // namespace

// full[rows[r]] (+)= packed[r] for the rows of one rank (thread = one float of the packed buffer)
__global__ void rt_rows_scatter_kernel(const float* __restrict__ packed, float* __restrict__ full, const int* __restrict__ rows, int n_rows,
                                       int row_floats) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_rows * row_floats) return;
    const int r = (int)(i / row_floats), k = (int)(i - (size_t)r * row_floats);
    full[(size_t)rows[r] * row_floats + k] += packed[i];
}
// dst[i] += src[i]
__global__ void rt_add_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

struct rt_comm {
    int n = 0;
    int transport = RT_COMM_RCCL;
    std::vector<int> devices;
    std::vector<rt_context*> ctx;
    std::vector<ncclComm_t> comms;   // RCCL transport only
    std::string err;
    // per-frame buffers (grow-only): every rank's accumulator; on the root the peers' staging copies, the full image
    std::vector<float*> accum;
    std::vector<size_t> accum_floats;
    std::vector<float*> staging;     // [rank] on the root's device (rank 0 unused)
    std::vector<size_t> staging_floats;
    float* full = nullptr;
    size_t full_floats = 0;
    int* d_rows = nullptr;           // root: row lists of all ranks, one after the other
    size_t d_rows_ints = 0;
    // state of the accumulated image on the root
    int width = 0, height = 0, band_rows = 0, partition = -1;
    int64_t samples_done = 0;
    std::vector<int> n_rows;         // rows per rank of the last render (ROWS)
    bool pending = false;            // the ranks hold a rendered frame that has not been combined yet
    unsigned long long rccl_reduces = 0, rccl_sendrecvs = 0;  // combine steps that went through RCCL (rt_comm_rccl_calls)
};

namespace {

int comm_err(rt_comm* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
#define COMM_HIP(c, expr)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return comm_err(c, RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define COMM_NCCL(c, expr)                                                                             \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) return comm_err(c, RT_ERR_DEVICE, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

// inside ncclGroupStart .. ncclGroupEnd: an error must close the group before it returns, or the next collective of
// this thread is queued into a group nobody ends
#define COMM_NCCL_G(c, expr)                                                                           \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) {                                                                       \
            (void)g_rccl.GroupEnd();                                                                   \
            return comm_err(c, RT_ERR_DEVICE, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));  \
        }                                                                                              \
    } while (0)
#define COMM_HIP_G(c, expr)                                                                            \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (void)g_rccl.GroupEnd();                                                                   \
            return comm_err(c, RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));      \
        }                                                                                              \
    } while (0)

int comm_grow(rt_comm* c, int device, float** buf, size_t* have, size_t need) {
    if (need <= *have) return RT_OK;
    COMM_HIP(c, hipSetDevice(device));
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    COMM_HIP(c, hipMalloc((void**)buf, std::max<size_t>(need, 4) * sizeof(float)));
    *have = need;
    return RT_OK;
}

// Run fn(rank) on one host thread per rank (rt_render and rt_upload_scene block); first failure wins.
template <class F>
int comm_parallel(rt_comm* c, F fn) {
    std::vector<int> rc((size_t)c->n, RT_OK);
    std::vector<std::thread> th;
    for (int r = 1; r < c->n; r++) th.emplace_back([&, r] { rc[(size_t)r] = fn(r); });
    rc[0] = fn(0);
    for (auto& t : th) t.join();
    for (int r = 0; r < c->n; r++)
        if (rc[(size_t)r] != RT_OK) return comm_err(c, rc[(size_t)r], "rank " + std::to_string(r) + ": " + rt_last_error(c->ctx[(size_t)r]));
    return RT_OK;
}

int comm_sync_all(rt_comm* c) {
    for (int r = 0; r < c->n; r++) {
        COMM_HIP(c, hipSetDevice(c->devices[(size_t)r]));
        COMM_HIP(c, hipStreamSynchronize(c->ctx[(size_t)r]->stream));
    }
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_band_rows_of_rank(int height, int band_rows, int n_ranks, int rank, int32_t* rows) {
    if (height < 0 || band_rows <= 0 || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return RT_ERR_INVALID;
    return band_rows_of_rank(height, band_rows, n_ranks, rank, rows);
}

int rt_host_gather_rows(const float* const* band_packed, int n_ranks, int width, int height, int band_rows, float* full) {
    if (!band_packed || !full || n_ranks <= 0 || width <= 0 || height < 0 || band_rows <= 0) return RT_ERR_INVALID;
    std::vector<int32_t> rows((size_t)std::max(height, 1));
    const size_t rf = (size_t)width * 3;
    for (int r = 0; r < n_ranks; r++) {
        const int n = band_rows_of_rank(height, band_rows, n_ranks, r, rows.data());
        if (n && !band_packed[r]) return RT_ERR_INVALID;
        for (int k = 0; k < n; k++) memcpy(full + (size_t)rows[(size_t)k] * rf, band_packed[r] + (size_t)k * rf, rf * sizeof(float));
    }
    return RT_OK;
}

int rt_host_reduce_sum(const float* const* accums, int n_ranks, size_t n_floats, float* out) {
    if (!accums || !out || n_ranks <= 0) return RT_ERR_INVALID;
    for (int r = 0; r < n_ranks; r++)
        if (!accums[r] && n_floats) return RT_ERR_INVALID;
    for (size_t i = 0; i < n_floats; i++) {  // ((a0 + a1) + a2) + ...: the order of the LOCAL transport
        float s = accums[0][i];
        for (int r = 1; r < n_ranks; r++) s += accums[r][i];
        out[i] = s;
    }
    return RT_OK;
}

int rt_comm_create(rt_comm** out, const int* device_ids, int n_devices, int transport) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    if (!device_ids || n_devices <= 0 || n_devices > RT_MAX_DEVICES || (transport != RT_COMM_RCCL && transport != RT_COMM_LOCAL))
        return RT_ERR_INVALID;
    rt_comm* c = new rt_comm();
    c->n = n_devices;
    c->transport = transport;
    c->devices.assign(device_ids, device_ids + n_devices);
    c->accum.assign((size_t)n_devices, nullptr);
    c->accum_floats.assign((size_t)n_devices, 0);
    c->staging.assign((size_t)n_devices, nullptr);
    c->staging_floats.assign((size_t)n_devices, 0);
    c->n_rows.assign((size_t)n_devices, 0);
    for (int r = 0; r < n_devices; r++) {
        rt_context* x = nullptr;
        if (rt_create(&x, device_ids[r]) != RT_OK) {
            rt_comm_destroy(c);
            return RT_ERR_DEVICE;
        }
        c->ctx.push_back(x);
    }
    if (transport == RT_COMM_RCCL) {
        std::string err;
        bool distinct = true;
        for (int a = 0; a < n_devices; a++)
            for (int b = a + 1; b < n_devices; b++) distinct = distinct && device_ids[a] != device_ids[b];
        if (!distinct || !g_rccl.load(err)) {  // RCCL wants one rank per device; without the library the LOCAL transport serves
            if (!distinct) { rt_comm_destroy(c); return RT_ERR_INVALID; }
            c->transport = RT_COMM_LOCAL;
            c->err = err + " -- using the local transport";
        } else {
            c->comms.assign((size_t)n_devices, nullptr);
            if (g_rccl.CommInitAll(c->comms.data(), n_devices, device_ids) != ncclSuccess) {
                c->comms.clear();
                rt_comm_destroy(c);
                return RT_ERR_DEVICE;
            }
        }
    }
    *out = c;
    return RT_OK;
}

void rt_comm_destroy(rt_comm* c) {
    if (!c) return;
    for (ncclComm_t k : c->comms)
        if (k) (void)g_rccl.CommDestroy(k);
    for (int r = 0; r < (int)c->ctx.size(); r++) {
        (void)hipSetDevice(c->devices[(size_t)r]);
        if (c->accum[(size_t)r]) {
            (void)rt_bind_accum(c->ctx[(size_t)r], nullptr, 0);
            (void)hipFree(c->accum[(size_t)r]);
        }
        rt_destroy(c->ctx[(size_t)r]);
    }
    if (!c->devices.empty()) (void)hipSetDevice(c->devices[0]);
    for (float* p : c->staging)
        if (p) (void)hipFree(p);
    if (c->full) (void)hipFree(c->full);
    if (c->d_rows) (void)hipFree(c->d_rows);
    delete c;
}

const char* rt_comm_last_error(const rt_comm* c) { return c ? c->err.c_str() : "null comm"; }
int rt_comm_size(const rt_comm* c) { return c ? c->n : 0; }
int rt_comm_transport(const rt_comm* c) { return c ? c->transport : -1; }
rt_context* rt_comm_context(rt_comm* c, int rank) { return (c && rank >= 0 && rank < c->n) ? c->ctx[(size_t)rank] : nullptr; }

int rt_comm_upload_scene(rt_comm* c, const rt_scene_desc* scene) {
    if (!c) return RT_ERR_INVALID;
    // the device layout (sub-trees, shortcut tree, light order) is built ONCE on all host threads; the ranks then only
    // allocate and copy, each on its own thread
    UploadTimer timer;
    HostLayout L;
    std::string err;
    const int rc = build_layout(scene, L, err, timer);
    if (rc != RT_OK) return comm_err(c, rc, err);
    return comm_parallel(c, [&](int r) { UploadTimer t; return upload_built(c->ctx[(size_t)r], scene, L, t); });
}

int rt_comm_reset(rt_comm* c) {
    if (!c) return RT_ERR_INVALID;
    if (c->full && c->full_floats) {
        COMM_HIP(c, hipSetDevice(c->devices[0]));
        COMM_HIP(c, hipMemsetAsync(c->full, 0, c->full_floats * sizeof(float), c->ctx[0]->stream));
        COMM_HIP(c, hipStreamSynchronize(c->ctx[0]->stream));
    }
    c->samples_done = 0;
    c->pending = false;
    return RT_OK;
}

// Combine step of the sample-slice partition: sum of every rank's accumulator -> added to the root's full image.
int rt_comm_reduce_sum(rt_comm* c) {
    if (!c) return RT_ERR_INVALID;
    if (!c->pending || c->partition != RT_PART_SAMPLES) return comm_err(c, RT_ERR_INVALID, "no sample-sliced frame to combine");
    const size_t n = (size_t)c->width * c->height * 3;
    hipStream_t s0 = c->ctx[0]->stream;
    if (c->transport == RT_COMM_RCCL) {  // (with ONE rank too: ncclReduce in place is then a no-op that still runs the transport)
        COMM_NCCL(c, g_rccl.GroupStart());
        for (int r = 0; r < c->n; r++) {
            COMM_HIP_G(c, hipSetDevice(c->devices[(size_t)r]));
            COMM_NCCL_G(c, g_rccl.Reduce(c->accum[(size_t)r], c->accum[(size_t)r], n, ncclFloat, ncclSum, 0, c->comms[(size_t)r], c->ctx[(size_t)r]->stream));
        }
        COMM_NCCL(c, g_rccl.GroupEnd());
        c->rccl_reduces++;
        COMM_HIP(c, hipSetDevice(c->devices[0]));
    } else {  // LOCAL: ((a0 + a1) + a2) + ... on the root
        COMM_HIP(c, hipSetDevice(c->devices[0]));
        for (int r = 1; r < c->n; r++) {
            int rc = comm_grow(c, c->devices[0], &c->staging[(size_t)r], &c->staging_floats[(size_t)r], n);
            if (rc != RT_OK) return rc;
            COMM_HIP(c, hipMemcpyPeerAsync(c->staging[(size_t)r], c->devices[0], c->accum[(size_t)r], c->devices[(size_t)r], n * sizeof(float), s0));
            hipLaunchKernelGGL(rt_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s0, c->accum[0], c->staging[(size_t)r], n);
        }
    }
    hipLaunchKernelGGL(rt_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s0, c->full, c->accum[0], n);
    COMM_HIP(c, hipGetLastError());
    int rc = comm_sync_all(c);
    if (rc != RT_OK) return rc;
    c->pending = false;
    return RT_OK;
}

// Combine step of the tile-row partition: every peer's band-packed rows -> the root, de-interleaved into the full image.
int rt_comm_gather_rows(rt_comm* c) {
    if (!c) return RT_ERR_INVALID;
    if (!c->pending || c->partition != RT_PART_ROWS) return comm_err(c, RT_ERR_INVALID, "no row-partitioned frame to combine");
    const size_t rf = (size_t)c->width * 3;
    hipStream_t s0 = c->ctx[0]->stream;
    // RCCL with ONE rank: the root's rows take the transport's path too (a grouped ncclSend / ncclRecv to itself into
    // its staging buffer), so that a one-GPU box runs the send / receive code; with peers the root's own rows stay put
    const bool self_send = c->transport == RT_COMM_RCCL && c->n == 1;
    for (int r = self_send ? 0 : 1; r < c->n; r++) {
        int rc = comm_grow(c, c->devices[0], &c->staging[(size_t)r], &c->staging_floats[(size_t)r], (size_t)c->n_rows[(size_t)r] * rf);
        if (rc != RT_OK) return rc;
    }
    if (c->transport == RT_COMM_RCCL) {
        COMM_NCCL(c, g_rccl.GroupStart());
        for (int r = self_send ? 0 : 1; r < c->n; r++) {
            if (!c->n_rows[(size_t)r]) continue;
            COMM_HIP_G(c, hipSetDevice(c->devices[(size_t)r]));
            COMM_NCCL_G(c, g_rccl.Send(c->accum[(size_t)r], (size_t)c->n_rows[(size_t)r] * rf, ncclFloat, 0, c->comms[(size_t)r], c->ctx[(size_t)r]->stream));
        }
        COMM_HIP_G(c, hipSetDevice(c->devices[0]));
        for (int r = self_send ? 0 : 1; r < c->n; r++)
            if (c->n_rows[(size_t)r])
                COMM_NCCL_G(c, g_rccl.Recv(c->staging[(size_t)r], (size_t)c->n_rows[(size_t)r] * rf, ncclFloat, r, c->comms[0], s0));
        COMM_NCCL(c, g_rccl.GroupEnd());
        c->rccl_sendrecvs++;
    } else {
        COMM_HIP(c, hipSetDevice(c->devices[0]));
        for (int r = 1; r < c->n; r++)
            if (c->n_rows[(size_t)r])
                COMM_HIP(c, hipMemcpyPeerAsync(c->staging[(size_t)r], c->devices[0], c->accum[(size_t)r], c->devices[(size_t)r],
                                               (size_t)c->n_rows[(size_t)r] * rf * sizeof(float), s0));
    }
    COMM_HIP(c, hipSetDevice(c->devices[0]));
    size_t off = 0;
    for (int r = 0; r < c->n; r++) {
        const size_t cnt = (size_t)c->n_rows[(size_t)r] * rf;
        if (cnt)
            hipLaunchKernelGGL(rt_rows_scatter_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s0, (r == 0 && !self_send) ? c->accum[0] : c->staging[(size_t)r],
                               c->full, c->d_rows + off, c->n_rows[(size_t)r], (int)rf);
        off += (size_t)c->n_rows[(size_t)r];
    }
    COMM_HIP(c, hipGetLastError());
    int rc = comm_sync_all(c);
    if (rc != RT_OK) return rc;
    c->pending = false;
    return RT_OK;
}

}  // extern "C"
namespace {
int comm_render_impl(rt_comm* c, const rt_render_params* p, int partition, int band_rows, rt_render_stats* stats) {
    if (!c || !p) return RT_ERR_INVALID;
    if (partition != RT_PART_ROWS && partition != RT_PART_SAMPLES) return comm_err(c, RT_ERR_INVALID, "unknown partition");
    if (p->cam.width <= 0 || p->cam.height <= 0 || p->sample_count < 0) return comm_err(c, RT_ERR_INVALID, "bad image size / sample count");
    if (band_rows <= 0) band_rows = 16;
    const int W = p->cam.width, H = p->cam.height;
    const size_t full_n = (size_t)W * H * 3;
    // a new image shape or partition starts a new accumulation on the root
    if (W != c->width || H != c->height) {
        int rc = comm_grow(c, c->devices[0], &c->full, &c->full_floats, full_n);
        if (rc != RT_OK) return rc;
        c->width = W;
        c->height = H;
        rc = rt_comm_reset(c);
        if (rc != RT_OK) return rc;
    }
    c->partition = partition;
    c->band_rows = band_rows;
    std::vector<rt_render_params> rp((size_t)c->n, *p);
    if (partition == RT_PART_ROWS) {
        std::vector<int32_t> rows((size_t)H * (size_t)c->n);  // the root's de-interleave table: every rank's rows, rank after rank
        size_t off = 0;
        for (int r = 0; r < c->n; r++) {
            c->n_rows[(size_t)r] = band_rows_of_rank(H, band_rows, c->n, r, rows.data() + off);
            off += (size_t)c->n_rows[(size_t)r];
            rp[(size_t)r].band_rows = band_rows;
            rp[(size_t)r].band_stride = c->n;
            rp[(size_t)r].band_index = r;
        }
        COMM_HIP(c, hipSetDevice(c->devices[0]));
        if (off > c->d_rows_ints) {
            if (c->d_rows) (void)hipFree(c->d_rows);
            c->d_rows = nullptr;
            c->d_rows_ints = 0;
            COMM_HIP(c, hipMalloc((void**)&c->d_rows, std::max<size_t>(off, 1) * sizeof(int)));
            c->d_rows_ints = off;
        }
        COMM_HIP(c, hipMemcpy(c->d_rows, rows.data(), off * sizeof(int), hipMemcpyHostToDevice));
    } else {
        for (int r = 0; r < c->n; r++) {  // sample slices: as even as the count allows
            const long long b = (long long)p->sample_count * r / c->n, e = (long long)p->sample_count * (r + 1) / c->n;
            rp[(size_t)r].sample_begin = p->sample_begin + (int)b;
            rp[(size_t)r].sample_count = (int)(e - b);
            rp[(size_t)r].band_rows = 0;
            rp[(size_t)r].band_stride = 1;
            rp[(size_t)r].band_index = 0;
            c->n_rows[(size_t)r] = H;
        }
    }
    // every rank renders into a zeroed accumulator of its own (bound, so the collective works on it in place)
    for (int r = 0; r < c->n; r++) {
        const size_t need = (size_t)c->n_rows[(size_t)r] * W * 3;
        const size_t before = c->accum_floats[(size_t)r];
        int rc = comm_grow(c, c->devices[(size_t)r], &c->accum[(size_t)r], &c->accum_floats[(size_t)r], std::max<size_t>(need, 4));
        if (rc != RT_OK) return rc;
        if (c->accum_floats[(size_t)r] != before || !c->ctx[(size_t)r]->d_accum_ext) {
            rc = rt_bind_accum(c->ctx[(size_t)r], c->accum[(size_t)r], c->accum_floats[(size_t)r]);
            if (rc != RT_OK) return comm_err(c, rc, rt_last_error(c->ctx[(size_t)r]));
        }
        COMM_HIP(c, hipSetDevice(c->devices[(size_t)r]));
        COMM_HIP(c, hipMemsetAsync(c->accum[(size_t)r], 0, std::max<size_t>(need, 4) * sizeof(float), c->ctx[(size_t)r]->stream));
        // (rt_render zeroes on a shape change by itself; the same shape twice in a row must start from zero too)
    }
    std::vector<rt_render_stats> st((size_t)c->n);
    // Every rank renders, THEN the one collective of the frame is issued (not from each rank's thread behind its own
    // render): a rank whose render fails must not leave its peers blocked inside a collective it never joins, and a
    // reduce cannot finish before its slowest input anyway.
    int rc = comm_parallel(c, [&](int r) { return render_impl(c->ctx[(size_t)r], &rp[(size_t)r], &st[(size_t)r]); });
    if (rc != RT_OK) return rc;
    c->pending = true;
    rc = partition == RT_PART_ROWS ? rt_comm_gather_rows(c) : rt_comm_reduce_sum(c);
    if (rc != RT_OK) return rc;
    c->samples_done += p->sample_count;
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (const rt_render_stats& s : st) {
            stats->kernel_ms = std::max(stats->kernel_ms, s.kernel_ms);
            stats->trace_ms = std::max(stats->trace_ms, s.trace_ms);
            stats->paths += s.paths; stats->rays += s.rays; stats->inner_visits += s.inner_visits; stats->leaf_visits += s.leaf_visits;
            stats->tri_tests += s.tri_tests; stats->hits += s.hits; stats->texel_fetches += s.texel_fetches; stats->light_tests += s.light_tests;
            stats->dev_inner_visits += s.dev_inner_visits; stats->dev_lds_inner_visits += s.dev_lds_inner_visits; stats->dev_tri_tests += s.dev_tri_tests;
            stats->dev_light_tests += s.dev_light_tests; stats->shortcut_rewalks += s.shortcut_rewalks;
            stats->iterations = std::max(stats->iterations, s.iterations);
            stats->trace_launches = std::max(stats->trace_launches, s.trace_launches);
        }
    }
    return RT_OK;
}
}  // namespace
extern "C" {

int rt_comm_render_v2(rt_comm* c, const rt_render_params* p, int partition, int band_rows, rt_render_stats* stats, size_t stats_size) {
    rt_render_stats full;
    memset(&full, 0, sizeof(full));
    const int rc = comm_render_impl(c, p, partition, band_rows, stats ? &full : nullptr);
    if (stats) memcpy(stats, &full, std::min(stats_size, sizeof(full)));
    return rc;
}
// The symbol round 2 exported: its callers hold round 2's struct.
int rt_comm_render(rt_comm* c, const rt_render_params* p, int partition, int band_rows, rt_render_stats* stats) {
    return rt_comm_render_v2(c, p, partition, band_rows, stats, RT_RENDER_STATS_V2_SIZE);
}

// Combine steps of this comm that went through RCCL so far: sum-reduces, grouped send / receive gathers.
int rt_comm_rccl_calls(const rt_comm* c, uint64_t* reduces, uint64_t* sendrecvs) {
    if (!c) return RT_ERR_INVALID;
    if (reduces) *reduces = c->rccl_reduces;
    if (sendrecvs) *sendrecvs = c->rccl_sendrecvs;
    return RT_OK;
}

int rt_comm_readback(rt_comm* c, float* rgb, size_t n_floats, int mode) {
    if (!c || !rgb) return RT_ERR_INVALID;
    if (mode < RT_READ_RAW_SUM || mode > RT_READ_TONEMAPPED) return comm_err(c, RT_ERR_INVALID, "bad readback mode");
    const size_t n = (size_t)c->width * c->height * 3;
    if (n == 0 || !c->full) return comm_err(c, RT_ERR_INVALID, "nothing rendered yet");
    if (n_floats != n) return comm_err(c, RT_ERR_INVALID, "n_floats does not match width*height*3 of the last render");
    COMM_HIP(c, hipSetDevice(c->devices[0]));
    rt_context* x = c->ctx[0];
    const float* src = c->full;
    if (mode != RT_READ_RAW_SUM) {
        int rc = ensure_scratch(x, n * sizeof(float));
        if (rc != RT_OK) return comm_err(c, rc, rt_last_error(x));
        const float samples = static_cast<float>(c->samples_done > 0 ? c->samples_done : 1);
        hipLaunchKernelGGL(rt_resolve_kernel, dim3((unsigned)((n / 3 + 255) / 256)), dim3(256), 0, x->stream, c->full, (float*)x->d_scratch, n / 3, samples, mode);
        COMM_HIP(c, hipGetLastError());
        src = (const float*)x->d_scratch;
    }
    COMM_HIP(c, hipMemcpyAsync(rgb, src, n * sizeof(float), hipMemcpyDeviceToHost, x->stream));
    COMM_HIP(c, hipStreamSynchronize(x->stream));
    return RT_OK;
}

}  // extern "C"
