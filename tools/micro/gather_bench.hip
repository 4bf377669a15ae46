// Micro-benchmark: how much does a divergent 64-byte record fetch cost per lane on gfx950?
//   A: every lane loads its own record with 4 x global_load_dwordx4 (what trav_step does)
//   B: the 4 lanes of a quad load ONE record per instruction (16 B each, one 64-B line per quad) and
//      take four instructions to serve the quad's four records; data exchanged with DPP-free __shfl
//   C: records in LDS (4 piece arrays), 4 x ds_read_b128 per lane
// Prints ns per record-fetch per CU for a table of N records (L2 resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned next_idx(unsigned s, unsigned mask) { return (s * 1664525u + 1013904223u) & mask; }

__global__ __launch_bounds__(1024) void kA(const float4* __restrict__ tab, unsigned mask, int iters, float* out) {
    unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        s = next_idx(s, 0xffffffffu);
        const unsigned idx = (s >> 8) & mask;
        const float4* w = tab + 4 * (size_t)idx;
        const float4 a = w[0], b = w[1], c = w[2], d = w[3];
        acc += a.x + b.y + c.z + d.w;
        s += __float_as_uint(acc) & 1u;  // dependent chain like a traversal
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(1024) void kB(const float4* __restrict__ tab, unsigned mask, int iters, float* out) {
    unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, q = lane & 3, qb = lane & ~3;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        s = next_idx(s, 0xffffffffu);
        const unsigned idx = (s >> 8) & mask;
        float4 r[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {  // instruction j serves quad member j: lane loads piece q of ITS record
            const unsigned other = __shfl(idx, qb + j, 64);
            r[j] = tab[4 * (size_t)other + q];
        }
        // lane needs piece p of its own record = r[q] held by lane qb+p
        float sum = 0.f;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            float4 v;
            // select r[q] of the source lane: every lane publishes r[k] for k = dest's q ... emulate with 4 shuffles per piece
            const float4 mine = q == 0 ? r[0] : (q == 1 ? r[1] : (q == 2 ? r[2] : r[3]));
            (void)mine;
            float4 cand[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                cand[k].x = __shfl(r[k].x, qb + p, 64); cand[k].y = __shfl(r[k].y, qb + p, 64);
                cand[k].z = __shfl(r[k].z, qb + p, 64); cand[k].w = __shfl(r[k].w, qb + p, 64);
            }
            v = q == 0 ? cand[0] : (q == 1 ? cand[1] : (q == 2 ? cand[2] : cand[3]));
            sum += p == 0 ? v.x : (p == 1 ? v.y : (p == 2 ? v.z : v.w));
        }
        acc += sum;
        s += __float_as_uint(acc) & 1u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

extern __shared__ float4 lds[];
__global__ __launch_bounds__(1024) void kC(const float4* __restrict__ tab, unsigned mask, int n, int iters, float* out) {
    for (int i = threadIdx.x; i < 4 * n; i += blockDim.x) lds[(i & 3) * n + (i >> 2)] = tab[i];
    __syncthreads();
    unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        s = next_idx(s, 0xffffffffu);
        const unsigned idx = (s >> 8) & mask;
        const float4 a = lds[idx], b = lds[n + idx], c = lds[2 * n + idx], d = lds[3 * n + idx];
        acc += a.x + b.y + c.z + d.w;
        s += __float_as_uint(acc) & 1u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// D: like A but only a fraction of lanes active (does the cost scale with active lanes?)
__global__ __launch_bounds__(1024) void kD(const float4* __restrict__ tab, unsigned mask, int iters, float* out, int active) {
    unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    if ((threadIdx.x & 63) < active)
        for (int i = 0; i < iters; i++) {
            s = next_idx(s, 0xffffffffu);
            const unsigned idx = (s >> 8) & mask;
            const float4* w = tab + 4 * (size_t)idx;
            const float4 a = w[0], b = w[1], c = w[2], d = w[3];
            acc += a.x + b.y + c.z + d.w;
            s += __float_as_uint(acc) & 1u;
        }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int n = 1024;  // records (64 KB: L1/L2 resident like a small BVH)
    std::vector<float> h(16 * n);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 97) * 0.01f;
    float4* tab; float* out;
    CHECK(hipMalloc(&tab, h.size() * 4));
    CHECK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int iters = 2000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void*)kC, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * n));
    for (int waves_per_cu : {4, 8, 16, 32}) {
        const int bs = waves_per_cu >= 16 ? 1024 : waves_per_cu * 64;
        const int blocks = 256 * (waves_per_cu * 64 / bs);
        CHECK(hipMalloc(&out, (size_t)blocks * bs * 4));
        auto run = [&](const char* name, auto launch, double lanes_frac) {
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double fetches = (double)blocks * bs * lanes_frac * iters;
            printf("waves/CU %2d %-28s %8.3f ms  %6.2f G record-fetches/s  %.2f ns per fetch per CU (%.1f cycles @2.4GHz)\n", waves_per_cu, name, ms,
                   fetches / ms / 1e6, ms * 1e6 / (fetches / 256), ms * 1e6 / (fetches / 256) * 2.4);
        };
        run("A per-lane 4 x dwordx4", [&] { hipLaunchKernelGGL(kA, dim3(blocks), dim3(bs), 0, 0, tab, n - 1, iters, out); }, 1.0);
        run("B quad-coalesced + shuffles", [&] { hipLaunchKernelGGL(kB, dim3(blocks), dim3(bs), 0, 0, tab, n - 1, iters, out); }, 1.0);
        if (waves_per_cu <= 16) run("C LDS 4 x ds_read_b128", [&] { hipLaunchKernelGGL(kC, dim3(256), dim3(bs), 64 * n, 0, tab, n - 1, n, iters, out); }, bs * 256.0 / ((double)blocks * bs));
        run("D per-lane, 16 of 64 lanes", [&] { hipLaunchKernelGGL(kD, dim3(blocks), dim3(bs), 0, 0, tab, n - 1, iters, out, 16); }, 0.25);
        CHECK(hipFree(out));
    }
    return 0;
}
