// valu_peak.hip -- how fast does one gfx950 SIMD issue plain (non-packed) f32 vector instructions?
// The answer is the "peak" of the VALU roofline in bench.py: the render kernel issues sub / mul / min / max /
// cndmask on 64-lane waves, nothing packed and no MFMA, so its ceiling is wave-instructions per cycle per SIMD.
//
//   hipcc -O3 --offload-arch=gfx950 -o valu_peak valu_peak.hip && ./valu_peak
//
// Each wave runs a loop of ITER x 64 independent instructions (8 accumulators, no instruction depends on the one
// before it); grid = 256 CUs x (waves per SIMD) x 4 SIMDs waves, block = 256 threads (one wave per SIMD).
// Output: cycles per wave-instruction per SIMD = kernel time x clock / (waves per SIMD x instructions per wave),
// with the clock taken from s_memtime / s_memrealtime inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void valu_loop(float* out, unsigned long long* clocks, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float k = 1.0000001f, m = 0.5f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (KIND == 0) {  // v_add_f32
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
            } else if (KIND == 1) {  // the render kernel's mix: sub, mul, min, max, max3, cndmask
                asm volatile("v_sub_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_min_f32 %2, %2, %9\n v_max_f32 %3, %3, %9\n"
                             "v_max3_f32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, vcc\n v_sub_f32 %6, %6, %9\n v_mul_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(m) : "vcc");
            } else if (KIND == 2) {  // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(m));
            } else {  // v_rcp_f32 (transcendental rate)
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

typedef void (*kfn)(float*, unsigned long long*, int);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 20000;
    const char* names[4] = {"v_add_f32", "sub/mul/min/max/max3/cndmask mix", "v_fma_f32", "v_rcp_f32"};
    const kfn fns[4] = {valu_loop<0>, valu_loop<1>, valu_loop<2>, valu_loop<3>};
    float* out;
    unsigned long long* clk;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&clk, (size_t)cus * 8 * 2 * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("device %s, %d CUs; %d x 64 independent instructions per wave\n", prop.gcnArchName, cus, iters);
    for (int kind = 0; kind < 4; kind++)
        for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD = 256-thread blocks per CU
            const int blocks = cus * wps;
            fns[kind]<<<blocks, 256>>>(out, clk, 100);  // warm-up
            CHECK(hipEventRecord(e0));
            fns[kind]<<<blocks, 256>>>(out, clk, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h((size_t)blocks * 2);
            CHECK(hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double cyc = 0, real = 0;
            for (int b = 0; b < blocks; b++) { cyc += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
            const double ghz = cyc / real * 0.1;  // s_memrealtime ticks at 100 MHz
            const double inst_per_wave = (double)iters * 64;
            // wall clock: every SIMD issues wps x inst_per_wave wave-instructions during the kernel (blocks of a launch
            // need not all be resident at once, so a block's own duration is not divided by wps)
            const double cyc_per_inst_simd = (double)ms * 1e-3 * ghz * 1e9 / (wps * inst_per_wave);
            printf("%-36s %d wave(s)/SIMD: %.3f ms, clock %.2f GHz, block lifetime %.2f cycles per instruction -> %.2f cycles per wave-instruction per SIMD"
                   " = %.1f lanes/cycle/SIMD\n", names[kind], wps, ms, ghz, cyc / blocks / inst_per_wave, cyc_per_inst_simd, 64.0 / cyc_per_inst_simd);
        }
    return 0;
}
