// rt_core.hip -- kernels and C-ABI (include/rt.h) of the MI355X path-tracing core.
//
// Replaces the render call of the reference: uploads src/main.cu:418-460, kernel
// render2 src/main.cu:155-407 (launch :472-482), readback :500-501.
//
// Kernel design (gfx950), details at each kernel and in DESIGN.md section 4:
//  * rt_render_paths_kernel (default) -- persistent 1024-thread blocks, one per CU, sharing an LDS copy of the top
//    of the tree.  The unit of work is one PATH (pixel, sample); a lane is traversing, or waits for a vertex pass,
//    and every iteration the wave votes and runs ONE kind of step (vertex pass / triangle test / inner-node
//    visit) for the lanes that wait for it.  Per-path radiance goes to a buffer,
//    rt_sum_samples_kernel adds it per pixel in sample order => bit-reproducible sums, no float atomics.
//  * rt_render_kernel (RT_F_LOCKSTEP) -- a wave owns an 8x8 tile, lane = pixel, "trace one ray, shade one hit"
//    in lock step with path regeneration.
//  * rt_logic_kernel + rt_trace_kernel (RT_F_WAVEFRONT) -- one global ray queue per bounce; the trace kernel is
//    also what rt_trace_rays runs.
//  * Traversal stack: per-lane short stack in LDS ([entry][lane] => conflict-free).  Node / triangle / material
//    fetches are 16-byte vector loads from the 64/48-byte records of rt_device.h.
//  * raw-sum accumulators + separate resolve kernels (needed for RCCL sum-reduce and progressive batches).
#include <hip/hip_runtime.h>

#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <xmmintrin.h>  // (host: the sub-tree sweep keeps a box in two SSE registers)

#define RT_NO_SIZED_MACROS  // (this file DEFINES rt_render & co.; callers get them as sizeof-passing macros)
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

#ifndef RT_MEGA_WAVES
#define RT_MEGA_WAVES 16  // ONE 1024-thread block per CU: its 16 waves share one LDS copy of the top of the tree.
#endif                    // (measured: 16 waves at 128 VGPRs beat 12 waves at 161 VGPRs, profiles/r01_ab_megakernel.txt)
template <bool STATS>
__global__ __launch_bounds__(64 * RT_MEGA_WAVES) void rt_render_kernel(Scene sc, Params P, float* __restrict__ accum,
                                                                        unsigned int* __restrict__ tile_counter,
                                                                        unsigned long long* __restrict__ stats, int n_cached,
                                                                        int stack_entries) {
    // dynamic LDS (rt_lds): [4 * n_cached node pieces][RT_MEGA_WAVES x stack_entries x 64 ints]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * n_cached; i += 64 * RT_MEGA_WAVES)  // item i>>2, piece i&3 (coalesced read)
        rt_lds[RT_LDS_SLOT(i >> 2, i & 3, n_cached)] = sc.items[i];
    int* stk = (int*)(rt_lds + RT_LDS_CACHE_FLOAT4S(n_cached)) + (size_t)wave * stack_entries * 64 + lane;
    __syncthreads();
    const bool use_bvh = (P.flags & RT_F_BVH) != 0;
    const bool spheres_on = (P.flags & RT_F_SPHERES) != 0 && sc.n_spheres > 0;
    Counters cn;
    cn.clear();
    [[maybe_unused]] const unsigned long long plife0 = RT_PROF_CLOCK();

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
        const int spp = in_image ? P.sample_count : 0;
        int k = 0;                                  // samples started
        bool alive = false;
        int bounce = 0;
        PathState ray;
        Rng rng;
        rng.init(P.seed, pixel, 0);

        if (P.max_bounces <= 0) {  // the bounce loop never runs: every sample contributes 0
            if (STATS) cn.paths += spp;
            k = spp;
        }
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
                const uint32_t bb = (RT_RNG_DIM_BOUNCE0 + RT_RNG_DIMS_PER_BOUNCE * (uint32_t)bounce) >> 2;
                int hitTri;
                float minD2;
                f3 hitPos;
                [[maybe_unused]] const unsigned long long pc0 = RT_PROF_CLOCK();
                trace_closest<STATS>(sc, ray.o, ray.d, use_bvh, stk, hitTri, minD2, hitPos, cn, n_cached);
                [[maybe_unused]] const unsigned long long pc1 = RT_PROF_CLOCK();
                RT_PROF_COUNT(cn, 2, pc1 - pc0);
                RT_PROF_COUNT(cn, 3, 1);
                RT_PROF_LANE(cn, 4);
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
                        shade_sphere<STATS>(sc, ray, sNrm, sPos, mat, rng, bb, cn);
                        handled = true;  // `continue` at main.cu:343: no renormalise
                    }
                }
                if (!handled) {
                    if (hitTri >= 0) {
                        if (STATS) cn.hits++;
                        const bool terminated = shade_triangle<STATS>(sc, ray, hitPos, hitTri, rng, bb, cn);
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
                RT_PROF_COUNT(cn, 5, RT_PROF_CLOCK() - pc1);
            }
        }
        if (in_image) {
            float* a = accum + ((size_t)r * P.width + x) * 3;
            a[0] += color.x;
            a[1] += color.y;
            a[2] += color.z;
        }
    }

    RT_PROF_COUNT(cn, 6, RT_PROF_CLOCK() - plife0);
    if (STATS) {  // wave-reduce, one atomic per counter per wave
        uint32_t v[12] = {cn.paths, cn.rays, cn.inner, cn.leaf, cn.tri, cn.hits, cn.texels, cn.lights, cn.dev_inner, cn.dev_cached, cn.dev_tri, cn.dev_lights};
#pragma unroll
        for (int i = 0; i < 12; i++) {
            unsigned long long s = v[i];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0 && s) atomicAdd(&stats[i], s);
        }
#ifdef RT_CYCLE_PROF
        for (int i = 0; i < 16; i++) {
            unsigned long long s = cn.prof[i];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0 && s) atomicAdd(&stats[16 + i], s);
        }
#endif
    }
}

// ------------------------------------------------------------------------------- path-parallel fused kernel
//
// Same arithmetic per path and the same per-pixel sum, different schedule.  rt_render_kernel above
// runs "trace one ray, shade one hit" in lock step: every lane of a wave waits for the longest
// traversal of each round (measured: 29.7 of 64 lanes active per traversal iteration) and a pixel's
// samples are one serial chain.  Here
//  * the unit of work is ONE PATH (pixel, sample).  A path writes its radiance to samples[path];
//    rt_sum_samples_kernel then adds a pixel's samples in sample order (c = 0; c += e_k; accum += c,
//    the very additions of the lock-step kernel), so the raw sums are bit-identical;
//  * a lane is in one of four phases and the wave keeps stepping the traversal of the lanes that
//    hold a ray; the others wait until RT_REGEN_MIN of them have gathered and then ONE "vertex pass"
//    serves them all: shade the hit (or the environment), store a finished path and take the next
//    one, generate its camera ray, start the next traversal;
//  * a wave takes paths in batches of RT_PATH_BATCH (one atomic) and hands them to lanes one by one.
// Path index p within a chunk of S samples: p = ((tile * S + k) * 64 + pixel-in-tile), so the 64
// lanes of a wave mostly hold neighbouring pixels of one sample.
#ifndef RT_REGEN_MIN
#define RT_REGEN_MIN 40
#endif
#ifndef RT_TRI_MIN
#define RT_TRI_MIN 12  // run a triangle-test iteration when this many lanes wait at a leaf (0 = mixed steps)
#endif
#ifndef RT_REGEN_MIN_RUNS
#define RT_REGEN_MIN_RUNS 44  // the two votes on a shortcut tree over runs (the bundled scenes; sweep in profiles/r03_ab.txt)
#endif
#ifndef RT_TRI_MIN_RUNS
#define RT_TRI_MIN_RUNS 16
#endif
#ifndef RT_BIG_LEAF
#define RT_BIG_LEAF 2           // leaves with at least this many triangles may get a sub-tree (rt_device.h) ...
#endif
#define RT_SUB_COST_INNER 0.8f  // ... where the estimate says so: cost of a node visit in triangle tests,
#define RT_SUB_MIN_TRIS 256     // ... in scenes of at least this many triangles (in a tiny tree the lanes of a wave walk
                                // in step, and more kinds of step only break that up: Cornell box -10 %),
#define RT_SUB_GAIN 1.0f        // ... by this factor at least,
#define RT_CULL_FAR 8.f          // padded boxes cull only for rays that start within this many scene scales of the origin
#define RT_ITEM_FRONT 8192       // inner items ordered by box area at the front of the item stream (what an LDS cache can hold, and more)
#define RT_REF_FRONT 1024        // with a shortcut tree: this many of the reference's biggest nodes stand in front of the shortcut walk's window
#define RT_LIGHT_TREE_FRONT_MAX 512  // wide nodes of a light tree that still goes into the LDS window (80 KB of it)
#define RT_WIDE_STACK 12         // stack entries a search of the WIDE shortcut tree may use for itself (a deeper path gives up and walks the reference's tree)
#define RT_SHORT_STACK_MAX 32    // stack entries the shortcut tree and the sub-trees below it may need together (256 B x 16 waves of LDS each)
#define RT_SHORT_FRONT_MAX (1 << 18)  // ... with at most this many entries (the 10M-triangle field, 4 spp frames: leaves 15.56 ms, 2 levels below them --
                                      // 262 k entries, 25 ms of build -- 14.43, 4 levels -- 1 M, 96 ms -- 14.43, 6 levels 14.49, runs -- 6.2 M, 0.5 s -- 13.8 to 14.5)
#define RT_SHORT_CUT_MAX 6       // ... and then over what stands at most this many levels below the reference's leaves
#define RT_SHORT_RUNS_MAX (1 << 20)  // shortcut tree over the runs of the item stream up to this many, over the reference's leaves above
#define RT_SHORT_SWEEP_MAX 2048  // shortcut tree: full surface-area sweep up to this many leaves per node (64 in trees over more than 2^18: build time), 32 bins above
#define RT_SUB_LEAF_MAX 8       // sub-leaves of more triangles than this are always split,
#define RT_SUB_SWEEP_MAX 4096   // full surface-area sweep up to this many triangles, median splits above (bounds the build time)
#ifndef RT_INNER_STEPS
#define RT_INNER_STEPS 4  // inner-node visits per vote (a vote costs about a tenth of a visit; 1 / 2 / 3 / 4 / 6 / 8: 31.9 / 30.3 / 30.0 / 29.8 / 31.3 / 30.6 ms)
#endif
#ifndef RT_INNER_STEPS_SHORT
#define RT_INNER_STEPS_SHORT 3  // ... on the shortcut walk (fewer, less alike walks: 2 / 3 / 4 / 6: 23.5 / 23.0 / 23.4 / 23.45 ms; the Cornell box, which has none, loses 6 % with 3)
#endif
#ifndef RT_INNER_ROLLED
#define RT_INNER_ROLLED 0
#endif
#ifndef RT_FINISH_MIN
#define RT_FINISH_MIN 12        // lanes back from a light query that get a (short) vertex pass of their own
#endif
#ifndef RT_WIDE7
#define RT_WIDE7 1  // the shortcut walk's LDS window: wide nodes only, 112 bytes each (0 = padded items, 160 bytes a node: experiments)
#endif
#ifndef RT_WIDE_STEPS
#define RT_WIDE_STEPS 2         // visits of wide (four-child) shortcut nodes per vote
#endif
#ifndef RT_WIDE_STEPS_TINY
#define RT_WIDE_STEPS_TINY 3    // ... in a scene that sits in the LDS entirely
#endif
#ifndef RT_INNER_STEPS_TINY
#define RT_INNER_STEPS_TINY 6   // ... on the shortcut walk of a scene that sits in the LDS entirely (the Cornell box, 256 spp: 4 / 5 / 6 / 7 / 8 / 12 = 109.1 / 109.0 / 97.3 / 98.4 / 98.8 / 99.8 ms at a vertex vote of 48)
#endif
#ifndef RT_REGEN_MIN_TINY
#define RT_REGEN_MIN_TINY 48    // vertex-pass vote there (40 / 44 / 48 / 52 / 56: 100.5 / 97.6 / 97.3 / 97.7 / 101.1 ms)
#endif
#ifndef RT_REGEN_MIN_COOP
#define RT_REGEN_MIN_COOP 40    // ... in scenes whose vertex passes run the wave-cooperative light loop (scene2_is.txt 32 spp: 40 / 44 / 48 / 52 / 56 = 69.4 / 68.1 / 67.4 / 67.4 / 69.0 ms)
#endif
#ifndef RT_NT_STORE
#define RT_NT_STORE 1  // a path's radiance is written once and read once, by another kernel: stored past the caches (-1 % frame time,
                       // more bytes at the HBM: 3.18 vs 2.93 GB per frame of scene.txt; profiles/r02_ab.txt)
#endif
#ifndef RT_HIT_PREFETCH
#define RT_HIT_PREFETCH 1  // a vertex pass fetches its hit triangles' item, leaf box and cold data in one go at its top (rt_device.h, HitFetch)
#endif
#ifndef RT_LEAF_STEPS
#define RT_LEAF_STEPS 1   // triangle tests per vote
#endif
#ifndef RT_TRACE_STEPS
#define RT_TRACE_STEPS 4  // traversal steps of the wavefront trace kernel per look at the queue (1 / 2 / 4: 85.2 / 81.4 / 79.6 ms)
#endif
#ifndef RT_PATH_BATCH
#define RT_PATH_BATCH 512
#endif
// A lane's phase lives in Trav::cur: >= 0 traversing (inner node below n_inner, else a leaf), RT_TRAV_DONE =
// closest hit known, vertex pending; RT_LANE_NEW = wants a path; RT_LANE_IDLE = no paths left.
#define RT_LANE_NEW (-2)
#define RT_LANE_IDLE (-3)

// BIG: the scene has leaf sub-trees (rt_device.h).  EXTRA: RT_F_VOLUME and / or RT_F_RAYMARCH
// (both off in the reference's config.h): seven more registers of path state (volume density + colour, march velocity).
// FAST: closest-hit searches go down the shortcut tree first (rt_device.h; the scene has one, no STATS of the reference's walk).
#ifndef RT_PATHS_MIN_WAVES
#define RT_PATHS_BOUNDS __launch_bounds__(64 * RT_MEGA_WAVES)
#else  // (experiments: blocks of fewer waves, several per CU, and a register budget for RT_PATHS_MIN_WAVES waves per SIMD)
#define RT_PATHS_BOUNDS __launch_bounds__(64 * RT_MEGA_WAVES, RT_PATHS_MIN_WAVES)
#endif
#ifndef RT_LDS_BUDGET_BYTES
#define RT_LDS_BUDGET_BYTES (160 * 1024 - 1024)  // of one block: leave a little for the runtime
#endif
template <bool STATS, bool BIG, bool EXTRA, bool COOP, bool TLDS = false, bool FAST = false>
__global__ RT_PATHS_BOUNDS void rt_render_paths_kernel(Scene sc, Params P, float* __restrict__ samples,
                                                                              unsigned int* __restrict__ path_counter,
                                                                              unsigned long long* __restrict__ stats, int n_cached,
                                                                              int stack_entries, int chunk_k0, int chunk_S,
                                                                              unsigned int n_paths, int n_mats_lds) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int cache_base = (FAST && !TLDS) ? sc.short_base : 0;  // the LDS holds items [cache_base, cache_base + n_cached) (TLDS: all of them)
    // W7: the window of the shortcut walk holds wide nodes only, seven float4s each (rt_device.h, WIDE7); n_cached = 2 x nodes.
    // (Not in the variant with light queries: measured 2.4 % slower there at the same number of nodes, profiles/r03_ab.txt.)
    constexpr bool W7 = RT_WIDE7 && FAST && !TLDS && !COOP;
    const size_t cache_f4 = W7 ? (size_t)RT_WIDE7_F4 * (size_t)(n_cached >> 1) : RT_LDS_CACHE_FLOAT4S(n_cached);
    if (W7) {
        for (int i = threadIdx.x; i < RT_WIDE7_F4 * (n_cached >> 1); i += 64 * RT_MEGA_WAVES) {
            const int w = i / RT_WIDE7_F4, p = i - w * RT_WIDE7_F4;
            rt_lds[i] = sc.items[4 * (size_t)cache_base + 8 * (size_t)w + p];
        }
    } else {
        for (int i = threadIdx.x; i < 4 * n_cached; i += 64 * RT_MEGA_WAVES) rt_lds[RT_LDS_SLOT(i >> 2, i & 3, n_cached)] = sc.items[4 * (size_t)cache_base + i];
    }
    int* stk = (int*)(rt_lds + cache_f4) + (size_t)wave * stack_entries * 64 + lane;
    // the material table behind the stacks (n_mats_lds = 0: too many materials, they stay in device memory)
    const int mats_lds = n_mats_lds > 0 ? (int)(cache_f4 + (size_t)RT_MEGA_WAVES * stack_entries * 16) : -1;
    for (int i = threadIdx.x; i < 3 * n_mats_lds; i += 64 * RT_MEGA_WAVES) rt_lds[mats_lds + i] = sc.mats[i];
    // COOP: behind the material table RT_LQ_CAP words per wave for the light loop's queue (wave_light_pdf_q)
    [[maybe_unused]] volatile unsigned* lqueue = (volatile unsigned*)(rt_lds + cache_f4 + (size_t)RT_MEGA_WAVES * stack_entries * 16 + 3 * (size_t)n_mats_lds) + wave * RT_LQ_CAP;
    __syncthreads();
    const bool use_bvh = (P.flags & RT_F_BVH) != 0;
    const bool spheres_on = (P.flags & RT_F_SPHERES) != 0 && sc.n_spheres > 0;
    const bool volume = EXTRA && (P.flags & RT_F_VOLUME) != 0;
    const bool marching = EXTRA && (P.flags & RT_F_RAYMARCH) != 0;
    Counters cn;
    cn.clear();
    [[maybe_unused]] const unsigned long long plife0 = RT_PROF_CLOCK();
    [[maybe_unused]] float volDensity = 0.0f;
    [[maybe_unused]] f3 volColor = mk(1.f, 1.f, 1.f), marchVel = mk(0.f, 0.f, 0.f);

    // wave-uniform path source: paths [batch_next, batch_end) are not handed out yet
    unsigned int batch_next = 0, batch_end = 0;
    bool paths_left = true;
    // lane state
    unsigned int path = 0;
    int bounce = 0;
    f3 thr = mk(1.f, 1.f, 1.f);
    float ri = 1.0f;
    Rng rng;
    rng.init(P.seed, 0, 0);
    Trav t;
    t.oxy = t.ixy = (v2f){0.f, 0.f};
    t.oz = t.iz = 0.f;
    t.d = mk(0.f, 0.f, 0.f);
    t.axis_parallel = false;
    t.set_hit(0.0f, mk(0.f, 0.f, 0.f));
    f3 ro = mk(0.f, 0.f, 0.f);  // origin of the ray that starts in the current vertex pass
    t.cur = RT_LANE_NEW;
    t.sp = 0;
    t.hitTri = -1;
    t.minD2 = 1e12f;
    t.rival = -1.0f;
    t.bestLeaf = -1;
    t.slack = 0.0f;

    for (;;) {
        const unsigned long long mNeed = __ballot((unsigned)(t.cur + 2) < 2u);  // RT_TRAV_DONE or RT_LANE_NEW
        const unsigned long long mTrav = __ballot(t.cur >= 0);
        // LTREE: reflectType5's sum over the lights runs as a search of the light tree (rt_device.h, LIGHT QUERY): the vertex pass that
        // shades the hit starts it, the lane walks the light tree with its ordinary steps and comes back to a later pass, which
        // finishes the bounce (finishing).  That second visit is short (the tail of reflectType5 and the start of the next search),
        // so RT_FINISH_MIN lanes waiting for it get a pass of their own, which serves them alone (finish_only).
        constexpr bool LTREE = FAST && COOP;
        const bool finishing = LTREE && t.cur == RT_TRAV_DONE && t.bestLeaf <= RT_LQ_RUNNING;
        // (a shortcut tree over runs -- the bundled scenes -- takes other votes than one over a frontier of sub-trees: RT_*_RUNS)
        const bool full_pass = __popcll(mNeed) >= ((FAST && TLDS) ? RT_REGEN_MIN_TINY : COOP ? RT_REGEN_MIN_COOP : (FAST && sc.short_runs) ? RT_REGEN_MIN_RUNS : RT_REGEN_MIN) || mTrav == 0;
        const bool finish_only = LTREE && !full_pass && __popcll(__ballot(finishing)) >= RT_FINISH_MIN;
        if (full_pass || finish_only) {
            if (mNeed == 0) break;  // every lane idle: no paths left
            [[maybe_unused]] const unsigned long long pc0 = RT_PROF_CLOCK();
            RT_PROF_COUNT(cn, 3, 1);
            if ((unsigned)(t.cur + 2) < 2u) RT_PROF_LANE(cn, 4);
            // PF: what the pass reads about the hit triangles of its lanes, fetched together up front (rt_device.h, HitFetch)
            constexpr bool PF = RT_HIT_PREFETCH && FAST && (!COOP || RT_HIT_PREFETCH == 2);
            HitFetch pf;
            const bool pf_on = PF && !finish_only && t.cur == RT_TRAV_DONE && !finishing && t.hitTri >= 0 && t.bestLeaf != -2;
            if (pf_on) hit_fetch<TLDS>(sc, t.hitTri, pf, rng, (RT_RNG_DIM_BOUNCE0 + RT_RNG_DIMS_PER_BOUNCE * (uint32_t)bounce) >> 2, n_cached);
            // (short_certified reads the box only for a lane with a hit whose search did not give up: pf_on)
            if (FAST && !finish_only && t.cur == RT_TRAV_DONE && !finishing && !short_certified<PF>(sc, t, pf.mn, pf.mx)) {
                // the shortcut walk's result does not prove what the reference's walk returns: this lane walks the
                // reference's tree now (main.cu:221-283) and comes back to a later vertex pass
                if (STATS) cn.rewalks++;
                trav_begin(sc, t, t.o(), t.d);
                t.rival = -1.0f;
            }
            bool go = false;        // this lane holds a ray whose traversal starts in this pass
            PathState ray;
            f3 e = mk(0.f, 0.f, 0.f);  // what this path adds to its pixel
            bool alive = true;
            const bool shading = t.cur == RT_TRAV_DONE && !finishing && !finish_only;
            bool deferred = false;  // this lane's bounce goes on in a later pass, behind its light query
            LightQuery lq;
            lq.pending = false;
            if (shading) {  // one iteration of the bounce loop after the closest-hit search, main.cu:305-389
                const uint32_t bb = (RT_RNG_DIM_BOUNCE0 + RT_RNG_DIMS_PER_BOUNCE * (uint32_t)bounce) >> 2;
                ray.o = t.o();
                ray.d = t.d;
                ray.thr = thr;
                ray.emission = mk(0.f, 0.f, 0.f);
                ray.ri = ri;
                ray.volDensity = EXTRA ? volDensity : 0.0f;
                ray.volColor = EXTRA ? volColor : mk(1.f, 1.f, 1.f);
                bool handled = false;
                // main.cu:338, 350 hand checkScatteredVolume the TRIANGLE hit position (zero without one), also for spheres
                const f3 triPos = (volume && t.hitTri >= 0) ? t.hit_pos() : mk(0.f, 0.f, 0.f);
                if (spheres_on) {
                    f3 sPos;
                    const int hitSphere = trav_spheres(sc, t, sPos);
                    if (hitSphere >= 0) {
                        if (!(volume && check_scattered_volume(ray, triPos, rng, (uint32_t)bounce))) {
                            const float4 s0 = sc.spheres[2 * hitSphere];
                            const f3 sNrm = normalize(sPos - mk(s0.x, s0.y, s0.z));
                            const int mat = __float_as_int(sc.spheres[2 * hitSphere + 1].x);
                            if (STATS) cn.hits++;
                            shade_sphere<STATS>(sc, ray, sNrm, sPos, mat, rng, bb, cn);
                        }
                        handled = true;  // `continue` at main.cu:339, 343: no renormalise
                    }
                }
                if (!handled) {
                    if (t.hitTri >= 0) {
                        if (volume && check_scattered_volume(ray, triPos, rng, (uint32_t)bounce)) {
                            // scattered inside the volume: `continue`, main.cu:350-352
                        } else {
                            if (STATS) cn.hits++;
                            if (shade_triangle<STATS, PF>(sc, ray, t.hit_pos(), t.hitTri, rng, bb, cn, COOP ? &lq : nullptr, mats_lds, pf)) {  // main.cu:357-360
                                e = ray.emission * ray.thr;
                                alive = false;
                            } else if (!COOP || !lq.pending) {
                                ray.d = normalize_keep(ray.d);  // main.cu:361
                            }
                        }
                    } else if (marching) {  // main.cu:364-372: nothing within this step, so march; the backdrop only far away
                        march_ray(ray.o, ray.d, marchVel);
                        if (len2(ray.o) > 20.0f * 20.0f) {
                            e = ld3(P.bg) * ray.thr * P.bg_brightness;
                            alive = false;
                        }
                    } else {  // main.cu:374-387
                        if ((P.flags & RT_F_SKYBOX) && P.env_w > 0) {
                            const long long envI = env_index(ray.d, P.env_w, P.env_h);
                            if (STATS) cn.texels++;
                            e = fetch_texel(sc, envI) * ray.thr * P.bg_brightness;
                        } else {
                            e = ld3(P.bg) * ray.thr * P.bg_brightness;
                        }
                        alive = false;
                    }
                }
            }
#ifdef RT_LQ_SKIP  // (timing experiments only: the sum over the lights taken as 0 -- a wrong image, the cost of everything else)
            if (LTREE && lq.pending) {
                finish_type5(sc, ray, lq, 0.0f);
                ray.d = normalize_keep(ray.d);
                lq.pending = false;
            }
#endif
            if (LTREE) {
                // pdf_cos > 2e-12: pdf = pdf_cos / 2 + pdf_light / 2 is then above reflectType5's 1e-12 whatever the lights add, so the
                // finishing pass never needs the incoming direction (material.h:324-327 keeps it when the pdf vanishes)
                if (lq.pending && sc.light_root >= 0 && lq.pdf_cos > 2e-12f && light_query_begin(sc, t, lq.hitPos, lq.outgoing, lq.pdf_cos, W7 ? 64 : 80)) {
                    thr = ray.thr * lq.clr;  // (ray.thr * clr) * (pdf_cos / pdf): the first product now, the second when the sum is known
                    ri = ray.ri;
                    if (STATS) cn.lights += (uint32_t)sc.n_lights;  // what the reference's loop tests
                    deferred = true;
                    lq.pending = false;
                }
                // What the tree cannot serve -- a ray it is not made for (light_query_begin), a vanishing pdf_cos, and afterwards a query
                // with three or more nonzero terms (their sum depends on the order) or one that ran out of stack -- is the reference's
                // own loop over every light, in its order (light_pdf_loop).  Rare; the wave waits for the lanes that run it.
                if (lq.pending) {
                    if (STATS) cn.dev_light_loops++;
                    const float pdf_light = light_pdf_loop<STATS>(sc, lq.hitPos, lq.outgoing, cn);
                    finish_type5(sc, ray, lq, pdf_light);
                    ray.d = normalize_keep(ray.d);  // main.cu:361
                    lq.pending = false;
                }
                if (finishing) {  // the tail of reflectType5 (material.h:324-335) and main.cu:361
                    if (t.hitTri > 2 || t.bestLeaf == RT_LQ_GAVE_UP) {
                        if (STATS) cn.dev_light_loops++;
                        Counters unused;  // (the deferral has counted this query's tests)
                        unused.clear();
                        t.minD2 = light_pdf_loop<false>(sc, t.o(), t.d, unused);
                    }
                    const float pdf_cos = t.hitT, pdf = 0.5f * pdf_cos + 0.5f * t.minD2;
                    ray.o = t.o();
                    ray.d = t.d;
                    ray.thr = thr * (pdf_cos / pdf);
                    ray.ri = ri;
                    ray.emission = mk(0.f, 0.f, 0.f);
                    ray.volDensity = 0.0f;
                    ray.volColor = mk(1.f, 1.f, 1.f);
                    ray.o = ray.o + ray.d * 1e-5f;
                    ray.d = normalize_keep(ray.d);
                }
            } else if (COOP && __ballot(lq.pending)) {  // material type 5 with many lights: its light loop, with the whole wave (rt_device.h)
                const float pdf_light = wave_light_pdf_q<STATS>(sc, lq, cn, lqueue);
                if (lq.pending) {
                    finish_type5(sc, ray, lq, pdf_light);
                    ray.d = normalize_keep(ray.d);  // main.cu:361
                }
            }
            if ((shading && !deferred) || finishing) {
                if (EXTRA) {
                    volDensity = ray.volDensity;
                    volColor = ray.volColor;
                }
                bounce++;
                if (bounce >= P.max_bounces) alive = false;  // path exhausted: contributes 0
                if (alive) {
                    ro = ray.o;
                    t.d = ray.d;
                    thr = ray.thr;
                    ri = ray.ri;
                    go = true;
                } else {
                    float* o = samples + 3 * (size_t)path;
#if RT_NT_STORE
                    __builtin_nontemporal_store(e.x, o);
                    __builtin_nontemporal_store(e.y, o + 1);
                    __builtin_nontemporal_store(e.z, o + 2);
#else
                    o[0] = e.x;
                    o[1] = e.y;
                    o[2] = e.z;
#endif
                    t.cur = RT_LANE_NEW;
                }
            }
            RT_PROF_COUNT(cn, 7, RT_PROF_CLOCK() - pc0);
            // lanes without a path take the next ones of the wave's batch
            bool needpath = t.cur == RT_LANE_NEW && !finish_only;
            int x = 0, y = 0;
            unsigned long long m = __ballot(needpath);
            while (m) {
                if (batch_next >= batch_end) {
                    if (!paths_left) break;
                    unsigned int b = 0;
                    if (lane == 0) b = atomicAdd(path_counter, (unsigned)RT_PATH_BATCH);
                    batch_next = __builtin_amdgcn_readfirstlane(b);
                    if (batch_next >= n_paths) { paths_left = false; batch_end = batch_next; break; }
                    batch_end = min(batch_next + (unsigned)RT_PATH_BATCH, n_paths);
                }
                const unsigned int avail = batch_end - batch_next;
                const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (needpath && rank < avail) {
                    path = batch_next + rank;
                    const unsigned int q = path >> 6, pi = path & 63u;
                    const unsigned int tile = q / (unsigned)chunk_S, k = q - tile * (unsigned)chunk_S;
                    const unsigned int ty = tile / (unsigned)P.tiles_x, tx = tile - ty * (unsigned)P.tiles_x;
                    x = (int)(tx * 8 + (pi & 7));
                    const int r = (int)(ty * 8 + (pi >> 3));
                    if (x < P.width && r < P.rows) {  // (a path of a pixel outside the image is consumed and the lane asks again)
                        y = row_to_y(P, r);
                        rng.pixel = (uint32_t)(x + y * P.width);  // main.cu:171
                        rng.sample = (uint32_t)(P.sample_begin + chunk_k0 + (int)k);
                        needpath = false;
                    }
                }
                batch_next += min(avail, (unsigned)__popcll(m));
                m = __ballot(needpath);
            }
            if (t.cur == RT_LANE_NEW && !finish_only) {
                if (needpath) {
                    t.cur = RT_LANE_IDLE;
                } else {  // camera sample (main.cu:178-217)
                    camera_ray(P, x, y, rng, ro, t.d);
                    thr = mk(1.f, 1.f, 1.f);
                    ri = 1.0f;
                    bounce = 0;
                    if (EXTRA) {  // Ray::Ray, ray.h:22-31
                        volDensity = 0.0f;
                        volColor = mk(1.f, 1.f, 1.f);
                        marchVel = t.d * 0.1f;
                    }
                    if (STATS) cn.paths++;
                    if (P.max_bounces > 0) {
                        go = true;
                    } else {  // the bounce loop never runs: the sample contributes 0
                        float* o = samples + 3 * (size_t)path;
                        o[0] = 0.f;
                        o[1] = 0.f;
                        o[2] = 0.f;
                    }
                }
            }
            if (go) {  // main.cu:221-236
                if (STATS) cn.rays++;
                trav_begin(sc, t, ro, t.d, marching ? len2(marchVel) : 1e12f);
                if (FAST) short_begin(sc, t, W7 ? 64 : 80);
                if (!use_bvh) trav_brute_force<STATS>(sc, t, cn);
            }
            RT_PROF_COUNT(cn, 5, RT_PROF_CLOCK() - pc0);
        }
        [[maybe_unused]] const unsigned long long pt0 = RT_PROF_CLOCK();
#if RT_TRI_MIN > 0
        // one KIND of step per iteration, by vote: triangle tests when enough lanes wait at a leaf (or none is at an
        // inner node), inner-node visits otherwise -- the iteration executes the code of that kind only
        const bool at_inner = trav_at_inner(sc, t);
        const bool at_leaf = t.cur >= sc.n_inner;  // a triangle or RT_TRAV_EMPTY
        const unsigned long long mLeaf = __ballot(at_leaf);
#if defined(RT_CYCLE_PROF) && RT_CYCLE_PROF != 2
        {   // lane census of this iteration: who steps, who waits for what
            const bool leaf_iter = __popcll(mLeaf) >= RT_TRI_MIN || __ballot(at_inner) == 0;
            const int base = leaf_iter ? 12 : 8;
            const unsigned long long nI = (unsigned long long)__popcll(__ballot(at_inner)), nL = (unsigned long long)__popcll(mLeaf),
                                     nV = (unsigned long long)__popcll(__ballot((unsigned)(t.cur + 2) < 2u));  // (votes by the whole wave)
            RT_PROF_COUNT(cn, base, 1);
            RT_PROF_COUNT(cn, base + 1, nI);
            RT_PROF_COUNT(cn, base + 2, nL);
            RT_PROF_COUNT(cn, base + 3, nV);
        }
#endif
        if (__popcll(mLeaf) >= ((FAST && !TLDS && !COOP && sc.short_runs) ? RT_TRI_MIN_RUNS : RT_TRI_MIN) || __ballot(at_inner) == 0) {
            RT_PROF_COUNT(cn, 13, 1);
            if (at_leaf) {
                RT_PROF_COUNT(cn, 0, 1);
                RT_PROF_LANE(cn, 1);
                RT_PROF_LANE(cn, 14);
                trav_step_leaf<STATS, BIG, TLDS, FAST, FAST && COOP>(sc, t, stk, cn, n_cached);
#pragma unroll
                for (int k = 1; k < RT_LEAF_STEPS; k++)
                    if (t.cur >= sc.n_inner) trav_step_leaf<STATS, BIG, TLDS, FAST, FAST && COOP>(sc, t, stk, cn, n_cached);
                // shortcut tree over runs (1.6 triangles each): a second test before the next vote (scene.txt -1.5 %, scene2.txt -1 %;
                // over leaves with their sub-trees, the 10M-triangle field, it loses 1 %)
                if (FAST && RT_LEAF_STEPS == 1 && sc.short_runs && t.cur >= sc.n_inner) trav_step_leaf<STATS, BIG, TLDS, FAST, FAST && COOP>(sc, t, stk, cn, n_cached);
            }
#if RT_CYCLE_PROF == 2
            RT_PROF_COUNT(cn, 10, RT_PROF_CLOCK() - pt0);
#endif
        } else {
            // wide shortcut nodes (rt_device.h) and two-child nodes (the reference's on a re-walk, sub-tree nodes below the
            // shortcut tree) are visited by separate code, each run only if some lane of the wave stands at such a node
            const bool at_wide = FAST && at_inner && trav_at_wide(sc, t);
            if (FAST && __ballot(at_wide)) {
                RT_PROF_COUNT(cn, 11, 1);
                if (at_wide) {
                    RT_PROF_COUNT(cn, 0, 1);
                    RT_PROF_LANE(cn, 1);
                    RT_PROF_LANE(cn, 15);
                    trav_step_wide<STATS, TLDS, W7>(sc, t, stk, cn, n_cached, cache_base);
#pragma unroll
                    for (int k = 1; k < (TLDS ? RT_WIDE_STEPS_TINY : RT_WIDE_STEPS); k++)
                        if (trav_at_wide(sc, t)) trav_step_wide<STATS, TLDS, W7>(sc, t, stk, cn, n_cached, cache_base);
                }
            }
#if RT_CYCLE_PROF == 2
            [[maybe_unused]] const unsigned long long pt1 = RT_PROF_CLOCK();
            RT_PROF_COUNT(cn, 8, pt1 - pt0);
            if (__ballot(at_inner && !at_wide)) RT_PROF_COUNT(cn, 12, 1);
#endif
            if (at_inner && !at_wide && (!FAST || trav_at_inner(sc, t))) {
                RT_PROF_COUNT(cn, 0, 1);
                RT_PROF_LANE(cn, 1);
#if RT_INNER_ROLLED  // (experiments: ONE copy of the visit's code in a counted loop)
#pragma unroll 1
                for (int k = 0; k < (FAST ? (TLDS ? RT_INNER_STEPS_TINY : RT_INNER_STEPS_SHORT) : RT_INNER_STEPS); k++)
                    if (trav_at_inner(sc, t) && !(FAST && trav_at_wide(sc, t))) trav_step_inner<STATS, BIG, TLDS>(sc, t, stk, cn, W7 ? 0 : n_cached, cache_base);
#else
                trav_step_inner<STATS, BIG, TLDS>(sc, t, stk, cn, W7 ? 0 : n_cached, cache_base);
#pragma unroll
                for (int k = 1; k < (FAST ? (TLDS ? RT_INNER_STEPS_TINY : RT_INNER_STEPS_SHORT) : RT_INNER_STEPS); k++)  // (more visits before the next vote: a vote costs a tenth of a step)
                    if (trav_at_inner(sc, t) && !(FAST && trav_at_wide(sc, t))) trav_step_inner<STATS, BIG, TLDS>(sc, t, stk, cn, W7 ? 0 : n_cached, cache_base);
#endif
            }
        }
#else
        if (t.cur >= 0) {
            RT_PROF_COUNT(cn, 0, 1);
            RT_PROF_LANE(cn, 1);
            trav_step<STATS, BIG>(sc, t, stk, cn, n_cached);
        }
#endif
        RT_PROF_COUNT(cn, 2, RT_PROF_CLOCK() - pt0);
    }

    RT_PROF_COUNT(cn, 6, RT_PROF_CLOCK() - plife0);
#if RT_CYCLE_PROF == 2
    if (!STATS)
        for (int i = 0; i < 16; i++) {
            unsigned long long s = cn.prof[i];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0 && s) atomicAdd(&stats[16 + i], s);
        }
#endif
    if (STATS) {
        uint32_t v[15] = {cn.paths, cn.rays, cn.inner, cn.leaf, cn.tri, cn.hits, cn.texels, cn.lights, cn.dev_inner, cn.dev_cached, cn.dev_tri, cn.dev_lights,
                          cn.rewalks, cn.dev_wide, cn.dev_light_loops};
#pragma unroll
        for (int i = 0; i < 15; i++) {
            unsigned long long s = v[i];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0 && s) atomicAdd(&stats[i], s);
        }
#ifdef RT_CYCLE_PROF
        for (int i = 0; i < 16; i++) {
            unsigned long long s = cn.prof[i];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0 && s) atomicAdd(&stats[16 + i], s);
        }
#endif
    }
}

#include "rt_pool_kernel.h"

// Adds the chunk's samples of every pixel, in sample order, to the pixel's running sum of this call (thread = pixel of a
// tile).  The running sum starts at 0 in the first chunk, is parked in `carry` between chunks and is added to the
// accumulator after the last one: accum += ((0 + e_0) + e_1) + ... however the call was cut into chunks.
__global__ void rt_sum_samples_kernel(const float* __restrict__ samples, float* __restrict__ accum, float* __restrict__ carry, Params P,
                                      int chunk_S, int first, int last) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)P.n_tiles * 64) return;
    const unsigned int tile = (unsigned int)(i >> 6), pi = (unsigned int)(i & 63);
    const unsigned int ty = tile / (unsigned)P.tiles_x, tx = tile - ty * (unsigned)P.tiles_x;
    const int x = (int)(tx * 8 + (pi & 7)), r = (int)(ty * 8 + (pi >> 3));
    if (x >= P.width || r >= P.rows) return;
    const float* s = samples + ((size_t)tile * chunk_S * 64 + pi) * 3;
    f3 c = mk(0.f, 0.f, 0.f);
    if (!first) c = ld3(carry + 3 * i);
    for (int k = 0; k < chunk_S; k++, s += 64 * 3) {
        c.x += s[0];
        c.y += s[1];
        c.z += s[2];
    }
    if (last) {
        float* a = accum + ((size_t)r * P.width + x) * 3;
        a[0] += c.x;
        a[1] += c.y;
        a[2] += c.z;
    } else {
        carry[3 * i] = c.x;
        carry[3 * i + 1] = c.y;
        carry[3 * i + 2] = c.z;
    }
}

// =============================================================================== wavefront pipeline
//
// Slot = one pixel of this call (8x8-tiled order so that neighbouring slots are neighbouring
// pixels).  A slot runs its samples one after the other (sample order per pixel = the
// reference's loop main.cu:178, so the per-pixel float sum is bit-identical), one ray in
// flight per slot.  Per iteration:
//   rt_logic_kernel : thread = slot.  Consumes the slot's hit record (shade / environment /
//                     accumulate), regenerates a camera ray when the path ended, and appends
//                     the next ray to a compacted ray queue (wave-aggregated atomic).
//   rt_trace_kernel : persistent waves; lanes pull rays from the queue with a shared cursor
//                     and REFILL as soon as enough lanes have finished, so a wave is not held
//                     hostage by its longest traversal.  Per-lane short stack in LDS.
// Slot state (SoA, 16-byte records, coalesced in the logic kernel):
//   st_dir[slot] = (d.xyz, refractiveIndex)      st_thr[slot] = (throughput.xyz, sample<<8 | bounce)
//   st_col[slot] = (colour sum.xyz, phase)       hits[slot]   = (hitPos.xyz, tri id | -1 miss | -2-i sphere i)
// Ray queue entry = 2 x float4: (o.xyz, slot) (d.xyz, 0).
struct Wavefront {
    float4* st_dir;
    float4* st_thr;
    float4* st_col;
    float4* hits;
    float4* rayq;
    unsigned int* qcount;  // [2] rays appended in iteration parity
    unsigned int* cursor;  // [2] rays handed out by the trace kernel
    int n_slots;
};

enum { SLOT_NEW = 0, SLOT_WAIT = 1, SLOT_DONE = 2 };

#define RT_LOGIC_BLOCK 1024  // one queue-append atomic per 1024 slots (same-address atomics serialise at ~12 ns each)

template <bool STATS>
__global__ __launch_bounds__(RT_LOGIC_BLOCK) void rt_logic_kernel(Scene sc, Params P, Wavefront wf, float* __restrict__ accum, int iter,
                                                                   unsigned long long* __restrict__ stats) {
    __shared__ unsigned int wave_count[RT_LOGIC_BLOCK / 64];
    __shared__ unsigned int block_base;
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int par = iter & 1;
    if (slot == 0) {  // the other parity's counters were last used by the previous trace launch
        wf.qcount[par ^ 1] = 0;
        wf.cursor[par ^ 1] = 0;
    }
    Counters cn;
    cn.clear();
    bool emit = false;
    f3 ro = mk(0.f, 0.f, 0.f), rd = mk(0.f, 0.f, 0.f);
    if (slot < wf.n_slots) {
        const int tile = slot >> 6, within = slot & 63;
        const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
        const int x = tx * 8 + (within & 7);
        const int r = ty * 8 + (within >> 3);
        const bool in_image = x < P.width && r < P.rows;
        int phase;
        f3 color = mk(0.f, 0.f, 0.f);
        if (iter == 0) {
            phase = in_image ? SLOT_NEW : SLOT_DONE;
            if (!in_image) wf.st_col[slot] = make_float4(0.f, 0.f, 0.f, __int_as_float(SLOT_DONE));
        } else {
            const float4 c = wf.st_col[slot];
            color = mk(c.x, c.y, c.z);
            phase = __float_as_int(c.w);
        }
        if (phase != SLOT_DONE) {
            const int y = row_to_y(P, r);
            const uint32_t pixel = (uint32_t)(x + y * P.width);  // main.cu:171
            int k = 0, bounce = 0;
            PathState ray;
            ray.emission = mk(0.f, 0.f, 0.f);
            ray.volDensity = 0.f;  // type-6 state is never read while volume scattering is off (config.h:9)
            ray.volColor = mk(1.f, 1.f, 1.f);
            Rng rng;
            bool need_regen = (phase == SLOT_NEW);
            if (phase == SLOT_WAIT) {  // one iteration of the bounce loop after its closest-hit query, main.cu:303-388
                const float4 sd = wf.st_dir[slot], stt = wf.st_thr[slot], h = wf.hits[slot];
                ray.d = mk(sd.x, sd.y, sd.z);
                ray.ri = sd.w;
                ray.thr = mk(stt.x, stt.y, stt.z);
                const int packed = __float_as_int(stt.w);
                k = packed >> 8;
                bounce = packed & 0xff;
                ray.o = mk(0.f, 0.f, 0.f);
                rng.init(P.seed, pixel, (uint32_t)(P.sample_begin + k - 1));
                const uint32_t bb = (RT_RNG_DIM_BOUNCE0 + RT_RNG_DIMS_PER_BOUNCE * (uint32_t)bounce) >> 2;
                const f3 hitPos = mk(h.x, h.y, h.z);
                const int tri = __float_as_int(h.w);
                bool terminated = false;
                if (tri <= -2) {  // sphere -2-i, main.cu:337-344
                    const int si = -2 - tri;
                    const float4 s0 = sc.spheres[2 * si];
                    const int mat = __float_as_int(sc.spheres[2 * si + 1].x);
                    if (STATS) cn.hits++;
                    shade_sphere<STATS>(sc, ray, normalize(hitPos - mk(s0.x, s0.y, s0.z)), hitPos, mat, rng, bb, cn);
                } else if (tri >= 0) {
                    if (STATS) cn.hits++;
                    terminated = shade_triangle<STATS>(sc, ray, hitPos, tri, rng, bb, cn);
                    if (terminated) {  // main.cu:357-360
                        const f3 e = ray.emission * ray.thr;
                        color.x += e.x; color.y += e.y; color.z += e.z;
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
                    terminated = true;
                }
                bounce++;
                if (bounce >= P.max_bounces) terminated = true;  // path exhausted: contributes 0
                if (terminated) need_regen = true;
                else { emit = true; ro = ray.o; rd = ray.d; }
            }
            if (need_regen) {
                if (P.max_bounces <= 0) {  // the bounce loop never runs: every sample contributes 0
                    if (STATS) cn.paths += P.sample_count - k;
                    k = P.sample_count;
                }
                if (k < P.sample_count) {  // camera sample k, main.cu:178-217
                    rng.init(P.seed, pixel, (uint32_t)(P.sample_begin + k));
                    k++;
                    camera_ray(P, x, y, rng, ray.o, ray.d);
                    ray.thr = mk(1.f, 1.f, 1.f);
                    ray.ri = 1.0f;
                    bounce = 0;
                    if (STATS) cn.paths++;
                    emit = true; ro = ray.o; rd = ray.d;
                } else {  // all samples of this pixel done: one plain add, this thread owns the pixel
                    float* a = accum + ((size_t)r * P.width + x) * 3;
                    a[0] += color.x;
                    a[1] += color.y;
                    a[2] += color.z;
                    phase = SLOT_DONE;
                }
            }
            if (emit) {
                phase = SLOT_WAIT;
                wf.st_dir[slot] = make_float4(rd.x, rd.y, rd.z, ray.ri);
                wf.st_thr[slot] = make_float4(ray.thr.x, ray.thr.y, ray.thr.z, __int_as_float((k << 8) | bounce));
            }
            wf.st_col[slot] = make_float4(color.x, color.y, color.z, __int_as_float(phase));
        }
    }
    // append to the ray queue: wave ballots -> LDS prefix over the block's waves -> ONE atomic per block
    const unsigned long long mask = __ballot(emit);
    if (lane == 0) wave_count[wave] = (unsigned int)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int total = 0;
        for (int w = 0; w < RT_LOGIC_BLOCK / 64; w++) {
            const unsigned int c = wave_count[w];
            wave_count[w] = total;
            total += c;
        }
        block_base = total ? atomicAdd(&wf.qcount[par], total) : 0u;
    }
    __syncthreads();
    if (emit) {
        const unsigned int pos = block_base + wave_count[wave] + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
        wf.rayq[2 * (size_t)pos] = make_float4(ro.x, ro.y, ro.z, __int_as_float(slot));
        wf.rayq[2 * (size_t)pos + 1] = make_float4(rd.x, rd.y, rd.z, 0.f);
    }
    if (STATS) {
        uint32_t v[5] = {cn.paths, cn.hits, cn.texels, cn.lights, cn.dev_lights};
        const int idx[5] = {0, 5, 6, 7, 11};
#pragma unroll
        for (int i = 0; i < 5; i++) {
            unsigned long long t = v[i];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0 && t) atomicAdd(&stats[idx[i]], t);
        }
    }
}

#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 16  // refill the wave when at least this many lanes are idle
#endif

// Persistent closest-hit kernel.  queue: n rays (2 x float4 each); results go to hits[slot].
// counts (optional): per-slot {inner, leaf, tri} visit counts for the rt_trace_rays test hook.
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 16  // waves per trace block: they share one LDS copy of the top of the tree
#endif
// dynamic LDS (rt_lds, declared in rt_device.h): [4 * n_cached node pieces][RT_TRACE_WAVES x stack_entries x 64 ints]

// FAST: the searches go down the shortcut tree first (rt_device.h), as in rt_render_paths_kernel.
template <bool STATS, bool FAST = false>
__global__ __launch_bounds__(64 * RT_TRACE_WAVES) void rt_trace_kernel(Scene sc, const float4* __restrict__ rayq,
                                                                        const unsigned int* __restrict__ n_rays_ptr,
                                                                        unsigned int* __restrict__ cursor,
                                                                        float4* __restrict__ hits, unsigned flags,
                                                                        int* __restrict__ counts,
                                                                        unsigned long long* __restrict__ stats, int n_cached,
                                                                        int stack_entries) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned int n_rays = *n_rays_ptr;
    if ((unsigned long long)blockIdx.x * RT_TRACE_WAVES * 64 >= n_rays) return;  // no wave of this block has a first chunk
    const int cache_base = FAST ? sc.short_base : 0;
    for (int i = threadIdx.x; i < 4 * n_cached; i += 64 * RT_TRACE_WAVES)  // item i>>2, piece i&3 (coalesced read)
        rt_lds[RT_LDS_SLOT(i >> 2, i & 3, n_cached)] = sc.items[4 * (size_t)cache_base + i];
    int* stk = (int*)(rt_lds + RT_LDS_CACHE_FLOAT4S(n_cached)) + (size_t)wave * stack_entries * 64 + lane;
    __syncthreads();
    const bool use_bvh = (flags & RT_F_BVH) != 0;
    const bool spheres_on = (flags & RT_F_SPHERES) != 0 && sc.n_spheres > 0;
    Counters cn, tot;
    cn.clear();
    tot.clear();
    Trav t;
    t.cur = RT_TRAV_DONE;
    bool has = false;
    int slot = 0;
    // Work distribution: the queue is cut into chunks.  Wave w owns chunk w outright (no atomic),
    // further chunks come from ONE shared cursor (a same-address atomic costs ~12 ns serialised,
    // so chunks are large when the queue is long).  Waves whose first chunk lies beyond the queue
    // leave at once.
    const unsigned int n_waves = gridDim.x * RT_TRACE_WAVES;
    const unsigned int gw = blockIdx.x * RT_TRACE_WAVES + wave;
    unsigned int chunk = 64;
    while (chunk < 512 && (unsigned long long)chunk * n_waves * 2 <= n_rays) chunk *= 2;
    unsigned int next = gw * chunk;                                  // wave-uniform: next ray of the local chunk
    unsigned int end = next + chunk < n_rays ? next + chunk : n_rays;  // wave-uniform: end of the local chunk
    bool exhausted = next >= n_rays;
    if (exhausted) next = end = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!has);
        const int n_idle = __popcll(idle);
        if (!exhausted && (n_idle >= RT_REFILL_MIN || n_idle == 64)) {
            if (next >= end) {  // local chunk used up: take another one
                unsigned int c = 0;
                if (lane == 0) c = atomicAdd(cursor, 1u);
                c = __builtin_amdgcn_readfirstlane(c);
                const unsigned long long start = ((unsigned long long)n_waves + c) * chunk;
                if (start >= n_rays) exhausted = true;
                else { next = (unsigned int)start; end = next + chunk < n_rays ? next + chunk : n_rays; }
            }
            if (!exhausted) {
                const unsigned int avail = end - next;
                const unsigned int rank = (unsigned int)__popcll(idle & ((1ull << lane) - 1ull));
                if (!has && rank < avail) {
                    const unsigned int idx = next + rank;
                    const float4 r0 = rayq[2 * (size_t)idx], r1 = rayq[2 * (size_t)idx + 1];
                    slot = __float_as_int(r0.w);
                    if (STATS) { cn.clear(); tot.rays++; }
                    trav_begin(sc, t, mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z));
                    if (FAST) short_begin(sc, t);
                    if (!use_bvh) trav_brute_force<STATS>(sc, t, cn);
                    has = true;
                }
                next += (unsigned int)n_idle < avail ? (unsigned int)n_idle : avail;
            }
        }
        if (!__any(has)) break;  // nothing in flight: the queue is exhausted as well
#pragma unroll
        for (int k = 0; k < RT_TRACE_STEPS; k++)  // (several steps before the next look at the queue)
            if (has && t.cur != RT_TRAV_DONE) trav_step<STATS, true, FAST>(sc, t, stk, cn, n_cached, cache_base);
        if (FAST && has && t.cur == RT_TRAV_DONE && !short_certified(sc, t)) {  // not certified: the reference's walk decides
            if (STATS) tot.rewalks++;
            trav_begin(sc, t, t.o(), t.d);
            t.rival = -1.0f;
        }
        if (has && t.cur == RT_TRAV_DONE) {  // query finished: spheres (main.cu:305-334), publish, free the lane
            int tri = t.hitTri;
            f3 hp = tri >= 0 ? t.hit_pos() : mk(0.f, 0.f, 0.f);
            if (spheres_on) {
                f3 sPos;
                const int si = trav_spheres(sc, t, sPos);
                if (si >= 0) { tri = -2 - si; hp = sPos; }
            }
            hits[slot] = make_float4(hp.x, hp.y, hp.z, __int_as_float(tri));
            if (STATS) {
                if (counts) { counts[slot * 3] = (int)cn.inner; counts[slot * 3 + 1] = (int)cn.leaf; counts[slot * 3 + 2] = (int)cn.tri; }
                tot.inner += cn.inner; tot.leaf += cn.leaf; tot.tri += cn.tri;
                tot.dev_inner += cn.dev_inner; tot.dev_cached += cn.dev_cached; tot.dev_tri += cn.dev_tri;
            }
            has = false;
        }
    }
    if (STATS && stats) {
        uint32_t v[7] = {tot.rays, tot.inner, tot.leaf, tot.tri, tot.dev_inner, tot.dev_cached, tot.dev_tri};
        const int idx[7] = {1, 2, 3, 4, 8, 9, 10};
#pragma unroll
        for (int i = 0; i < 7; i++) {
            unsigned long long s = v[i];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0 && s) atomicAdd(&stats[idx[i]], s);
        }
    }
}

// Post-pass of the rt_trace_rays hook: distance and interpolated shading normal of each hit.
__global__ void rt_hitinfo_kernel(Scene sc, const float* __restrict__ rays, const float4* __restrict__ hits, size_t n, int* prim,
                                  float* t, float* hit_pos, float* normal3) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 h = hits[i];
    const int tri = __float_as_int(h.w);
    const f3 o = ld3(rays + i * 6), pos = mk(h.x, h.y, h.z);
    if (prim) prim[i] = tri >= 0 ? tri_id(sc, tri) : tri;  // the reference's triangle id
    if (t) t[i] = tri >= 0 ? sqrtf(len2(pos - o)) : -1.0f;  // sqrt of the closest-hit metric, main.cu:257
    if (hit_pos) { hit_pos[i * 3] = tri >= 0 ? pos.x : 0.f; hit_pos[i * 3 + 1] = tri >= 0 ? pos.y : 0.f; hit_pos[i * 3 + 2] = tri >= 0 ? pos.z : 0.f; }
    if (normal3) {
        f3 nn = mk(0.f, 0.f, 0.f);
        if (tri >= 0) nn = hit_frame(sc, tri, pos).normal;
        normal3[i * 3] = nn.x; normal3[i * 3 + 1] = nn.y; normal3[i * 3 + 2] = nn.z;
    }
}

// rays (n x 6 floats) -> queue entries with slot = ray index
__global__ void rt_fill_queue_kernel(const float* __restrict__ rays, size_t n, float4* __restrict__ rayq) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rayq[2 * i] = make_float4(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2], __int_as_float((int)i));
    rayq[2 * i + 1] = make_float4(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5], 0.f);
}

// The same for rt_trace_rays_noalloc: the rays are in device memory already, and the kernel also sets the two words the
// trace kernel starts from (ctl[0] = ray count, ctl[1] = cursor of the chunks handed out), so that no copy from host
// memory is needed.  ctl is the head of the static scratch array, which keeps the last call's contents.
__global__ void rt_fill_queue_ctl_kernel(const float* __restrict__ rays, size_t n, float4* __restrict__ rayq, unsigned int* __restrict__ ctl) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { ctl[0] = (unsigned int)n; ctl[1] = 0u; }
    if (i >= n) return;
    rayq[2 * i] = make_float4(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2], __int_as_float((int)i));
    rayq[2 * i + 1] = make_float4(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5], 0.f);
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

// average + toneMap (main.cu:392-406) + the 8-bit conversion of Texture::saveImgData (texture.cpp:53-56)
__global__ void rt_resolve_rgb8_kernel(const float* __restrict__ accum, unsigned char* __restrict__ out, size_t n_pixels, float samples) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    f3 c = mk(accum[i * 3], accum[i * 3 + 1], accum[i * 3 + 2]);
    c.x /= samples; c.y /= samples; c.z /= samples;
    c = tone_map(c);
    out[i * 3] = (unsigned char)(int)(c.x * 255);
    out[i * 3 + 1] = (unsigned char)(int)(c.y * 255);
    out[i * 3 + 2] = (unsigned char)(int)(c.z * 255);
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
    Counters cn;
    cn.clear();
    bool term = true;
    if (tri >= 0 && tri < sc.n_tris) term = shade_triangle<false>(sc, ray, hitPos, sc.tri_place[tri], rng, RT_RNG_DIM_BOUNCE0 >> 2, cn);
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
            const uint4 w = rng.block(__float_as_uint(q[3]) >> 2);
            out[r * 4] = __uint_as_float(w.x); out[r * 4 + 1] = __uint_as_float(w.y);
            out[r * 4 + 2] = __uint_as_float(w.z); out[r * 4 + 3] = __uint_as_float(w.w);
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

// Scratch of rt_trace_rays_noalloc (include/rt.h): control words, ray queue and hit records of one call, in a
// file-scope device array, so that the entry point needs no allocation.  Every GPU the library runs on holds its own
// copy (12 MiB) until the process ends.  In host code the array's NAME is only a stand-in for hipGetSymbolAddress:
// rt_create looks the address up once per device (noalloc_scratch_of) and everything below uses that pointer.
#define RT_NOALLOC_MAX_RAYS RT_TRACE_NOALLOC_MAX_RAYS  // (the limit include/rt.h states sizes the array)
#define RT_NOALLOC_CTL_BYTES 256
#define RT_NOALLOC_BYTES (RT_NOALLOC_CTL_BYTES + (size_t)RT_NOALLOC_MAX_RAYS * (32 + 16))  // queue entry 32 B + hit record 16 B per ray
static __device__ __align__(256) unsigned char rt_noalloc_scratch[RT_NOALLOC_BYTES];

#define RT_MAX_DEVICES 64
static void* g_noalloc_ptr[RT_MAX_DEVICES];  // device address of rt_noalloc_scratch on each device, set once by rt_create
static std::mutex g_noalloc_mutex;

// Device `dev` must be current.  Returns hipSuccess and the device's own address of the array.
static hipError_t noalloc_scratch_of(int dev, void** out) {
    *out = nullptr;
    if (dev < 0 || dev >= RT_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (!g_noalloc_ptr[dev]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(rt_noalloc_scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_noalloc_ptr[dev] = p;
    }
    *out = g_noalloc_ptr[dev];
    return hipSuccess;
}

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
    void* d_items = nullptr; void* d_cold = nullptr; void* d_place = nullptr; void* d_refboxes = nullptr;
    void* d_mats = nullptr; void* d_lights = nullptr; void* d_spheres = nullptr; void* d_tex = nullptr;
    int bvh_depth = 0;
    int short_depth = 0;  // stack entries of the shortcut walk (0: no shortcut tree)

    // accumulators
    float* d_accum_own = nullptr;
    size_t accum_own_floats = 0;
    float* d_accum_ext = nullptr;
    size_t accum_ext_floats = 0;
    int acc_rows = 0, acc_width = 0;
    int64_t samples_done = 0;

    unsigned int* d_tile_counter = nullptr;  // [0] megakernel tile counter, [4..5] qcount, [6..7] cursor, [8] trace hook count
    unsigned long long* d_stats = nullptr;

    // wavefront state (sized to the largest call so far)
    Wavefront wf{};
    void* d_wf = nullptr;
    int wf_slots = 0;
    int trace_lds_set = 0;   // dynamic-LDS limit already granted to the trace kernels
    int mega_lds_set = 0;
    float* d_samples = nullptr;  // per-path radiance of the chunk in flight (path-parallel fused kernel), grow-only
    size_t samples_floats = 0;
    std::vector<hipEvent_t> ev_pool;
    unsigned int* h_pinned = nullptr;
    void* noalloc_scratch = nullptr;  // this device's address of rt_noalloc_scratch
    void* d_scratch = nullptr;   // grow-only device scratch of the readbacks and test hooks (no allocation per call)
    size_t scratch_bytes = 0;
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
    void** ptrs[] = {&c->d_items, &c->d_place, &c->d_refboxes, &c->d_cold, &c->d_mats, &c->d_lights, &c->d_spheres, &c->d_tex};
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

// Grow-only scratch of the context: a progressive-preview loop (render, read back, render, ...) allocates once.
int ensure_scratch(rt_context* c, size_t bytes) {
    if (bytes <= c->scratch_bytes) return RT_OK;
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    c->d_scratch = nullptr;
    c->scratch_bytes = 0;
    if (hipMalloc(&c->d_scratch, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return set_err(c, RT_ERR_NOMEM, "device scratch allocation failed");
    }
    c->scratch_bytes = bytes;
    return RT_OK;
}

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
    if ((p->flags & (RT_F_VOLUME | RT_F_RAYMARCH)) && (p->flags & (RT_F_LOCKSTEP | RT_F_WAVEFRONT)))
        return set_err(c, RT_ERR_INVALID, "volume scattering / ray marching run on the default fused kernel only");
    if (p->band_stride > 1 && (p->band_rows <= 0 || p->band_index < 0 || p->band_index >= p->band_stride))
        return set_err(c, RT_ERR_INVALID, "bad band partition");
    // the wavefront pipeline keeps (sample << 8 | bounce) in one word of the slot state (rt_logic_kernel)
    if ((p->flags & RT_F_WAVEFRONT) && (p->max_bounces > 255 || p->sample_count >= (1 << 23)))
        return set_err(c, RT_ERR_LIMIT, "wavefront pipeline: at most 255 bounces and 2^23 - 1 samples per call");
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
    bool ok = noalloc_scratch_of(device_id, &c->noalloc_scratch) == hipSuccess &&
              hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess &&
              hipMalloc((void**)&c->d_tile_counter, 256) == hipSuccess &&
              hipHostMalloc((void**)&c->h_pinned, 64, hipHostMallocDefault) == hipSuccess &&
              hipMalloc((void**)&c->d_stats, 32 * sizeof(unsigned long long)) == hipSuccess;
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
    if (c->d_samples) (void)hipFree(c->d_samples);
    if (c->d_wf) (void)hipFree(c->d_wf);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
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

namespace {
struct UploadTimer {  // RT_UPLOAD_TIMING=1: wall time of the stages of rt_upload_scene on stderr (experiments only)
    bool on = getenv("RT_UPLOAD_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void stage(const char* name) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[rt upload] %-28s %8.1f ms\n", name, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
}  // namespace

namespace {
// Everything rt_upload_scene copies to the device, built in host memory (no HIP call): the item stream with the
// leaf sub-trees, cold triangle data, decoded materials, light records with their group bounds, spheres.
// A float array that is NOT cleared when it is sized (1.6 GB of zeroes for a 10M-triangle scene, each byte written again
// right after): whoever sizes it writes every element.
// Big host arrays come from here: 2 MB-aligned and advised to use huge pages (advice only: refused or unsupported changes
// nothing).  The 10 M-triangle scene touches some 4 GB of fresh host memory while its layout is built; with 4 KB pages
// that is a million first-touch faults (measured on the item stream alone: 58 -> 34 ms).  RT_NO_HUGE_PAGES=1: plain malloc.
static void* host_big_alloc(size_t bytes) {
    static const bool plain = getenv("RT_NO_HUGE_PAGES") != nullptr;
    const size_t huge = (size_t)2 << 20;
    bytes = std::max<size_t>(bytes, 1);
    if (bytes < 4 * huge || plain) return malloc(bytes);
    const size_t rounded = (bytes + huge - 1) / huge * huge;
    void* q = aligned_alloc(huge, rounded);
    if (q) madvise(q, rounded, MADV_HUGEPAGE);
    return q;
}
extern "C++" {
template <class T>
struct HostHuge {  // allocator of BigVec<T>
    using value_type = T;
    HostHuge() = default;
    template <class U> HostHuge(const HostHuge<U>&) {}
    T* allocate(size_t n) {
        void* q = host_big_alloc(n * sizeof(T));
        if (!q) throw std::bad_alloc();
        return (T*)q;
    }
    void deallocate(T* q, size_t) { free(q); }
    template <class U> bool operator==(const HostHuge<U>&) const { return true; }
    template <class U> bool operator!=(const HostHuge<U>&) const { return false; }
};
// ... and of RawVec<T>: elements of a trivial T are left as they come (no fill when the vector is sized)
template <class T>
struct HostRaw : HostHuge<T> {
    HostRaw() = default;
    template <class U> HostRaw(const HostRaw<U>&) {}
    template <class U> struct rebind { using other = HostRaw<U>; };
    template <class U, class... A>
    void construct(U* q, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)q) U;
        else ::new ((void*)q) U(std::forward<A>(a)...);
    }
};
template <class T> using BigVec = std::vector<T, HostHuge<T>>;
template <class T> using RawVec = std::vector<T, HostRaw<T>>;
}
struct RawFloats {  // floats that are sized without a fill
    struct Free { void operator()(float* q) const { free(q); } };
    std::unique_ptr<float[], Free> p;
    size_t n = 0;
    void size_to(size_t count) {
        p.reset((float*)host_big_alloc(count * sizeof(float)));
        if (!p) throw std::bad_alloc();
        n = count;
    }
    float* data() { return p.get(); }
    const float* data() const { return p.get(); }
    size_t size() const { return n; }
    float& operator[](size_t i) { return p[i]; }
};
struct HostLayout {
    RawFloats items, cold;
    std::vector<float> mats, lights, spheres;
    RawVec<int> place;  // reference triangle id -> place in the item stream (sized without a fill: every triangle has a place)
    std::vector<float> ref_boxes;  // the reference's node boxes as uploaded, 8 floats per node (the shortcut tree's certificate reads the leaf's)
    int n_inner = 0, n_sub = 0, n_tris = 0, root_entry = RT_TRAV_DONE, depth = 0;
    bool short_runs = false, short_single = false;
    int short_cut = 0;
    int n_short = 0, short_root = RT_TRAV_DONE, short_base = 0, short_depth = 0;  // shortcut tree (rt_device.h): nodes, root item, first item of its LDS window, stack entries
    int light_root = RT_TRAV_DONE, n_light_items = 0, light_need = 0;  // light tree (wide nodes) over the light items behind the triangles; stack entries its worst path needs
    int n_wide = 0, wide_need = 0, below_short = 0;  // wide (four-child) shortcut nodes, two items each from short_base on; stack entries their worst path needs; levels of sub-trees below the shortcut tree
    size_t light_box_off = 0, light_place_off = 0;  // floats into `lights`
    float scene_scale = 1.f, box_pad = 0.f;
};

// What does not depend on the layout -- the cold triangle data (the reference's order) and the texture pool, 45 % of a big scene's
// bytes -- is formed and copied to the device by a thread of its own WHILE build_layout runs (rt_upload_scene only).
// order[0..n) = the indices 0 .. n-1 sorted by key, ties by index: a byte-wise radix sort from the lowest byte up (each pass
// is stable, so equal keys keep their index order; a byte that is the same in every key is skipped: the centroids of one
// leaf share their high bytes).  tmp: n ints of scratch.  The builders' sweeps sort a few hundred keys per leaf, three
// times: this is 2-3 x faster there than std::sort on (key, index) pairs and gives the same order.
static void order_by_key32(const uint32_t* key, int n, int* order, int* tmp) {
    for (int k = 0; k < n; k++) order[k] = k;
    int *src = order, *dst = tmp;
    for (int shift = 0; shift < 32; shift += 8) {
        unsigned cnt[257];
        memset(cnt, 0, sizeof(cnt));
        for (int k = 0; k < n; k++) cnt[((key[src[k]] >> shift) & 255u) + 1]++;
        if (n > 0 && cnt[((key[src[0]] >> shift) & 255u) + 1] == (unsigned)n) continue;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (int k = 0; k < n; k++) dst[cnt[(key[src[k]] >> shift) & 255u]++] = src[k];
        std::swap(src, dst);
    }
    if (src != order) memcpy(order, src, (size_t)n * sizeof(int));
}

// The item of the triangle at place p of the item stream (rt_device.h): vertex 0, the two edges, flags, end of its leaf, id, leaf.
static inline void fill_tri_item(float* h, const float* tris, int tri, unsigned char flags, int leaf_end, int leaf_node) {
    const float* t = tris + (size_t)tri * RT_TRI_FLOATS;
    h[0] = t[0]; h[1] = t[1]; h[2] = t[2];
    h[3] = t[3] - t[0]; h[4] = t[4] - t[1]; h[5] = t[5] - t[2];   // edge1 = v1 - v0, main.cu:43
    h[6] = t[6] - t[0]; h[7] = t[7] - t[1]; h[8] = t[8] - t[2];   // edge2 = v2 - v0, main.cu:44
    h[9] = i2f((int)flags); h[10] = i2f(leaf_end); h[11] = i2f(tri);
    h[12] = i2f(leaf_node);
    h[13] = h[14] = h[15] = 0.f;
}

struct EarlyUpload {
    rt_context* c = nullptr;
    std::thread th;
    bool started = false;
    int rc = RT_OK;
    std::string err;
    void* d_cold = nullptr;
    void* d_tex = nullptr;
    // ... and, once the triangles' order is final (after the sub-tree build), their items: written to a host array of their
    // own and copied to a device array of their own while the shortcut tree is built; upload_built moves them into place
    // behind the nodes' items on the device (where the nodes end is not known before the shortcut tree is)
    std::thread th_tris;
    RawFloats tri_items;
    void* d_tri_items = nullptr;
    int rc_tris = RT_OK;
    std::string err_tris;
    void start_tris(const float* tris, int ntri, const int* at, const unsigned char* tflags, const int* leaf_end, const int* leaf_node, int* place) {
        th_tris = std::thread([=] {
            auto fail = [&](const char* what, hipError_t e) { rc_tris = RT_ERR_DEVICE; err_tris = std::string(what) + ": " + hipGetErrorString(e); };
            hipError_t e = hipSetDevice(c->device);
            if (e != hipSuccess) return fail("hipSetDevice", e);
            tri_items.size_to((size_t)ntri * 16);
            const int nt = 4;
            std::vector<std::thread> w;
            auto fill = [&](int wi) {
                for (size_t p = (size_t)ntri * (size_t)wi / (size_t)nt; p < (size_t)ntri * (size_t)(wi + 1) / (size_t)nt; p++)
                    place[(size_t)at[p]] = (int)p, fill_tri_item(&tri_items[p * 16], tris, at[p], tflags[p], leaf_end[p], leaf_node[p]);
            };
            for (int wi = 1; wi < nt; wi++) w.emplace_back(fill, wi);
            fill(0);
            for (auto& t : w) t.join();
            if ((e = hipMalloc(&d_tri_items, tri_items.size() * 4)) != hipSuccess) return fail("hipMalloc(triangle items)", e);
            if ((e = hipMemcpy(d_tri_items, tri_items.data(), tri_items.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(triangle items)", e);
        });
    }
    void start(const rt_scene_desc* s, const float* tris, int ntri, RawFloats* cold) {
        started = true;
        th = std::thread([this, s, tris, ntri, cold] {
            auto fail = [&](const char* what, hipError_t e) { rc = RT_ERR_DEVICE; err = std::string(what) + ": " + hipGetErrorString(e); };
            hipError_t e = hipSetDevice(c->device);
            if (e != hipSuccess) return fail("hipSetDevice", e);
            const size_t tex_bytes = s->n_tex_floats * 4;
            if ((e = hipMalloc(&d_tex, tex_bytes ? tex_bytes : 16)) != hipSuccess) return fail("hipMalloc(textures)", e);
            if (tex_bytes && (e = hipMemcpy(d_tex, s->textures, tex_bytes, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(textures)", e);
            cold->size_to((size_t)std::max(ntri, 1) * 16);
            if (ntri == 0)
                for (int k = 0; k < 16; k++) (*cold)[(size_t)k] = 0.f;
            const int nt = ntri >= (1 << 16) ? 4 : 1;
            std::vector<std::thread> w;
            auto fill = [&](int wi) {
                for (size_t p = (size_t)ntri * (size_t)wi / (size_t)nt; p < (size_t)ntri * (size_t)(wi + 1) / (size_t)nt; p++) {
                    const float* u = tris + p * RT_TRI_FLOATS;
                    float* cd = &(*cold)[p * 16];
                    for (int k = 0; k < 15; k++) cd[k] = u[9 + k];
                    cd[15] = i2f(static_cast<int>(u[24]));  // main.cu:353
                }
            };
            for (int wi = 1; wi < nt; wi++) w.emplace_back(fill, wi);
            fill(0);
            for (auto& t : w) t.join();
            if ((e = hipMalloc(&d_cold, cold->size() * 4)) != hipSuccess) return fail("hipMalloc(cold data)", e);
            if ((e = hipMemcpy(d_cold, cold->data(), cold->size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(cold data)", e);
        });
    }
    int finish() {
        if (started && th.joinable()) th.join();
        if (th_tris.joinable()) th_tris.join();
        if (rc == RT_OK && rc_tris != RT_OK) { rc = rc_tris; err = err_tris; }
        return rc;
    }
    void discard() {  // the upload did not go through: give the early copies back
        finish();
        if (d_cold) (void)hipFree(d_cold);
        if (d_tex) (void)hipFree(d_tex);
        if (d_tri_items) (void)hipFree(d_tri_items);
        d_cold = d_tex = d_tri_items = nullptr;
    }
};

int build_layout(const rt_scene_desc* s, HostLayout& L, std::string& err, UploadTimer& timer, EarlyUpload* early = nullptr) {
    auto fail = [&](int code, const std::string& msg) { err = msg; return code; };
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    if (s->n_tri_floats % RT_TRI_FLOATS) return fail(RT_ERR_INVALID, "n_tri_floats is not a multiple of 25");
    if ((s->n_tri_floats && !s->tris) || (s->n_nodes && !s->nodes) || (s->n_mats && !s->mats) ||
        (s->n_tex_floats && !s->textures) || (s->n_light_tris && !s->light_tris) || s->n_nodes < 0 ||
        s->n_spheres < 0 || s->n_light_tris < 0)
        return fail(RT_ERR_INVALID, "inconsistent scene arrays");
    if (s->n_tri_floats / RT_TRI_FLOATS > 0x7fffffffu / RT_TRI_FLOATS)
        return fail(RT_ERR_INVALID, "too many triangles for int32 float offsets");
    const size_t prefix = s->n_spheres > 0 ? 1 + 5 * (size_t)s->n_spheres : 0;
    const float* tris = s->tris + prefix;
    const int ntri = (int)(s->n_tri_floats / RT_TRI_FLOATS);

    // ---- BVH + triangles: reference AoS nodes / 25-float records -> one item stream
    //      (same topology, BFS numbering of the inner nodes, leaf order of the triangles)
    const int nn = s->n_nodes;
    std::vector<int> inner_index(nn, -1);
    int n_inner = 0;
    for (int i = 0; i < nn; i++) {
        const rt_bvh_node& n = s->nodes[i];
        if (n.child_a == -1) {
            if (n.tri_offs < 0 || n.tri_len < 0 || n.tri_offs % RT_TRI_FLOATS || n.tri_len % RT_TRI_FLOATS ||
                (size_t)n.tri_offs + (size_t)n.tri_len > s->n_tri_floats)
                return fail(RT_ERR_INVALID, "BVH leaf " + std::to_string(i) + " has a bad triangle range");
        } else {
            // children must come after the parent (BFS numbering): guarantees a finite, acyclic walk
            if (n.child_a <= i || n.child_a >= nn || n.child_b <= i || n.child_b >= nn)
                return fail(RT_ERR_INVALID, "BVH node " + std::to_string(i) + " has bad child indices");
            inner_index[i] = n_inner++;
        }
    }
    if ((long long)n_inner + ntri > 0x7fffffffLL / 4) return fail(RT_ERR_LIMIT, "scene too large for 32-bit item indices");
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
        return fail(RT_ERR_LIMIT, "BVH depth " + std::to_string(max_depth) + " exceeds the traversal stack (" +
                                           std::to_string(RT_STACK_ENTRIES) + ")");
    if (early) early->start(s, tris, ntri, &L.cold);
    timer.stage("checks, depths");
    // ---- leaf sub-trees (rt_device.h): a leaf with at least big_leaf triangles may get a tree of its own where the
    //      surface-area estimate says it pays (build_sub below): padded boxes, the triangles reordered inside the
    //      leaf only, built on the host threads
    int big_leaf = RT_BIG_LEAF;
    float cost_inner = RT_SUB_COST_INNER;
    if (const char* e = getenv("RT_BIG_LEAF")) big_leaf = std::max(2, atoi(e));       // experiments only
    if (const char* e = getenv("RT_SUB_COST_INNER")) cost_inner = (float)atof(e);      // experiments only
    float gain = RT_SUB_GAIN;
    if (const char* e = getenv("RT_SUB_GAIN")) gain = (float)atof(e);                  // experiments only
    int sweep_max = RT_SUB_SWEEP_MAX;
    if (const char* e = getenv("RT_SUB_SWEEP_MAX")) sweep_max = std::max(2, atoi(e));  // experiments only
    int sub_min_tris = RT_SUB_MIN_TRIS;
    if (const char* e = getenv("RT_SUB_MIN_TRIS")) sub_min_tris = atoi(e);             // tests: sub-trees in small scenes too
    struct SubNode {  // e >= 0: sub node, e < 0: ~(place of a sub-leaf's first triangle)
        float box[2][6]; int e[2]; int cnt;
        SubNode() {}  // (no zero-fill when the merged vector of millions of them is sized: every element is written)
    };
    // Per-triangle arrays are sized WITHOUT a fill (HostRaw): the area pass below writes their first values on all host
    // threads (one thread's fill of 170 MB, and its page faults, were a fifth of that stage for 10 M triangles).
    const size_t ntri1 = (size_t)std::max(ntri, 1);
    RawVec<int> at(ntri1);  // place -> triangle
    RawVec<int> leaf_end(ntri1);
    RawVec<int> leaf_node(ntri1);  // place -> the reference leaf (node index) that holds it
    RawVec<unsigned char> tflags(ntri1);
    std::vector<int> sub_root((size_t)std::max(nn, 1), -1);
    at[0] = leaf_end[0] = leaf_node[0] = 0;
    tflags[0] = 0;
    auto centroid = [&](int tri, int k) {
        const float* t = tris + (size_t)tri * RT_TRI_FLOATS;
        return (t[k] + t[3 + k] + t[6 + k]) * (1.0f / 3.0f);
    };
    auto grow = [&](float* box, int tri) {
        const float* t = tris + (size_t)tri * RT_TRI_FLOATS;
        for (int k = 0; k < 3; k++) {
            box[k] = std::min(box[k], std::min(t[k], std::min(t[3 + k], t[6 + k])));
            box[3 + k] = std::max(box[3 + k], std::max(t[k], std::max(t[3 + k], t[6 + k])));
        }
    };
    auto half_area = [](const float* b) {
        const float x = b[3] - b[0], y = b[4] - b[1], z = b[5] - b[2];
        return x * y + y * z + z * x;
    };
    // chance that a ray which reaches a node's box reaches a child's: the ratio of the surface areas, plus the share
    // of all rays that START on a triangle of the child (every bounce does, and then the child's box is reached for
    // sure: what makes sub-trees over wall-sized triangles a loss)
    RawVec<float> tri_area(ntri1);
    tri_area[0] = 0.f;
    double total_area = 0.0;
    float scene_scale = 0.f;  // largest |coordinate| of the scene
    int n_workers_max = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("RT_HOST_THREADS")) n_workers_max = std::min(16, std::max(1, atoi(e)));
    {   // in fixed chunks of 2^18 triangles whose partial sums are added in chunk order: the same total on any number of threads
        const size_t chunk = (size_t)1 << 18, nch = ((size_t)ntri + chunk - 1) / chunk;
        std::vector<double> part_area(std::max<size_t>(nch, 1), 0.0);
        std::vector<float> part_scale(std::max<size_t>(nch, 1), 0.f);
        std::atomic<size_t> next{0};
        auto run = [&] {
            for (size_t c; (c = next.fetch_add(1)) < nch;) {
                double a = 0.0;
                float sc_max = 0.f;
                for (size_t i = c * chunk; i < std::min((size_t)ntri, (c + 1) * chunk); i++) {
                    const float* t = tris + i * RT_TRI_FLOATS;
                    for (int k = 0; k < 9; k++) sc_max = std::max(sc_max, std::fabs(t[k]));
                    const H3 e1{t[3] - t[0], t[4] - t[1], t[5] - t[2]}, e2{t[6] - t[0], t[7] - t[1], t[8] - t[2]};
                    tri_area[i] = 0.5f * hlen(hcross(e1, e2));
                    a += tri_area[i];
                    at[i] = (int)i;
                    leaf_end[i] = leaf_node[i] = 0;  // (a triangle outside every leaf keeps these)
                    tflags[i] = 0;
                }
                part_area[c] = a;
                part_scale[c] = sc_max;
            }
        };
        std::vector<std::thread> th;
        for (int wi = 1; wi < (nch > 1 ? n_workers_max : 1); wi++) th.emplace_back(run);
        run();
        for (auto& t : th) t.join();
        for (size_t c = 0; c < nch; c++) {
            total_area += part_area[c];
            scene_scale = std::max(scene_scale, part_scale[c]);
        }
    }
    const float inv_total_area = total_area > 0.0 ? (float)(1.0 / total_area) : 0.f;
    if (!(scene_scale > 0.f) || !(scene_scale < 1e30f)) scene_scale = 1.f;
    // Boxes this library adds (sub-trees, light groups) are padded by 1e-5 of the scene's scale S (largest |coordinate|),
    // and cull only for rays that start within RT_CULL_FAR x S of the origin on every axis.  The slab test forms
    // (bound - o) * (1 / d): a few ulp of the distance to the box, at most 4 x 1.2e-7 x (8 S + S) = 4.5e-6 S for such
    // a ray -- and the triangle test is blurred by as much.  Rays from farther away test every triangle of a leaf,
    // as the reference does.
    const float box_pad = 1e-5f * scene_scale;
    const float kEmpty[6] = {3e38f, 3e38f, 3e38f, -3e38f, -3e38f, -3e38f};
    const bool getenv_old_sweep = getenv("RT_SUB_RESORT") != nullptr;  // (tests: the sort-at-every-node form of the sweep, same result)
    struct SubWork {  // what one host thread builds: leaves are independent (each reorders its own range of `at`)
        std::vector<SubNode> subs;
        std::vector<int> sorted, best_sorted;
        std::vector<uint64_t> keys;
        std::vector<uint32_t> keys32;
        // sweep_sub: the node of at most sweep_max triangles being built: its triangles, their boxes and areas, and per
        // axis the local indices in sorted order (position p of `at` <-> entry p - base)
        int base = 0;
        std::vector<int> ltri, ord[3], scratch;
        std::vector<float> lbox, larea;
        std::vector<unsigned char> side;
        std::vector<float> suffix;
        std::vector<std::pair<int, int>> roots;  // (leaf node, root in subs)
        std::vector<int> boxed;                  // candidate leaves in the order they were built ...
        std::vector<float> boxes;                // ... and the padded box of each one's triangles (6 floats)
        int deepest = 0;
        int leaf_deepest = 0, max_below_leaf = 0;  // deepest level of the sub-tree being built; most levels any sub-tree adds below its leaf
    };
    // The sweep builder below a node of at most sweep_max triangles.  Its triangles are sorted ONCE per axis (by centroid,
    // ties by triangle id: as one 64-bit key, the float's bits made monotonic, -0 folded into +0); a split keeps the
    // three orders by stable partition, so a child's orders are what sorting the child would give (the keys are
    // distinct) -- the same splits, leaves and triangle order as sorting at every node, without the sorting.
    // Per-triangle centroids are not needed after the sort; boxes and areas sit in small local arrays.
    auto sweep_sub = [&](auto&& self, SubWork& w, int lo, int hi, int d, float* box, bool root) -> int {
        const int n = hi - lo, o = lo - w.base;
        // A box lives in two SSE registers (mins, maxs; the fourth lane stays 0): the same min / max / subtract / multiply /
        // add per component as the scalar form of build_sub below, in the same order -- the same floats, a third of the
        // instructions on the sweep's dependent chains.
        const __m128 empty_lo = _mm_setr_ps(3e38f, 3e38f, 3e38f, 0.f), empty_hi = _mm_setr_ps(-3e38f, -3e38f, -3e38f, 0.f);
        auto half_area4 = [](__m128 mn, __m128 mx) {  // x * y + y * z + z * x of the extents, summed left to right
            const __m128 e = _mm_sub_ps(mx, mn);
            const __m128 p = _mm_mul_ps(e, _mm_shuffle_ps(e, e, _MM_SHUFFLE(3, 0, 2, 1)));  // (xy, yz, zx, 0)
            return _mm_cvtss_f32(_mm_add_ss(_mm_add_ss(p, _mm_shuffle_ps(p, p, 1)), _mm_shuffle_ps(p, p, 2)));
        };
        const float* lb = w.lbox.data();
        __m128 tlo = empty_lo, thi = empty_hi;
        for (int k = 0; k < n; k++) {
            const float* b = lb + 8 * (size_t)w.ord[0][(size_t)(o + k)];
            tlo = _mm_min_ps(_mm_loadu_ps(b), tlo);  // (min_ps(a, b) = a < b ? a : b: std::min(tight, b) with its operands swapped)
            thi = _mm_max_ps(_mm_loadu_ps(b + 4), thi);
        }
        float t_lo[4], t_hi[4];
        _mm_storeu_ps(t_lo, tlo);
        _mm_storeu_ps(t_hi, thi);
        for (int k = 0; k < 3; k++) { box[k] = t_lo[k]; box[3 + k] = t_hi[k]; }
        for (int k = 0; k < 3; k++) { box[k] -= box_pad; box[3 + k] += box_pad; }
        w.deepest = std::max(w.deepest, d);
        w.leaf_deepest = std::max(w.leaf_deepest, d);
        int mid = -1;
        const float pa = half_area4(tlo, thi);
        if (n >= 2 && d < RT_STACK_ENTRIES && pa > 0.f) {
            float best = 3e38f;  // expected triangle tests below this node if it is split
            int best_axis = -1, best_k = -1;
            for (int axis = 0; axis < 3; axis++) {
                const int* so = &w.ord[axis][(size_t)o];
                __m128 blo = empty_lo, bhi = empty_hi;
                float ar = 0.f;
                for (int k = n - 1; k >= 1; k--) {  // box area and triangle area of so[k..n)
                    const float* b = lb + 8 * (size_t)so[k];
                    blo = _mm_min_ps(_mm_loadu_ps(b), blo);
                    bhi = _mm_max_ps(_mm_loadu_ps(b + 4), bhi);
                    ar += w.larea[(size_t)so[k]];
                    w.suffix[2 * (size_t)k] = half_area4(blo, bhi);
                    w.suffix[2 * (size_t)k + 1] = ar;
                }
                blo = empty_lo;
                bhi = empty_hi;
                ar = 0.f;
                for (int k = 1; k < n; k++) {  // left = the first k
                    const float* b = lb + 8 * (size_t)so[k - 1];
                    blo = _mm_min_ps(_mm_loadu_ps(b), blo);
                    bhi = _mm_max_ps(_mm_loadu_ps(b + 4), bhi);
                    ar += w.larea[(size_t)so[k - 1]];
                    const float cst = std::min(1.f, half_area4(blo, bhi) / pa + ar * inv_total_area) * (float)k +
                                      std::min(1.f, w.suffix[2 * (size_t)k] / pa + w.suffix[2 * (size_t)k + 1] * inv_total_area) * (float)(n - k);
                    if (cst < best) { best = cst; best_axis = axis; best_k = k; }
                }
            }
            if (best_axis >= 0) {
                const int* so = &w.ord[best_axis][(size_t)o];
                for (int k = 0; k < n; k++) at[(size_t)(lo + k)] = w.ltri[(size_t)so[k]];  // the node's triangles in the order of the best axis
                mid = lo + best_k;
                // walk n triangles, or one more node and the triangles of the children that are reached
                if (n <= RT_SUB_LEAF_MAX && !(cost_inner + best < gain * (float)n)) mid = -1;
                if (mid >= 0) {
                    for (int k = 0; k < n; k++) w.side[(size_t)so[k]] = (unsigned char)(k < best_k ? 1 : 0);
                    for (int axis = 0; axis < 3; axis++) {
                        if (axis == best_axis) continue;
                        int* oa = &w.ord[axis][(size_t)o];
                        int nl = 0, nr = 0;
                        for (int k = 0; k < n; k++) {
                            const int t = oa[k];
                            if (w.side[(size_t)t]) oa[nl++] = t;      // (never past the read position)
                            else w.scratch[(size_t)nr++] = t;
                        }
                        memcpy(oa + nl, w.scratch.data(), (size_t)nr * sizeof(int));
                    }
                }
            }
        }
        if (mid < 0) {
            if (root) {
                std::sort(at.begin() + lo, at.begin() + hi);
                return -1 - 0x40000000;
            }
            for (int p = lo; p < hi; p++) tflags[(size_t)p] = (unsigned char)(4 | (p == hi - 1 ? 1 : 0));
            return ~lo;
        }
        const int idx = (int)w.subs.size();
        w.subs.push_back(SubNode());
        float ba[6], bb[6];
        const int ea = self(self, w, lo, mid, d + 1, ba, false), eb = self(self, w, mid, hi, d + 1, bb, false);
        SubNode& nd = w.subs[(size_t)idx];
        memcpy(nd.box[0], ba, sizeof(ba));
        memcpy(nd.box[1], bb, sizeof(bb));
        nd.e[0] = ea;
        nd.e[1] = eb;
        nd.cnt = 0;
        return idx;
    };
    auto sweep_begin = [&](SubWork& w, int lo, int hi) {  // local copies and the three sorted orders of at[lo, hi)
        const int n = hi - lo;
        w.base = lo;
        w.ltri.resize((size_t)n);
        w.lbox.resize(8 * (size_t)n);  // per triangle: mins, 0, maxs, 0
        w.larea.resize((size_t)n);
        w.side.resize((size_t)n);
        w.scratch.resize((size_t)n);
        w.suffix.resize(2 * ((size_t)n + 1));
        w.keys.resize((size_t)n);
        for (int k = 0; k < n; k++) w.ltri[(size_t)k] = at[(size_t)(lo + k)];
        // local indices in triangle-id order (they already are below a reference leaf): the index then breaks ties like the id
        if (!std::is_sorted(w.ltri.begin(), w.ltri.end())) std::sort(w.ltri.begin(), w.ltri.end());
        for (int k = 0; k < n; k++) {
            const int tri = w.ltri[(size_t)k];
            float b6[6];
            memcpy(b6, kEmpty, sizeof(b6));
            grow(b6, tri);
            float* b = &w.lbox[8 * (size_t)k];
            b[0] = b6[0]; b[1] = b6[1]; b[2] = b6[2]; b[3] = 0.f;
            b[4] = b6[3]; b[5] = b6[4]; b[6] = b6[5]; b[7] = 0.f;
            w.larea[(size_t)k] = tri_area[(size_t)tri];
        }
        w.keys32.resize((size_t)n);
        for (int axis = 0; axis < 3; axis++) {
            for (int k = 0; k < n; k++) {
                const int tri = w.ltri[(size_t)k];
                const float cc = centroid(tri, axis) + 0.0f;
                uint32_t b;
                memcpy(&b, &cc, 4);
                b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
                w.keys32[(size_t)k] = b;
                w.keys[(size_t)k] = ((uint64_t)b << 32) | (uint32_t)k;
            }
            w.ord[axis].resize((size_t)n);
            if (n >= 48) {
                order_by_key32(w.keys32.data(), n, w.ord[axis].data(), w.scratch.data());
                continue;
            }
            std::sort(w.keys.begin(), w.keys.end());
            for (int k = 0; k < n; k++) w.ord[axis][(size_t)k] = (int)(uint32_t)w.keys[(size_t)k];
        }
    };
    auto build_sub = [&](auto&& self, SubWork& w, int lo, int hi, int d, float* box, bool root) -> int {
        const int n = hi - lo;
        if (n <= sweep_max && !getenv_old_sweep) {
            sweep_begin(w, lo, hi);
            return sweep_sub(sweep_sub, w, lo, hi, d, box, root);
        }
        float tight[6];
        memcpy(tight, kEmpty, sizeof(tight));
        for (int p = lo; p < hi; p++) grow(tight, at[(size_t)p]);
        memcpy(box, tight, sizeof(tight));
        for (int k = 0; k < 3; k++) { box[k] -= box_pad; box[3 + k] += box_pad; }  // a sub-tree box must never make the ray miss a triangle it hits
        w.deepest = std::max(w.deepest, d);
        w.leaf_deepest = std::max(w.leaf_deepest, d);
        int mid = -1;
        const float pa = half_area(tight);
        if (n >= 2 && d < RT_STACK_ENTRIES && pa > 0.f) {
            float best = 3e38f;  // expected triangle tests below this node if it is split
            if (n <= sweep_max) {
                int best_axis = -1, best_k = -1;
                for (int axis = 0; axis < 3; axis++) {
                    // by centroid, ties by triangle id: as ONE 64-bit key (the float's bits made monotonic, -0 folded into +0)
                    w.keys.resize((size_t)n);
                    for (int p = lo; p < hi; p++) {
                        const int tri = at[(size_t)p];
                        const float cc = centroid(tri, axis) + 0.0f;
                        uint32_t b;
                        memcpy(&b, &cc, 4);
                        b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
                        w.keys[(size_t)(p - lo)] = ((uint64_t)b << 32) | (uint32_t)tri;
                    }
                    std::sort(w.keys.begin(), w.keys.end());
                    w.sorted.resize((size_t)n);
                    for (int k = 0; k < n; k++) w.sorted[(size_t)k] = (int)(uint32_t)w.keys[(size_t)k];
                    w.suffix.assign(2 * ((size_t)n + 1), 0.f);  // box area and triangle area of w.sorted[k..n)
                    float bx[6], ar = 0.f;
                    memcpy(bx, kEmpty, sizeof(bx));
                    for (int k = n - 1; k >= 1; k--) {
                        grow(bx, w.sorted[(size_t)k]);
                        ar += tri_area[(size_t)w.sorted[(size_t)k]];
                        w.suffix[2 * (size_t)k] = half_area(bx);
                        w.suffix[2 * (size_t)k + 1] = ar;
                    }
                    memcpy(bx, kEmpty, sizeof(bx));
                    ar = 0.f;
                    for (int k = 1; k < n; k++) {  // left = the first k
                        grow(bx, w.sorted[(size_t)k - 1]);
                        ar += tri_area[(size_t)w.sorted[(size_t)k - 1]];
                        const float cst = std::min(1.f, half_area(bx) / pa + ar * inv_total_area) * (float)k +
                                          std::min(1.f, w.suffix[2 * (size_t)k] / pa + w.suffix[2 * (size_t)k + 1] * inv_total_area) * (float)(n - k);
                        if (cst < best) { best = cst; best_axis = axis; best_k = k; }
                    }
                    if (best_axis == axis) w.best_sorted = w.sorted;  // (this axis holds the best split so far: keep its order)
                }
                std::copy(w.best_sorted.begin(), w.best_sorted.end(), at.begin() + lo);
                mid = lo + best_k;
            } else {
                float cmn[3] = {3e38f, 3e38f, 3e38f}, cmx[3] = {-3e38f, -3e38f, -3e38f};
                for (int p = lo; p < hi; p++)
                    for (int k = 0; k < 3; k++) {
                        const float cc = centroid(at[(size_t)p], k);
                        cmn[k] = std::min(cmn[k], cc);
                        cmx[k] = std::max(cmx[k], cc);
                    }
                int axis = 0;
                for (int k = 1; k < 3; k++)
                    if (cmx[k] - cmn[k] > cmx[axis] - cmn[axis]) axis = k;
                mid = lo + n / 2;
                std::nth_element(at.begin() + lo, at.begin() + mid, at.begin() + hi, [&](int a, int b) {
                    const float ca = centroid(a, axis), cb = centroid(b, axis);
                    return ca < cb || (ca == cb && a < b);
                });
                float bl[6], br[6], al = 0.f, ar = 0.f;
                memcpy(bl, kEmpty, sizeof(bl));
                memcpy(br, kEmpty, sizeof(br));
                for (int p = lo; p < mid; p++) { grow(bl, at[(size_t)p]); al += tri_area[(size_t)at[(size_t)p]]; }
                for (int p = mid; p < hi; p++) { grow(br, at[(size_t)p]); ar += tri_area[(size_t)at[(size_t)p]]; }
                best = std::min(1.f, half_area(bl) / pa + al * inv_total_area) * (float)(mid - lo) +
                       std::min(1.f, half_area(br) / pa + ar * inv_total_area) * (float)(hi - mid);
            }
            // walk n triangles, or one more node and the triangles of the children that are reached
            if (n <= RT_SUB_LEAF_MAX && !(cost_inner + best < gain * (float)n)) mid = -1;
        }
        if (mid < 0) {
            if (root) {  // not worth a sub-tree: the leaf stays as it is, in the reference's order (the sweep has sorted it)
                std::sort(at.begin() + lo, at.begin() + hi);
                return -1 - 0x40000000;
            }
            for (int p = lo; p < hi; p++) tflags[(size_t)p] = (unsigned char)(4 | (p == hi - 1 ? 1 : 0));
            return ~lo;
        }
        const int idx = (int)w.subs.size();
        w.subs.push_back(SubNode());
        float ba[6], bb[6];
        const int ea = self(self, w, lo, mid, d + 1, ba, false), eb = self(self, w, mid, hi, d + 1, bb, false);
        SubNode& nd = w.subs[(size_t)idx];
        memcpy(nd.box[0], ba, sizeof(ba));
        memcpy(nd.box[1], bb, sizeof(bb));
        nd.e[0] = ea;
        nd.e[1] = eb;
        nd.cnt = 0;
        return idx;
    };
    std::vector<int> candidates;
    {
        std::vector<int> leaves;  // (node indices; then sorted by their first triangle)
        for (int i = 0; i < nn; i++) {
            const rt_bvh_node& n = s->nodes[i];
            if (n.child_a != -1) continue;
            if (n.tri_len > 0) leaves.push_back(i);
            // (the traversal stack must have room below this leaf)
            if (n.tri_len / RT_TRI_FLOATS >= big_leaf && ntri >= sub_min_tris && depth[i] < RT_STACK_ENTRIES) candidates.push_back(i);
        }
        auto by_first = [&](int a, int b) { return s->nodes[a].tri_offs < s->nodes[b].tri_offs; };
        if (!std::is_sorted(leaves.begin(), leaves.end(), by_first)) std::sort(leaves.begin(), leaves.end(), by_first);
        // a triangle in two leaves would need two "last" flags: the reference builder never does that
        for (size_t k = 1; k < leaves.size(); k++)
            if (s->nodes[leaves[k]].tri_offs < s->nodes[leaves[k - 1]].tri_offs + s->nodes[leaves[k - 1]].tri_len)
                return fail(RT_ERR_INVALID, "BVH leaves overlap in the triangle array");
        // the leaves' ranges are disjoint: their flags are written by all host threads, a contiguous run of leaves each
        const int nw = leaves.size() >= 4096 ? n_workers_max : 1;
        auto fill = [&](int wi) {
            for (size_t li = leaves.size() * (size_t)wi / (size_t)nw; li < leaves.size() * (size_t)(wi + 1) / (size_t)nw; li++) {
                const int i = leaves[li];
                const int first = s->nodes[i].tri_offs / RT_TRI_FLOATS, cnt = s->nodes[i].tri_len / RT_TRI_FLOATS;
                for (int k = 0; k < cnt; k++) {
                    tflags[(size_t)first + k] = (unsigned char)((k == cnt - 1 ? 1 : 0) | (k == 0 ? 2 : 0));
                    leaf_end[(size_t)first + k] = first + cnt;
                    leaf_node[(size_t)first + k] = i;
                }
            }
        };
        std::vector<std::thread> th;
        for (int wi = 1; wi < nw; wi++) th.emplace_back(fill, wi);
        fill(0);
        for (auto& t : th) t.join();
    }
    timer.stage("areas, flags, candidates");
    // The candidates are cut into n_workers contiguous PARTS (a fixed number: the result does not depend on the thread
    // count), several per host thread, and the threads take parts as they come free: leaves differ in cost, and with one
    // part per thread the slowest thread took 1.15 x the mean.
    const int n_threads = candidates.size() >= 4096 ? n_workers_max : 1;
    const int n_workers = candidates.size() >= 4096 ? 128 : 1;
    auto run_parts = [&](auto&& fn) {
        std::atomic<int> next{0};
        auto body = [&] {
            for (int wi; (wi = next.fetch_add(1)) < n_workers;) fn(wi);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < n_threads; t++) th.emplace_back(body);
        body();
        for (auto& t : th) t.join();
    };
    std::vector<SubWork> work((size_t)n_workers);
    {
        auto run = [&](int wi) {
            SubWork& w = work[(size_t)wi];
            const size_t lo = candidates.size() * (size_t)wi / (size_t)n_workers, hi = candidates.size() * (size_t)(wi + 1) / (size_t)n_workers;
            for (size_t ci = lo; ci < hi; ci++) {
                const int i = candidates[ci];
                const int first = s->nodes[i].tri_offs / RT_TRI_FLOATS, cnt = s->nodes[i].tri_len / RT_TRI_FLOATS;
                float box[6];
                w.leaf_deepest = depth[i];
                const int root = build_sub(build_sub, w, first, first + cnt, depth[i], box, true);
                w.boxed.push_back(i);
                w.boxes.insert(w.boxes.end(), box, box + 6);
                if (root >= 0) {
                    w.subs[(size_t)root].cnt = cnt;
                    w.roots.push_back({i, root});
                    w.max_below_leaf = std::max(w.max_below_leaf, w.leaf_deepest - depth[i]);
                }
            }
        };
        run_parts(run);
    }
    timer.stage("sub-tree build (threads)");
    // fn(chunk, lo, hi) over [0, n) cut into n_workers_max ranges, one host thread each (one range when n is small)
    auto par_ranges = [&](size_t n, auto&& fn) {
        const int nw = n >= ((size_t)1 << 16) ? n_workers_max : 1;
        std::vector<std::thread> th;
        for (int wi = 1; wi < nw; wi++) th.emplace_back([&, wi] { fn(wi, n * (size_t)wi / (size_t)nw, n * (size_t)(wi + 1) / (size_t)nw); });
        fn(0, (size_t)0, n / (size_t)nw);
        for (auto& t : th) t.join();
        return nw;
    };
    BigVec<SubNode> subs;  // the workers' nodes one after the other (contiguous leaf ranges: same order as one thread's)
    int deepest = max_depth;    // of the whole structure: sizes the traversal stack
    {
        std::vector<size_t> off((size_t)n_workers + 1, 0);
        for (int wi = 0; wi < n_workers; wi++) off[(size_t)wi + 1] = off[(size_t)wi] + work[(size_t)wi].subs.size();
        subs.resize(off[(size_t)n_workers]);
        auto copy = [&](int wi) {
            const SubWork& w = work[(size_t)wi];
            const int o = (int)off[(size_t)wi];
            for (size_t k = 0; k < w.subs.size(); k++) {
                SubNode nd = w.subs[k];
                for (int c = 0; c < 2; c++)
                    if (nd.e[c] >= 0) nd.e[c] += o;
                subs[(size_t)o + k] = nd;
            }
        };
        run_parts(copy);
        for (int wi = 0; wi < n_workers; wi++) {
            for (const auto& r : work[(size_t)wi].roots) sub_root[(size_t)r.first] = (int)off[(size_t)wi] + r.second;
            deepest = std::max(deepest, work[(size_t)wi].deepest);
        }
    }
    const int n_top = n_inner, n_sub = (int)subs.size();
    L.place.resize((size_t)std::max(ntri, 1));
    L.place[0] = 0;  // (no triangles: one entry, defined)
    // the triangles' order, flags and leaves are final from here on: their items can be written and copied meanwhile
    int early_tris_min = 1 << 20;  // (below: the thread costs more than the copy)
    if (const char* e = getenv("RT_EARLY_TRIS_MIN")) early_tris_min = std::max(1, atoi(e));  // tests: small scenes take this way too
    const bool early_tris = early && early->started && ntri >= early_tris_min && !getenv("RT_NO_EARLY_TRIS");
    if (early_tris) early->start_tris(tris, ntri, at.data(), tflags.data(), leaf_end.data(), leaf_node.data(), L.place.data());
    struct JoinTris {  // that thread reads arrays of this function: it has ended before they go, whichever way the function is left
        EarlyUpload* e;
        ~JoinTris() { if (e && e->th_tris.joinable()) e->th_tris.join(); }
    } join_tris{early_tris ? early : nullptr};
    timer.stage("sub-tree nodes merged");
    // ---- lights: ImportanceTri constructor (importanceTri.h:19-25) evaluated once.  With enough lights for the
    //      wave-cooperative loop (rt_device.h) the records are stored in SPATIAL order (Morton code of the centroid), so
    //      that a group of RT_LIGHT_GROUP consecutive records is a small patch with a tight box; place[i] says where the
    //      reference's i-th light went, and every record carries its reference index (the sum over the lights is formed
    //      in that order, see wave_light_pdf).
    const int nl = s->n_light_tris;
    if (nl >= (1 << 24)) return fail(RT_ERR_LIMIT, "more than 2^24 - 1 importance-sampled triangles (the light loop packs a light index into 24 bits)");
    const int n_light_groups = (std::max(nl, 1) + RT_LIGHT_GROUP - 1) / RT_LIGHT_GROUP;
    const size_t light_box_off = (size_t)n_light_groups * RT_LIGHT_GROUP * 4 * RT_LIGHT_PLANES;  // padded bounds per RT_LIGHT_GROUP lights behind the records (whole groups)
    const size_t light_place_off = light_box_off + (size_t)n_light_groups * 8;  // int place[nl] behind the boxes
    std::vector<float>& lights = L.lights;
    lights.assign(light_place_off + (size_t)((std::max(nl, 1) + 3) & ~3), 0.f);
    for (int g = 0; g < n_light_groups; g++)
        for (int k = 0; k < 3; k++) {
            lights[light_box_off + (size_t)g * 8 + k] = 3e38f;
            lights[light_box_off + (size_t)g * 8 + 4 + k] = -3e38f;
        }
    for (int i = 0; i < nl; i++) {
        const int off = s->light_tris[i];
        if (off < 0 || off % RT_TRI_FLOATS || (size_t)off + RT_TRI_FLOATS > s->n_tri_floats)
            return fail(RT_ERR_INVALID, "light triangle offset out of range");
    }
    std::vector<int> light_order((size_t)std::max(nl, 1));  // light_order[j] = reference index of the light stored at j
    for (int i = 0; i < nl; i++) light_order[(size_t)i] = i;
    if (nl >= RT_LIGHTS_COOP_MIN && !getenv("RT_LIGHTS_UNSORTED")) {
        float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
        std::vector<float> cen((size_t)nl * 3);
        for (int i = 0; i < nl; i++) {
            const float* t = tris + s->light_tris[i];
            for (int k = 0; k < 3; k++) {
                const float cc = (t[k] + t[3 + k] + t[6 + k]) * (1.0f / 3.0f);
                cen[(size_t)i * 3 + k] = cc;
                lo[k] = std::min(lo[k], cc);
                hi[k] = std::max(hi[k], cc);
            }
        }
        auto spread = [](unsigned v) {  // 10 bits -> every third bit
            v &= 1023u;
            v = (v | (v << 16)) & 0x030000FFu;
            v = (v | (v << 8)) & 0x0300F00Fu;
            v = (v | (v << 4)) & 0x030C30C3u;
            v = (v | (v << 2)) & 0x09249249u;
            return v;
        };
        std::vector<unsigned> code((size_t)nl);
        for (int i = 0; i < nl; i++) {
            unsigned q[3];
            for (int k = 0; k < 3; k++) {
                const float w = hi[k] - lo[k];
                const float u = w > 0.f ? (cen[(size_t)i * 3 + k] - lo[k]) / w : 0.f;
                q[k] = (unsigned)std::min(1023.f, std::max(0.f, u * 1024.f));
            }
            code[(size_t)i] = spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
        }
        std::stable_sort(light_order.begin(), light_order.begin() + nl, [&](int a, int b) { return code[(size_t)a] < code[(size_t)b]; });
    }
    for (int j = 0; j < nl; j++) {
        const int i = light_order[(size_t)j];
        const float* t = tris + s->light_tris[i];
        const H3 p1{t[0], t[1], t[2]}, p2{t[3], t[4], t[5]}, p3{t[6], t[7], t[8]};
        const H3 e1 = hsub(p2, p1), e2 = hsub(p3, p1);
        const H3 cr = hcross(e1, e2);
        const H3 nrm = hnormalize(cr);
        const float area = 0.5f * hlen(cr);
        float Lr[24];
        {   // bounding sphere of the triangle about its centroid, radius padded (wave_light_pdf_q drops a light only if the ray misses it)
            const H3 cen{(p1.x + p2.x + p3.x) * (1.0f / 3.0f), (p1.y + p2.y + p3.y) * (1.0f / 3.0f), (p1.z + p2.z + p3.z) * (1.0f / 3.0f)};
            const float r = std::max(hlen(hsub(p1, cen)), std::max(hlen(hsub(p2, cen)), hlen(hsub(p3, cen))));
            Lr[20] = cen.x; Lr[21] = cen.y; Lr[22] = cen.z; Lr[23] = r * 1.001f + box_pad;
        }
        Lr[0] = p1.x; Lr[1] = p1.y; Lr[2] = p1.z; Lr[3] = e1.x; Lr[4] = e1.y; Lr[5] = e1.z;
        Lr[6] = e2.x; Lr[7] = e2.y; Lr[8] = e2.z; Lr[9] = nrm.x; Lr[10] = nrm.y; Lr[11] = nrm.z;
        Lr[12] = area; Lr[13] = p2.x; Lr[14] = p2.y; Lr[15] = p2.z; Lr[16] = p3.x; Lr[17] = p3.y; Lr[18] = p3.z;
        Lr[19] = i2f(i);                                    // the light's place in the reference's list
        for (int k = 0; k < 24; k++) lights[RT_LIGHT_F4(j, k / 4) * 4 + (size_t)(k % 4)] = Lr[k];  // plane by plane per group (rt_device.h)
        lights[light_place_off + (size_t)i] = i2f(j);
        float* B = &lights[light_box_off + (size_t)(j / RT_LIGHT_GROUP) * 8];
        for (int k = 0; k < 3; k++) {
            B[k] = std::min(B[k], std::min(t[k], std::min(t[3 + k], t[6 + k])));
            B[4 + k] = std::max(B[4 + k], std::max(t[k], std::max(t[3 + k], t[6 + k])));
        }
    }
    for (int g = 0; g < n_light_groups && s->n_light_tris > 0; g++) {  // pad: the culling test must never say no wrongly
        float* B = &lights[light_box_off + (size_t)g * 8];
        for (int k = 0; k < 3; k++) {
            B[k] -= box_pad;
            B[4 + k] += box_pad;
        }
    }

    // ---- SHORTCUT TREE (rt_device.h): a surface-area tree of this library's own over the reference's non-empty LEAVES
    //      (each with whatever hangs below it: its triangles or its sub-tree), padded boxes, walked with the sub-trees'
    //      slack rule.  It finds the closest hit in fewer visits than the reference's mean-split tree; a per-ray
    //      certificate (exact box test of the hit's reference leaf against the runner-up from another leaf) says whether
    //      the reference's own walk must return the same triangle, and the ray walks the reference's tree if not.
    //      Needs the reference's boxes to be nested (child inside parent), as its builder makes them.
    struct ShortPrim { float b[6]; float c[3]; float cost; int node; };
    struct ShortNode { float box[2][6]; int e[2]; };  // e >= 0: shortcut node, e < 0: ~(ShortPrim::node of what hangs there)
    BigVec<ShortNode> shorts;
    int short_root = -1, short_levels = 0, below_leaf = 0;
    bool short_runs = false, short_single = false;
    int short_cut = 0;  // levels below the reference's leaves at which the shortcut tree's own leaves stand (when it is not over runs)
    for (int wi = 0; wi < n_workers; wi++) below_leaf = std::max(below_leaf, work[(size_t)wi].max_below_leaf);
    {
        // a scene too small for sub-trees (the Cornell box): the shortcut tree over its SINGLE triangles -- a shortcut search
        // then leaves a leaf after one test, whatever the 'last' flags of the reference's leaves say (Scene::short_single)
        const bool tiny = ntri < sub_min_tris || getenv("RT_SHORT_SINGLE");  // (RT_SHORT_SINGLE: experiments only)
        bool want = nn > 1 && ntri >= 2 && !getenv("RT_NO_SHORTCUT") && !(tiny && getenv("RT_NO_SHORTCUT_TINY"));
        for (int i = 0; i < nn && want; i++) {
            const rt_bvh_node& n = s->nodes[i];
            if (n.child_a == -1) continue;
            for (int ch : {n.child_a, n.child_b})
                for (int k = 0; k < 3; k++)
                    if (!(s->nodes[ch].bbox_min[k] >= n.bbox_min[k] && s->nodes[ch].bbox_max[k] <= n.bbox_max[k])) want = false;
        }
        // What hangs below the shortcut tree: the RUNS of the item stream -- triangles tested one after the other up to a
        // 'last' flag: a sub-leaf, or a whole reference leaf without a sub-tree -- so that the walk never enters a sub-tree;
        // in a scene with more than RT_SHORT_RUNS_MAX of them the reference's leaves with their sub-trees (a smaller build).
        BigVec<ShortPrim> prims;
        if (want) {
            size_t runs_max = RT_SHORT_RUNS_MAX;
            if (const char* e = getenv("RT_SHORT_RUNS_MAX")) runs_max = (size_t)atoll(e);  // experiments only
            std::vector<size_t> c_runs((size_t)n_workers_max + 1, 0), c_leaves((size_t)n_workers_max + 1, 0);
            const int nr = par_ranges((size_t)nn, [&](int wi, size_t lo, size_t hi) {  // count per range of nodes, prefix, fill
                size_t r = 0, l = 0;
                for (size_t i = lo; i < hi; i++) {
                    const rt_bvh_node& n = s->nodes[i];
                    if (n.child_a != -1 || n.tri_len == 0) continue;
                    const size_t first = (size_t)(n.tri_offs / RT_TRI_FLOATS), cnt = (size_t)(n.tri_len / RT_TRI_FLOATS);
                    l++;
                    if (tiny) r += cnt;
                    else
                        for (size_t k = 0; k < cnt; k++) r += tflags[first + k] & 1;
                }
                c_runs[(size_t)wi + 1] = r;
                c_leaves[(size_t)wi + 1] = l;
            });
            for (int wi = 0; wi < nr; wi++) { c_runs[(size_t)wi + 1] += c_runs[(size_t)wi]; c_leaves[(size_t)wi + 1] += c_leaves[(size_t)wi]; }
            short_runs = tiny || (c_runs[(size_t)nr] <= runs_max && !getenv("RT_SHORT_LEAVES"));
            short_single = tiny;
            // boxes the sub-tree build has formed already (padded once; the shortcut builder pads again): a candidate leaf's
            // and, through the sub-tree nodes, every sub-leaf's -- the triangles are not read again
            std::vector<const float*> leaf_box((size_t)nn, nullptr);
            for (const SubWork& w : work)
                for (size_t k = 0; k < w.boxed.size(); k++) leaf_box[(size_t)w.boxed[k]] = &w.boxes[6 * k];
            auto set_prim = [&](ShortPrim& p, const float* known, int first_place, int len, int code, float cost) {
                if (known) {
                    memcpy(p.b, known, sizeof(p.b));
                } else {
                    memcpy(p.b, kEmpty, sizeof(p.b));
                    for (int j = 0; j < len; j++) grow(p.b, at[(size_t)first_place + j]);
                }
                for (int c = 0; c < 3; c++) p.c[c] = 0.5f * (p.b[c] + p.b[3 + c]);
                p.cost = cost;   // steps a ray spends below the box, roughly
                p.node = code;   // what hangs there: 2 * (first place of a run of triangles), or 2 * (sub-tree node) + 1
            };
            if (short_runs) {
                prims.resize(c_runs[(size_t)nr]);
                BigVec<int> run_src;  // place of a sub-leaf's first triangle -> 2 * sub node + side
                if (n_sub > 0) {
                    run_src.assign((size_t)ntri, -1);
                    par_ranges((size_t)n_sub, [&](int, size_t lo, size_t hi) {
                        for (size_t j = lo; j < hi; j++)
                            for (int k = 0; k < 2; k++)
                                if (subs[j].e[k] < 0) run_src[(size_t)~subs[j].e[k]] = (int)(2 * j) + k;
                    });
                }
                par_ranges((size_t)nn, [&](int wi, size_t lo, size_t hi) {
                    size_t o = c_runs[(size_t)wi];
                    for (size_t i = lo; i < hi; i++) {
                        const rt_bvh_node& n = s->nodes[i];
                        if (n.child_a != -1 || n.tri_len == 0) continue;
                        const int first = n.tri_offs / RT_TRI_FLOATS, cnt = n.tri_len / RT_TRI_FLOATS;
                        const bool whole = !short_single && sub_root[i] < 0;  // the leaf as one unit
                        for (int k = 0; k < cnt;) {
                            int len = short_single ? 1 : cnt - k;
                            if (!whole && !short_single)
                                for (len = 1; !(tflags[(size_t)first + k + len - 1] & 1); len++) {}
                            const int src = (short_single || whole || run_src.empty()) ? -1 : run_src[(size_t)first + k];
                            set_prim(prims[o++], short_single ? nullptr : whole ? leaf_box[i] : (src >= 0 ? subs[(size_t)(src >> 1)].box[src & 1] : nullptr),
                                     first + k, len, 2 * (first + k), (float)len);
                            k += len;
                        }
                    }
                });
                below_leaf = 0;
            } else {
                // Too many runs for the build: the shortcut tree stands over a FRONTIER of the leaves' sub-trees instead -- what
                // is found `cut` levels below each reference leaf (sub-tree nodes there, runs met on the way); cut = 0 is the
                // leaves themselves.  The deepest cut whose frontier has at most RT_SHORT_FRONT_MAX entries is taken.
                const int D = RT_SHORT_CUT_MAX;
                typedef std::array<size_t, RT_SHORT_CUT_MAX + 1> Counts;
                auto count_below = [&](auto&& self, int entry, int k, Counts& c) -> void {
                    if (entry < 0) {
                        for (int d = k; d <= D; d++) c[(size_t)d]++;
                        return;
                    }
                    c[(size_t)k]++;
                    if (k < D) {
                        self(self, subs[(size_t)entry].e[0], k + 1, c);
                        self(self, subs[(size_t)entry].e[1], k + 1, c);
                    }
                };
                std::vector<Counts> c_front((size_t)n_workers_max + 1);
                for (auto& c : c_front) c.fill(0);
                par_ranges((size_t)nn, [&](int wi, size_t lo, size_t hi) {
                    Counts c;
                    c.fill(0);
                    for (size_t i = lo; i < hi; i++) {
                        const rt_bvh_node& n = s->nodes[i];
                        if (n.child_a != -1 || n.tri_len == 0) continue;
                        count_below(count_below, sub_root[i] >= 0 ? sub_root[i] : -1, 0, c);
                    }
                    c_front[(size_t)wi + 1] = c;
                });
                for (int wi = 0; wi < nr; wi++)
                    for (int d = 0; d <= D; d++) c_front[(size_t)wi + 1][(size_t)d] += c_front[(size_t)wi][(size_t)d];
                int cut = 0;
                for (int d = 1; d <= D; d++)
                    if (c_front[(size_t)nr][(size_t)d] <= (size_t)RT_SHORT_FRONT_MAX) cut = d;
                if (const char* e = getenv("RT_SHORT_CUT")) cut = std::min(D, std::max(0, atoi(e)));  // tests, experiments
                prims.resize(c_front[(size_t)nr][(size_t)cut]);
                par_ranges((size_t)nn, [&](int wi, size_t lo, size_t hi) {
                    size_t o = c_front[(size_t)wi][(size_t)cut];
                    auto emit = [&](auto&& self, int entry, int k, const float* box, int first, int cnt) -> void {
                        if (entry < 0 && k == 0) {  // a leaf without a sub-tree: its triangles as one run
                            set_prim(prims[o++], box, first, cnt, 2 * first, 1.0f + (float)std::min(cnt, 4));
                        } else if (entry < 0) {     // a sub-leaf met above the cut
                            int len = 1;
                            while (!(tflags[(size_t)~entry + (size_t)len - 1] & 1)) len++;
                            set_prim(prims[o++], box, ~entry, len, 2 * ~entry, (float)len);
                        } else if (k == cut) {      // a sub-tree node on the cut: the walk goes on below it
                            set_prim(prims[o++], box, 0, 0, 2 * entry + 1, k == 0 ? 1.0f + (float)std::min(cnt, 4) : 3.0f);
                        } else {
                            self(self, subs[(size_t)entry].e[0], k + 1, subs[(size_t)entry].box[0], first, cnt);
                            self(self, subs[(size_t)entry].e[1], k + 1, subs[(size_t)entry].box[1], first, cnt);
                        }
                    };
                    for (size_t i = lo; i < hi; i++) {
                        const rt_bvh_node& n = s->nodes[i];
                        if (n.child_a != -1 || n.tri_len == 0) continue;
                        emit(emit, sub_root[i] >= 0 ? sub_root[i] : -1, 0, leaf_box[i], n.tri_offs / RT_TRI_FLOATS, n.tri_len / RT_TRI_FLOATS);
                    }
                });
                below_leaf = std::max(0, below_leaf - cut);
                short_cut = cut;
            }
        }
        timer.stage("shortcut tree: leaves / runs and their boxes");
        const int np = (int)prims.size();
        int lg = 0;
        while ((1 << lg) < np) lg++;
        // levels the shortcut tree may have: the stack holds RT_SHORT_STACK_MAX entries and the sub-trees need below_leaf of them
        const int levels_max = std::min(RT_SHORT_STACK_MAX - below_leaf, lg + 8);
        if (np >= 2 && lg <= levels_max) {
            shorts.resize((size_t)np - 1);  // the node that splits [lo, hi) at mid is shorts[mid - 1]: no two nodes share a split point
            std::atomic<int> deepest_short{0};
            const int NB = 32;              // bins of the centroid range per axis
            int short_sweep_max = np <= (1 << 18) ? RT_SHORT_SWEEP_MAX : 64;
            if (const char* e = getenv("RT_SHORT_SWEEP_MAX")) short_sweep_max = std::max(2, atoi(e));  // experiments only
            struct Bins { float bb[3][32][6], bw[3][32]; int bn[3][32]; float ub[6], cmn[3], cmx[3]; };
            // A pass over prims[lo, hi) in fixed chunks of 2^13, the chunks' results folded in chunk order: the same result on
            // any number of threads (the float sums included); big nodes only, where the recursion has few branches yet
            // (from 2^15 leaves on: the 10 M-triangle field has 260 k, and its first four levels were 30 of the stage's 41 ms
            // on one, two, four and eight threads).
            auto chunked = [&](int lo, int hi, auto&& init, auto&& one, auto&& fold, auto& total) {
                const int CH = 1 << 13, n = hi - lo, nch = (n + CH - 1) / CH;
                if (n < (1 << 15)) {
                    init(total);
                    for (int p = lo; p < hi; p++) one(total, prims[(size_t)p]);
                    return;
                }
                std::vector<typename std::remove_reference<decltype(total)>::type> part((size_t)nch);
                std::atomic<int> next{0};
                auto run = [&] {
                    for (int c; (c = next.fetch_add(1)) < nch;) {
                        init(part[(size_t)c]);
                        for (int p = lo + c * CH; p < std::min(hi, lo + (c + 1) * CH); p++) one(part[(size_t)c], prims[(size_t)p]);
                    }
                };
                std::vector<std::thread> th;
                for (int wi = 1; wi < n_workers_max; wi++) th.emplace_back(run);
                run();
                for (auto& t : th) t.join();
                init(total);
                for (int c = 0; c < nch; c++) fold(total, part[(size_t)c]);
            };
            auto less_on = [](int axis) {  // by centroid, ties by what hangs there: on the floats' bits made monotonic, a total order whatever the values (NaN included)
                return [axis](const ShortPrim& a, const ShortPrim& b) {
                    auto key = [](float f) {
                        f += 0.0f;  // (-0 -> +0)
                        uint32_t u;
                        memcpy(&u, &f, 4);
                        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                    };
                    const uint32_t ka = key(a.c[axis]), kb = key(b.c[axis]);
                    return ka < kb || (ka == kb && a.node < b.node);
                };
            };
            // The full sweep below a node of at most short_sweep_max leaves, as the sub-tree builder does it (sweep_sub above):
            // the node's leaves are sorted ONCE per axis (centroid, ties by what hangs there: one 64-bit key) and a split keeps
            // the three orders by stable partition -- the same splits as sorting the 44-byte records at every node (the form
            // below, RT_SHORT_RESORT=1: tests), which was most of this stage for the 10 M-triangle field.
            struct ShortSweep {
                int base = 0;
                std::vector<float> lb, lcost, lc, suffix;  // per local leaf: box (mins, 0, maxs, 0), cost, centroid
                std::vector<int> lnode, ord[3], scratch;
                std::vector<unsigned char> side;
                std::vector<uint64_t> keys;
                std::vector<uint32_t> keys32;
            };
            auto ckey = [](float f) {  // the float's bits made monotonic, -0 folded into +0
                f += 0.0f;
                uint32_t u;
                memcpy(&u, &f, 4);
                return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            };
            auto sweep_short_begin = [&](ShortSweep& w, int lo, int hi) {
                const int n = hi - lo;
                w.base = lo;
                w.keys.resize((size_t)n);
                for (int k = 0; k < n; k++) w.keys[(size_t)k] = ((uint64_t)((uint32_t)prims[(size_t)(lo + k)].node ^ 0x80000000u) << 32) | (uint32_t)k;
                std::sort(w.keys.begin(), w.keys.end());  // local index order = order of what hangs there: the index then breaks ties like it
                w.lb.resize(8 * (size_t)n); w.lcost.resize((size_t)n); w.lc.resize(3 * (size_t)n); w.lnode.resize((size_t)n);
                w.side.resize((size_t)n); w.scratch.resize((size_t)n); w.suffix.resize(2 * ((size_t)n + 1));
                for (int k = 0; k < n; k++) {
                    const ShortPrim& p = prims[(size_t)(lo + (int)(uint32_t)w.keys[(size_t)k])];
                    float* b = &w.lb[8 * (size_t)k];
                    b[0] = p.b[0]; b[1] = p.b[1]; b[2] = p.b[2]; b[3] = 0.f;
                    b[4] = p.b[3]; b[5] = p.b[4]; b[6] = p.b[5]; b[7] = 0.f;
                    w.lcost[(size_t)k] = p.cost;
                    w.lnode[(size_t)k] = p.node;
                    for (int c = 0; c < 3; c++) w.lc[3 * (size_t)k + (size_t)c] = p.c[c];
                }
                w.keys32.resize((size_t)n);
                for (int axis = 0; axis < 3; axis++) {
                    w.ord[axis].resize((size_t)n);
                    if (n >= 48) {
                        for (int k = 0; k < n; k++) w.keys32[(size_t)k] = ckey(w.lc[3 * (size_t)k + (size_t)axis]);
                        order_by_key32(w.keys32.data(), n, w.ord[axis].data(), w.scratch.data());
                        continue;
                    }
                    for (int k = 0; k < n; k++) w.keys[(size_t)k] = ((uint64_t)ckey(w.lc[3 * (size_t)k + (size_t)axis]) << 32) | (uint32_t)k;
                    std::sort(w.keys.begin(), w.keys.end());
                    for (int k = 0; k < n; k++) w.ord[axis][(size_t)k] = (int)(uint32_t)w.keys[(size_t)k];
                }
            };
            auto sweep_short = [&](auto&& self, ShortSweep& w, int lo, int hi, int d, float* box) -> int {
                const int n = hi - lo, o = lo - w.base;
                int seen = deepest_short.load();
                while (seen < d && !deepest_short.compare_exchange_weak(seen, d)) {}
                if (n == 1) {
                    const float* b = &w.lb[8 * (size_t)w.ord[0][(size_t)o]];
                    for (int k = 0; k < 3; k++) { box[k] = b[k] - box_pad; box[3 + k] = b[4 + k] + box_pad; }
                    return ~w.lnode[(size_t)w.ord[0][(size_t)o]];
                }
                const int room = levels_max - d - 1;
                const int side_max = room >= 30 ? n : std::min(n, 1 << room);
                const int k_lo = std::max(1, n - side_max), k_hi = std::min(n - 1, side_max);
                const __m128 empty_lo = _mm_setr_ps(3e38f, 3e38f, 3e38f, 0.f), empty_hi = _mm_setr_ps(-3e38f, -3e38f, -3e38f, 0.f);
                auto half_area4 = [](__m128 mn, __m128 mx) {  // x * y + y * z + z * x of the extents, summed left to right
                    const __m128 e = _mm_sub_ps(mx, mn);
                    const __m128 p = _mm_mul_ps(e, _mm_shuffle_ps(e, e, _MM_SHUFFLE(3, 0, 2, 1)));
                    return _mm_cvtss_f32(_mm_add_ss(_mm_add_ss(p, _mm_shuffle_ps(p, p, 1)), _mm_shuffle_ps(p, p, 2)));
                };
                const float* lb = w.lb.data();
                __m128 tlo = empty_lo, thi = empty_hi;
                for (int k = 0; k < n; k++) {
                    const float* b = lb + 8 * (size_t)w.ord[0][(size_t)(o + k)];
                    tlo = _mm_min_ps(_mm_loadu_ps(b), tlo);
                    thi = _mm_max_ps(_mm_loadu_ps(b + 4), thi);
                }
                float t_lo[4], t_hi[4];
                _mm_storeu_ps(t_lo, tlo);
                _mm_storeu_ps(t_hi, thi);
                for (int k = 0; k < 3; k++) { box[k] = t_lo[k] - box_pad; box[3 + k] = t_hi[k] + box_pad; }
                int best_axis = -1, best_k = -1;
                float best = 3e38f;
                for (int axis = 0; axis < 3; axis++) {
                    const int* so = &w.ord[axis][(size_t)o];
                    __m128 blo = empty_lo, bhi = empty_hi;
                    float cw = 0.f;
                    for (int k = n - 1; k >= 1; k--) {
                        const float* b = lb + 8 * (size_t)so[k];
                        blo = _mm_min_ps(_mm_loadu_ps(b), blo);
                        bhi = _mm_max_ps(_mm_loadu_ps(b + 4), bhi);
                        cw += w.lcost[(size_t)so[k]];
                        w.suffix[2 * (size_t)k] = half_area4(blo, bhi);
                        w.suffix[2 * (size_t)k + 1] = cw;
                    }
                    blo = empty_lo;
                    bhi = empty_hi;
                    cw = 0.f;
                    for (int k = 1; k < n; k++) {
                        const float* b = lb + 8 * (size_t)so[k - 1];
                        blo = _mm_min_ps(_mm_loadu_ps(b), blo);
                        bhi = _mm_max_ps(_mm_loadu_ps(b + 4), bhi);
                        cw += w.lcost[(size_t)so[k - 1]];
                        if (k < k_lo || k > k_hi) continue;
                        const float cst = half_area4(blo, bhi) * cw + w.suffix[2 * (size_t)k] * w.suffix[2 * (size_t)k + 1];
                        if (cst < best) { best = cst; best_axis = axis; best_k = k; }
                    }
                }
                if (best_axis < 0) {  // no admissible surface-area split: halves along the widest axis of the centroids
                    float ext[3];
                    for (int a = 0; a < 3; a++)
                        ext[a] = w.lc[3 * (size_t)w.ord[a][(size_t)(o + n - 1)] + (size_t)a] - w.lc[3 * (size_t)w.ord[a][(size_t)o] + (size_t)a];
                    best_axis = 0;
                    for (int a = 1; a < 3; a++)
                        if (ext[a] > ext[best_axis]) best_axis = a;
                    best_k = n / 2;
                }
                {
                    const int* so = &w.ord[best_axis][(size_t)o];
                    for (int k = 0; k < n; k++) w.side[(size_t)so[k]] = (unsigned char)(k < best_k ? 1 : 0);
                    for (int axis = 0; axis < 3; axis++) {
                        if (axis == best_axis) continue;
                        int* oa = &w.ord[axis][(size_t)o];
                        int nl = 0, nr = 0;
                        for (int k = 0; k < n; k++) {
                            const int t = oa[k];
                            if (w.side[(size_t)t]) oa[nl++] = t;
                            else w.scratch[(size_t)nr++] = t;
                        }
                        memcpy(oa + nl, w.scratch.data(), (size_t)nr * sizeof(int));
                    }
                }
                const int mid = lo + best_k;
                float ba[6], bb2[6];
                const int ea = self(self, w, lo, mid, d + 1, ba), eb = self(self, w, mid, hi, d + 1, bb2);
                ShortNode& nd = shorts[(size_t)mid - 1];
                memcpy(nd.box[0], ba, sizeof(ba));
                memcpy(nd.box[1], bb2, sizeof(bb2));
                nd.e[0] = ea;
                nd.e[1] = eb;
                return mid - 1;
            };
            const bool short_resort = getenv("RT_SHORT_RESORT") != nullptr;
            // returns the entry of the node over prims[lo, hi) standing at level d (root: 0) and its padded box
            auto build_short = [&](auto&& self, int lo, int hi, int d, float* box) -> int {
                const int n = hi - lo;
                int seen = deepest_short.load();
                while (seen < d && !deepest_short.compare_exchange_weak(seen, d)) {}
                if (n >= 2 && n <= short_sweep_max && !short_resort) {
                    ShortSweep w;
                    sweep_short_begin(w, lo, hi);
                    return sweep_short(sweep_short, w, lo, hi, d, box);
                }
                if (n == 1) {
                    memcpy(box, prims[(size_t)lo].b, 6 * sizeof(float));
                    for (int k = 0; k < 3; k++) { box[k] -= box_pad; box[3 + k] += box_pad; }
                    return ~prims[(size_t)lo].node;
                }
                // a side may keep at most 2^(levels_max - d - 1) leaves: what is left of the levels can still hold them
                const int room = levels_max - d - 1;
                const int side_max = room >= 30 ? n : std::min(n, 1 << room);
                const int k_lo = std::max(1, n - side_max), k_hi = std::min(n - 1, side_max);  // left = the first k, k in [k_lo, k_hi]
                int best_axis = -1, best_k = -1;
                float best = 3e38f;
                float cmn[3], cmx[3];
                if (n <= short_sweep_max) {  // every split position along every axis
                    std::vector<float> suffix(2 * ((size_t)n + 1));
                    memcpy(box, kEmpty, 6 * sizeof(float));
                    for (int k = 0; k < 3; k++) { cmn[k] = 3e38f; cmx[k] = -3e38f; }
                    for (int p = lo; p < hi; p++) {
                        const ShortPrim& pr = prims[(size_t)p];
                        for (int c = 0; c < 3; c++) {
                            box[c] = std::min(box[c], pr.b[c]); box[3 + c] = std::max(box[3 + c], pr.b[3 + c]);
                            cmn[c] = std::min(cmn[c], pr.c[c]); cmx[c] = std::max(cmx[c], pr.c[c]);
                        }
                    }
                    for (int axis = 0; axis < 3; axis++) {
                        std::sort(prims.begin() + lo, prims.begin() + hi, less_on(axis));
                        float bx[6], w = 0.f;
                        memcpy(bx, kEmpty, sizeof(bx));
                        for (int k = n - 1; k >= 1; k--) {
                            const ShortPrim& p = prims[(size_t)(lo + k)];
                            for (int c = 0; c < 3; c++) { bx[c] = std::min(bx[c], p.b[c]); bx[3 + c] = std::max(bx[3 + c], p.b[3 + c]); }
                            w += p.cost;
                            suffix[2 * k] = half_area(bx);
                            suffix[2 * k + 1] = w;
                        }
                        memcpy(bx, kEmpty, sizeof(bx));
                        w = 0.f;
                        for (int k = 1; k < n; k++) {
                            const ShortPrim& p = prims[(size_t)(lo + k - 1)];
                            for (int c = 0; c < 3; c++) { bx[c] = std::min(bx[c], p.b[c]); bx[3 + c] = std::max(bx[3 + c], p.b[3 + c]); }
                            w += p.cost;
                            if (k < k_lo || k > k_hi) continue;
                            const float cst = half_area(bx) * w + suffix[2 * k] * suffix[2 * k + 1];
                            if (cst < best) { best = cst; best_axis = axis; best_k = k; }
                        }
                    }
                    if (best_axis >= 0 && best_axis != 2) std::sort(prims.begin() + lo, prims.begin() + hi, less_on(best_axis));
                } else {  // 32 bins of the centroid range per axis; the leaves are then partitioned, not sorted
                    struct Ext { float ub[6], cmn[3], cmx[3]; } ext;
                    chunked(lo, hi,
                            [&](Ext& e) { memcpy(e.ub, kEmpty, sizeof(e.ub)); for (int k = 0; k < 3; k++) { e.cmn[k] = 3e38f; e.cmx[k] = -3e38f; } },
                            [&](Ext& e, const ShortPrim& pr) {
                                for (int c = 0; c < 3; c++) {
                                    e.ub[c] = std::min(e.ub[c], pr.b[c]); e.ub[3 + c] = std::max(e.ub[3 + c], pr.b[3 + c]);
                                    e.cmn[c] = std::min(e.cmn[c], pr.c[c]); e.cmx[c] = std::max(e.cmx[c], pr.c[c]);
                                }
                            },
                            [&](Ext& e, const Ext& q) {
                                for (int c = 0; c < 3; c++) {
                                    e.ub[c] = std::min(e.ub[c], q.ub[c]); e.ub[3 + c] = std::max(e.ub[3 + c], q.ub[3 + c]);
                                    e.cmn[c] = std::min(e.cmn[c], q.cmn[c]); e.cmx[c] = std::max(e.cmx[c], q.cmx[c]);
                                }
                            },
                            ext);
                    memcpy(box, ext.ub, sizeof(ext.ub));
                    memcpy(cmn, ext.cmn, sizeof(cmn));
                    memcpy(cmx, ext.cmx, sizeof(cmx));
                    float scale[3];
                    for (int k = 0; k < 3; k++) scale[k] = cmx[k] > cmn[k] ? (float)NB / (cmx[k] - cmn[k]) : 0.f;
                    auto bin_of = [&](const ShortPrim& p, int axis) {
                        return std::min(NB - 1, std::max(0, (int)((p.c[axis] - cmn[axis]) * scale[axis])));
                    };
                    std::unique_ptr<Bins> bins(new Bins);
                    chunked(lo, hi,
                            [&](Bins& B) {
                                for (int a = 0; a < 3; a++)
                                    for (int b = 0; b < NB; b++) { memcpy(B.bb[a][b], kEmpty, 6 * sizeof(float)); B.bw[a][b] = 0.f; B.bn[a][b] = 0; }
                            },
                            [&](Bins& B, const ShortPrim& pr) {
                                for (int a = 0; a < 3; a++) {
                                    const int b = bin_of(pr, a);
                                    float* bx = B.bb[a][b];
                                    for (int c = 0; c < 3; c++) { bx[c] = std::min(bx[c], pr.b[c]); bx[3 + c] = std::max(bx[3 + c], pr.b[3 + c]); }
                                    B.bw[a][b] += pr.cost;
                                    B.bn[a][b]++;
                                }
                            },
                            [&](Bins& B, const Bins& Q) {
                                for (int a = 0; a < 3; a++)
                                    for (int b = 0; b < NB; b++) {
                                        for (int c = 0; c < 3; c++) {
                                            B.bb[a][b][c] = std::min(B.bb[a][b][c], Q.bb[a][b][c]);
                                            B.bb[a][b][3 + c] = std::max(B.bb[a][b][3 + c], Q.bb[a][b][3 + c]);
                                        }
                                        B.bw[a][b] += Q.bw[a][b];
                                        B.bn[a][b] += Q.bn[a][b];
                                    }
                            },
                            *bins);
                    int best_bin = -1;
                    for (int axis = 0; axis < 3; axis++) {
                        if (!(cmx[axis] > cmn[axis])) continue;
                        const auto& bb = bins->bb[axis];
                        const float* bw = bins->bw[axis];
                        const int* bn = bins->bn[axis];
                        float ra[32], rw[32], bx[6], w = 0.f;
                        memcpy(bx, kEmpty, sizeof(bx));
                        for (int b = NB - 1; b >= 1; b--) {
                            for (int c = 0; c < 3; c++) { bx[c] = std::min(bx[c], bb[b][c]); bx[3 + c] = std::max(bx[3 + c], bb[b][3 + c]); }
                            w += bw[b];
                            ra[b] = half_area(bx);
                            rw[b] = w;
                        }
                        memcpy(bx, kEmpty, sizeof(bx));
                        w = 0.f;
                        int k = 0;
                        for (int b = 1; b < NB; b++) {  // left = bins [0, b)
                            for (int c = 0; c < 3; c++) { bx[c] = std::min(bx[c], bb[b - 1][c]); bx[3 + c] = std::max(bx[3 + c], bb[b - 1][3 + c]); }
                            w += bw[b - 1];
                            k += bn[b - 1];
                            if (k < k_lo || k > k_hi) continue;
                            const float cst = half_area(bx) * w + ra[b] * rw[b];
                            if (cst < best) { best = cst; best_axis = axis; best_k = k; best_bin = b; }
                        }
                    }
                    if (best_axis >= 0) {
                        const int axis = best_axis, bsplit = best_bin;
                        std::partition(prims.begin() + lo, prims.begin() + hi, [&](const ShortPrim& a) { return bin_of(a, axis) < bsplit; });
                    }
                }
                for (int k = 0; k < 3; k++) { box[k] -= box_pad; box[3 + k] += box_pad; }
                if (best_axis < 0) {  // no admissible surface-area split (or every centroid in one place): halves along the widest axis
                    int axis = 0;
                    for (int k = 1; k < 3; k++)
                        if (cmx[k] - cmn[k] > cmx[axis] - cmn[axis]) axis = k;
                    best_k = n / 2;
                    std::nth_element(prims.begin() + lo, prims.begin() + lo + best_k, prims.begin() + hi, less_on(axis));
                }
                const int mid = lo + best_k;
                float ba[6], bb2[6];
                int ea, eb;
                if (n >= (1 << 14) && d < 4) {  // the big halves on two host threads (disjoint ranges of prims and of shorts)
                    int ea_t = 0;
                    std::thread th([&] { ea_t = self(self, lo, mid, d + 1, ba); });
                    eb = self(self, mid, hi, d + 1, bb2);
                    th.join();
                    ea = ea_t;
                } else {
                    ea = self(self, lo, mid, d + 1, ba);
                    eb = self(self, mid, hi, d + 1, bb2);
                }
                ShortNode& nd = shorts[(size_t)mid - 1];
                memcpy(nd.box[0], ba, sizeof(ba));
                memcpy(nd.box[1], bb2, sizeof(bb2));
                nd.e[0] = ea;
                nd.e[1] = eb;
                return mid - 1;
            };
            float root_box[6];
            short_root = build_short(build_short, 0, np, 0, root_box);
            short_levels = deepest_short.load();
        }
    }
    // ---- WIDE SHORTCUT NODES (rt_device.h): the binary tree collapsed into nodes of up to four children -- the child with
    //      the biggest box is replaced by its own two children until there are four (or only leaves are left).  The
    //      binary nodes are then not emitted.  RT_SHORT_BINARY keeps them (experiments, tests of the two-child form).
    struct WideNode {  // e >= 0: wide node; e < 0: ~(ShortPrim::node of what hangs there); slots >= nk unused
        float box[4][6]; int e[4]; int nk;
        WideNode() {}  // (no zero-fill when a vector of millions of them is sized: whoever sizes it writes every element)
    };
    BigVec<WideNode> wides;
    int wide_root = -1, wide_need = 0;  // (stack entries a search of the wide tree can need: sum of (children - 1) down the worst path)
    int n_wide_top = 0;                 // wide nodes of the shortcut tree proper (the others: wide forms of sub-tree parts, huge scenes)
    timer.stage("shortcut tree: two-child build");
    if (short_root >= 0 && !getenv("RT_SHORT_BINARY")) {
        wides.reserve(shorts.size() / 2 + 1);
        auto collapse = [&](auto&& self, int bnode, int& need) -> int {
            const float* kbox[4];
            int ke[4], nk = 2;
            kbox[0] = shorts[(size_t)bnode].box[0]; ke[0] = shorts[(size_t)bnode].e[0];
            kbox[1] = shorts[(size_t)bnode].box[1]; ke[1] = shorts[(size_t)bnode].e[1];
            while (nk < 4) {
                int best = -1;
                float ba = -1.f;
                for (int i = 0; i < nk; i++)
                    if (ke[i] >= 0 && half_area(kbox[i]) > ba) { ba = half_area(kbox[i]); best = i; }
                if (best < 0) break;
                const ShortNode& c = shorts[(size_t)ke[best]];
                kbox[best] = c.box[0]; ke[best] = c.e[0];
                kbox[nk] = c.box[1]; ke[nk] = c.e[1];
                nk++;
            }
            const int idx = (int)wides.size();
            wides.push_back(WideNode());
            WideNode nd;
            nd.nk = nk;
            int below = 0;
            for (int i = 0; i < 4; i++) {
                if (i < nk) {
                    memcpy(nd.box[i], kbox[i], 6 * sizeof(float));
                    int nb = 0;
                    nd.e[i] = ke[i] >= 0 ? self(self, ke[i], nb) : ke[i];
                    below = std::max(below, nb);
                } else {
                    for (int k = 0; k < 6; k++) nd.box[i][k] = 3e38f;  // min = max far outside: no ray reaches it (rt_device.h)
                    nd.e[i] = 0;
                }
            }
            wides[(size_t)idx] = nd;
            need = nk - 1 + below;
            return idx;
        };
        wide_root = collapse(collapse, short_root, wide_need);
        shorts.clear();
        n_wide_top = (int)wides.size();
        timer.stage("shortcut tree: collapse to wide nodes");
        if (!short_runs && !getenv("RT_WIDE_TOP_ONLY")) {
            // The shortcut tree stands over a frontier of the leaves' sub-trees (huge scenes): what hangs below the frontier is
            // collapsed into wide nodes as well -- a second, four-child form of those parts of the sub-trees, for the shortcut walk
            // alone (the reference's walk keeps the two-child nodes) --, so that a shortcut search meets ONE kind of inner node.
            struct Task { int w, i, sub; };
            std::vector<Task> tasks;
            for (int w = 0; w < n_wide_top; w++)
                for (int i = 0; i < wides[(size_t)w].nk; i++) {
                    const int e = wides[(size_t)w].e[i];
                    if (e < 0 && (~e & 1)) tasks.push_back({w, i, ~e >> 1});
                }
            auto collapse_sub = [&](auto&& self, auto& out, int bnode, int& need) -> int {
                const float* kbox[4];
                int ke[4], nk = 2;
                kbox[0] = subs[(size_t)bnode].box[0]; ke[0] = subs[(size_t)bnode].e[0];
                kbox[1] = subs[(size_t)bnode].box[1]; ke[1] = subs[(size_t)bnode].e[1];
                while (nk < 4) {
                    int best = -1;
                    float ba = -1.f;
                    for (int i = 0; i < nk; i++)
                        if (ke[i] >= 0 && half_area(kbox[i]) > ba) { ba = half_area(kbox[i]); best = i; }
                    if (best < 0) break;
                    const SubNode& c = subs[(size_t)ke[best]];
                    kbox[best] = c.box[0]; ke[best] = c.e[0];
                    kbox[nk] = c.box[1]; ke[nk] = c.e[1];
                    nk++;
                }
                const int idx = (int)out.size();
                out.push_back(WideNode());
                WideNode nd;
                nd.nk = nk;
                int below = 0;
                for (int i = 0; i < 4; i++) {
                    if (i < nk) {
                        memcpy(nd.box[i], kbox[i], 6 * sizeof(float));
                        int nb = 0;
                        nd.e[i] = ke[i] >= 0 ? self(self, out, ke[i], nb) : ~(2 * ~ke[i]);  // (a sub-leaf: the run of triangles at place ~ke)
                        below = std::max(below, nb);
                    } else {
                        for (int k = 0; k < 6; k++) nd.box[i][k] = 3e38f;
                        nd.e[i] = 0;
                    }
                }
                out[(size_t)idx] = nd;
                need = nk - 1 + below;
                return idx;
            };
            // (in contiguous parts of the task list, several per thread, taken as threads come free; merged in part order: the
            //  same numbering as one thread's)
            const int nw = tasks.size() >= 4096 ? 128 : 1;
            std::vector<BigVec<WideNode>> loc((size_t)nw);
            std::vector<int> task_root(tasks.size(), 0), need_max((size_t)nw, 0), task_worker(tasks.size(), 0);
            {
                std::atomic<int> next{0};
                auto body = [&] {
                    for (int wi; (wi = next.fetch_add(1)) < nw;) {
                        const size_t lo = tasks.size() * (size_t)wi / (size_t)nw, hi = tasks.size() * (size_t)(wi + 1) / (size_t)nw;
                        // (two-child nodes collapse to about 0.45 wide nodes each; the tasks are spread evenly: no regrowth in the common case)
                        loc[(size_t)wi].reserve((size_t)((double)subs.size() * 0.5 * (double)(hi - lo) / (double)std::max<size_t>(tasks.size(), 1)) + 64);
                        for (size_t k = lo; k < hi; k++) {
                            int need = 0;
                            task_root[k] = collapse_sub(collapse_sub, loc[(size_t)wi], tasks[k].sub, need);
                            task_worker[k] = wi;
                            need_max[(size_t)wi] = std::max(need_max[(size_t)wi], need);
                        }
                    }
                };
                std::vector<std::thread> th;
                for (int t = 1; t < (nw > 1 ? n_workers_max : 1); t++) th.emplace_back(body);
                body();
                for (auto& t : th) t.join();
            }
            timer.stage("shortcut tree: wide forms of the sub-tree parts (threads)");
            std::vector<size_t> off((size_t)nw + 1, (size_t)n_wide_top);
            for (int wi = 0; wi < nw; wi++) off[(size_t)wi + 1] = off[(size_t)wi] + loc[(size_t)wi].size();
            wides.resize(off[(size_t)nw]);
            {
                std::atomic<int> next{0};
                auto copy = [&] {
                    for (int wi; (wi = next.fetch_add(1)) < nw;) {
                        const int o = (int)off[(size_t)wi];
                        for (size_t k = 0; k < loc[(size_t)wi].size(); k++) {
                            WideNode nd = loc[(size_t)wi][k];
                            for (int i = 0; i < nd.nk; i++)
                                if (nd.e[i] >= 0) nd.e[i] += o;
                            wides[(size_t)o + k] = nd;
                        }
                    }
                };
                std::vector<std::thread> th;
                for (int t = 1; t < (nw > 1 ? n_workers_max : 1); t++) th.emplace_back(copy);
                copy();
                for (auto& t : th) t.join();
            }
            int sub_need = 0;
            for (size_t k = 0; k < tasks.size(); k++) wides[(size_t)tasks[k].w].e[tasks[k].i] = (int)off[(size_t)task_worker[k]] + task_root[k];
            for (int wi = 0; wi < nw; wi++) sub_need = std::max(sub_need, need_max[(size_t)wi]);
            timer.stage("shortcut tree: wide forms merged");
            wide_need += sub_need;   // (an upper bound: the deepest frontier part below the deepest path of the shortcut tree)
            below_leaf = 0;          // a shortcut search no longer enters the two-child sub-trees
        }
    }
    // ---- LIGHT TREE (rt_device.h): with enough importance-sampled triangles for it to pay, a tree of wide nodes over the light
    //      triangles (median splits of the centroids, padded boxes) whose leaves are single LIGHT ITEMS behind the triangles of
    //      the item stream.  The fused kernel runs reflectType5's sum over all lights (material.h:315-323) as an ALL-HITS search
    //      of this tree with the lane's ordinary traversal steps, instead of testing the lights one by one.
    int light_root = -1, n_light_items = 0, light_need = 0;
    if (n_wide_top > 0 && nl >= RT_LIGHTS_COOP_MIN && !getenv("RT_NO_LIGHT_TREE")) {
        std::vector<std::array<float, 9>> lb((size_t)nl);  // per stored light j: box (6), centroid (3)
        for (int j = 0; j < nl; j++) {
            const float* t = tris + s->light_tris[light_order[(size_t)j]];
            for (int k = 0; k < 3; k++) {
                lb[(size_t)j][(size_t)k] = std::min(t[k], std::min(t[3 + k], t[6 + k]));
                lb[(size_t)j][(size_t)(3 + k)] = std::max(t[k], std::max(t[3 + k], t[6 + k]));
                lb[(size_t)j][(size_t)(6 + k)] = (t[k] + t[3 + k] + t[6 + k]) * (1.0f / 3.0f);
            }
        }
        std::vector<int> lid((size_t)nl);
        for (int j = 0; j < nl; j++) lid[(size_t)j] = j;
        std::vector<ShortNode> lnodes;
        lnodes.reserve((size_t)nl);
        const bool light_sah = !getenv("RT_LIGHT_TREE_MEDIAN");  // (experiments: median splits of the centroids along the widest axis)
        // returns the entry over lid[lo, hi) (>= 0: node in lnodes; < 0: ~(2 * place of the light's item)) and its padded box
        auto build_light = [&](auto&& self, int lo, int hi, float* box) -> int {
            float u[6] = {3e38f, 3e38f, 3e38f, -3e38f, -3e38f, -3e38f}, cmn[3] = {3e38f, 3e38f, 3e38f}, cmx[3] = {-3e38f, -3e38f, -3e38f};
            for (int p = lo; p < hi; p++) {
                const auto& b = lb[(size_t)lid[(size_t)p]];
                for (int c = 0; c < 3; c++) {
                    u[c] = std::min(u[c], b[(size_t)c]); u[3 + c] = std::max(u[3 + c], b[(size_t)(3 + c)]);
                    cmn[c] = std::min(cmn[c], b[(size_t)(6 + c)]); cmx[c] = std::max(cmx[c], b[(size_t)(6 + c)]);
                }
            }
            for (int c = 0; c < 3; c++) { box[c] = u[c] - box_pad; box[3 + c] = u[3 + c] + box_pad; }
            if (hi - lo == 1) return ~(2 * (ntri + 1 + lid[(size_t)lo]));
            int axis = 0;
            for (int c = 1; c < 3; c++)
                if (cmx[c] - cmn[c] > cmx[axis] - cmn[axis]) axis = c;
            int mid = lo + (hi - lo) / 2;
            const int n = hi - lo;
            if (light_sah && n > 2) {
                // surface-area split: an all-hits query visits a node whenever the ray meets its box, i.e. in proportion to the box's
                // area, and nothing is ever pruned by distance -- the sum of the children's areas weighted by what hangs below them is
                // exactly what a query pays.  Full sweep over the three axes (a few hundred lights: nothing).
                float best = 3e38f;
                int best_axis = -1, best_k = -1;
                std::vector<int> ord((size_t)n);
                std::vector<float> suf((size_t)n + 1);
                for (int c = 0; c < 3; c++) {
                    for (int k = 0; k < n; k++) ord[(size_t)k] = lid[(size_t)(lo + k)];
                    std::sort(ord.begin(), ord.end(), [&](int a, int b) {
                        const float ca = lb[(size_t)a][(size_t)(6 + c)], cb = lb[(size_t)b][(size_t)(6 + c)];
                        return ca < cb || (ca == cb && a < b);
                    });
                    float bx[6] = {3e38f, 3e38f, 3e38f, -3e38f, -3e38f, -3e38f};
                    for (int k = n - 1; k >= 1; k--) {
                        const auto& b = lb[(size_t)ord[(size_t)k]];
                        for (int a = 0; a < 3; a++) { bx[a] = std::min(bx[a], b[(size_t)a]); bx[3 + a] = std::max(bx[3 + a], b[(size_t)(3 + a)]); }
                        suf[(size_t)k] = half_area(bx);
                    }
                    float bl[6] = {3e38f, 3e38f, 3e38f, -3e38f, -3e38f, -3e38f};
                    for (int k = 1; k < n; k++) {
                        const auto& b = lb[(size_t)ord[(size_t)k - 1]];
                        for (int a = 0; a < 3; a++) { bl[a] = std::min(bl[a], b[(size_t)a]); bl[3 + a] = std::max(bl[3 + a], b[(size_t)(3 + a)]); }
                        const float cst = half_area(bl) * (float)k + suf[(size_t)k] * (float)(n - k);
                        if (cst < best) { best = cst; best_axis = c; best_k = k; }
                    }
                }
                if (best_axis >= 0) { axis = best_axis; mid = lo + best_k; }
            }
            std::nth_element(lid.begin() + lo, lid.begin() + mid, lid.begin() + hi, [&](int a, int b) {
                const float ca = lb[(size_t)a][(size_t)(6 + axis)], cb = lb[(size_t)b][(size_t)(6 + axis)];
                return ca < cb || (ca == cb && a < b);
            });
            const int idx = (int)lnodes.size();
            lnodes.push_back(ShortNode());
            float ba[6], bb2[6];
            const int ea = self(self, lo, mid, ba), eb = self(self, mid, hi, bb2);
            ShortNode& nd = lnodes[(size_t)idx];
            memcpy(nd.box[0], ba, sizeof(ba));
            memcpy(nd.box[1], bb2, sizeof(bb2));
            nd.e[0] = ea;
            nd.e[1] = eb;
            return idx;
        };
        float root_box[6];
        const int lroot = build_light(build_light, 0, nl, root_box);  // (nl >= 32: an inner node)
        auto collapse_light = [&](auto&& self, int bnode, int& need) -> int {
            const float* kbox[4];
            int ke[4], nk = 2;
            kbox[0] = lnodes[(size_t)bnode].box[0]; ke[0] = lnodes[(size_t)bnode].e[0];
            kbox[1] = lnodes[(size_t)bnode].box[1]; ke[1] = lnodes[(size_t)bnode].e[1];
            while (nk < 4) {
                int best = -1;
                float ba = -1.f;
                for (int i = 0; i < nk; i++)
                    if (ke[i] >= 0 && half_area(kbox[i]) > ba) { ba = half_area(kbox[i]); best = i; }
                if (best < 0) break;
                const ShortNode& c = lnodes[(size_t)ke[best]];
                kbox[best] = c.box[0]; ke[best] = c.e[0];
                kbox[nk] = c.box[1]; ke[nk] = c.e[1];
                nk++;
            }
            const int idx = (int)wides.size();
            wides.push_back(WideNode());
            WideNode nd;
            nd.nk = nk;
            int below = 0;
            for (int i = 0; i < 4; i++) {
                if (i < nk) {
                    memcpy(nd.box[i], kbox[i], 6 * sizeof(float));
                    int nb = 0;
                    nd.e[i] = ke[i] >= 0 ? self(self, ke[i], nb) : ke[i];
                    below = std::max(below, nb);
                } else {
                    for (int k = 0; k < 6; k++) nd.box[i][k] = 3e38f;
                    nd.e[i] = 0;
                }
            }
            wides[(size_t)idx] = nd;
            need = nk - 1 + below;
            return idx;
        };
        light_root = collapse_light(collapse_light, lroot, light_need);
        n_light_items = nl;
    }
    const int n_wide = (int)wides.size();
    const int n_short = (int)shorts.size();
    timer.stage("shortcut tree");
    n_inner = n_top + n_sub + n_short + 2 * n_wide;
    // Item index of every inner node, the reference's and the sub-trees' alike.  The RT_ITEM_FRONT nodes with the biggest
    // boxes come first, biggest first (ties: the reference's nodes in their order, then the sub-tree nodes in theirs): a
    // random ray meets a box in proportion to its area, and the LDS node cache of the kernels holds the items from the
    // front.  Behind them the reference's BFS order and the sub-trees' depth-first order stay as they are, which keeps
    // neighbours in the tree close in memory.  "Node q" below: q < nn is the reference's node q, q >= nn sub-tree node q - nn.
    BigVec<int> sub_index((size_t)std::max(n_sub, 1), 0), short_index((size_t)std::max(n_short, 1), 0), wide_index;
    int short_base = 0;
    {
        struct Key { float area; int kind, id; };
        auto bigger = [](const Key& a, const Key& b) { return a.area != b.area ? a.area > b.area : (a.kind != b.kind ? a.kind < b.kind : a.id < b.id); };
        // "node q": q < nn the reference's node q, then the sub-tree nodes, then the shortcut nodes
        const size_t nq = (size_t)nn + (size_t)n_sub + (size_t)n_short;
        auto union_area = [&](const float* A, const float* B) {
            float u[6];
            for (int c = 0; c < 3; c++) {
                u[c] = std::min(A[c], B[c]);
                u[3 + c] = std::max(A[3 + c], B[3 + c]);
            }
            return half_area(u);
        };
        auto key_of = [&](size_t q, Key& k) -> bool {  // false: a leaf of the reference's tree (no item of its own)
            if (q < (size_t)nn) {
                const rt_bvh_node& nd = s->nodes[q];
                if (nd.child_a == -1) return false;
                const float b[6] = {nd.bbox_min[0], nd.bbox_min[1], nd.bbox_min[2], nd.bbox_max[0], nd.bbox_max[1], nd.bbox_max[2]};
                k = {q == 0 ? 3e38f : half_area(b), 0, (int)q};  // (the root first, whatever its box)
                return true;
            }
            if (q < (size_t)nn + (size_t)n_sub) {
                const SubNode& nd = subs[q - (size_t)nn];
                k = {union_area(nd.box[0], nd.box[1]), 1, (int)(q - (size_t)nn)};
                return true;
            }
            const int j = (int)(q - (size_t)nn - (size_t)n_sub);
            const ShortNode& nd = shorts[(size_t)j];
            k = {j == short_root ? 3e38f : union_area(nd.box[0], nd.box[1]), 2, j};
            return true;
        };
        const size_t front = getenv("RT_ITEMS_BFS") ? 0 : std::min<size_t>(RT_ITEM_FRONT, (size_t)n_inner);  // (RT_ITEMS_BFS: experiments only)
        // the `count` biggest nodes of the kinds in `kinds` (bit per kind), biggest first: every range keeps its own
        // `count` biggest; the biggest of those are the biggest of all (the order is total)
        auto select = [&](int kinds, size_t count) {
            std::vector<std::vector<Key>> top((size_t)n_workers_max);
            if (count)
                par_ranges(nq, [&](int wi, size_t lo, size_t hi) {
                    std::vector<Key>& v = top[(size_t)wi];
                    Key k;
                    for (size_t q = lo; q < hi; q++) {
                        if (!key_of(q, k) || !((kinds >> k.kind) & 1)) continue;
                        v.push_back(k);
                        if (v.size() >= 4 * count) {
                            std::nth_element(v.begin(), v.begin() + (std::ptrdiff_t)count, v.end(), bigger);
                            v.resize(count);
                        }
                    }
                });
            std::vector<Key> big;
            for (auto& v : top) big.insert(big.end(), v.begin(), v.end());
            if (big.size() > count) {
                std::nth_element(big.begin(), big.begin() + (std::ptrdiff_t)count, big.end(), bigger);
                big.resize(count);
            }
            std::sort(big.begin(), big.end(), bigger);
            return big;
        };
        // Without a shortcut tree: reference and sub-tree nodes in ONE order by area (what the reference's walk meets most).
        // With one the kernels cache one of two windows of the item stream: the reference's walk items [0, n_cached) --
        // RT_REF_FRONT of its own biggest nodes, then the window below --, the shortcut walk items [short_base, short_base +
        // n_cached): the biggest shortcut nodes (and sub-tree nodes, where it enters sub-trees), which is all it visits.
        std::vector<Key> big;
        size_t rest_base = 0;  // first item index of the nodes that are not in `big`
        if (n_wide > 0) {
            // [0, short_base): the reference's biggest nodes (its walk's window starts there); [short_base, short_base + 2 n_wide): the
            // wide nodes, ONE index range (trav_at_wide), the `front` biggest first by box area (the shortcut walk's window is the
            // head of that range), the others in tree order; then every other reference node and the sub-tree nodes
            big = select(1, front ? std::min<size_t>(RT_REF_FRONT, (size_t)n_top) : 0);
            short_base = (int)big.size();
            std::vector<std::pair<float, int>> wk((size_t)n_wide_top);  // (the wide forms of sub-tree parts have the smaller boxes: never in front)
            for (int w = 0; w < n_wide_top; w++) {
                float u[6] = {3e38f, 3e38f, 3e38f, -3e38f, -3e38f, -3e38f};
                for (int i = 0; i < wides[(size_t)w].nk; i++)
                    for (int c = 0; c < 3; c++) {
                        u[c] = std::min(u[c], wides[(size_t)w].box[i][c]);
                        u[3 + c] = std::max(u[3 + c], wides[(size_t)w].box[i][3 + c]);
                    }
                wk[(size_t)w] = {w == wide_root ? 3e38f : half_area(u), w};
            }
            const size_t nf = std::min<size_t>(front, (size_t)n_wide_top);
            auto wbigger = [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; };
            std::nth_element(wk.begin(), wk.begin() + (std::ptrdiff_t)nf, wk.end(), wbigger);
            std::sort(wk.begin(), wk.begin() + (std::ptrdiff_t)nf, wbigger);
            wide_index.assign((size_t)n_wide, -1);
            int next = short_base;
            // a small light tree goes in FRONT of everything: a light query visits each of its few nodes as often as a closest-hit
            // search visits the top of the shortcut tree, and the front of the wide range is what the LDS window holds
            if (light_root >= 0 && n_wide - light_root <= RT_LIGHT_TREE_FRONT_MAX && !getenv("RT_LIGHT_TREE_BACK"))
                for (int w = light_root; w < n_wide; w++) { wide_index[(size_t)w] = next; next += 2; }
            for (size_t k = 0; k < nf; k++) { wide_index[(size_t)wk[k].second] = next; next += 2; }
            for (int w = 0; w < n_wide; w++)
                if (wide_index[(size_t)w] < 0) { wide_index[(size_t)w] = next; next += 2; }
            rest_base = (size_t)short_base + 2 * (size_t)n_wide;
        } else if (n_short > 0 && front) {
            big = select(1, std::min<size_t>(RT_REF_FRONT, (size_t)n_top));
            short_base = (int)big.size();
            const std::vector<Key> rest = select(short_runs ? 4 : 2 | 4, front);  // (over runs the shortcut walk never enters a sub-tree)
            big.insert(big.end(), rest.begin(), rest.end());
        } else {
            big = select(1 | 2 | 4, front);
        }
        if (n_wide == 0) rest_base = big.size();
        BigVec<char> taken(std::max<size_t>(nq, 1), 0);
        for (size_t k = 0; k < big.size(); k++) {
            const Key& b = big[k];
            taken[b.kind == 0 ? (size_t)b.id : (b.kind == 1 ? (size_t)nn + (size_t)b.id : (size_t)nn + (size_t)n_sub + (size_t)b.id)] = 1;
            (b.kind == 0 ? inner_index[(size_t)b.id] : (b.kind == 1 ? sub_index[(size_t)b.id] : short_index[(size_t)b.id])) = (int)k;
        }
        // the others: front + their rank among the nodes not taken (count per range, prefix, assign)
        std::vector<size_t> cnt((size_t)n_workers_max + 1, 0);
        const int nr = par_ranges(nq, [&](int wi, size_t lo, size_t hi) {
            size_t c = 0;
            for (size_t q = lo; q < hi; q++)
                if (!taken[q] && (q >= (size_t)nn || s->nodes[q].child_a != -1)) c++;
            cnt[(size_t)wi + 1] = c;
        });
        for (int wi = 0; wi < nr; wi++) cnt[(size_t)wi + 1] += cnt[(size_t)wi];
        par_ranges(nq, [&](int wi, size_t lo, size_t hi) {
            size_t k = rest_base + cnt[(size_t)wi];
            for (size_t q = lo; q < hi; q++) {
                if (taken[q]) continue;
                if (q < (size_t)nn) {
                    if (s->nodes[q].child_a != -1) inner_index[q] = (int)k++;
                } else if (q < (size_t)nn + (size_t)n_sub) {
                    sub_index[q - (size_t)nn] = (int)k++;
                } else {
                    short_index[q - (size_t)nn - (size_t)n_sub] = (int)k++;
                }
            }
        });
    }
    if ((long long)n_inner + ntri > 0x7fffffffLL / 4) return fail(RT_ERR_LIMIT, "scene too large for 32-bit item indices");

    auto entry_of = [&](int node) -> int {
        const rt_bvh_node& n = s->nodes[node];
        if (n.child_a != -1) return inner_index[node];
        if (n.tri_len == 0) return RT_TRAV_EMPTY;
        if (sub_root[(size_t)node] >= 0) return sub_index[(size_t)sub_root[(size_t)node]];
        return n_inner + n.tri_offs / RT_TRI_FLOATS;
    };
    timer.stage("merge, order");
    RawFloats& items = L.items;
    RawFloats& cold = L.cold;
    RawVec<int>& place = L.place;
    items.size_to(((size_t)n_inner + (size_t)std::max(ntri, 0) + 1 + (size_t)n_light_items) * 16);  // (+ one item of padding behind the last triangle, then the light items)
    if (!early) cold.size_to((size_t)std::max(ntri, 1) * 16);  // (early: the other thread sizes, fills and uploads it)
    for (int k = 0; k < 16; k++) items[((size_t)n_inner + (size_t)ntri) * 16 + (size_t)k] = 0.f;
    if (ntri == 0 && !early)
        for (int k = 0; k < 16; k++) cold[(size_t)k] = 0.f;
    if (!early_tris) par_ranges((size_t)ntri, [&](int, size_t lo, size_t hi) {  // (early_tris: the other thread writes them, and `place`)
        for (size_t p = lo; p < hi; p++) {
            const int tri = at[p];
            place[(size_t)tri] = (int)p;
            fill_tri_item(&items[((size_t)n_inner + p) * 16], tris, tri, tflags[p], leaf_end[p], leaf_node[p]);
            if (early) continue;
            const float* u = tris + p * RT_TRI_FLOATS;                    // cold data stays in the reference's order
            float* cd = &cold[p * 16];
            for (int k = 0; k < 15; k++) cd[k] = u[9 + k];
            cd[15] = i2f(static_cast<int>(u[24]));  // main.cu:353
        }
    });
    par_ranges((size_t)nn, [&](int, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const rt_bvh_node& n = s->nodes[i];
            if (n.child_a == -1) continue;
            float* w = &items[(size_t)inner_index[i] * 16];
            const rt_bvh_node& A = s->nodes[n.child_a];
            const rt_bvh_node& B = s->nodes[n.child_b];
            w[0] = A.bbox_min[0]; w[1] = A.bbox_min[1]; w[2] = A.bbox_max[0]; w[3] = A.bbox_max[1];
            w[4] = B.bbox_min[0]; w[5] = B.bbox_min[1]; w[6] = B.bbox_max[0]; w[7] = B.bbox_max[1];
            w[8] = A.bbox_min[2]; w[9] = A.bbox_max[2]; w[10] = B.bbox_min[2]; w[11] = B.bbox_max[2];
            w[12] = i2f(entry_of(n.child_a));
            w[13] = i2f(entry_of(n.child_b));
            w[14] = i2f((int)i);
            w[15] = 0.f;
        }
    });
    par_ranges((size_t)n_sub, [&](int, size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; j++) {
            const SubNode& nd = subs[j];
            float* w = &items[(size_t)sub_index[j] * 16];
            const float *A = nd.box[0], *B = nd.box[1];
            w[0] = A[0]; w[1] = A[1]; w[2] = A[3]; w[3] = A[4];
            w[4] = B[0]; w[5] = B[1]; w[6] = B[3]; w[7] = B[4];
            w[8] = A[2]; w[9] = A[5]; w[10] = B[2]; w[11] = B[5];
            for (int k = 0; k < 2; k++) w[12 + k] = i2f(nd.e[k] >= 0 ? sub_index[(size_t)nd.e[k]] : n_inner + ~nd.e[k]);
            w[14] = i2f(nd.cnt);
            w[15] = i2f(1);
        }
    });
    par_ranges((size_t)n_short, [&](int, size_t lo, size_t hi) {  // shortcut nodes: sub-tree nodes that stand for no leaf of their own
        for (size_t j = lo; j < hi; j++) {
            const ShortNode& nd = shorts[j];
            float* w = &items[(size_t)short_index[j] * 16];
            const float *A = nd.box[0], *B = nd.box[1];
            w[0] = A[0]; w[1] = A[1]; w[2] = A[3]; w[3] = A[4];
            w[4] = B[0]; w[5] = B[1]; w[6] = B[3]; w[7] = B[4];
            w[8] = A[2]; w[9] = A[5]; w[10] = B[2]; w[11] = B[5];
            for (int k = 0; k < 2; k++)  // (what hangs below: see ShortPrim::node)
                w[12 + k] = i2f(nd.e[k] >= 0 ? short_index[(size_t)nd.e[k]] : ((~nd.e[k] & 1) ? sub_index[(size_t)(~nd.e[k] >> 1)] : n_inner + (~nd.e[k] >> 1)));
            w[14] = i2f(0);
            w[15] = i2f(2);
        }
    });
    for (int j = 0; j < n_light_items; j++) {  // light items (rt_device.h): the light's triangle as the triangle test reads it + normal, area, reference index
        float* h = &items[((size_t)n_inner + (size_t)ntri + 1 + (size_t)j) * 16];
        const float* Lr[6];
        for (int k = 0; k < 6; k++) Lr[k] = &lights[RT_LIGHT_F4(j, k) * 4];
        h[0] = Lr[0][0]; h[1] = Lr[0][1]; h[2] = Lr[0][2];                    // p1
        h[3] = Lr[0][3]; h[4] = Lr[1][0]; h[5] = Lr[1][1];                    // e1
        h[6] = Lr[1][2]; h[7] = Lr[1][3]; h[8] = Lr[2][0];                    // e2
        h[9] = i2f(1);                                                       // flags: last (and only) triangle of its leaf
        h[10] = Lr[3][0];                                                    // area
        h[11] = Lr[4][3];                                                    // the light's place in the reference's list (int bits)
        h[12] = Lr[2][1]; h[13] = Lr[2][2]; h[14] = Lr[2][3]; h[15] = 0.f;   // normal
    }
    par_ranges((size_t)n_wide, [&](int, size_t lo, size_t hi) {  // wide shortcut nodes: two items each (layout in rt_device.h)
        for (size_t j = lo; j < hi; j++) {
            const WideNode& nd = wides[j];
            float* w = &items[(size_t)wide_index[j] * 16];
            for (int i = 0; i < 4; i++) {
                for (int c = 0; c < 3; c++) {
                    w[4 * c + i] = nd.box[i][c];            // item 0: min.x[4], min.y[4], min.z[4]
                    w[16 + 4 * c + i] = nd.box[i][3 + c];   // item 1: max.x[4], max.y[4], max.z[4]
                }
                int e = RT_TRAV_EMPTY;
                if (i < nd.nk) e = nd.e[i] >= 0 ? wide_index[(size_t)nd.e[i]] : ((~nd.e[i] & 1) ? sub_index[(size_t)(~nd.e[i] >> 1)] : n_inner + (~nd.e[i] >> 1));
                w[12 + i] = i2f(e);
            }
            w[28] = i2f(nd.nk); w[29] = i2f(0); w[30] = i2f(0); w[31] = i2f(3);
        }
    });
    L.ref_boxes.assign((size_t)std::max(nn, 1) * 8, 0.f);
    for (int i = 0; i < nn; i++)
        for (int k = 0; k < 3; k++) {
            L.ref_boxes[(size_t)i * 8 + k] = s->nodes[i].bbox_min[k];
            L.ref_boxes[(size_t)i * 8 + 4 + k] = s->nodes[i].bbox_max[k];
        }

    timer.stage("item stream, cold data");
    // ---- materials: decoded colour (Vec3::fromColorInt, vec3.h:118-123)
    std::vector<float>& mats = L.mats;
    mats.assign(std::max<size_t>(s->n_mats, 1) * 12, 0.f);
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

    // ---- spheres
    std::vector<float>& spheres = L.spheres;
    spheres.assign((size_t)std::max(s->n_spheres, 1) * 8, 0.f);
    for (int i = 0; i < s->n_spheres; i++) {
        const float* q = s->tris + 1 + 5 * (size_t)i;
        float* d = &spheres[(size_t)i * 8];
        d[0] = q[0]; d[1] = q[1]; d[2] = q[2]; d[3] = q[3];
        d[4] = i2f(static_cast<int>(q[4]));
    }

    L.root_entry = nn > 0 ? entry_of(0) : RT_TRAV_DONE;
    L.n_inner = n_inner;
    L.n_sub = n_sub;
    L.n_tris = ntri;
    L.depth = deepest;
    L.n_short = n_wide > 0 ? n_wide : n_short;
    L.n_wide = n_wide;
    L.short_root = wide_root >= 0 ? wide_index[(size_t)wide_root] : (short_root >= 0 ? short_index[(size_t)short_root] : RT_TRAV_DONE);
    L.wide_need = wide_need;
    L.light_root = light_root >= 0 ? wide_index[(size_t)light_root] : RT_TRAV_DONE;
    L.n_light_items = n_light_items;
    L.light_need = light_need;
    L.below_short = below_leaf;
    L.short_base = short_base;
    L.short_runs = short_runs;
    L.short_single = short_single;
    L.short_cut = short_cut;
    // stack entries of a shortcut search.  The wide tree's worst path may need many (three per level); the search is given
    // RT_WIDE_STACK of them and gives up beyond (trav_step_wide: the ray then walks the reference's tree)
    // (+ 2: a visit wants room for three pushes even where the node has two children)
    int wide_stack = RT_WIDE_STACK;
    if (const char* e = getenv("RT_WIDE_STACK")) wide_stack = std::max(5, atoi(e));  // experiments only
    L.short_depth = n_wide > 0 ? std::min(wide_need + 2, wide_stack) + below_leaf : (n_short > 0 ? short_levels + below_leaf : 0);
    L.light_box_off = light_box_off;
    L.light_place_off = light_place_off;
    L.scene_scale = scene_scale;
    L.box_pad = box_pad;
    timer.stage("materials, lights, spheres");
    return RT_OK;
}
}  // namespace

static int prime_trace_kernels(rt_context* c);
// The device half of rt_upload_scene: allocations and copies of a layout that is built (rt_comm_upload_scene builds it
// once for all its ranks).
static int upload_built(rt_context* c, const rt_scene_desc* s, const HostLayout& L, UploadTimer& timer, EarlyUpload* early = nullptr) {
    int rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const int nn = s->n_nodes;
    free_scene(c);
    if ((rc = upload_buf(c, &c->d_place, L.place.data(), L.place.size() * 4)) != RT_OK) return rc;
    if ((rc = upload_buf(c, &c->d_refboxes, L.ref_boxes.data(), L.ref_boxes.size() * 4)) != RT_OK) return rc;
    if (early && early->d_tri_items) {
        // the triangles' items are on the device already: the nodes' items in front of them and what stands behind them
        // (padding, light items) come from the host, the triangles' move into place on the device
        const size_t tri_lo = (size_t)L.n_inner * 64, tri_hi = tri_lo + (size_t)L.n_tris * 64, total = L.items.size() * 4;
        const char* host = (const char*)L.items.data();
        HIP_TRY(c, hipMalloc(&c->d_items, total));
        if (tri_lo) HIP_TRY(c, hipMemcpy(c->d_items, host, tri_lo, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy((char*)c->d_items + tri_hi, host + tri_hi, total - tri_hi, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy((char*)c->d_items + tri_lo, early->d_tri_items, tri_hi - tri_lo, hipMemcpyDeviceToDevice));
        HIP_TRY(c, hipFree(early->d_tri_items));
        early->d_tri_items = nullptr;
    } else if ((rc = upload_buf(c, &c->d_items, L.items.data(), L.items.size() * 4)) != RT_OK) return rc;
    if (early) {  // (copied while the layout was built)
        c->d_cold = early->d_cold;
        c->d_tex = early->d_tex;
        early->d_cold = early->d_tex = nullptr;
    } else if ((rc = upload_buf(c, &c->d_cold, L.cold.data(), L.cold.size() * 4)) != RT_OK) return rc;
    if ((rc = upload_buf(c, &c->d_mats, L.mats.data(), L.mats.size() * 4)) != RT_OK) return rc;
    if ((rc = upload_buf(c, &c->d_lights, L.lights.data(), L.lights.size() * 4)) != RT_OK) return rc;
    if ((rc = upload_buf(c, &c->d_spheres, L.spheres.data(), L.spheres.size() * 4)) != RT_OK) return rc;
    if (!early && (rc = upload_buf(c, &c->d_tex, s->textures, s->n_tex_floats * 4)) != RT_OK) return rc;

    Scene& sc = c->sc;
    sc.items = (const float4*)c->d_items;
    sc.tri_place = (const int*)c->d_place;
    sc.ref_boxes = (const float4*)c->d_refboxes;
    sc.short_root = L.short_root;
    sc.short_base = L.short_base;
    sc.short_runs = L.short_runs ? 1 : 0;
    sc.short_single = L.short_single ? 1 : 0;
    sc.light_root = L.light_root;
    sc.n_light_items = L.n_light_items;
    sc.wide_lo = L.short_base;
    sc.wide_n = 2 * L.n_wide;
    // (the stack has max(depth, short_depth) entries: trace_cfg; the sub-trees below the shortcut tree push one entry per level)
    sc.wide_sp_max = std::max(L.depth, L.short_depth) - L.below_short - 3;
    sc.n_inner = L.n_inner;
    sc.n_sub = L.n_sub;
    sc.cull_far = RT_CULL_FAR * L.scene_scale;
    sc.tri_cold = (const float4*)c->d_cold;
    sc.mats = (const float4*)c->d_mats;
    sc.lights = (const float4*)c->d_lights;
    sc.light_boxes = sc.lights + L.light_box_off / 4;
    sc.light_place = (const int*)((const float*)c->d_lights + L.light_place_off);
    sc.spheres = (const float4*)c->d_spheres;
    sc.tex = (const float*)c->d_tex;
    sc.n_tex = (long long)s->n_tex_floats;
    sc.n_tris = L.n_tris;
    sc.n_mats = (int)s->n_mats;
    sc.n_lights = s->n_light_tris;
    sc.n_spheres = s->n_spheres;
    sc.has_root = nn > 0;
    sc.root_entry = L.root_entry;
    for (int k = 0; k < 3; k++) {
        sc.root_min[k] = nn > 0 ? s->nodes[0].bbox_min[k] : 0.f;
        sc.root_max[k] = nn > 0 ? s->nodes[0].bbox_max[k] : 0.f;
    }
    timer.stage("device allocations, copies");
    c->bvh_depth = L.depth;
    c->short_depth = L.short_depth;
    c->has_scene = true;
    return prime_trace_kernels(c);
}

int rt_upload_scene(rt_context* c, const rt_scene_desc* s) {
    if (!c) return RT_ERR_INVALID;
    UploadTimer timer;
    HostLayout L;
    std::string err;
    EarlyUpload early;
    early.c = c;
    int rc = build_layout(s, L, err, timer, getenv("RT_NO_EARLY_UPLOAD") ? nullptr : &early);
    if (rc != RT_OK) {
        early.discard();
        return set_err(c, rc, err);
    }
    if (early.started && (rc = early.finish()) != RT_OK) {
        const std::string msg = early.err;
        early.discard();
        return set_err(c, rc, msg);
    }
    timer.stage("early copies joined (cold data, textures)");
    rc = upload_built(c, s, L, timer, early.started ? &early : nullptr);
    if (rc != RT_OK) early.discard();
    return rc;
}

// Host-side inspection of the device layout (include/rt.h): no device needed.
struct rt_layout {
    HostLayout L;
};

int rt_layout_create(const rt_scene_desc* scene, rt_layout** out, char* err_buf, size_t err_len) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    rt_layout* h = new (std::nothrow) rt_layout();
    if (!h) return RT_ERR_NOMEM;
    std::string err;
    UploadTimer timer;
    const int rc = build_layout(scene, h->L, err, timer);
    if (rc != RT_OK) {
        if (err_buf && err_len) snprintf(err_buf, err_len, "%s", err.c_str());
        delete h;
        return rc;
    }
    *out = h;
    return RT_OK;
}

int rt_layout_get_v2(const rt_layout* h, rt_layout_info* info_out, size_t info_size, const float** items, const int32_t** places) {
    if (!h) return RT_ERR_INVALID;
    if (info_out) {
        rt_layout_info full, *info = &full;
        memset(&full, 0, sizeof(full));
        info->n_inner = h->L.n_inner;
        info->n_sub = h->L.n_sub;
        info->n_tris = h->L.n_tris;
        info->root_entry = h->L.root_entry;
        info->depth = h->L.depth;
        info->box_pad = h->L.box_pad;
        info->cull_far = RT_CULL_FAR * h->L.scene_scale;
        info->n_short = h->L.n_short;
        info->short_root = h->L.short_root;
        info->short_base = h->L.short_base;
        info->short_depth = h->L.short_depth;
        info->short_single = h->L.short_single ? 1 : 0;
        info->short_cut = h->L.short_cut;
        info->n_wide = h->L.n_wide;
        info->wide_need = h->L.wide_need;
        info->wide_sp_max = std::max(h->L.depth, h->L.short_depth) - h->L.below_short - 3;
        info->light_root = h->L.light_root;
        info->n_light_items = h->L.n_light_items;
        info->light_need = h->L.light_need;
        memcpy(info_out, &full, std::min(info_size, sizeof(full)));
    }
    if (items) *items = h->L.items.data();
    if (places) *places = h->L.place.data();
    return RT_OK;
}

// The symbol round 1 exported: fills round 1's fields only.
int rt_layout_get(const rt_layout* h, rt_layout_info* info, const float** items, const int32_t** places) {
    return rt_layout_get_v2(h, info, RT_LAYOUT_INFO_V1_SIZE, items, places);
}

void rt_layout_destroy(rt_layout* h) { delete h; }

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

int rt_release_scratch(rt_context* c) {
    if (!c) return RT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_samples) (void)hipFree(c->d_samples);
    c->d_samples = nullptr;
    c->samples_floats = 0;
    if (c->d_wf) (void)hipFree(c->d_wf);
    c->d_wf = nullptr;
    c->wf_slots = 0;
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    c->d_scratch = nullptr;
    c->scratch_bytes = 0;
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

namespace {

int ensure_wavefront(rt_context* c, int n_slots) {
    if (n_slots <= c->wf_slots) {
        c->wf.n_slots = n_slots;
        return RT_OK;
    }
    if (c->d_wf) (void)hipFree(c->d_wf);
    c->d_wf = nullptr;
    c->wf_slots = 0;
    // st_dir, st_thr, st_col, hits: 16 B each; ray queue: 32 B  => 96 B per slot
    const size_t bytes = (size_t)n_slots * 96;
    HIP_TRY(c, hipMalloc(&c->d_wf, bytes));
    float4* base = (float4*)c->d_wf;
    c->wf.st_dir = base;
    c->wf.st_thr = base + (size_t)n_slots;
    c->wf.st_col = base + 2 * (size_t)n_slots;
    c->wf.hits = base + 3 * (size_t)n_slots;
    c->wf.rayq = base + 4 * (size_t)n_slots;
    c->wf.qcount = c->d_tile_counter + 4;
    c->wf.cursor = c->d_tile_counter + 6;
    c->wf.n_slots = n_slots;
    c->wf_slots = n_slots;
    return RT_OK;
}

// Launch shape of the trace kernel for the uploaded scene: one 1024-thread block per CU whose LDS
// (160 KiB) holds the per-lane stacks (bvh depth entries) and as much of the top of the tree as fits.
struct TraceCfg {
    int grid, n_cached, stack_entries, n_mats_lds;
    size_t lds_bytes;
};

#define RT_LDS_MATS_MAX 128  // material tables up to this size go to LDS in the path kernel (48 B each)
// compact_wide (the path kernel's shortcut walk): the window holds wide nodes only, 112 bytes each (rt_device.h, WIDE7), unless the whole
// item stream fits (then the kernel variant that keeps everything in the LDS runs, with the padded records).
TraceCfg trace_cfg(rt_context* c, int waves = RT_TRACE_WAVES, bool with_mats = false, bool fast = false, bool compact_wide = false) {
    TraceCfg t;
    // (shortcut walk: a search that is not certified walks the reference's tree with the same stack)
    t.stack_entries = std::max(fast ? std::max(c->bvh_depth, c->short_depth) : c->bvh_depth, 1);
    t.n_mats_lds = (with_mats && c->sc.n_mats <= RT_LDS_MATS_MAX && !getenv("RT_NO_LDS_MATS")) ? c->sc.n_mats : 0;
    const size_t queue_bytes = (with_mats && c->sc.n_lights >= RT_LIGHTS_COOP_MIN) ? (size_t)waves * RT_LQ_CAP * 4 : 0;  // the light loop's queue
    const size_t stack_bytes = (size_t)waves * t.stack_entries * 64 * sizeof(int) + (size_t)t.n_mats_lds * 48 + queue_bytes;
    const size_t budget = RT_LDS_BUDGET_BYTES;
    size_t cache_nodes = budget > stack_bytes ? (budget - stack_bytes) / (RT_LDS_CACHE_FLOAT4S(1) * 16) : 0;
    const char* cap = getenv("RT_LDS_NODES");  // experiments only
    if (cap) cache_nodes = std::min<size_t>(cache_nodes, (size_t)std::max(0, atoi(cap)));
    t.n_cached = (int)std::min<size_t>(cache_nodes, (size_t)(c->sc.n_inner - (fast ? c->sc.short_base : 0)));
    // a scene whose whole item stream fits (the Cornell box): the triangles too, so that a leaf step reads the LDS
    // instead of L1 (path-parallel kernel only: its leaf step knows)
    const size_t all_items = (size_t)c->sc.n_inner + (size_t)c->sc.n_tris + (c->sc.n_light_items ? 1 + (size_t)c->sc.n_light_items : 0);  // (the light items stand behind the triangles)
    const bool everything = with_mats && c->sc.n_tris > 0 && c->sc.n_sub == 0 && all_items <= cache_nodes && !getenv("RT_NO_LDS_TRIS");
    if (everything) t.n_cached = (int)all_items;
    t.lds_bytes = RT_LDS_CACHE_FLOAT4S(t.n_cached) * 16 + stack_bytes;
    if (RT_WIDE7 && fast && compact_wide && !everything) {
        size_t wide_nodes = budget > stack_bytes ? (budget - stack_bytes) / (RT_WIDE7_F4 * 16) : 0;
        if (cap) wide_nodes = std::min<size_t>(wide_nodes, (size_t)std::max(0, atoi(cap)) / 2);
        wide_nodes = std::min<size_t>(wide_nodes, (size_t)c->sc.wide_n / 2);
        t.n_cached = (int)(2 * wide_nodes);
        t.lds_bytes = wide_nodes * RT_WIDE7_F4 * 16 + stack_bytes;
    }
    t.grid = c->num_cus;
    return t;
}

// rt_trace_rays / rt_trace_rays_noalloc: the shortcut tree unless the caller asks for the reference's walk or brute force
bool trace_uses_shortcut(rt_context* c, uint32_t flags) {
    return c->sc.short_root >= 0 && (flags & RT_F_BVH) && !(flags & RT_F_REFWALK) && !getenv("RT_REFWALK");
}

int set_trace_lds(rt_context* c, const TraceCfg& t) {
    if ((size_t)c->trace_lds_set >= t.lds_bytes) return RT_OK;
    HIP_TRY(c, hipFuncSetAttribute((const void*)rt_trace_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds_bytes));
    HIP_TRY(c, hipFuncSetAttribute((const void*)rt_trace_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds_bytes));
    HIP_TRY(c, hipFuncSetAttribute((const void*)rt_trace_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    c->trace_lds_set = (int)t.lds_bytes;
    return RT_OK;
}

}  // namespace
// The dynamic-LDS attribute of the trace kernels for this scene, set at upload: rt_trace_rays_noalloc then only launches.
static int prime_trace_kernels(rt_context* c) {
    int rc = set_trace_lds(c, trace_cfg(c, RT_TRACE_WAVES, false, false));
    if (rc == RT_OK && c->sc.short_root != RT_TRAV_DONE) rc = set_trace_lds(c, trace_cfg(c, RT_TRACE_WAVES, false, true));
    return rc;
}
namespace {

typedef void (*paths_fn)(Scene, Params, float*, unsigned int*, unsigned long long*, int, int, int, int, unsigned int, int);
paths_fn paths_kernel_variant(bool stats, bool big, bool extra, bool coop, bool tlds = false, bool fast = false) {
    // the shortcut walk: plain triangle scenes with sub-tree style nodes (its own), no extras
    static const paths_fn shortcut[4] = {rt_render_paths_kernel<false, true, false, false, false, true>, rt_render_paths_kernel<true, true, false, false, false, true>,
                                         rt_render_paths_kernel<false, true, false, true, false, true>,  rt_render_paths_kernel<true, true, false, true, false, true>};
    // ... of a scene whose whole item stream sits in the LDS
    static const paths_fn shortcut_tiny[4] = {rt_render_paths_kernel<false, true, false, false, true, true>, rt_render_paths_kernel<true, true, false, false, true, true>,
                                              rt_render_paths_kernel<false, true, false, true, true, true>,  rt_render_paths_kernel<true, true, false, true, true, true>};
    if (fast && tlds) return shortcut_tiny[(stats ? 1 : 0) | (coop ? 2 : 0)];
    if (fast) return shortcut[(stats ? 1 : 0) | (coop ? 2 : 0)];
#define RT_V4(C) \
    rt_render_paths_kernel<false, false, false, C>, rt_render_paths_kernel<true, false, false, C>, \
    rt_render_paths_kernel<false, true, false, C>,  rt_render_paths_kernel<true, true, false, C>,  \
    rt_render_paths_kernel<false, false, true, C>,  rt_render_paths_kernel<true, false, true, C>,  \
    rt_render_paths_kernel<false, true, true, C>,   rt_render_paths_kernel<true, true, true, C>
    static const paths_fn table[16] = {RT_V4(false), RT_V4(true)};
#undef RT_V4
    // the whole item stream in LDS (tiny scenes: no sub-trees there): leaf steps read the LDS
    static const paths_fn tiny[8] = {rt_render_paths_kernel<false, false, false, false, true>, rt_render_paths_kernel<true, false, false, false, true>,
                                     rt_render_paths_kernel<false, false, true, false, true>,  rt_render_paths_kernel<true, false, true, false, true>,
                                     rt_render_paths_kernel<false, false, false, true, true>,  rt_render_paths_kernel<true, false, false, true, true>,
                                     rt_render_paths_kernel<false, false, true, true, true>,   rt_render_paths_kernel<true, false, true, true, true>};
    if (tlds && !big) return tiny[(stats ? 1 : 0) | (extra ? 2 : 0) | (coop ? 4 : 0)];
    return table[(stats ? 1 : 0) | (big ? 2 : 0) | (extra ? 4 : 0) | (coop ? 8 : 0)];
}

// rt_render_pool_kernel (rt_pool_kernel.h): n_trav traversal waves with stacks, the others shade; the rings' bytes come
// out of the stacks the vertex waves do not need and, beyond that, out of the node cache.
struct PoolCfg {
    int n_trav, cap_ready, cap_hit, cap_miss, population, n_cached, n_mats_lds, stack_entries;
    size_t lds_bytes;
};
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static int pow2_floor(int v) {
    int p = 1;
    while (p * 2 <= v) p *= 2;
    return p;
}
PoolCfg pool_cfg(rt_context* c) {
    PoolCfg p;
    p.n_trav = std::min(RT_MEGA_WAVES - 1, std::max(1, env_int("RT_POOL_TRAV", 12)));
    p.cap_ready = pow2_floor(std::max(64, env_int("RT_POOL_READY", 256)));
    p.cap_hit = pow2_floor(std::max(64, env_int("RT_POOL_HIT", 128)));
    p.cap_miss = pow2_floor(std::max(64, env_int("RT_POOL_MISS", 128)));
    const int max_surplus = p.cap_ready + std::min(p.cap_hit, p.cap_miss) - 64;
    p.population = p.n_trav * 64 + std::min(max_surplus, std::max(64, env_int("RT_POOL_SURPLUS", 320)));
    p.stack_entries = std::max(c->bvh_depth, 1);
    p.n_mats_lds = (c->sc.n_mats <= RT_LDS_MATS_MAX) ? c->sc.n_mats : 0;
    const size_t fixed = (size_t)p.n_trav * p.stack_entries * 256 + (size_t)p.n_mats_lds * 48 + (size_t)p.cap_ready * 16 * RT_POOL_READY_F4 +
                         (size_t)p.cap_hit * 16 * RT_POOL_HIT_F4 + (size_t)p.cap_miss * 16 * RT_POOL_MISS_F4 + 16 * RT_POOL_CTL_F4;
    const size_t budget = 160 * 1024 - 1024;
    size_t cache_nodes = budget > fixed ? (budget - fixed) / (RT_LDS_CACHE_FLOAT4S(1) * 16) : 0;
    if (const char* e = getenv("RT_LDS_NODES")) cache_nodes = std::min<size_t>(cache_nodes, (size_t)std::max(0, atoi(e)));  // experiments only
    p.n_cached = (int)std::min<size_t>(cache_nodes, (size_t)c->sc.n_inner);
    p.lds_bytes = RT_LDS_CACHE_FLOAT4S(p.n_cached) * 16 + fixed;
    return p;
}
typedef void (*pool_fn)(Scene, Params, float*, unsigned int*, unsigned long long*, int, int, int, int, unsigned int, int, int, int, int, int, int);
pool_fn pool_kernel_variant(bool stats, bool big, bool coop) {
    static const pool_fn table[8] = {rt_render_pool_kernel<false, false, false>, rt_render_pool_kernel<true, false, false>,
                                     rt_render_pool_kernel<false, true, false>,  rt_render_pool_kernel<true, true, false>,
                                     rt_render_pool_kernel<false, false, true>,  rt_render_pool_kernel<true, false, true>,
                                     rt_render_pool_kernel<false, true, true>,   rt_render_pool_kernel<true, true, true>};
    return table[(stats ? 1 : 0) | (big ? 2 : 0) | (coop ? 4 : 0)];
}

int set_mega_lds(rt_context* c, const TraceCfg& t) {
    if ((size_t)c->mega_lds_set >= t.lds_bytes) return RT_OK;
    HIP_TRY(c, hipFuncSetAttribute((const void*)rt_render_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds_bytes));
    HIP_TRY(c, hipFuncSetAttribute((const void*)rt_render_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds_bytes));
    for (int v = 0; v < 32; v++)
        HIP_TRY(c, hipFuncSetAttribute((const void*)paths_kernel_variant((v & 1) != 0, (v & 2) != 0, (v & 4) != 0, (v & 8) != 0, (v & 16) != 0),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds_bytes));
    for (int v = 0; v < 8; v++)
        HIP_TRY(c, hipFuncSetAttribute((const void*)paths_kernel_variant((v & 1) != 0, true, false, (v & 2) != 0, (v & 4) != 0, true),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, RT_LDS_BUDGET_BYTES));
    for (int v = 0; v < 8; v++)
        HIP_TRY(c, hipFuncSetAttribute((const void*)pool_kernel_variant((v & 1) != 0, (v & 2) != 0, (v & 4) != 0),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    c->mega_lds_set = (int)t.lds_bytes;
    return RT_OK;
}

// Lock-step fused kernel (one lane owns a pixel for the whole call).
int render_lockstep(rt_context* c, const Params& P, float* accum, bool want_stats) {
    HIP_TRY(c, hipMemsetAsync(c->d_tile_counter, 0, sizeof(unsigned int), c->stream));
    // persistent grid: ONE block of RT_MEGA_WAVES waves per CU (its register count allows no more), never more than the work
    const TraceCfg tc = trace_cfg(c, RT_MEGA_WAVES);
    int rc = set_mega_lds(c, tc);
    if (rc != RT_OK) return rc;
    typedef void (*mega_fn)(Scene, Params, float*, unsigned int*, unsigned long long*, int, int);
    const mega_fn kfn = want_stats ? rt_render_kernel<true> : rt_render_kernel<false>;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kfn, 64 * RT_MEGA_WAVES, tc.lds_bytes) != hipSuccess || occ < 1) occ = 1;
    const int want_blocks = (P.n_tiles + RT_MEGA_WAVES - 1) / RT_MEGA_WAVES;
    const int blocks = std::max(1, std::min(c->num_cus * occ, want_blocks));
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(64 * RT_MEGA_WAVES), tc.lds_bytes, c->stream, c->sc, P, accum, c->d_tile_counter,
                       c->d_stats, tc.n_cached, tc.stack_entries);
    HIP_TRY(c, hipGetLastError());
    return RT_OK;
}

// Path-parallel fused kernel: the call's samples in chunks of S; per chunk one persistent launch that
// writes one radiance per path, and one pass that adds them to the accumulator in sample order.
int render_paths(rt_context* c, const Params& P, float* accum, bool want_stats, bool profile, rt_render_stats* out) {
    // the block-level ray pool (RT_F_POOL / RT_POOL=1): plain triangle scenes (no spheres, volume, ray marching, brute force)
    const bool pool_wanted = (P.flags & RT_F_POOL) || (getenv("RT_POOL") && atoi(getenv("RT_POOL")) != 0);
    const bool pool = pool_wanted && !(P.flags & (RT_F_VOLUME | RT_F_RAYMARCH)) && (P.flags & RT_F_BVH) &&
                      !((P.flags & RT_F_SPHERES) && c->sc.n_spheres > 0);
    // the shortcut tree (rt_device.h) unless the call asks for the reference's walk, or counts it (RT_F_STATS)
    const bool fast = c->sc.short_root >= 0 && !pool && !(P.flags & (RT_F_REFWALK | RT_F_VOLUME | RT_F_RAYMARCH)) && (P.flags & RT_F_BVH) &&
                      (!want_stats || (P.flags & RT_F_SHORTCUT_STATS)) && !getenv("RT_REFWALK") &&
                      // (many lights but no light tree -- the two-child shortcut form of the experiments: the shortcut kernel would run the
                      // plain loop over all lights per lane; the reference-walk kernel has the wave-cooperative loop)
                      !(c->sc.n_lights >= RT_LIGHTS_COOP_MIN && c->sc.light_root < 0);
    const TraceCfg tc = trace_cfg(c, RT_MEGA_WAVES, true, fast, c->sc.n_lights < RT_LIGHTS_COOP_MIN);
    int rc = set_mega_lds(c, tc);
    if (rc != RT_OK) return rc;
    const size_t slots = (size_t)P.n_tiles * 64;  // pixels incl. the padding of edge tiles
    size_t budget = (size_t)8 << 30;              // bytes of per-path radiance kept in flight (HBM is 288 GB)
    if (const char* e = getenv("RT_SAMPLE_BUF_MB")) budget = (size_t)std::max(1, atoi(e)) << 20;  // experiments only
    long long S = std::min<long long>(P.sample_count, std::max<long long>(1, (long long)(budget / (slots * 12))));
    S = std::min<long long>(S, (long long)(0x7fffffffu / slots));  // 32-bit path counter
    if (S < 1) return set_err(c, RT_ERR_LIMIT, "image too large for the path-parallel kernel");
    // S samples per pixel + one more slot per pixel for the running sum carried between chunks
    while (c->samples_floats < slots * (size_t)(S + 1) * 3) {
        if (c->d_samples) (void)hipFree(c->d_samples);
        c->d_samples = nullptr;
        c->samples_floats = 0;
        if (hipMalloc((void**)&c->d_samples, slots * (size_t)(S + 1) * 12) == hipSuccess) {
            c->samples_floats = slots * (size_t)(S + 1) * 3;
            break;
        }
        (void)hipGetLastError();
        if (S == 1) return set_err(c, RT_ERR_NOMEM, "no device memory for the per-path radiance buffer");
        S = (S + 1) / 2;
    }
    const paths_fn kfn = paths_kernel_variant(want_stats, c->sc.n_sub > 0, (P.flags & (RT_F_VOLUME | RT_F_RAYMARCH)) != 0,
                                            c->sc.n_lights >= RT_LIGHTS_COOP_MIN, tc.n_cached > c->sc.n_inner, fast);
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kfn, 64 * RT_MEGA_WAVES, tc.lds_bytes) != hipSuccess || occ < 1) occ = 1;
    const PoolCfg pc = pool ? pool_cfg(c) : PoolCfg();
    const pool_fn pfn = pool_kernel_variant(want_stats, c->sc.n_sub > 0, c->sc.n_lights >= RT_LIGHTS_COOP_MIN);
    if (pool) occ = 1;
    size_t ev_used = 0;
    for (long long k0 = 0; k0 < P.sample_count; k0 += S) {
        const int s = (int)std::min<long long>(S, P.sample_count - k0);
        const unsigned int n_paths = (unsigned int)(slots * (size_t)s);
        HIP_TRY(c, hipMemsetAsync(c->d_tile_counter, 0, (pool ? 3 : 1) * sizeof(unsigned int), c->stream));  // ([2]: the pool kernel's give-up flag)
        const long long want_blocks = ((long long)n_paths + 64 * RT_MEGA_WAVES - 1) / (64 * RT_MEGA_WAVES);
        const int blocks = (int)std::max<long long>(1, std::min<long long>((long long)c->num_cus * occ, want_blocks));
        const bool timed = profile && ev_used + 2 <= 4096;
        if (timed) {
            while (c->ev_pool.size() < ev_used + 2) {
                hipEvent_t e;
                HIP_TRY(c, hipEventCreate(&e));
                c->ev_pool.push_back(e);
            }
            HIP_TRY(c, hipEventRecord(c->ev_pool[ev_used], c->stream));
        }
        if (pool)
            hipLaunchKernelGGL(pfn, dim3(blocks), dim3(64 * RT_MEGA_WAVES), pc.lds_bytes, c->stream, c->sc, P, c->d_samples, c->d_tile_counter,
                               c->d_stats, pc.n_cached, pc.stack_entries, (int)k0, s, n_paths, pc.n_mats_lds, pc.n_trav, pc.cap_ready,
                               pc.cap_hit, pc.cap_miss, pc.population);
        else
            hipLaunchKernelGGL(kfn, dim3(blocks), dim3(64 * RT_MEGA_WAVES), tc.lds_bytes, c->stream, c->sc, P, c->d_samples, c->d_tile_counter,
                               c->d_stats, tc.n_cached, tc.stack_entries, (int)k0, s, n_paths, tc.n_mats_lds);
        if (timed) {
            HIP_TRY(c, hipEventRecord(c->ev_pool[ev_used + 1], c->stream));
            ev_used += 2;
        }
        hipLaunchKernelGGL(rt_sum_samples_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, c->stream, c->d_samples, accum,
                           c->d_samples + slots * (size_t)S * 3, P, s, (int)(k0 == 0), (int)(k0 + s >= P.sample_count));
    }
    HIP_TRY(c, hipGetLastError());
    if (pool) {  // a wave that waited for seconds gave up (rt_pool_kernel.h): the sums are incomplete
        HIP_TRY(c, hipMemcpyAsync(c->h_pinned, c->d_tile_counter + 2, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->h_pinned[0] != 0) return set_err(c, RT_ERR_DEVICE, "the ray pool kernel gave up waiting (watchdog)");
    }
    if (out && profile && ev_used) {  // RT_F_PROFILE: summed duration of the path kernel's launches alone
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        double sum = 0.0;
        for (size_t i = 0; i + 1 < ev_used; i += 2) {
            float ms = 0.f;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_pool[i], c->ev_pool[i + 1]));
            sum += ms;
        }
        out->trace_ms = sum;
        out->trace_launches = (uint32_t)(ev_used / 2);
    }
    return RT_OK;
}

int render_megakernel(rt_context* c, const Params& P, float* accum, bool want_stats, bool profile, rt_render_stats* out) {
    return (P.flags & RT_F_LOCKSTEP) ? render_lockstep(c, P, accum, want_stats) : render_paths(c, P, accum, want_stats, profile, out);
}

// Wavefront loop: logic -> trace -> logic -> ... until no slot emits a ray.
int render_wavefront(rt_context* c, const Params& P, float* accum, bool want_stats, bool profile, rt_render_stats* out) {
    const int n_slots = P.n_tiles * 64;
    int rc = ensure_wavefront(c, n_slots);
    if (rc != RT_OK) return rc;
    HIP_TRY(c, hipMemsetAsync(c->d_tile_counter + 4, 0, 4 * sizeof(unsigned int), c->stream));
    const long long bound = (long long)P.sample_count * std::max(P.max_bounces, 0) + 2;  // each slot: <= spp*bounces rays
    const int logic_blocks = (n_slots + RT_LOGIC_BLOCK - 1) / RT_LOGIC_BLOCK;
    const TraceCfg tc = trace_cfg(c);
    if ((rc = set_trace_lds(c, tc)) != RT_OK) return rc;
    const int check_every = 8;
    size_t ev_used = 0;
    const size_t ev_cap = 4096;
    long long it = 0;
    for (; it < bound; it++) {
        const int par = (int)(it & 1);
        if (want_stats)
            hipLaunchKernelGGL(rt_logic_kernel<true>, dim3(logic_blocks), dim3(RT_LOGIC_BLOCK), 0, c->stream, c->sc, P, c->wf, accum, (int)(it > 0x3fffffff ? 2 + par : it), c->d_stats);
        else
            hipLaunchKernelGGL(rt_logic_kernel<false>, dim3(logic_blocks), dim3(RT_LOGIC_BLOCK), 0, c->stream, c->sc, P, c->wf, accum, (int)(it > 0x3fffffff ? 2 + par : it), c->d_stats);
        const bool timed = profile && ev_used + 2 <= ev_cap;
        if (timed) {
            while (c->ev_pool.size() < ev_used + 2) {
                hipEvent_t e;
                HIP_TRY(c, hipEventCreate(&e));
                c->ev_pool.push_back(e);
            }
            HIP_TRY(c, hipEventRecord(c->ev_pool[ev_used], c->stream));
        }
        if (want_stats)
            hipLaunchKernelGGL(rt_trace_kernel<true>, dim3(tc.grid), dim3(64 * RT_TRACE_WAVES), tc.lds_bytes, c->stream, c->sc, c->wf.rayq,
                               c->wf.qcount + par, c->wf.cursor + par, c->wf.hits, P.flags, (int*)nullptr, c->d_stats, tc.n_cached, tc.stack_entries);
        else
            hipLaunchKernelGGL(rt_trace_kernel<false>, dim3(tc.grid), dim3(64 * RT_TRACE_WAVES), tc.lds_bytes, c->stream, c->sc, c->wf.rayq,
                               c->wf.qcount + par, c->wf.cursor + par, c->wf.hits, P.flags, (int*)nullptr, c->d_stats, tc.n_cached, tc.stack_entries);
        if (timed) {
            HIP_TRY(c, hipEventRecord(c->ev_pool[ev_used + 1], c->stream));
            ev_used += 2;
        }
        if ((it + 1) % check_every == 0) {  // has the pipeline drained?
            HIP_TRY(c, hipMemcpyAsync(c->h_pinned, c->wf.qcount + par, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (c->h_pinned[0] == 0) { it++; break; }
        }
    }
    HIP_TRY(c, hipGetLastError());
    if (out) {
        out->iterations = (uint32_t)it;
        if (profile && ev_used) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            double sum = 0.0;
            for (size_t i = 0; i + 1 < ev_used; i += 2) {
                float ms = 0.f;
                HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_pool[i], c->ev_pool[i + 1]));
                sum += ms;
            }
            out->trace_ms = sum;
            out->trace_launches = (uint32_t)(ev_used / 2);
        }
    }
    return RT_OK;
}

}  // namespace

}  // extern "C"
namespace {
int render_impl(rt_context* c, const rt_render_params* p, rt_render_stats* stats) {
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
    const bool profile = (P.flags & RT_F_PROFILE) != 0;
    if (stats) memset(stats, 0, sizeof(*stats));
#if RT_CYCLE_PROF == 2
    if (true) HIP_TRY(c, hipMemsetAsync(c->d_stats, 0, 32 * sizeof(unsigned long long), c->stream));
#else
    if (want_stats) HIP_TRY(c, hipMemsetAsync(c->d_stats, 0, 32 * sizeof(unsigned long long), c->stream));
#endif
    const bool work = P.n_tiles > 0 && P.sample_count > 0;
    if (work) {
        HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
        if (!(P.flags & RT_F_WAVEFRONT)) rc = render_megakernel(c, P, accum, want_stats, profile, stats);
        else rc = render_wavefront(c, P, accum, want_stats, profile, stats);
        if (rc != RT_OK) return rc;
        HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->samples_done += P.sample_count;
    if (stats) {
        if (work) {
            float ms = 0.f;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
            stats->kernel_ms = ms;
        }
#if RT_CYCLE_PROF == 2
        if (!want_stats && work) {  // the clocks of the kernel without counters: where a wave's cycles go
            unsigned long long h[32];
            HIP_TRY(c, hipMemcpy(h, c->d_stats, sizeof(h), hipMemcpyDeviceToHost));
            const double life = (double)std::max<unsigned long long>(h[22], 1);
            fprintf(stderr, "[rt prof2] kernel %.3f ms; wave lifetime split: traversal %.3f (wide part %.3f, two-child part %.3f, leaf iterations %.3f), vertex passes %.3f (shading part %.3f); "
                            "iterations: wide %llu (%.1f lanes), two-child %llu, leaf %llu (%.1f lanes); vertex passes %llu (%.1f lanes); cycles per wide iteration %.0f, per leaf iteration %.0f, per vertex pass %.0f\n",
                    stats->kernel_ms, h[18] / life, h[24] / life, (h[18] - h[24] - h[26]) / life, h[26] / life, h[21] / life, h[23] / life,
                    h[27], h[27] ? (double)h[31] / h[27] : 0.0, h[28], h[29], h[29] ? (double)h[30] / h[29] : 0.0, h[19], h[19] ? (double)h[20] / h[19] : 0.0,
                    h[27] ? (double)h[24] / h[27] : 0.0, h[29] ? (double)h[26] / h[29] : 0.0, h[19] ? (double)h[21] / h[19] : 0.0);
        }
#endif
        if (want_stats) {
            unsigned long long h[32];
            HIP_TRY(c, hipMemcpy(h, c->d_stats, sizeof(h), hipMemcpyDeviceToHost));
#ifdef RT_POOL_CENSUS
            if (h[16])
                fprintf(stderr, "[rt pool] traversal waves: %llu iterations, lanes walking %.1f / finished %.1f / empty %.1f; exchanges in %.3f of them, %.1f rays taken each; "
                                "%llu idle polls; leaf iterations %.3f (%.1f lanes at a leaf).  vertex waves: %llu idle polls, %llu shade jobs of %.1f, %llu close/open jobs of %.1f + %.1f\n",
                        h[16], (double)h[17] / h[16], (double)h[18] / h[16], (double)h[19] / h[16], (double)h[20] / h[16], h[20] ? (double)h[21] / h[20] : 0.0, h[22],
                        (double)h[23] / h[16], h[23] ? (double)h[24] / h[23] : 0.0, h[25], h[26], h[26] ? (double)h[27] / h[26] : 0.0, h[28],
                        h[28] ? (double)h[29] / h[28] : 0.0, h[28] ? (double)h[30] / h[28] : 0.0);
#endif
#ifdef RT_CYCLE_PROF
            fprintf(stderr, "[rt prof] trav: %llu wave-iterations, %llu lane-steps (%.1f lanes/iteration), %llu cycles (%.0f per iteration); "
                            "vertex: %llu passes, %llu lanes (%.1f per pass), %llu cycles (%.0f per pass)(shading part %.0f); wave lifetimes %llu cycles; kernel %.3f ms\n",
                    h[16], h[17], h[16] ? (double)h[17] / h[16] : 0.0, h[18], h[16] ? (double)h[18] / h[16] : 0.0, h[19], h[20],
                    h[19] ? (double)h[20] / h[19] : 0.0, h[21], h[19] ? (double)h[21] / h[19] : 0.0, h[19] ? (double)h[23] / h[19] : 0.0, h[22], stats->kernel_ms);
            fprintf(stderr, "[rt prof] census per iteration (lanes at inner / at leaf / waiting for a vertex pass): inner iterations %llu: %.1f / %.1f / %.1f; "
                            "leaf iterations %llu: %.1f / %.1f / %.1f\n",
                    h[24], h[24] ? (double)h[25] / h[24] : 0.0, h[24] ? (double)h[26] / h[24] : 0.0, h[24] ? (double)h[27] / h[24] : 0.0,
                    h[28], h[28] ? (double)h[29] / h[28] : 0.0, h[28] ? (double)h[30] / h[28] : 0.0, h[28] ? (double)h[31] / h[28] : 0.0);
#endif
            stats->paths = h[0]; stats->rays = h[1]; stats->inner_visits = h[2]; stats->leaf_visits = h[3];
            stats->tri_tests = h[4]; stats->hits = h[5]; stats->texel_fetches = h[6]; stats->light_tests = h[7];
            stats->dev_inner_visits = h[8]; stats->dev_lds_inner_visits = h[9]; stats->dev_tri_tests = h[10]; stats->dev_light_tests = h[11]; stats->shortcut_rewalks = h[12]; stats->dev_wide_visits = h[13]; stats->dev_light_loops = h[14];
        }
    }
    return RT_OK;
}
}  // namespace
extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

// The caller says how large ITS rt_render_stats is: the library fills a struct of its own and copies that many bytes
// (the structs only ever grow at the end), so a caller compiled against an older rt.h is never written past its struct.
int rt_render_v2(rt_context* c, const rt_render_params* p, rt_render_stats* stats, size_t stats_size) {
    rt_render_stats full;
    memset(&full, 0, sizeof(full));
    const int rc = render_impl(c, p, stats ? &full : nullptr);
    if (stats) memcpy(stats, &full, std::min(stats_size, sizeof(full)));
    return rc;
}
// The symbol round 1 exported: its callers hold round 1's struct (RT_RENDER_STATS_V1_SIZE bytes).
int rt_render(rt_context* c, const rt_render_params* p, rt_render_stats* stats) { return rt_render_v2(c, p, stats, RT_RENDER_STATS_V1_SIZE); }

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

int rt_readback_rgb8(rt_context* c, unsigned char* rgb8, size_t n_bytes) {
    if (!c || !rgb8) return RT_ERR_INVALID;
    const size_t n = (size_t)c->acc_rows * c->acc_width * 3;
    if (n == 0 || !accum_ptr(c)) return set_err(c, RT_ERR_INVALID, "nothing rendered yet");
    if (n_bytes != n) return set_err(c, RT_ERR_INVALID, "n_bytes does not match rows*width*3 of the last render");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_scratch(c, n);
    if (rc != RT_OK) return rc;
    unsigned char* tmp = (unsigned char*)c->d_scratch;
    const size_t npix = n / 3;
    const float samples = static_cast<float>(c->samples_done > 0 ? c->samples_done : 1);
    hipLaunchKernelGGL(rt_resolve_rgb8_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, c->stream, accum_ptr(c), tmp, npix, samples);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(rgb8, tmp, n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = set_err(c, RT_ERR_DEVICE, std::string("8-bit readback: ") + hipGetErrorString(e));
    return rc;
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
    int rc = ensure_scratch(c, n * sizeof(float));
    if (rc != RT_OK) return rc;
    float* tmp = (float*)c->d_scratch;
    rc = rt_resolve_device(c, tmp, n, mode);
    if (rc == RT_OK) {
        hipError_t e = hipMemcpy(rgb, tmp, n * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = set_err(c, RT_ERR_DEVICE, std::string("readback copy: ") + hipGetErrorString(e));
    }
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
    if (n > 0x7fffffffu / 8) return set_err(c, RT_ERR_INVALID, "too many rays in one call");
    HIP_TRY(c, hipSetDevice(c->device));
    // one block of the context's grow-only scratch, cut into 256-byte aligned pieces
    const size_t sizes[8] = {n * 24, n * 32, n * 16, n * 4, n * 4, n * 12, n * 12, n * 12};
    size_t offs[8], total = 0;
    for (int i = 0; i < 8; i++) { offs[i] = total; total += (sizes[i] + 255) & ~(size_t)255; }
    int rc = ensure_scratch(c, total);
    if (rc != RT_OK) return rc;
    char* base = (char*)c->d_scratch;
    float* d_rays = (float*)(base + offs[0]);
    float4* d_q = (float4*)(base + offs[1]);
    float4* d_hits = (float4*)(base + offs[2]);
    int* d_prim = (int*)(base + offs[3]);
    float* d_t = (float*)(base + offs[4]);
    float* d_pos = (float*)(base + offs[5]);
    float* d_nrm = (float*)(base + offs[6]);
    int* d_cnt = (int*)(base + offs[7]);
    HIP_TRY(c, hipMemcpyAsync(d_rays, org_dir, n * 24, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(rt_fill_queue_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_rays, n, d_q);
    unsigned int* d_n = c->d_tile_counter + 8;  // [8] = ray count, [9] = cursor
    const unsigned int hn[2] = {(unsigned int)n, 0u};
    HIP_TRY(c, hipMemcpyAsync(d_n, hn, sizeof(hn), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(d_cnt, 0, n * 12, c->stream));
    // per-ray visit counts describe the reference's walk: only then (or on request) is it walked for every ray
    const bool fast = trace_uses_shortcut(c, flags) && !counts;
    const TraceCfg tc = trace_cfg(c, RT_TRACE_WAVES, false, fast);
    rc = set_trace_lds(c, tc);
    if (rc != RT_OK) return rc;
    // the same persistent kernel the renderer uses (with refill), restricted to closest hit (+ spheres if asked)
    if (fast)
        hipLaunchKernelGGL((rt_trace_kernel<false, true>), dim3(tc.grid), dim3(64 * RT_TRACE_WAVES), tc.lds_bytes, c->stream, c->sc, d_q, d_n, d_n + 1,
                           d_hits, flags, (int*)nullptr, (unsigned long long*)nullptr, tc.n_cached, tc.stack_entries);
    else
        hipLaunchKernelGGL(rt_trace_kernel<true>, dim3(tc.grid), dim3(64 * RT_TRACE_WAVES), tc.lds_bytes, c->stream, c->sc, d_q, d_n, d_n + 1, d_hits,
                           flags, d_cnt, (unsigned long long*)nullptr, tc.n_cached, tc.stack_entries);
    hipLaunchKernelGGL(rt_hitinfo_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->sc, d_rays, d_hits, n, d_prim,
                       d_t, d_pos, d_nrm);
    HIP_TRY(c, hipGetLastError());
    if (prim_id) HIP_TRY(c, hipMemcpyAsync(prim_id, d_prim, n * 4, hipMemcpyDeviceToHost, c->stream));
    if (t) HIP_TRY(c, hipMemcpyAsync(t, d_t, n * 4, hipMemcpyDeviceToHost, c->stream));
    if (hit_pos) HIP_TRY(c, hipMemcpyAsync(hit_pos, d_pos, n * 12, hipMemcpyDeviceToHost, c->stream));
    if (normal3) HIP_TRY(c, hipMemcpyAsync(normal3, d_nrm, n * 12, hipMemcpyDeviceToHost, c->stream));
    if (counts) HIP_TRY(c, hipMemcpyAsync(counts, d_cnt, n * 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

// rt_trace_rays for callers that cannot allocate (include/rt.h): rays and results in DEVICE memory, queue / hit records /
// control words in the static scratch array of this device, no hipMalloc, hipFree or host copy.  Same kernels as rt_trace_rays.
int rt_trace_rays_noalloc(rt_context* c, const float* dev_org_dir, size_t n, uint32_t flags, int32_t* dev_prim_id, float* dev_t,
                          float* dev_hit_pos, float* dev_normal3, int sync) {
    if (!c) return RT_ERR_INVALID;
    if (!c->has_scene) return set_err(c, RT_ERR_NO_SCENE, "rt_trace_rays_noalloc before rt_upload_scene");
    if (n > (size_t)RT_NOALLOC_MAX_RAYS)
        return set_err(c, RT_ERR_LIMIT, "rt_trace_rays_noalloc: at most " + std::to_string(RT_NOALLOC_MAX_RAYS) + " rays per call (static scratch)");
    if (n == 0) return RT_OK;
    if (!dev_org_dir) return set_err(c, RT_ERR_INVALID, "null rays");
    if (!c->noalloc_scratch) return set_err(c, RT_ERR_DEVICE, "static scratch address unknown");
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned char* base = (unsigned char*)c->noalloc_scratch;   // the pointer from hipGetSymbolAddress, never the array's name
    unsigned int* ctl = (unsigned int*)base;
    float4* d_q = (float4*)(base + RT_NOALLOC_CTL_BYTES);
    float4* d_hits = (float4*)(base + RT_NOALLOC_CTL_BYTES + (size_t)RT_NOALLOC_MAX_RAYS * 32);
    const bool fast = trace_uses_shortcut(c, flags);
    const TraceCfg tc = trace_cfg(c, RT_TRACE_WAVES, false, fast);
    int rc = set_trace_lds(c, tc);  // (a function attribute, set once per context: no allocation)
    if (rc != RT_OK) return rc;
    // the queue and the control words are rewritten for every call (the array keeps the previous call's contents);
    // hit records are written for every ray of this call before rt_hitinfo_kernel reads them
    hipLaunchKernelGGL(rt_fill_queue_ctl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, dev_org_dir, n, d_q, ctl);
    if (fast)
        hipLaunchKernelGGL((rt_trace_kernel<false, true>), dim3(tc.grid), dim3(64 * RT_TRACE_WAVES), tc.lds_bytes, c->stream, c->sc, d_q, ctl, ctl + 1,
                           d_hits, flags, (int*)nullptr, (unsigned long long*)nullptr, tc.n_cached, tc.stack_entries);
    else
        hipLaunchKernelGGL(rt_trace_kernel<false>, dim3(tc.grid), dim3(64 * RT_TRACE_WAVES), tc.lds_bytes, c->stream, c->sc, d_q, ctl, ctl + 1, d_hits,
                           flags, (int*)nullptr, (unsigned long long*)nullptr, tc.n_cached, tc.stack_entries);
    hipLaunchKernelGGL(rt_hitinfo_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->sc, dev_org_dir, d_hits, n, dev_prim_id,
                       dev_t, dev_hit_pos, dev_normal3);
    HIP_TRY(c, hipGetLastError());
    if (sync) HIP_TRY(c, hipStreamSynchronize(c->stream));
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

#include "rt_comm_impl.h"
