// rt_pool_kernel.h -- the path-parallel fused kernel with the rays regrouped ACROSS the 16 waves of the block through
// the LDS (part of rt_core.hip; included there behind rt_render_paths_kernel).
//
// rt_render_paths_kernel lets every wave do everything: a lane whose traversal is over waits until 40 lanes of its
// wave have gathered, and the vertex pass that then serves them runs the shading code for the hits, the environment
// code for the misses and the camera code for the new paths one after the other, each for a part of its lanes.  Here
// the waves of a block have roles:
//  * TRAVERSAL waves (n_trav of the 16) only walk the tree.  A lane whose closest-hit search is over writes what is
//    left of its ray -- 64 bytes for a hit, 32 for a miss -- into a ring in the LDS and takes the next ready ray (80
//    bytes: origin, direction, 1/direction, the pruning constants, throughput, path identity) out of another; it
//    never waits for shading;
//  * VERTEX waves take 64 records of ONE kind at a time: 64 hits to shade (Material::reflect, main.cu:346-361), or
//    64 finished paths to close (environment, main.cu:374-387) and replace by 64 new camera rays (main.cu:178-217),
//    and write the rays that go on into the ready ring, root box tested and 1/d formed.
// A ray without a stack is all that ever changes hands (a search in progress stays in its lane), so the rings are
// small; the traversal stacks of the vertex waves are not needed and pay for them.  The arithmetic per path is that of
// rt_render_paths_kernel operation by operation (same helpers), a path still writes samples[path], and
// rt_sum_samples_kernel adds them: the sums and counters are bit-identical whatever wave served which ray.
//
// Rings: multi-producer / multi-consumer, four counters each {reserved tail, committed tail, reserved head,
// committed head}; the leader lane of a wave reserves a range by compare-and-swap, the lanes copy their records, the
// leader commits in reservation order.  Population: `population` rays exist per block (tokens + rays); a vertex wave
// turns tokens into camera rays while the launch has paths left.  With population <= 64 n_trav + cap_ready +
// min(cap_hit, cap_miss) the rings cannot dead-lock: in the state where nobody can move every traversal lane holds a
// finished ray, the ready ring and a finished ring are full -- more rays than exist.
#pragma once

#define RT_POOL_READY_F4 5
#define RT_POOL_HIT_F4 4
#define RT_POOL_MISS_F4 2
#define RT_POOL_CTL_F4 4  // 16 control words
// control words
#define RT_PC_READY 0
#define RT_PC_HIT 4
#define RT_PC_MISS 8
#define RT_PC_TOKENS 12
#define RT_PC_EXHAUSTED 13
#define RT_PC_ABORT 14
#ifndef RT_POOL_XCH_MIN
#define RT_POOL_XCH_MIN 16  // a traversal wave goes to the rings when this many of its lanes have nothing to walk
#endif
#ifndef RT_POOL_LOW_WATER
#define RT_POOL_LOW_WATER 128  // fewer ready rays than this: a vertex wave no longer waits for a full batch of 64
#endif
#define RT_POOL_SPIN_LIMIT (1u << 24)  // idle polls (of >= 512 cycles each) before a wave gives up: seconds, not a frame

#define RT_WG __HIP_MEMORY_SCOPE_WORKGROUP
#ifdef RT_POOL_CENSUS  // experiment builds: what the waves of each role do (leader lane counts, flushed with the RT_F_STATS counters)
#define RT_PCEN(i, v) do { if (STATS && lane == 0) pcen[i] += (unsigned long long)(v); } while (0)
#else
#define RT_PCEN(i, v) do {} while (0)
#endif

__device__ __forceinline__ unsigned pool_ld(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, RT_WG); }
__device__ __forceinline__ unsigned pool_ld_acq(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, RT_WG); }

// Call with the wave converged.  Reserves up to `want` slots to write; returns the first slot, `granted` may be 0.
__device__ __forceinline__ unsigned pool_push_reserve(unsigned* ctl, unsigned want, unsigned cap, unsigned& granted) {
    unsigned base = 0, n = 0;
    if ((threadIdx.x & 63) == 0) {
        for (;;) {
            const unsigned t = pool_ld(&ctl[0]);
            const unsigned h = pool_ld_acq(&ctl[3]);
            const unsigned room = cap - (t - h);
            n = min(want, room);
            if (n == 0) break;
            unsigned expect = t;
            if (__hip_atomic_compare_exchange_strong(&ctl[0], &expect, t + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, RT_WG)) {
                base = t;
                break;
            }
        }
    }
    granted = (unsigned)__builtin_amdgcn_readfirstlane((int)n);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)base);
}
__device__ __forceinline__ void pool_push_commit(unsigned* ctl, unsigned base, unsigned n) {
    if ((threadIdx.x & 63) == 0) {
        while (pool_ld(&ctl[1]) != base) __builtin_amdgcn_s_sleep(1);  // the reservations before this one commit first (they only copy)
        __hip_atomic_store(&ctl[1], base + n, __ATOMIC_RELEASE, RT_WG);
    }
}
// Reserves up to `want` committed records to read (none if fewer than min_take are there).
__device__ __forceinline__ unsigned pool_pop_reserve(unsigned* ctl, unsigned want, unsigned min_take, unsigned& granted) {
    unsigned base = 0, n = 0;
    if ((threadIdx.x & 63) == 0) {
        for (;;) {
            const unsigned h = pool_ld(&ctl[2]);
            const unsigned c = pool_ld_acq(&ctl[1]);
            const unsigned avail = c - h;
            n = min(avail, want);
            if (n == 0 || avail < min_take) { n = 0; break; }
            unsigned expect = h;
            if (__hip_atomic_compare_exchange_strong(&ctl[2], &expect, h + n, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, RT_WG)) {
                base = h;
                break;
            }
        }
    }
    granted = (unsigned)__builtin_amdgcn_readfirstlane((int)n);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)base);
}
__device__ __forceinline__ void pool_pop_commit(unsigned* ctl, unsigned base, unsigned n) {
    if ((threadIdx.x & 63) == 0) {
        while (pool_ld(&ctl[3]) != base) __builtin_amdgcn_s_sleep(1);
        __hip_atomic_store(&ctl[3], base + n, __ATOMIC_RELEASE, RT_WG);
    }
}
__device__ __forceinline__ unsigned pool_rank(unsigned long long m) {  // this lane's rank among the set bits below it
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ float4 pool_read(int f4_index) {  // ds_read_b128, kept apart from global loads (rt_device.h)
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef const volatile __attribute__((address_space(3))) v4f* lds_v4f;
    const v4f a = ((lds_v4f)rt_lds)[f4_index];
    return make_float4(a.x, a.y, a.z, a.w);
}
__device__ __forceinline__ void pool_write(int f4_index, float4 v) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef volatile __attribute__((address_space(3))) v4f* lds_v4f;
    ((lds_v4f)rt_lds)[f4_index] = (v4f){v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ float pool_fi(int i) { return __int_as_float(i); }
__device__ __forceinline__ float pool_fu(unsigned i) { return __uint_as_float(i); }

struct PoolGeom {  // where things are in the LDS (float4 units) and how big
    int ready_f4, hit_f4, miss_f4;
    unsigned cap_ready, cap_hit, cap_miss;  // powers of two
    unsigned* ctl;
    unsigned population, n_vertex_waves;
    unsigned* abort_flag;  // device memory: a wave that gave up says so
};

__device__ __forceinline__ bool pool_finished(const PoolGeom& g) {
    return pool_ld(&g.ctl[RT_PC_TOKENS]) == g.population && pool_ld(&g.ctl[RT_PC_EXHAUSTED]) == g.n_vertex_waves;
}
__device__ __forceinline__ bool pool_aborted(const PoolGeom& g) { return pool_ld(&g.ctl[RT_PC_ABORT]) != 0; }
__device__ __forceinline__ void pool_abort(const PoolGeom& g) {
    if ((threadIdx.x & 63) == 0) {
        __hip_atomic_store(&g.ctl[RT_PC_ABORT], 1u, __ATOMIC_RELAXED, RT_WG);
        atomicExch(g.abort_flag, 1u);
    }
}

// A ray that goes on: main.cu:221-236 (trav_begin's arithmetic) done here, by a full wave, and the result parked.
struct ReadyRay {
    f3 o, d, thr;
    float ri;
    unsigned path, pixel, sample;
    int bounce;
};
__device__ __forceinline__ void pool_write_ready(const Scene& sc, const PoolGeom& g, unsigned slot, const ReadyRay& r) {
    const f3 inv = mk(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    const float osum = fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z);
    const float slack = fmaxf(fabsf(r.o.x), fmaxf(fabsf(r.o.y), fabsf(r.o.z))) <= sc.cull_far ? 1e-5f * osum * osum : INFINITY;
    const float invD2 = 1.00001f / len2(r.d);
    const int cur0 = (sc.has_root && aabb_distance(r.o, inv, ld3(sc.root_min), ld3(sc.root_max)) <= 1e12f) ? sc.root_entry : RT_TRAV_DONE;
    const int b = g.ready_f4 + (int)(slot & (g.cap_ready - 1)) * RT_POOL_READY_F4;
    pool_write(b + 0, make_float4(r.o.x, r.o.y, r.o.z, r.d.x));
    pool_write(b + 1, make_float4(r.d.y, r.d.z, inv.x, inv.y));
    pool_write(b + 2, make_float4(inv.z, slack, invD2, r.thr.x));
    pool_write(b + 3, make_float4(r.thr.y, r.thr.z, r.ri, pool_fu(r.path)));
    pool_write(b + 4, make_float4(pool_fi(r.bounce), pool_fu(r.pixel), pool_fu(r.sample), pool_fi(cur0)));
}

// Vertex wave: the lanes in `mask` hold a ray that goes on; park them all (waits for room: the traversal lanes drain
// the ring).  False if the wave gave up.
__device__ __forceinline__ bool pool_push_ready(const Scene& sc, const PoolGeom& g, bool have, const ReadyRay& r) {
    unsigned long long mask = __ballot(have);
    unsigned spins = 0;
    while (mask) {
        unsigned granted;
        const unsigned base = pool_push_reserve(g.ctl + RT_PC_READY, (unsigned)__popcll(mask), g.cap_ready, granted);
        if (granted) {
            const unsigned rank = pool_rank(mask);
            const bool mine = have && ((mask >> (threadIdx.x & 63)) & 1) && rank < granted;
            if (mine) pool_write_ready(sc, g, base + rank, r);
            pool_push_commit(g.ctl + RT_PC_READY, base, granted);
            mask &= ~__ballot(mine);
            spins = 0;
        } else {
            if (pool_aborted(g)) return false;
            if (++spins > RT_POOL_SPIN_LIMIT) { pool_abort(g); return false; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    return true;
}

template <bool STATS, bool BIG, bool COOP>
__global__ __launch_bounds__(64 * RT_MEGA_WAVES) void rt_render_pool_kernel(Scene sc, Params P, float* __restrict__ samples,
                                                                             unsigned int* __restrict__ path_counter,
                                                                             unsigned long long* __restrict__ stats, int n_cached,
                                                                             int stack_entries, int chunk_k0, int chunk_S,
                                                                             unsigned int n_paths, int n_mats_lds, int n_trav,
                                                                             int cap_ready, int cap_hit, int cap_miss, int population) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * n_cached; i += 64 * RT_MEGA_WAVES) rt_lds[RT_LDS_SLOT(i >> 2, i & 3, n_cached)] = sc.items[i];
    const int stacks_f4 = (int)RT_LDS_CACHE_FLOAT4S(n_cached);
    const int mats_f4 = stacks_f4 + n_trav * stack_entries * 16;
    const int mats_lds = n_mats_lds > 0 ? mats_f4 : -1;
    for (int i = threadIdx.x; i < 3 * n_mats_lds; i += 64 * RT_MEGA_WAVES) rt_lds[mats_f4 + i] = sc.mats[i];
    PoolGeom g;
    g.ready_f4 = mats_f4 + 3 * n_mats_lds;
    g.hit_f4 = g.ready_f4 + cap_ready * RT_POOL_READY_F4;
    g.miss_f4 = g.hit_f4 + cap_hit * RT_POOL_HIT_F4;
    g.cap_ready = (unsigned)cap_ready;
    g.cap_hit = (unsigned)cap_hit;
    g.cap_miss = (unsigned)cap_miss;
    g.ctl = (unsigned*)(rt_lds + g.miss_f4 + cap_miss * RT_POOL_MISS_F4);
    g.population = (unsigned)population;
    g.n_vertex_waves = (unsigned)(RT_MEGA_WAVES - n_trav);
    g.abort_flag = path_counter + 2;
    if (threadIdx.x < 4 * RT_POOL_CTL_F4) g.ctl[threadIdx.x] = threadIdx.x == RT_PC_TOKENS ? (unsigned)population : 0u;
    __syncthreads();  // (the only barrier: from here on the waves of the block run on their own)
    Counters cn;
    cn.clear();
#ifdef RT_POOL_CENSUS
    unsigned long long pcen[16];
    for (int i = 0; i < 16; i++) pcen[i] = 0;
#endif

    if (wave < n_trav) {
        // ------------------------------------------------------------------ traversal wave
        int* stk = (int*)(rt_lds + stacks_f4) + (size_t)wave * stack_entries * 64 + lane;
        Trav t;
        t.oxy = t.ixy = (v2f){0.f, 0.f};
        t.oz = t.iz = 0.f;
        t.d = mk(0.f, 0.f, 0.f);
        t.axis_parallel = false;
        t.set_hit(0.0f, mk(0.f, 0.f, 0.f));
        t.cur = RT_LANE_IDLE;
        t.sp = 0;
        t.hitTri = -1;
        t.slack = 0.f;
        t.invD2 = 1.f;
        t.minD2 = 1e12f;
        t.subLim = 1e12f;
        f3 thr = mk(1.f, 1.f, 1.f);
        float ri = 1.0f;
        unsigned path = 0, pixel = 0, sample = 0;
        int bounce = 0;
        unsigned spins = 0;
        int cooldown = 0;
        for (;;) {
            const unsigned long long mDone = __ballot(t.cur == RT_TRAV_DONE), mIdle = __ballot(t.cur == RT_LANE_IDLE);
            unsigned long long mTrav = __ballot(t.cur >= 0);
            RT_PCEN(0, 1);
            RT_PCEN(1, __popcll(mTrav));
            RT_PCEN(2, __popcll(mDone));
            RT_PCEN(3, __popcll(mIdle));
            if ((__popcll(mDone) + __popcll(mIdle) >= RT_POOL_XCH_MIN && cooldown == 0) || mTrav == 0) {
                RT_PCEN(4, 1);
                bool moved = false;
                if (mDone) {  // what is left of a finished search goes to the vertex waves
                    const bool isHit = t.cur == RT_TRAV_DONE && t.hitTri >= 0, isMiss = t.cur == RT_TRAV_DONE && t.hitTri < 0;
                    const unsigned long long mh = __ballot(isHit);
                    if (mh) {
                        unsigned granted;
                        const unsigned base = pool_push_reserve(g.ctl + RT_PC_HIT, (unsigned)__popcll(mh), g.cap_hit, granted);
                        if (granted) {
                            const unsigned rank = pool_rank(mh);
                            if (isHit && rank < granted) {
                                const int b = g.hit_f4 + (int)((base + rank) & (g.cap_hit - 1)) * RT_POOL_HIT_F4;
                                pool_write(b + 0, make_float4(t.oxy.x, t.oxy.y, t.oz, t.d.x));
                                pool_write(b + 1, make_float4(t.d.y, t.d.z, t.hitT, pool_fi(t.hitTri)));
                                pool_write(b + 2, make_float4(thr.x, thr.y, thr.z, ri));
                                pool_write(b + 3, make_float4(pool_fu(path), pool_fi(bounce), pool_fu(pixel), pool_fu(sample)));
                                t.cur = RT_LANE_IDLE;
                            }
                            pool_push_commit(g.ctl + RT_PC_HIT, base, granted);
                            moved = true;
                        }
                    }
                    const unsigned long long mm = __ballot(isMiss);
                    if (mm) {
                        unsigned granted;
                        const unsigned base = pool_push_reserve(g.ctl + RT_PC_MISS, (unsigned)__popcll(mm), g.cap_miss, granted);
                        if (granted) {
                            const unsigned rank = pool_rank(mm);
                            if (isMiss && rank < granted) {
                                const int b = g.miss_f4 + (int)((base + rank) & (g.cap_miss - 1)) * RT_POOL_MISS_F4;
                                pool_write(b + 0, make_float4(t.d.x, t.d.y, t.d.z, thr.x));
                                pool_write(b + 1, make_float4(thr.y, thr.z, pool_fu(path), 0.f));
                                t.cur = RT_LANE_IDLE;
                            }
                            pool_push_commit(g.ctl + RT_PC_MISS, base, granted);
                            moved = true;
                        }
                    }
                }
                const bool idle = t.cur == RT_LANE_IDLE;
                const unsigned long long mi = __ballot(idle);
                if (mi) {  // and the lanes without a ray take the next ready ones
                    unsigned granted;
                    const unsigned base = pool_pop_reserve(g.ctl + RT_PC_READY, (unsigned)__popcll(mi), 1u, granted);
                    if (granted) {
                        const unsigned rank = pool_rank(mi);
                        if (idle && rank < granted) {
                            const int b = g.ready_f4 + (int)((base + rank) & (g.cap_ready - 1)) * RT_POOL_READY_F4;
                            const float4 r0 = pool_read(b + 0), r1 = pool_read(b + 1), r2 = pool_read(b + 2), r3 = pool_read(b + 3),
                                         r4 = pool_read(b + 4);
                            t.oxy = (v2f){r0.x, r0.y};
                            t.oz = r0.z;
                            t.d = mk(r0.w, r1.x, r1.y);
                            t.ixy = (v2f){r1.z, r1.w};
                            t.iz = r2.x;
                            t.axis_parallel = !(fabsf(t.ixy.x) < INFINITY && fabsf(t.ixy.y) < INFINITY && fabsf(t.iz) < INFINITY);
                            t.sp = 0;
                            t.hitTri = -1;
                            t.slack = r2.y;
                            t.invD2 = r2.z;
                            t.set_min(1e12f);
                            t.set_hit(0.0f, mk(0.f, 0.f, 0.f));
                            thr = mk(r2.w, r3.x, r3.y);
                            ri = r3.z;
                            path = __float_as_uint(r3.w);
                            bounce = __float_as_int(r4.x);
                            pixel = __float_as_uint(r4.y);
                            sample = __float_as_uint(r4.z);
                            t.cur = __float_as_int(r4.w);
                        }
                        pool_pop_commit(g.ctl + RT_PC_READY, base, granted);
                        moved = true;
                        RT_PCEN(5, granted);
                    }
                }
                mTrav = __ballot(t.cur >= 0);
                if (mTrav == 0) {
                    if (moved) { spins = 0; continue; }  // (e.g. rays whose root test failed: finished at once)
                    if (__ballot(t.cur == RT_TRAV_DONE) == 0 && pool_finished(g)) break;
                    if (pool_aborted(g)) break;
                    if (++spins > RT_POOL_SPIN_LIMIT) { pool_abort(g); break; }
                    RT_PCEN(6, 1);
                    __builtin_amdgcn_s_sleep(8);
                    continue;
                }
                spins = 0;
                cooldown = moved ? 0 : 2;  // nothing to take and no room: walk on for a while before asking again
            } else if (cooldown > 0) {
                cooldown--;
            }
            // one KIND of step per iteration, by vote (rt_render_paths_kernel)
            const bool at_inner = trav_at_inner(sc, t);
            const bool at_leaf = t.cur >= sc.n_inner;  // a triangle or RT_TRAV_EMPTY
            const unsigned long long mLeaf = __ballot(at_leaf);
            if (__popcll(mLeaf) >= RT_TRI_MIN || __ballot(at_inner) == 0) {
                RT_PCEN(7, 1);
                RT_PCEN(8, __popcll(mLeaf));
                if (at_leaf) {
                    trav_step_leaf<STATS, BIG>(sc, t, stk, cn);
#pragma unroll
                    for (int k = 1; k < RT_LEAF_STEPS; k++)
                        if (t.cur >= sc.n_inner) trav_step_leaf<STATS, BIG>(sc, t, stk, cn);
                }
            } else {
                if (at_inner) {
                    trav_step_inner<STATS, BIG>(sc, t, stk, cn, n_cached);
#pragma unroll
                    for (int k = 1; k < RT_INNER_STEPS; k++)
                        if (trav_at_inner(sc, t)) trav_step_inner<STATS, BIG>(sc, t, stk, cn, n_cached);
                }
            }
        }
    } else {
        // ------------------------------------------------------------------ vertex wave
        unsigned int batch_next = 0, batch_end = 0;  // wave-uniform path source, as in rt_render_paths_kernel
        bool paths_left = true, counted = false;
        unsigned spins = 0;
        Rng rng;
        rng.init(P.seed, 0, 0);
        for (;;) {
            const unsigned ah = pool_ld_acq(&g.ctl[RT_PC_HIT + 1]) - pool_ld(&g.ctl[RT_PC_HIT + 2]);
            const unsigned am = pool_ld_acq(&g.ctl[RT_PC_MISS + 1]) - pool_ld(&g.ctl[RT_PC_MISS + 2]);
            const unsigned tok = pool_ld(&g.ctl[RT_PC_TOKENS]);
            const unsigned backlog = pool_ld(&g.ctl[RT_PC_READY + 1]) - pool_ld(&g.ctl[RT_PC_READY + 2]);
            const bool can_new = paths_left || batch_next < batch_end;
            const unsigned closable = am + (can_new ? tok : 0u);
            int job = 0;  // 1 shade 64 hits, 2 close / open paths
            if ((int)ah >= 64) job = 1;
            else if ((int)closable >= 64) job = 2;
            else if ((int)backlog < RT_POOL_LOW_WATER) job = ((int)ah > 0 && ah >= closable) ? 1 : ((int)closable > 0 ? 2 : 0);
            job = __builtin_amdgcn_readfirstlane(job);
            if (job == 0) {
                if (!can_new && !counted) {
                    counted = true;
                    if (lane == 0) __hip_atomic_fetch_add(&g.ctl[RT_PC_EXHAUSTED], 1u, __ATOMIC_RELAXED, RT_WG);
                }
                if (pool_finished(g) || pool_aborted(g)) break;
                if (++spins > RT_POOL_SPIN_LIMIT) { pool_abort(g); break; }
                RT_PCEN(9, 1);
                __builtin_amdgcn_s_sleep(8);
                continue;
            }
            bool have = false;  // this lane ends the job with a ray that goes on
            ReadyRay nr;
            nr.o = nr.d = mk(0.f, 0.f, 0.f);
            nr.thr = mk(1.f, 1.f, 1.f);
            nr.ri = 1.0f;
            nr.path = nr.pixel = nr.sample = 0;
            nr.bounce = 0;
            unsigned freed = 0;  // paths that ended in this lane without a successor
            if (job == 1) {
                unsigned n;
                const unsigned base = pool_pop_reserve(g.ctl + RT_PC_HIT, 64u, 1u, n);
                if (n == 0) continue;  // another vertex wave was faster
                const bool mine = (unsigned)lane < n;
                float4 h0 = make_float4(0.f, 0.f, 0.f, 0.f), h1 = h0, h2 = h0, h3 = h0;
                if (mine) {
                    const int b = g.hit_f4 + (int)((base + (unsigned)lane) & (g.cap_hit - 1)) * RT_POOL_HIT_F4;
                    h0 = pool_read(b + 0); h1 = pool_read(b + 1); h2 = pool_read(b + 2); h3 = pool_read(b + 3);
                }
                pool_pop_commit(g.ctl + RT_PC_HIT, base, n);
                spins = 0;
                RT_PCEN(10, 1);
                RT_PCEN(11, n);
                // one iteration of the bounce loop after the closest-hit search, main.cu:346-361 (rt_render_paths_kernel's shading branch)
                PathState ray;
                ray.o = mk(h0.x, h0.y, h0.z);
                ray.d = mk(h0.w, h1.x, h1.y);
                ray.thr = mk(h2.x, h2.y, h2.z);
                ray.emission = mk(0.f, 0.f, 0.f);
                ray.ri = h2.w;
                ray.volDensity = 0.0f;
                ray.volColor = mk(1.f, 1.f, 1.f);
                const unsigned path = __float_as_uint(h3.x);
                int bounce = __float_as_int(h3.y);
                rng.pixel = __float_as_uint(h3.z);
                rng.sample = __float_as_uint(h3.w);
                f3 e = mk(0.f, 0.f, 0.f);
                bool alive = true;
                LightQuery lq;
                lq.pending = false;
                if (mine) {
                    const uint32_t bb = (RT_RNG_DIM_BOUNCE0 + RT_RNG_DIMS_PER_BOUNCE * (uint32_t)bounce) >> 2;
                    const f3 hitPos = ray.o + ray.d * h1.z;  // Trav::hit_pos
                    if (STATS) cn.hits++;
                    if (shade_triangle<STATS>(sc, ray, hitPos, __float_as_int(h1.w), rng, bb, cn, COOP ? &lq : nullptr, mats_lds)) {
                        e = ray.emission * ray.thr;
                        alive = false;
                    } else if (!COOP || !lq.pending) {
                        ray.d = normalize_keep(ray.d);  // main.cu:361
                    }
                }
                if (COOP && __ballot(lq.pending)) {
                    const float pdf_light = wave_light_pdf<STATS>(sc, lq, cn);
                    if (lq.pending) {
                        finish_type5(sc, ray, lq, pdf_light);
                        ray.d = normalize_keep(ray.d);
                    }
                }
                if (mine) {
                    bounce++;
                    if (bounce >= P.max_bounces) alive = false;  // path exhausted: contributes 0
                    if (alive) {
                        have = true;
                        nr.o = ray.o;
                        nr.d = ray.d;
                        nr.thr = ray.thr;
                        nr.ri = ray.ri;
                        nr.path = path;
                        nr.pixel = rng.pixel;
                        nr.sample = rng.sample;
                        nr.bounce = bounce;
                        if (STATS) cn.rays++;
                    } else {
                        float* o = samples + 3 * (size_t)path;
                        o[0] = e.x;
                        o[1] = e.y;
                        o[2] = e.z;
                        freed = 1;
                    }
                }
            } else {
                // finished paths (main.cu:374-387), each replaced by a new one (main.cu:178-217) while the launch has paths
                unsigned nm;
                const unsigned base = pool_pop_reserve(g.ctl + RT_PC_MISS, 64u, 1u, nm);
                const bool closing = (unsigned)lane < nm;
                if (nm) {
                    float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0;
                    if (closing) {
                        const int b = g.miss_f4 + (int)((base + (unsigned)lane) & (g.cap_miss - 1)) * RT_POOL_MISS_F4;
                        m0 = pool_read(b + 0); m1 = pool_read(b + 1);
                    }
                    pool_pop_commit(g.ctl + RT_PC_MISS, base, nm);
                    if (closing) {
                        const f3 d = mk(m0.x, m0.y, m0.z), thr = mk(m0.w, m1.x, m1.y);
                        f3 e;
                        if ((P.flags & RT_F_SKYBOX) && P.env_w > 0) {
                            const long long envI = env_index(d, P.env_w, P.env_h);
                            if (STATS) cn.texels++;
                            e = fetch_texel(sc, envI) * thr * P.bg_brightness;
                        } else {
                            e = ld3(P.bg) * thr * P.bg_brightness;
                        }
                        float* o = samples + 3 * (size_t)__float_as_uint(m1.z);
                        o[0] = e.x;
                        o[1] = e.y;
                        o[2] = e.z;
                    }
                }
                // tokens for the other lanes
                unsigned nt = 0;
                if (can_new && nm < 64u) {
                    if (lane == 0) {
                        for (;;) {
                            const unsigned have_tok = pool_ld(&g.ctl[RT_PC_TOKENS]);
                            nt = min(64u - nm, have_tok);
                            if (nt == 0) break;
                            unsigned expect = have_tok;
                            if (__hip_atomic_compare_exchange_strong(&g.ctl[RT_PC_TOKENS], &expect, have_tok - nt, __ATOMIC_RELAXED,
                                                                     __ATOMIC_RELAXED, RT_WG))
                                break;
                        }
                    }
                    nt = (unsigned)__builtin_amdgcn_readfirstlane((int)nt);
                }
                if (nm + nt == 0) continue;
                spins = 0;
                RT_PCEN(12, 1);
                RT_PCEN(13, nm);
                RT_PCEN(14, nt);
                const bool holder = (unsigned)lane < nm + nt;  // this lane may open a path
                bool needpath = holder && can_new;
                int x = 0, y = 0;
                unsigned int path = 0;
                unsigned long long m = __ballot(needpath);
                while (m) {  // (rt_render_paths_kernel's hand-out, batch by batch)
                    if (batch_next >= batch_end) {
                        if (!paths_left) break;
                        unsigned int b = 0;
                        if (lane == 0) b = atomicAdd(path_counter, (unsigned)RT_PATH_BATCH);
                        batch_next = __builtin_amdgcn_readfirstlane(b);
                        if (batch_next >= n_paths) { paths_left = false; batch_end = batch_next; break; }
                        batch_end = min(batch_next + (unsigned)RT_PATH_BATCH, n_paths);
                    }
                    const unsigned int avail = batch_end - batch_next;
                    const unsigned int rank = pool_rank(m);
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
                if (holder) {
                    if (needpath || !can_new) {
                        freed = 1;  // nothing left to open: the slot goes back
                    } else {
                        f3 ro, rd;
                        camera_ray(P, x, y, rng, ro, rd);
                        if (STATS) cn.paths++;
                        if (P.max_bounces > 0) {
                            have = true;
                            nr.o = ro;
                            nr.d = rd;
                            nr.path = path;
                            nr.pixel = rng.pixel;
                            nr.sample = rng.sample;
                            if (STATS) cn.rays++;
                        } else {  // the bounce loop never runs: the sample contributes 0
                            float* o = samples + 3 * (size_t)path;
                            o[0] = 0.f;
                            o[1] = 0.f;
                            o[2] = 0.f;
                            freed = 1;
                        }
                    }
                }
            }
            const unsigned nfreed = (unsigned)__popcll(__ballot(freed != 0));
            if (nfreed && lane == 0) __hip_atomic_fetch_add(&g.ctl[RT_PC_TOKENS], nfreed, __ATOMIC_RELEASE, RT_WG);
            if (!pool_push_ready(sc, g, have, nr)) break;
        }
    }

    if (STATS) {
        uint32_t v[12] = {cn.paths, cn.rays, cn.inner, cn.leaf, cn.tri, cn.hits, cn.texels, cn.lights, cn.dev_inner, cn.dev_cached, cn.dev_tri, cn.dev_lights};
#pragma unroll
        for (int i = 0; i < 12; i++) {
            unsigned long long s = v[i];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0 && s) atomicAdd(&stats[i], s);
        }
#ifdef RT_POOL_CENSUS
        if (lane == 0)
            for (int i = 0; i < 16; i++)
                if (pcen[i]) atomicAdd(&stats[16 + i], pcen[i]);
#endif
    }
}
