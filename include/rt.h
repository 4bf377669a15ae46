/*
 * rt.h -- C-ABI of the MI355X (gfx950) path-tracing core.
 *
 * Drop-in boundary for the render call of EricWittrock/Ray-Tracer: the kernel
 * launch `render2<<<grid,block>>>(...)` (reference src/main.cu:472-482), the
 * uploads that feed it (src/main.cu:418-460) and the readback after it
 * (src/main.cu:500-501).  Every array crosses this boundary in the
 * reference's own layout (25-float triangle records, 40-byte BVH nodes,
 * 48-byte materials, one float RGB texture pool, int light list); the
 * library converts them to its gfx950 layouts at upload.
 *
 * Plain C: POD structs, pointers and sizes only.  Every function returns
 * RT_OK (0) or a negative rt_status; nothing exits or throws across the ABI
 * (the reference prints and continues, src/main.cu:484-492).
 *
 * A context is bound to one HIP device and is not thread-safe.  There is no
 * CPU fallback: without a usable HIP device rt_create fails.
 */
#ifndef RT_H_
#define RT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ABI version --------------------------------------------------------
 * rt_render_stats and rt_layout_info grow at the END from round to round.  A caller says how large ITS struct is and
 * the library writes at most that many bytes: rt_render_v2 / rt_comm_render_v2 / rt_layout_get_v2 take the size,
 * and a C caller that includes this header gets them through the function-like macros below, with sizeof of the struct
 * it was compiled with.  The plain symbols rt_render / rt_comm_render / rt_layout_get stay exported for binaries built
 * against an older rt.h and fill only the fields that header had (RT_*_V1_SIZE / _V2_SIZE bytes).
 * rt_abi_version() = RT_ABI_VERSION of the library that is loaded. */
#define RT_ABI_VERSION 3
#define RT_RENDER_STATS_V1_SIZE 88 /* round 1: kernel_ms .. trace_launches */
#define RT_RENDER_STATS_V2_SIZE 128 /* round 2: + dev_inner_visits .. shortcut_rewalks (what rt_comm_render was born with) */
#define RT_LAYOUT_INFO_V1_SIZE 28  /* round 1: n_inner .. cull_far */
int rt_abi_version(void);

/* ---- status codes ------------------------------------------------------- */
typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument / malformed scene              */
    RT_ERR_DEVICE = -2,    /* HIP error (message in rt_last_error)        */
    RT_ERR_NO_SCENE = -3,  /* render/trace before rt_upload_scene         */
    RT_ERR_LIMIT = -4,     /* scene exceeds a built-in limit (BVH depth…) */
    RT_ERR_NOMEM = -5
} rt_status;

/* ---- reference layouts -------------------------------------------------- */

/* BVH::BVHNode, reference src/BVHcreator.h:23-34 (40 bytes, AoS).
 * child_a == -1 marks a leaf; tri_offs / tri_len are FLOAT offsets/lengths
 * into the triangle array (25 floats per triangle). */
typedef struct rt_bvh_node {
    float bbox_min[3];
    float bbox_max[3];
    int32_t child_a;
    int32_t child_b;
    int32_t tri_offs;
    int32_t tri_len;
} rt_bvh_node;

/* class Material, reference src/material.h:9-22 (48 bytes with the natural
 * padding of the original: char type at 0, int color at 4, ... u16 width at
 * 32, u16 height at 34, pointer at 40).  The pointer member is never
 * dereferenced by the reference kernel and is ignored here. */
typedef struct rt_material {
    int8_t type;
    int8_t pad0_[3];
    int32_t color;          /* 0xRRGGBB, vec3.h:111-123 */
    float p1, p2, p3;
    int32_t image_offset[3]; /* float offsets into the texture pool, -1 = none */
    uint16_t image_width;
    uint16_t image_height;
    int32_t pad1_;
    uint64_t textures_unused;
} rt_material;

/* class SceneConfigs, reference src/sceneConfigs.h:4-18 (80 bytes). */
typedef struct rt_camera {
    int32_t width;
    int32_t height;
    int32_t num_samples;
    float focal_length;
    float aperture;
    float focus_distance;
    float pos[3];
    float rot[3];
    float vel[3];
    float ang_vel[3];
    int32_t env_width;   /* 0 = no environment map (constant background) */
    int32_t env_height;
} rt_camera;

#define RT_TRI_FLOATS 25 /* model.h:135-162: v0 v1 v2 n0 n1 n2 uv0 uv1 uv2 matId+0.001 */

/* The argument list of render2 (main.cu:155) as host arrays.
 * `tris` starts with the optional sphere prefix of sceneParser.cpp:443-453:
 * when n_spheres > 0 the first 1 + 5*n_spheres floats are
 * [N+0.0001, (x,y,z,r,mat+0.0001) x N] and triangle records follow; BVH
 * tri_offs and light_tris are relative to the first triangle record
 * (main.cu:161-168).  n_tri_floats counts the triangle floats only. */
typedef struct rt_scene_desc {
    const float* tris;
    size_t n_tri_floats;
    int32_t n_spheres;
    const rt_bvh_node* nodes;
    int32_t n_nodes;
    const rt_material* mats;
    size_t n_mats;
    const float* textures; /* env map (if any) at offset 0, sceneParser.cpp:482-495 */
    size_t n_tex_floats;
    const int32_t* light_tris; /* float offsets of importance-sampled triangles */
    int32_t n_light_tris;
} rt_scene_desc;

/* ---- render parameters -------------------------------------------------- */

/* Runtime versions of the compile-time switches of reference src/config.h. */
enum {
    RT_F_SKYBOX = 1 << 0,       /* ENABLE_SKYBOX        config.h:5  */
    RT_F_ANTIALIAS = 1 << 1,    /* ENABLE_ANTIALIASING  config.h:11 */
    RT_F_DEFOCUS = 1 << 2,      /* ENABLE_DEFOCUS_BLUR  config.h:10 */
    RT_F_BVH = 1 << 3,          /* ENABLE_BVH           config.h:16 (off = brute force, main.cu:285-301) */
    RT_F_SPHERES = 1 << 4,      /* ENABLE_SPHERES       config.h:13 */
    RT_F_VOLUME = 1 << 5,       /* ENABLE_VOLUME_SCATTERING config.h:9: homogeneous scattering inside type-6 volumes
                                   (checkScatteredVolume, main.cu:127-141, 338, 350).  Fused kernel only. */
    RT_F_RAYMARCH = 1 << 6,     /* ENABLE_RAY_MARCHING  config.h:14: rays bent by a point mass at the origin (marchRay,
                                   main.cu:143-153, 225-227, 364-372); max_bounces is the step count
                                   (RAY_MARCHING_MAX_STEPS, config.h:15).  Fused kernel only. */
    RT_F_STATS = 1 << 8,        /* count rays / node visits / triangle tests */
    RT_F_MEGAKERNEL = 1 << 9,   /* force the fused persistent kernel (path-parallel: lanes take (pixel, sample) paths
                                   one by one, per-path radiance summed per pixel in sample order afterwards) */
    RT_F_LOCKSTEP = 1 << 12,    /* fused kernel in lock step instead: one lane owns a pixel for the whole call, no
                                   per-path radiance buffer (slower; bit-identical sums) */
    RT_F_WAVEFRONT = 1 << 11,   /* force the wavefront pipeline (logic kernel + persistent trace kernel);
                                   with neither bit the library picks (currently the fused kernel) */
    RT_F_POOL = 1 << 13,        /* fused path-parallel kernel with the rays regrouped across the 16 waves of a block: some
                                   waves only walk the tree, the others shade 64 hits / open 64 paths at a time, rays
                                   change hands through rings in the LDS (rt_pool_kernel.h).  Experimental (slower on the
                                   bundled scenes, bit-identical sums); plain triangle scenes only -- with RT_F_SPHERES
                                   and spheres, RT_F_VOLUME, RT_F_RAYMARCH or without RT_F_BVH the ordinary fused
                                   kernel runs.  Also switched on by the environment variable RT_POOL=1. */
    RT_F_REFWALK = 1 << 14,     /* every closest-hit search walks the reference's own tree.  Without it the fused
                                   path-parallel kernel searches a surface-area tree of the library's own over the
                                   reference's leaves first (the SHORTCUT TREE, rt_device.h) and keeps the result only
                                   where a per-ray certificate shows that the reference's walk returns the same
                                   triangle; other rays walk the reference's tree after all.  Same sums bit for bit,
                                   fewer node visits.  Implied by RT_F_STATS (whose counters describe the reference's
                                   walk), RT_F_VOLUME, RT_F_RAYMARCH, brute force and the other schedules. */
    RT_F_SHORTCUT_STATS = 1 << 15, /* with RT_F_STATS: keep the shortcut tree; the dev_* counters and shortcut_rewalks then
                                   say what that walk does, inner_visits / leaf_visits / tri_tests are not the reference's */
    RT_F_PROFILE = 1 << 10,     /* HIP-event timing of every launch of the dominant kernel (rt_render_stats.trace_ms) */
    RT_F_DEFAULT = RT_F_SKYBOX | RT_F_ANTIALIAS | RT_F_DEFOCUS | RT_F_BVH
};

typedef struct rt_render_params {
    rt_camera cam;        /* cam.num_samples is NOT used here; see sample_count */
    int32_t max_bounces;  /* MAX_BOUNCES, config.h:3 (default 6) */
    uint32_t flags;       /* RT_F_* */
    uint32_t seed;        /* reference: 37811 (main.cu:171) */
    int32_t sample_begin; /* first sample index rendered by this call */
    int32_t sample_count; /* samples added to the accumulators by this call */
    /* Tile-row partition: image rows are cut into bands of band_rows rows;
     * this call renders band b iff b % band_stride == band_index.  The
     * accumulator then holds only those rows, packed band after band.
     * band_stride <= 1 renders the whole image. */
    int32_t band_rows;
    int32_t band_stride;
    int32_t band_index;
    float background[3];       /* BACKGROUND_COLOR config.h:7 (1,1,0.05) */
    float background_brightness; /* BACKGROUND_BRIGHTNESS config.h:6 (0.5) */
} rt_render_params;

typedef struct rt_render_stats {
    double kernel_ms;       /* HIP-event time of the render kernel(s) of this call */
    uint64_t paths;         /* camera samples started */
    uint64_t rays;          /* closest-hit queries (primary + continued bounces), main.cu:220 */
    uint64_t inner_visits;  /* inner nodes popped, main.cu:265-279 */
    uint64_t leaf_visits;   /* leaves popped, main.cu:246-264 */
    uint64_t tri_tests;     /* rayTriangleIntersect calls, main.cu:256 */
    uint64_t hits;          /* bounces that hit a triangle or sphere */
    uint64_t texel_fetches; /* material texel fetches + env fetches */
    uint64_t light_tests;   /* ImportanceTri::pdf_value calls, material.h:315-323 */
    double trace_ms;        /* RT_F_PROFILE: summed HIP-event time of the launches of rt_trace_kernel (wavefront)
                               or rt_render_paths_kernel (fused) */
    uint32_t iterations;    /* wavefront iterations (logic + trace launch pairs) of this call */
    uint32_t trace_launches; /* launches covered by trace_ms */
    /* What the DEVICE really did (the counters above describe the reference's walk: a leaf sub-tree's root is
     * credited with its leaf visit and all the leaf's triangle tests, rt_device.h): */
    uint64_t dev_inner_visits;     /* inner items visited: reference nodes and sub-tree nodes, 64 B each */
    uint64_t dev_lds_inner_visits; /* of those, served by the LDS node cache instead of L2 / HBM */
    uint64_t dev_tri_tests;        /* triangle tests really run, 48 B each */
    uint64_t dev_light_tests;      /* light-triangle pdf tests really run (after the culling of light groups), 64 B each */
    uint64_t shortcut_rewalks;     /* searches whose shortcut-tree result was not certified and that walked the reference's tree again */
    /* round 3 */
    uint64_t dev_wide_visits;      /* visits of wide (four-child) shortcut nodes: 128 B each, counted as TWO items in dev_inner_visits /
                                      dev_lds_inner_visits */
    uint64_t dev_light_loops;      /* type-5 bounces whose sum over the lights the light tree did not serve (rays it is not made for,
                                      three or more nonzero terms, stack room): summed by the loop over every light instead */
} rt_render_stats;

enum { RT_READ_RAW_SUM = 0, RT_READ_AVERAGE = 1, RT_READ_TONEMAPPED = 2 };

typedef struct rt_context rt_context;

/* ---- entry points ------------------------------------------------------- */

/* Replaces the device set-up of main(): src/main.cu:418-420.  device_id < 0
 * uses the current HIP device. */
int rt_create(rt_context** out, int device_id);
void rt_destroy(rt_context* ctx);
const char* rt_last_error(const rt_context* ctx);

/* Replaces the six cudaMalloc+cudaMemcpy uploads, src/main.cu:418-460.
 * Copies and converts; the caller keeps ownership of every host array. May
 * be called again to replace the scene. */
int rt_upload_scene(rt_context* ctx, const rt_scene_desc* scene);

/* Replaces the launch + sync of render2, src/main.cu:463-492.  Adds
 * params->sample_count samples per pixel to the raw-sum accumulators
 * (re-entrant: progressive batches and sample slices).  Blocking.
 * stats may be NULL. */
int rt_render(rt_context* ctx, const rt_render_params* params, rt_render_stats* stats); /* round 1's symbol: fills RT_RENDER_STATS_V1_SIZE bytes */
int rt_render_v2(rt_context* ctx, const rt_render_params* params, rt_render_stats* stats, size_t stats_size);
#ifndef RT_NO_SIZED_MACROS
#define rt_render(ctx, params, stats) rt_render_v2((ctx), (params), (stats), sizeof(rt_render_stats))
#endif

/* Replaces the D2H copy src/main.cu:500-501 (and, with RT_READ_TONEMAPPED,
 * the `/numSamples` + toneMap of main.cu:392-406 that the reference does
 * inside the kernel).  n_floats must equal rows*width*3 of the last render. */
int rt_readback(rt_context* ctx, float* rgb, size_t n_floats, int mode);

/* The image the reference writes as PNG: average, toneMap (main.cu:392-406) and the 8-bit conversion of
 * Texture::saveImgData (texture.cpp:53-56: static_cast<unsigned char>(v * 255), no clamp) done on the device; a
 * quarter of the bytes of rt_readback cross PCIe.  n_bytes must equal rows*width*3 of the last render. */
int rt_readback_rgb8(rt_context* ctx, unsigned char* rgb8, size_t n_bytes);

/* Zero the accumulators and the sample counter. */
int rt_reset_accum(rt_context* ctx);

/* Give back the device memory the context keeps between calls for its own use: the per-path radiance buffer of the
 * fused kernel (up to 8 GiB for a large frame at many samples per call), the wavefront pipeline's ray queues and the
 * readback scratch.  They grow on demand and are otherwise held until rt_destroy; a long-lived context that has
 * rendered one big frame calls this before it idles.  Scene, accumulators and sample count are untouched; the next
 * render allocates again.  No reference counterpart (the reference frees everything at the end of main, main.cu:508-513). */
int rt_release_scratch(rt_context* ctx);

/* Use caller-owned DEVICE memory (n_floats >= rows*width*3) as the raw-sum
 * accumulator, so that the host layer can run its RCCL reduce/gather on it
 * directly.  dev_ptr == NULL returns to the internal buffer.  No reference
 * counterpart (the reference is single-GPU). */
int rt_bind_accum(rt_context* ctx, float* dev_ptr, size_t n_floats);

/* Tell the context how many samples the bound accumulator now holds (after
 * an external sum-reduce of sample slices). */
int rt_set_sample_count(rt_context* ctx, int64_t samples);

/* Launch on this hipStream_t instead of the context's own stream. */
int rt_set_stream(rt_context* ctx, void* hip_stream);

/* Resolve the accumulators into device memory (dev_out: rows*width*3 floats)
 * with the given RT_READ_* mode; no host copy. */
int rt_resolve_device(rt_context* ctx, float* dev_out, size_t n_floats, int mode);

/* Test hook: closest hit of explicit rays through the traversal of
 * main.cu:221-301.  org_dir = n x 6 floats (origin, direction) on the host.
 * prim_id = hitTriIndex/25 or -1; t = |hitPos - origin| (sqrt of the
 * reference's closest-hit metric, main.cu:257), hit_pos = 3 floats,
 * normal3 = interpolated shading normal (material.h:51-52).  Output pointers
 * may be NULL.  counts (may be NULL) = per-ray {inner, leaf, tri} visit
 * counts, 3 x int32 per ray. */
int rt_trace_rays(rt_context* ctx, const float* org_dir, size_t n, uint32_t flags,
                  int32_t* prim_id, float* t, float* hit_pos, float* normal3, int32_t* counts);

/* rt_trace_rays for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE (its main() allocates freely,
 * src/main.cu:418-460): it is for tight serving loops, harnesses that forbid hipMalloc and code meant to run under HIP
 * graph capture (the call only launches: the kernels' attributes are set at rt_upload_scene; capture itself is untested).
 * The call makes no hipMalloc / hipFree and no host copy: rays in and results out are DEVICE pointers (org_dir n x 6 floats; prim_id, t, hit_pos n x 3, normal3 n x 3 as in rt_trace_rays, each may be NULL), and the
 * ray queue, hit records and control words of the trace kernel live in a file-scope device array of the library
 * (12 MiB per GPU, held until the process ends).  n > RT_TRACE_NOALLOC_MAX_RAYS is refused with RT_ERR_LIMIT.
 * Work is enqueued on the context's stream (rt_set_stream); sync != 0 waits for it before returning.
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two contexts or two streams) race. */
#define RT_TRACE_NOALLOC_MAX_RAYS (1 << 18)
int rt_trace_rays_noalloc(rt_context* ctx, const float* dev_org_dir, size_t n, uint32_t flags, int32_t* dev_prim_id,
                          float* dev_t, float* dev_hit_pos, float* dev_normal3, int sync);

/* ---- several GPUs of one node from one process ----------------------------------------------------------
 * NO COUNTERPART IN THE REFERENCE (one GPU, one launch, src/main.cu:472-482).  BASELINE.json asks for the image to be
 * partitioned over the 8 GPUs of a node by tile rows (RCCL gather) or by sample slices (RCCL sum-reduce) of the
 * float accumulators.  An rt_comm owns one rt_context per device, renders on all of them at once (one host thread
 * each) and combines on rank 0 with ONE collective per call: grouped ncclSend / ncclRecv of the band-packed rows + a
 * de-interleave kernel, or ncclReduce(sum).  librccl is loaded at run time.  RT_COMM_LOCAL moves the same data
 * with hipMemcpyPeerAsync and sums in rank order ((a0 + a1) + a2 ...): used when librccl is missing and by the tests
 * on a one-GPU box (several ranks on ONE device, which RCCL refuses).  Rows: bit-identical to one GPU.  Samples: the
 * difference to one GPU is the order of the float additions.  rt_comm_render accumulates into the root's image like
 * rt_render does (rt_comm_reset starts over); band_rows <= 0 means 16. */
enum { RT_PART_ROWS = 0, RT_PART_SAMPLES = 1 };
enum { RT_COMM_RCCL = 0, RT_COMM_LOCAL = 1 };
typedef struct rt_comm rt_comm;
int rt_comm_create(rt_comm** out, const int* device_ids, int n_devices, int transport);
void rt_comm_destroy(rt_comm* comm);
const char* rt_comm_last_error(const rt_comm* comm);
int rt_comm_size(const rt_comm* comm);
int rt_comm_transport(const rt_comm* comm);               /* the transport in use (RT_COMM_LOCAL if librccl could not be loaded) */
rt_context* rt_comm_context(rt_comm* comm, int rank);     /* owned by the comm */
int rt_comm_upload_scene(rt_comm* comm, const rt_scene_desc* scene);   /* replicated on every device */
int rt_comm_render(rt_comm* comm, const rt_render_params* params, int partition, int band_rows, rt_render_stats* stats); /* round 2's symbol: fills RT_RENDER_STATS_V2_SIZE bytes */
int rt_comm_render_v2(rt_comm* comm, const rt_render_params* params, int partition, int band_rows, rt_render_stats* stats, size_t stats_size);
#ifndef RT_NO_SIZED_MACROS
#define rt_comm_render(comm, params, partition, band_rows, stats) rt_comm_render_v2((comm), (params), (partition), (band_rows), (stats), sizeof(rt_render_stats))
#endif
int rt_comm_reduce_sum(rt_comm* comm);    /* the combine steps alone (rt_comm_render calls them) */
int rt_comm_gather_rows(rt_comm* comm);
int rt_comm_reset(rt_comm* comm);
int rt_comm_readback(rt_comm* comm, float* rgb, size_t n_floats, int mode);   /* the root's full image, RT_READ_* */
/* Combine steps of this comm that went through RCCL so far (ncclReduce calls; grouped ncclSend / ncclRecv gathers).  With
 * the RCCL transport a comm of ONE rank still runs both: the reduce in place, the gather as a send to itself. */
int rt_comm_rccl_calls(const rt_comm* comm, uint64_t* reduces, uint64_t* sendrecvs);
/* Host-side statements of the two combine steps, no device needed (what the CPU tests check the index arithmetic
 * and the summation order against): rows of `rank` under the band interleave (returns their number; rows may be
 * NULL), the de-interleave of band-packed buffers, the rank-ordered sum. */
int rt_band_rows_of_rank(int height, int band_rows, int n_ranks, int rank, int32_t* rows);
int rt_host_gather_rows(const float* const* band_packed, int n_ranks, int width, int height, int band_rows, float* full);
int rt_host_reduce_sum(const float* const* accums, int n_ranks, size_t n_floats, float* out);

/* Inspection hook, no device needed: the layout rt_upload_scene copies to the
 * GPU, built in host memory.  The reference has no counterpart (it uploads
 * its arrays as they are, main.cu:418-460); this exists so that the item
 * stream with its leaf sub-trees can be checked on a
 * machine without a GPU.  items = 16 floats per item (layout in
 * csrc/rt_device.h): items [0, n_inner) are inner nodes, the triangles
 * follow; places[id] = place of the reference's triangle id among them.
 * Pointers stay valid until rt_layout_destroy. */
typedef struct rt_layout rt_layout;
typedef struct rt_layout_info {
    int32_t n_inner;     /* inner items: the reference's nodes and the sub-tree nodes */
    int32_t n_sub;       /* sub-tree nodes among them */
    int32_t n_tris;
    int32_t root_entry;  /* item index of the root, -1 without nodes */
    int32_t depth;       /* of the whole structure = traversal stack entries */
    float box_pad;       /* padding of every box the library adds */
    float cull_far;      /* padded boxes cull only for rays starting within this of the origin */
    int32_t n_short;     /* shortcut-tree nodes among the inner items (0: none built) */
    int32_t short_root;  /* item index of the shortcut tree's root, -1 without one */
    int32_t short_base;  /* first item of the window the shortcut walk caches in LDS */
    int32_t short_depth; /* traversal stack entries of the shortcut walk */
    int32_t short_single; /* the shortcut tree stands over single triangles (scenes too small for sub-trees) */
    int32_t short_cut;    /* huge scenes: levels below the reference's leaves at which the shortcut tree's leaves stand */
    /* round 3 */
    int32_t n_wide;       /* the shortcut tree's nodes have up to FOUR children and take two items each, items [short_base,
                             short_base + 2 n_wide) (layout in csrc/rt_device.h); 0 = two-child shortcut nodes; n_short == n_wide then */
    int32_t wide_need;    /* stack entries the worst path of the wide tree needs */
    int32_t wide_sp_max;  /* a wide visit with more entries than this on the stack gives up (the ray walks the reference's tree) */
    int32_t light_root;   /* item of the light tree's root (a wide node), -1 without one: the all-hits search that stands for
                             reflectType5's loop over the lights (material.h:315-323) */
    int32_t n_light_items; /* its leaves: light items [n_inner + n_tris + 1, + n_light_items), one per importance-sampled triangle */
    int32_t light_need;   /* stack entries its worst path needs */
} rt_layout_info;
int rt_layout_create(const rt_scene_desc* scene, rt_layout** out, char* err_buf, size_t err_len);
int rt_layout_get(const rt_layout* layout, rt_layout_info* info, const float** items, const int32_t** places); /* round 1's symbol: fills RT_LAYOUT_INFO_V1_SIZE bytes */
int rt_layout_get_v2(const rt_layout* layout, rt_layout_info* info, size_t info_size, const float** items, const int32_t** places);
#ifndef RT_NO_SIZED_MACROS
#define rt_layout_get(layout, info, items, places) rt_layout_get_v2((layout), (info), sizeof(rt_layout_info), (items), (places))
#endif
void rt_layout_destroy(rt_layout* layout);

/* Test hook: evaluate device-side primitives on n host records.
 * Each op reads `in_stride` floats and writes `out_stride` floats per record
 * (layouts in DESIGN.md "device unit ops"). */
enum {
    RT_OP_TRI_INTERSECT = 1, /* main.cu:41-70      in 15: o d v0 v1 v2      out 4: hit pos           */
    RT_OP_AABB_DIST = 2,     /* main.cu:73-117     in 12: o d min max       out 1                    */
    RT_OP_SPHERE = 3,        /* sphere.h:30-46     in 10: o d c r           out 7: hit pos normal    */
    RT_OP_FLOOR = 4,         /* floor.h:8-21       in 7:  o d elevation     out 7: hit pos normal    */
    RT_OP_ITRI_PDF = 5,      /* importanceTri.h:67 in 15: o d p1 p2 p3      out 1                    */
    RT_OP_ITRI_SAMPLE = 6,   /* importanceTri.h:79 in 14: o r1 r2 p1 p2 p3  out 3                    */
    RT_OP_ATTRACTOR_PDF = 7, /* rayAttractor.h:39  in 15: o d pos u v       out 1                    */
    RT_OP_ATTRACTOR_SAMPLE = 8, /* rayAttractor.h:50 in 14: o r1 r2 pos u v out 3                    */
    RT_OP_TONEMAP = 9,       /* main.cu:23-38      in 3                     out 3                    */
    RT_OP_RNG = 10,          /* counter RNG        in 4 (as u32 bits: seed pixel sample dim) out 4 (u32 bits) */
    RT_OP_ENV_INDEX = 11,    /* main.cu:376-382    in 5: d envW envH        out 1 (int bits)         */
    RT_OP_CAMERA_RAY = 12,   /* main.cu:173-217    in 3 (u32 bits: x y sample), camera/flags/seed from params: out 6 */
    RT_OP_PMATH = 13         /* portable libm      in 3: fn(int bits) x y   out 2
                                fn 0 sincos(x) -> (sin, cos); 1 asin(x); 2 acos(x); 3 atan2(x, y); 4 log(x) */
};
/* params is only read by RT_OP_CAMERA_RAY (may be NULL otherwise). */
int rt_debug_op(rt_context* ctx, int op, const float* in, size_t n, float* out,
                const rt_render_params* params);

/* Test hook: one Material::reflect (material.h:41-106) per record against the
 * uploaded scene.  in = 16 floats per record:
 *   o(3) d(3) throughput(3) refractiveIndex hitPos(3) triId(int bits)
 *   rngKey: pixel(u32 bits) sample(u32 bits)   [bounce = 0, params->seed]
 * out = 16 floats: o(3) d(3, before the kernel's renormalise main.cu:361)
 *   throughput(3) emission(3) refractiveIndex volumeDensity terminated(int bits) pad */
int rt_shade_hits(rt_context* ctx, const float* in, size_t n, uint32_t seed, float* out);

/* ---- counter-based RNG (replaces cuRAND XORWOW, main.cu:170-171) ---------
 * draw(seed, pixel, sample, dim) = Philox4x32-10(
 *     counter = {pixel, sample, dim >> 2, RT_RNG_TAG}, key = {seed, RT_RNG_KEY1})[dim & 3]
 * pixel = x + y*width in full-image coordinates (main.cu:171), sample = absolute
 * sample index.  uniform = ((draw >> 8) + 1) * 2^-24 in (0,1] like curand_uniform.
 * normal  = sqrtf(-2 logf(u_a)) * cosf(RT_TWO_PI_F * u_b) from two consecutive dims.
 * Dimensions: 0 animation time (main.cu:179), 1-2 anti-alias jitter (:200-201),
 * 4-5 lens theta, r (:120-121); bounce j owns dims 8+8j .. 15+8j, consumed in
 * the reference's call order inside Material::reflect / Sphere::hitSphere;
 * the volume check of bounce j (checkScatteredVolume, main.cu:127-141, RT_F_VOLUME)
 * owns dims RT_RNG_DIM_VOLUME0 + 4j .. +2: free-flight draw, Vec3::randSphere r1, r2. */
#define RT_RNG_TAG 0x52544131u
#define RT_RNG_KEY1 0x9E3779B9u
#define RT_RNG_DIM_CAMERA 0
#define RT_RNG_DIM_LENS 4
#define RT_RNG_DIM_BOUNCE0 8
#define RT_RNG_DIMS_PER_BOUNCE 8
#define RT_RNG_DIM_VOLUME0 0x80000000u
#define RT_RNG_DIMS_PER_VOLUME 4

#ifdef __cplusplus
}
#endif
#endif /* RT_H_ */
