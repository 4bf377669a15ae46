// rtcli -- the caller of the render core: what main() is in the reference (src/main.cu:410-517):
//   parse scene -> assemble arrays (+BVH) -> rt_upload_scene -> rt_render -> rt_readback -> image file.
// Unlike the reference (hard-coded SCENE_CONFIG_PATH / OUTPUT_IMAGE_PATH macros, config.h:20-21, argv ignored)
// everything is a command-line option.
//
//   rtcli --scene scenes/scene.txt --asset-root assets --out out.png [--spp N] [--bounces N] [--size WxH]
//         [--env file.hdr] [--hdr] [--seed N] [--spheres] [--no-bvh] [--wavefront|--lockstep] [--volume] [--raymarch] [--fast-bvh] [--ref-walk] [--stats]
//         [--devices 0-7 | 0,2,5] [--partition rows|samples]     several GPUs of the node (rt_comm_*: RCCL gather / sum-reduce)
//         [--json [--steps K] [--warmup W]]    with --devices: time K frames (render on every device + the collective) and
//                                              print ONE JSON line with bench.py's fields instead of writing an image
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt.h"
#include "rt_scene.h"

static int usage() {
    fprintf(stderr, "usage: rtcli --scene FILE [--asset-root DIR] [--out FILE.png|.ppm|.hdr|.pfm] [--spp N] [--bounces N]\n"
                    "             [--size WxH] [--env FILE.hdr] [--hdr] [--seed N] [--spheres] [--no-bvh] [--wavefront|--lockstep] [--volume] [--raymarch] [--fast-bvh] [--ref-walk] [--stats]\n"
                    "             [--devices 0-7|0,2,5] [--partition rows|samples] [--json [--steps K] [--warmup W]]\n");
    return 2;
}

// "0-7" or "0,2,5" (or a mix) -> device ids
static bool parse_devices(const std::string& spec, std::vector<int>& out) {
    size_t i = 0;
    while (i < spec.size()) {
        size_t j = spec.find(',', i);
        if (j == std::string::npos) j = spec.size();
        const std::string part = spec.substr(i, j - i);
        int a = 0, b = 0;
        if (sscanf(part.c_str(), "%d-%d", &a, &b) == 2) {
            if (a < 0 || b < a) return false;
            for (int d = a; d <= b; d++) out.push_back(d);
        } else if (sscanf(part.c_str(), "%d", &a) == 1 && a >= 0) {
            out.push_back(a);
        } else {
            return false;
        }
        i = j + 1;
    }
    return !out.empty();
}

int main(int argc, char** argv) {
    std::string scene, asset_root, out = "output.png", env;
    int spp = -1, bounces = -1, W = 0, H = 0, device = 0;
    unsigned seed = 37811;
    std::vector<int> devices;
    int partition = RT_PART_ROWS;
    bool json = false;
    int steps = 5, warmup = 1;
    bool hdr = false, spheres = false, bvh = true, wavefront = false, lockstep = false, volume = false, raymarch = false, fast_bvh = false, want_stats = false, ref_walk = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--scene") scene = next();
        else if (a == "--asset-root") asset_root = next();
        else if (a == "--out") out = next();
        else if (a == "--env") env = next();
        else if (a == "--spp") spp = atoi(next());
        else if (a == "--bounces") bounces = atoi(next());
        else if (a == "--seed") seed = (unsigned)strtoul(next(), nullptr, 10);
        else if (a == "--device") device = atoi(next());
        else if (a == "--devices") { if (!parse_devices(next(), devices)) return usage(); }
        else if (a == "--partition") { const std::string v = next(); if (v == "rows") partition = RT_PART_ROWS; else if (v == "samples") partition = RT_PART_SAMPLES; else return usage(); }
        else if (a == "--size") { if (sscanf(next(), "%dx%d", &W, &H) != 2) return usage(); }
        else if (a == "--hdr") hdr = true;            // HDRI_EXPORT, config.h:12
        else if (a == "--spheres") spheres = true;    // ENABLE_SPHERES, config.h:13
        else if (a == "--no-bvh") bvh = false;        // ENABLE_BVH, config.h:16
        else if (a == "--wavefront") wavefront = true;
        else if (a == "--lockstep") lockstep = true;
        else if (a == "--volume") volume = true;
        else if (a == "--fast-bvh") fast_bvh = true;
        else if (a == "--raymarch") raymarch = true;
        else if (a == "--ref-walk") ref_walk = true;   // every ray on the reference's own tree (RT_F_REFWALK; same image, slower)
        else if (a == "--stats") want_stats = true;
        else if (a == "--json") json = true;
        else if (a == "--steps") steps = atoi(next());
        else if (a == "--warmup") warmup = atoi(next());
        else return usage();
    }
    if (scene.empty()) return usage();
    if (bounces < 0) bounces = raymarch ? 100 : 6;  // RAY_MARCHING_MAX_STEPS config.h:15 / MAX_BOUNCES config.h:3

    rt_host_scene* hs = nullptr;
    if (rt_scene_load(scene.c_str(), asset_root.empty() ? nullptr : asset_root.c_str(), env.empty() ? nullptr : env.c_str(),
                      spheres ? 1 : 0, 0, 0, &hs) != RT_OK) {
        fprintf(stderr, "scene: %s\n", rt_scene_last_error());
        return 1;
    }
    rt_scene_desc desc;
    rt_camera cam;
    if (fast_bvh && rt_scene_rebuild_bvh(hs, RT_BVH_SAH, 0, 0) != RT_OK) {  // non-parity fast mode
        fprintf(stderr, "rebuild: %s\n", rt_scene_last_error());
        return 1;
    }
    rt_scene_get_desc(hs, &desc);
    rt_scene_get_camera(hs, &cam);
    FILE* info = json ? stderr : stdout;  // (--json: stdout carries the one JSON line only)
    fprintf(info, "Number of triangles: %zu, BVH nodes: %d, materials: %zu\n", desc.n_tri_floats / RT_TRI_FLOATS, desc.n_nodes, desc.n_mats);
    fprintf(info, "num important tris: %d\n", desc.n_light_tris);          // main.cu:442
    fprintf(info, "Image data length: %zu\n", desc.n_tex_floats);          // main.cu:456

    rt_render_params p;
    memset(&p, 0, sizeof(p));
    p.cam = cam;
    if (W > 0 && H > 0) { p.cam.width = W; p.cam.height = H; }
    p.max_bounces = bounces;
    p.flags = RT_F_SKYBOX | RT_F_ANTIALIAS | RT_F_DEFOCUS | (bvh ? RT_F_BVH : 0) | (spheres ? RT_F_SPHERES : 0) |
              (wavefront ? RT_F_WAVEFRONT : 0) | (lockstep ? RT_F_LOCKSTEP : 0) | (volume ? RT_F_VOLUME : 0) |
              (raymarch ? RT_F_RAYMARCH : 0) | (want_stats ? RT_F_STATS : 0) | (ref_walk ? RT_F_REFWALK : 0);
    p.seed = seed;
    p.sample_begin = 0;
    p.sample_count = spp > 0 ? spp : cam.num_samples;
    p.band_stride = 1;
    p.background[0] = 1.0f; p.background[1] = 1.0f; p.background[2] = 0.05f;  // config.h:7
    p.background_brightness = 0.5f;                                          // config.h:6
    std::vector<float> img((size_t)p.cam.width * p.cam.height * 3);
    rt_render_stats st;

    if (!devices.empty()) {  // several GPUs: image rows (RCCL gather) or sample slices (RCCL sum-reduce) over the devices
        rt_comm* comm = nullptr;
        if (rt_comm_create(&comm, devices.data(), (int)devices.size(), RT_COMM_RCCL) != RT_OK) {
            fprintf(stderr, "rt_comm_create failed: devices not usable (each id once, all present)\n");
            return 1;
        }
        fprintf(info, "%d devices, %s, transport %s\n", rt_comm_size(comm), partition == RT_PART_ROWS ? "image rows in bands of 16" : "sample slices",
               rt_comm_transport(comm) == RT_COMM_RCCL ? "RCCL" : "local copies");
        if (rt_comm_upload_scene(comm, &desc) != RT_OK) { fprintf(stderr, "upload: %s\n", rt_comm_last_error(comm)); return 1; }
        if (json) {  // the native host's bench line: frames of (render on every device + ONE collective), wall clock around K of them
            rt_render_params pc = p;
            pc.flags |= RT_F_STATS;  // one untimed frame with counters: rays per frame (deterministic)
            if (rt_comm_render(comm, &pc, partition, 16, &st) != RT_OK) { fprintf(stderr, "render: %s\n", rt_comm_last_error(comm)); return 1; }
            const unsigned long long rays = st.rays, paths = st.paths;
            for (int k = 0; k < warmup; k++)
                if (rt_comm_reset(comm) != RT_OK || rt_comm_render(comm, &p, partition, 16, nullptr) != RT_OK) { fprintf(stderr, "render: %s\n", rt_comm_last_error(comm)); return 1; }
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < steps; k++)
                if (rt_comm_reset(comm) != RT_OK || rt_comm_render(comm, &p, partition, 16, nullptr) != RT_OK) { fprintf(stderr, "render: %s\n", rt_comm_last_error(comm)); return 1; }
            const double step_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (steps > 0 ? steps : 1);
            uint64_t n_reduce = 0, n_sendrecv = 0;
            rt_comm_rccl_calls(comm, &n_reduce, &n_sendrecv);
            printf("{\"metric\": \"Mrays/sec (incl. bounces)\", \"value\": %.2f, \"unit\": \"Mrays/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, "
                   "\"ms_per_step\": %.3f, \"higher_is_better\": true, \"scaling\": \"strong\", \"vs_baseline\": null, \"dtype\": \"f32\", "
                   "\"data\": \"bundled scene geometry/textures, synthetic camera batch\", \"config\": {\"workload\": \"%s (%zu triangles)\", \"host\": \"rtcli (one process, one thread per device, rt_comm_*)\", "
                   "\"width\": %d, \"height\": %d, \"spp_total\": %d, \"max_bounces\": %d, \"partition\": \"%s\", \"ranks_seen_by_rccl\": %d, \"transport\": \"%s\", "
                   "\"rccl_reduces\": %llu, \"rccl_sendrecv_gathers\": %llu, \"rays_per_step\": %llu, \"paths_per_step\": %llu}}\n",
                   rays / (step_ms * 1e3), rt_comm_size(comm), steps, warmup, step_ms, scene.c_str(), desc.n_tri_floats / RT_TRI_FLOATS, p.cam.width, p.cam.height,
                   p.sample_count, p.max_bounces, partition == RT_PART_ROWS ? "rows" : "samples", rt_comm_size(comm),
                   rt_comm_transport(comm) == RT_COMM_RCCL ? "rccl" : "local", (unsigned long long)n_reduce, (unsigned long long)n_sendrecv, rays, paths);
            rt_comm_destroy(comm);
            rt_scene_free(hs);
            return 0;
        }
        printf("begin rendering\n");
        const auto t0 = std::chrono::steady_clock::now();
        if (rt_comm_render(comm, &p, partition, 16, &st) != RT_OK) { fprintf(stderr, "render: %s\n", rt_comm_last_error(comm)); return 1; }
        const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("done rendering\nRender time: %g ms (slowest device's kernels %g ms)\n", wall_ms, st.kernel_ms);
        if (want_stats)
            printf("paths %llu rays %llu (%.1f Mrays/s)\n", (unsigned long long)st.paths, (unsigned long long)st.rays, st.rays / (wall_ms * 1e3));
        if (rt_comm_readback(comm, img.data(), img.size(), hdr ? RT_READ_AVERAGE : RT_READ_TONEMAPPED) != RT_OK) {
            fprintf(stderr, "readback: %s\n", rt_comm_last_error(comm));
            return 1;
        }
        if (rt_image_save(out.c_str(), img.data(), p.cam.width, p.cam.height) != RT_OK) { fprintf(stderr, "save: %s\n", rt_scene_last_error()); return 1; }
        printf("Saved output\n");
        rt_comm_destroy(comm);
        rt_scene_free(hs);
        return 0;
    }

    rt_context* ctx = nullptr;
    if (rt_create(&ctx, device) != RT_OK) {
        fprintf(stderr, "rt_create failed: no usable HIP device (there is no CPU fallback)\n");
        return 1;
    }
    if (rt_upload_scene(ctx, &desc) != RT_OK) {
        fprintf(stderr, "upload: %s\n", rt_last_error(ctx));
        return 1;
    }
    printf("begin rendering\n");                                             // main.cu:470
    if (rt_render(ctx, &p, &st) != RT_OK) {
        fprintf(stderr, "render: %s\n", rt_last_error(ctx));
        return 1;
    }
    printf("done rendering\nRender time: %g ms\n", st.kernel_ms);            // main.cu:493, 498
    if (want_stats)
        printf("paths %llu rays %llu (%.1f Mrays/s) inner %llu leaf %llu tri %llu\n", (unsigned long long)st.paths,
               (unsigned long long)st.rays, st.rays / (st.kernel_ms * 1e3), (unsigned long long)st.inner_visits,
               (unsigned long long)st.leaf_visits, (unsigned long long)st.tri_tests);
    if (rt_readback(ctx, img.data(), img.size(), hdr ? RT_READ_AVERAGE : RT_READ_TONEMAPPED) != RT_OK) {
        fprintf(stderr, "readback: %s\n", rt_last_error(ctx));
        return 1;
    }
    if (rt_image_save(out.c_str(), img.data(), p.cam.width, p.cam.height) != RT_OK) {
        fprintf(stderr, "save: %s\n", rt_scene_last_error());
        return 1;
    }
    printf("Saved output\n");                                                // main.cu:515
    rt_destroy(ctx);
    rt_scene_free(hs);
    return 0;
}
