/*
 * rt_scene.h -- host-side scene assembly (plain C++ behind a C-ABI, no GPU).
 *
 * The producers of render2's arguments in the reference:
 *   SceneParser::parseFromFile      src/sceneParser.cpp:22-171   (text DSL)
 *   SceneParser::getTriangleData    src/sceneParser.cpp:398-475  (instances -> records -> BVH -> light list)
 *   SceneParser::getMaterialData    src/sceneParser.cpp:477-517  (materials + one float RGB texture pool)
 *   Model::loadMesh                 src/model.h:47-178           (OBJ -> 25-float records)
 *   BVH::createBVH                  src/BVHcreator.cpp:240-256   (mean-split BVH2, BFS numbering)
 *   NoiseTexture::generate          src/noiseTexture.cpp:22-40
 *   Texture::Texture / saveImgData  src/texture.cpp:7-18, 49-60
 * This is host code on purpose (north_star: "Host code stays C/C++").  It
 * produces exactly the arrays rt_upload_scene (rt.h) consumes.
 */
#ifndef RT_SCENE_H_
#define RT_SCENE_H_

#include "rt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_host_scene rt_host_scene;

/* Thread-local message of the last failing rt_scene_xxx / rt_bvh_xxx / rt_image_xxx call. */
const char* rt_scene_last_error(void);

/* Parse a scene file of the reference's DSL and assemble every array.
 * asset_root != NULL: every PATH is reduced to its basename (after the last
 * '/' or '\\') and looked up under asset_root (the bundled scene files hold
 * absolute Windows paths).  env_override != NULL replaces the file named by
 * SET_BACKGROUND (the 4K HDRs named by the bundled scenes are not in the
 * reference checkout).  enable_spheres mirrors ENABLE_SPHERES (config.h:13).
 * bvh_max_depth / bvh_min_tris <= 0 use the reference's 16 / 4
 * (config.h:17-18). */
int rt_scene_load(const char* scene_path, const char* asset_root, const char* env_override,
                  int enable_spheres, int bvh_max_depth, int bvh_min_tris, rt_host_scene** out);

/* Assemble a scene from caller-made triangle records (n_floats = 25 * n) and
 * materials; builds the BVH (reordering a private copy of the records) and
 * the light list from records whose material id carries the +1000 marker
 * (sceneParser.cpp:408-410, 455-470).  textures may be NULL. */
int rt_scene_from_records(const float* tris, size_t n_floats, const rt_material* mats, size_t n_mats,
                          const float* textures, size_t n_tex_floats, int env_w, int env_h,
                          int bvh_max_depth, int bvh_min_tris, rt_host_scene** out);

/* Synthetic "instanced mesh field" (BASELINE.json config 4): `instances`
 * copies of the OBJ at obj_path on a jittered sqrt(n) x sqrt(n) grid with
 * random Y rotations, materials cycling Lambert / glossy / metal, flattened
 * like getTriangleData; plus a ground quad.  Environment = env_path
 * up-sampled by integer factor env_upsample (nearest), or none if NULL. */
int rt_scene_synth_field(const char* obj_path, int instances, uint32_t seed, const char* env_path,
                         int env_upsample, int bvh_max_depth, int bvh_min_tris, rt_host_scene** out);

void rt_scene_free(rt_host_scene* s);

/* Views into the scene (valid until rt_scene_free). */
int rt_scene_get_desc(const rt_host_scene* s, rt_scene_desc* out);
int rt_scene_get_camera(const rt_host_scene* s, rt_camera* out);
/* Deepest leaf level (root = 0) of the BVH. */
int rt_scene_bvh_depth(const rt_host_scene* s);

/* BVH::createBVH (BVHcreator.cpp:240-256): reorders tris in place (leaf
 * contiguous, BFS leaf order) and returns a malloc'ed node array (free with
 * rt_host_free). */
int rt_bvh_build(float* tris, size_t n_floats, int max_depth, int min_tris,
                 rt_bvh_node** out_nodes, int32_t* out_n_nodes);

/* NON-PARITY fast mode (no counterpart in the reference, which stops at BVH_DEPTH 16 with mean splits,
 * config.h:17): a binned-SAH BVH2 in the same node / triangle layout, leaves of at most leaf_tris triangles
 * (<= 0: 2), depth at most max_depth (<= 0: 24 = the traversal stack of rt.h).  rt_upload_scene takes it like
 * any other tree; results differ from the reference's tree only in visit counts and in which of two equally
 * distant triangles wins. */
int rt_bvh_build_sah(float* tris, size_t n_floats, int max_depth, int leaf_tris,
                     rt_bvh_node** out_nodes, int32_t* out_n_nodes);

/* Rebuild the BVH of an assembled scene in place (records reordered, light list refreshed).
 * mode RT_BVH_REFERENCE: createBVH as rt_bvh_build (leaf_tris = BVH_NODE_MIN_TRIS); RT_BVH_SAH: as above. */
enum { RT_BVH_REFERENCE = 0, RT_BVH_SAH = 1 };
int rt_scene_rebuild_bvh(rt_host_scene* s, int mode, int max_depth, int leaf_tris);

/* Model::loadMesh (model.h:47-178); returns malloc'ed records. Also accepts
 * "f v//n", "f v/t" and "f v" faces (the reference only reads "f v/t/n"). */
int rt_obj_load(const char* path, const float translation[3], float scale, const float rotation[3],
                int material_index, float** out_tris, size_t* out_n_floats);

/* NoiseTexture::generate (noiseTexture.cpp:22-40): w*h*3 floats. */
int rt_noise_texture(int w, int h, float* out_rgb);

/* Texture::Texture (texture.cpp:7-18 = stbi_loadf(...,3)): linear float RGB.
 * Formats: Radiance .hdr (RGBE, RLE), binary .ppm (8-bit, linearised with
 * gamma 2.2 as stbi_loadf does for LDR input), .pfm.  malloc'ed result. */
int rt_image_load(const char* path, float** out_rgb, int* w, int* h);

/* Texture::saveImgData (texture.cpp:49-60).  By extension: .png / .ppm = 8-bit
 * with the reference's unclamped `(unsigned char)(v*255)` cast, .pfm = raw
 * float, .hdr = Radiance RGBE (flat). */
int rt_image_save(const char* path, const float* rgb, int w, int h);
/* Texture::saveImgData(const char*, const unsigned char*, int, int) (texture.cpp:64-66): 8-bit RGB as .png / .ppm. */
int rt_image_save_rgb8(const char* path, const unsigned char* rgb8, int w, int h);

void rt_host_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* RT_SCENE_H_ */
