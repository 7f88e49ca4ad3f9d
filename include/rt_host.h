/*
 * rt_host.h -- C ABI of librt_host.so: the host-side mirror of the reference's
 * scene-setup API, i.e. the code that stays on the CPU and FEEDS the trace/shade
 * path: Materials.hs, Objects.hs, BenchmarkScene*.hs, sahBVH
 * (BoundingVolumeHierarchy.hs:34-94), configToWorld/getCam (Main.hs:104-126),
 * generateGrids (RayTracer.hs:462-475) and writePPM (Main.hs:128-138).
 *
 * In a deployment the Haskell host does all of this itself; this
 * library exists because there is no GHC in the build image, so the harness,
 * the tests and bench.py need a native producer of the same inputs.  No GPU code.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* `Config` (src/Main.hs:50-62) minus cScene. */
typedef struct rth_config {
  int32_t image_w, image_h;
  int32_t view_w, view_h, view_d;
  int32_t reflection_depth;
  int32_t anti_aliasing;        /* rays per pixel; wAntiAliasing = floor(sqrt) */
  float lens;
  float up[3], eye[3], look_at[3];
} rth_config;

typedef struct rth_scene rth_scene;

/* Built-in scenes: "bench1".."bench5" (src/BenchmarkScene*.hs with their lights),
 * "kat_tris" (benchmarks/Main.hs:10-13), "mesh:<n>" = generated station-style
 * triangle mesh of about n triangles with the 8 point lights of bench6World
 * (src/Main.hs:239-246).  Returns NULL for an unknown name. */
rth_scene *rth_scene_builtin(const char *name);

/* Scene from caller-provided objects.  `order` lists the objects in [Object]
 * list order (leaf_refs encoding: RT_REF_TRIANGLE | index, or sphere index);
 * object_id fields are overwritten with the list position. */
rth_scene *rth_scene_create(const rt_sphere *spheres, int32_t n_spheres,
                            const rt_triangle *triangles, int32_t n_triangles,
                            const uint32_t *order, int32_t n_objects,
                            const rt_material *materials, int32_t n_materials,
                            const rt_light *lights, int32_t n_lights);
void rth_scene_free(rth_scene *s);

/* parseObj + convertMesh + fromMesh (src/HaObj.hs:63-130, src/Convert.hs:15-78): the objects
 * of a .obj/.mtl pair in the reference's dialect, in the reference's [Object] order (groups
 * with `s off`/0 first, then smooth groups; one material per usemtl group; degenerate faces
 * dropped; flat normals).  No lights.  On failure returns NULL and writes the reason
 * (parse error, unknown material, non-triangle face, invalid illum) to err. */
rth_scene *rth_scene_load_obj(const char *obj_path, char *err, size_t err_len);
int rth_scene_add_light(rth_scene *s, const rt_light *light);
/* the eight point lights of bench6World (src/Main.hs:239-246) */
int rth_scene_add_bench6_lights(rth_scene *s);

/* Build the tree.  builder 0 = sahBVH (BoundingVolumeHierarchy.hs:34-51),
 * 1 = noBVH (one leaf with every object, BoundingVolumeHierarchy.hs:130-131). */
int rth_scene_build_bvh(rth_scene *s, int builder);

/* Borrowed view (valid until the scene is freed or rebuilt). */
int rth_scene_view(const rth_scene *s, rt_scene *out);
int32_t rth_scene_n_objects(const rth_scene *s);
int32_t rth_scene_bvh_depth(const rth_scene *s);

/* benchNConfig (src/Main.hs:151-235): "bench1".."bench6". */
int rth_default_config(const char *name, rth_config *out);

/* configToWorld + getCam (src/Main.hs:104-126), scalar part. */
int rth_config_to_camera(const rth_config *cfg, rt_camera *out);

/* Sample table with the structure of generateGrids/getGrids/getGridR
 * (src/RayTracer.hs:462-475, 215-222): n_grids x aa*aa x F6.  out holds
 * n_grids*aa*aa*6 floats.  The random stream is OURS (MT19937-64 seeded with
 * `seed`), not the reference's unpinned package stack. */
int rth_generate_grids(uint64_t seed, int32_t n_grids, int32_t aa, float *out);

/* generateGrids for `rng = pureMT seed`, SEED-LEVEL: the stream a GHC build of
 * the reference produces from that seed, restated from the PUBLISHED algorithms of the three
 * packages it imports (none is in /root/reference; ray-tracer.cabal:39-41 gives no versions;
 * pinned here to the ones current when the reference was written, 2014/15):
 *   mersenne-random-pure64 0.2.0.4  pureMT = init_genrand64; randomWord64 = genrand64_int64;
 *       randomDouble = (w `div` 2048) / 2^53; `next` = randomInt = the same word as Int;
 *       genRange = (minBound, maxBound)
 *   random 1.1                      randomR (0,i) on Int = randomIvalInteger: with b = 2^64 >=
 *       1000 (i+1) ONE draw x, result (x - minBound) `mod` (i+1)
 *   random-shuffle 0.0.4            shuffle' xs n g: r_j = randomR (0, n-j) for j = 1..n-1 threaded
 *       through g; output = repeatedly extract the r_j-th of what remains
 * and from the call sites: grid k draws its 2 aa^2 floats from `randomFloats rng` at position
 * 2 aa^2 k and shuffles with `iterate (snd . next) rng !! k`, i.e. the SAME stream entered k
 * words in, for both shuffles (src/RayTracer.hs:462-481).  PARITY UNPINNED: no GHC, no package
 * sources and no recorded (seed, output) pair exist to check this against. */
int rth_generate_grids_pure64(uint64_t seed, int32_t n_grids, int32_t aa, float *out);
/* The first n words of that stream (genrand64_int64 after init_genrand64(seed)), for known-answer tests. */
int rth_mt19937_64(uint64_t seed, int32_t n, uint64_t *out);

/* writePPM (src/Main.hs:128-138) and its 8-bit quantisation. */
int rth_write_ppm(const char *path, int32_t w, int32_t h, const float *rgb);
void rth_quantize(const float *rgb, size_t n, uint8_t *out);
/* Output stage after the GPU's rt_quantize_async: the same pixels as writePPM's P3 text, as
 * binary P6 (3 bytes per pixel instead of up to 12 characters). */
int rth_write_ppm_binary(const char *path, int32_t w, int32_t h, const uint8_t *rgb8);

/* Material table of Materials.hs by name ("whiteDull", "glass", ...). */
int rth_material_by_name(const char *name, rt_material *out);

/* makeTriangle / calcNormal (src/Objects.hs:13-19). */
void rth_make_triangle(const float a[3], const float b[3], const float c[3], int32_t material,
                       rt_triangle *out);

#ifdef __cplusplus
}
#endif
#endif
