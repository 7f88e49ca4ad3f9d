/* rt_harness.c -- the C ABI exercised from plain C (no Python, no torch): what a Haskell host's
 * `foreign import ccall` stubs call, in the order Main.run would (src/Main.hs:32-47 of the
 * reference): build the World (host library: scene, sahBVH, getCam, generateGrids), hand it to the
 * device library, render, write the image.
 *
 *   rt_harness <scene> <width> <height> <rays per pixel> <out.ppm> [n_gpus]
 *
 * With n_gpus >= 1 the frame is rendered through rt_multi_* (tiles over the GPUs, RCCL gather) and
 * compared byte for byte with the single-context frame.  Prints one line of key=value pairs.
 * Built by __graft_entry__.build() with gcc; links librt_mi355x.so and librt_host.so only. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_host.h"
#include "rt_mi355x.h"

#define CHECK(call, ctx)                                                              \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != RT_OK) {                                                               \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, (ctx) ? rt_last_error(ctx) : ""); \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s scene width height rays_per_pixel out.ppm [n_gpus]\n", argv[0]);
    return 2;
  }
  const char *name = argv[1];
  const int w = atoi(argv[2]), h = atoi(argv[3]), rpp = atoi(argv[4]);
  const int n_gpus = argc > 6 ? atoi(argv[6]) : 0;

  rth_scene *scene = rth_scene_builtin(name);               /* bench1Objects + bench1Lights ...  */
  if (!scene || rth_scene_build_bvh(scene, 0) != 0) {       /* sahBVH                            */
    fprintf(stderr, "cannot build scene %s\n", name);
    return 1;
  }
  rt_scene view;
  rth_scene_view(scene, &view);
  rth_config cfg;
  if (rth_default_config(name, &cfg) != 0) return 1;        /* benchNConfig                      */
  cfg.image_w = w;
  cfg.image_h = h;
  cfg.anti_aliasing = rpp;
  rt_camera cam;
  if (rth_config_to_camera(&cfg, &cam) != 0) return 1;      /* configToWorld / getCam            */
  const int spp = cam.aa * cam.aa, n_grids = w + 10;
  float *grids = malloc((size_t)n_grids * spp * 6 * sizeof(float));
  float *rgb = malloc((size_t)w * h * 3 * sizeof(float));
  if (!grids || !rgb || rth_generate_grids(2024, n_grids, cam.aa, grids) != 0) return 1;

  rt_ctx *ctx = NULL;
  if (rt_init(0, &ctx) != RT_OK) {
    fprintf(stderr, "rt_init failed: no GPU?\n");
    return 1;
  }
  CHECK(rt_upload_scene(ctx, &view), ctx);
  CHECK(rt_set_camera(ctx, &cam), ctx);
  CHECK(rt_set_samples(ctx, grids, n_grids, spp), ctx);
  rt_stats st;
  CHECK(rt_render(ctx, 0, rgb, NULL, &st), ctx);             /* render rng world                  */
  if (rth_write_ppm(argv[5], w, h, rgb) != 0) return 1;     /* writePPM                          */
  unsigned long long rays = st.rays_primary + st.rays_shadow + st.rays_reflection + st.rays_refraction;
  char line[1024];   /* printed in one piece at the end: RCCL writes a banner of its own to stdout */
  int len = snprintf(line, sizeof line, "scene=%s size=%dx%d spp=%d rays=%llu primary_hits=%llu kernel_ms=%.3f", name, w,
                     h, spp, rays, (unsigned long long)st.primary_hits, st.ms_kernel);

  if (n_gpus >= 1) {
    rt_multi *m = NULL;
    float *rgb_m = malloc((size_t)w * h * 3 * sizeof(float));
    uint8_t *rgb8 = malloc((size_t)w * h * 3), *ref8 = malloc((size_t)w * h * 3);
    int rc = rt_multi_init(n_gpus, NULL, &m);
    if (rc != RT_OK) {
      fprintf(stderr, "\nrt_multi_init failed (%d): %s\n", rc, rt_multi_last_error(m));
      return 1;
    }
    rt_stats ms;
    if (rt_multi_upload_scene(m, &view) || rt_multi_set_camera(m, &cam) ||
        rt_multi_set_samples(m, grids, n_grids, spp) || rt_multi_render(m, 0, rgb_m, NULL, &ms) ||
        rt_multi_render(m, 0, NULL, rgb8, NULL)) {
      fprintf(stderr, "\nrt_multi failed: %s\n", rt_multi_last_error(m));
      return 1;
    }
    rth_quantize(rgb, (size_t)w * h * 3, ref8);
    snprintf(line + len, sizeof line - (size_t)len,
             " multi_gpus=%d multi_frame_identical=%d multi_8bit_identical=%d multi_rays=%llu", n_gpus,
             memcmp(rgb, rgb_m, (size_t)w * h * 3 * sizeof(float)) == 0, memcmp(ref8, rgb8, (size_t)w * h * 3) == 0,
             (unsigned long long)(ms.rays_primary + ms.rays_shadow + ms.rays_reflection + ms.rays_refraction));
    rt_multi_shutdown(m);
    free(rgb_m); free(rgb8); free(ref8);
  }
  printf("\n%s\n", line);
  rt_shutdown(ctx);
  rth_scene_free(scene);
  free(grids);
  free(rgb);
  return 0;
}
