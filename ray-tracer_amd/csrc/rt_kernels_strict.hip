// Strict arithmetic build of the trace/shade kernels: compile with
// -ffp-contract=off so no multiply-add is fused (the reference's GHC/SSE code never
// fuses) and hit decisions are bit-identical to the CPU oracle.
#define RT_NS strict
#include "rt_kernels.inl"

namespace rtdev {
void launch_render_strict(const RenderParams &p, const LaunchCfg &cfg, hipStream_t s) {
  strict::launch(p, cfg, s);
}
void launch_material_prep(const float *materials12, float *materials_dev, int n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(strict::material_prep_kernel, dim3((n + 63) / 64), dim3(64), 0, s, materials12, materials_dev, n);
}
int occupancy_strict(const LaunchCfg &cfg) {
  return strict::occupancy(cfg.prune, cfg.count, cfg.lds_scene, cfg.mlight, cfg.five_waves, cfg.block, cfg.lds_bytes,
                           cfg.lds_scene_bytes);
}
}  // namespace rtdev
