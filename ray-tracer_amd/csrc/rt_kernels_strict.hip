// Strict arithmetic build of the trace/shade kernels: compile with
// -ffp-contract=off so no multiply-add is fused (the reference's GHC/SSE code never
// fuses) and hit decisions are bit-identical to the CPU oracle.
#define RT_NS strict
#include "rt_kernels.inl"

namespace rtdev {
void launch_render_strict(const RenderParams &p, const LaunchCfg &cfg, hipStream_t s) {
  strict::launch(p, cfg, s);
}
int occupancy_strict(const LaunchCfg &cfg) {
  return strict::occupancy(cfg.prune, cfg.count, cfg.lds_scene, cfg.mlight, cfg.five_waves, cfg.block, cfg.lds_bytes,
                           cfg.lds_scene_bytes);
}
}  // namespace rtdev
