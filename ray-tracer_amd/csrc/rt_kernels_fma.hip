// FMA build of the trace/shade kernels: the same source compiled with
// -ffp-contract=fast, so the Geometry3 vector math contracts to v_fma_f32
// (BASELINE.json north_star).  Selected with RT_FLAG_FMA.
#define RT_NS fma
#define RT_FAST_MATH 1  // slab quotients by reciprocal multiply
#include "rt_kernels.inl"

namespace rtdev {
void launch_render_fma(const RenderParams &p, const LaunchCfg &cfg, hipStream_t s) {
  fma::launch(p, cfg, s);
}
int occupancy_fma(const LaunchCfg &cfg) {
  return fma::occupancy(cfg.prune, cfg.count, cfg.lds_scene, cfg.mlight, cfg.five_waves, cfg.block, cfg.lds_bytes,
                           cfg.lds_scene_bytes);
}
}  // namespace rtdev
